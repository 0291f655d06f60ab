"""lazy_load (lit_llama_amd/checkpoint.py): the streaming reader of torch.save checkpoints that replaces the
reference's lazy_load / NotYetLoadedTensor (lit_llama/utils.py:166-344).  Host logic only: runs on CPU."""
import warnings

import pytest
import torch

from lit_llama_amd import synth, tp
from lit_llama_amd.checkpoint import LazyTensor, lazy_load
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import lazy_load as lazy_load_from_utils

warnings.filterwarnings("ignore", message="The given NumPy array is not writable")


def _sample_state():
    gen = torch.Generator().manual_seed(0)
    packed = torch.randint(0, 256, (24, 40), generator=gen, dtype=torch.uint8).t()  # [40, 24] stride (1, 40)
    return {
        "a.weight": torch.randn((48, 32), generator=gen),
        "a.bf16": torch.randn((16, 8), generator=gen).to(torch.bfloat16),
        "q.quant_weight": packed,
        "q.scales": torch.rand((40, 1), generator=gen).to(torch.bfloat16),
        "scalar": torch.tensor(3.5),
        "ints": torch.arange(10, dtype=torch.int32),
        "empty": torch.zeros((0, 4)),
        "param": torch.nn.Parameter(torch.randn((5, 3), generator=gen)),
        "view": torch.randn((10, 10), generator=gen)[2:7, 1:4],  # a non-trivial offset / stride into a bigger storage
    }


def test_lazy_load_reads_nothing_until_asked_and_matches_torch_load(tmp_path):
    sd = _sample_state()
    path = tmp_path / "ckpt.pth"
    torch.save(sd, path)
    ref = torch.load(path, weights_only=False)
    with lazy_load(path) as ck:
        reader = ck  # the dict
        assert list(ck.keys()) == list(sd.keys())
        assert all(isinstance(v, LazyTensor) for v in ck.values())
        for k, v in ck.items():
            assert v.shape == ref[k].shape and v.dtype == ref[k].dtype and v.dim() == ref[k].dim()
        ll = [v for v in ck.values()][0]._storage.reader
        assert ll.bytes_mapped == 0, "metadata access must not touch tensor bytes"
        w = ck["a.weight"].materialize()
        assert ll.bytes_mapped == 48 * 32 * 4
        assert torch.equal(w, ref["a.weight"])
        for k, v in ck.items():
            t = v.materialize()
            assert t.shape == ref[k].shape and t.stride() == ref[k].stride(), k
            assert torch.equal(t, ref[k].data if isinstance(ref[k], torch.nn.Parameter) else ref[k]), k
        assert isinstance(ck["param"].materialize(), torch.nn.Parameter)
        assert ck["q.quant_weight"].stride() == (1, 40)
    assert reader is ck


def test_lazy_load_hands_out_what_the_reference_lazy_load_hands_out(golden):
    """tests/golden/lazy_ckpt.pth read by the REFERENCE's lazy_load (lit_llama/utils.py:166-344; recorded by
    `python oracle/gen_golden.py --lazy-load` in tests/golden/lazy_load.npz): same keys in the same order, every entry a
    not-yet-loaded tensor with `_load_tensor()`, same dtype / shape / stride / Parameter-ness / values — including the
    transposed quant_weight view and two views of one storage."""
    import numpy as np
    from pathlib import Path

    g = golden("lazy_load")
    ckpt = Path(__file__).resolve().parent / "golden" / "lazy_ckpt.pth"
    with lazy_load(ckpt) as ck:
        assert list(ck) == [str(k) for k in g["keys"]]
        for k, v in ck.items():
            assert hasattr(v, "_load_tensor"), k      # the attribute the reference's callers test for (quantization.py, adapter.py:177)
            dtype_s, shape_s, stride_s, is_param = (str(x) for x in g[k + "/meta"])
            assert str(v.dtype) == dtype_s and str(tuple(v.shape)) == shape_s, k
            t = v._load_tensor()
            assert str(t.dtype) == dtype_s and str(tuple(t.shape)) == shape_s and str(tuple(t.stride())) == stride_s, k
            assert str(isinstance(t, torch.nn.Parameter)) == is_param, k
            raw = t.detach().contiguous()
            raw = raw.view(torch.int16) if raw.dtype in (torch.bfloat16, torch.float16) else raw
            assert np.array_equal(raw.numpy(), g[k + "/values"]), k


class ATensor(torch.Tensor):  # (module level: the pickle of the checkpoint names the class)
    pass


def test_the_reference_s_own_lazy_load_tests_hold(tmp_path):
    """/root/reference tests/test_utils.py:12-49 restated for lit_llama_amd.utils.lazy_load: a module's state dict loads
    through `load_state_dict(lazy dict)` and computes the same; plain tensors, Parameters and Tensor SUBCLASSES come back
    equal from `_load_tensor()`."""
    m = torch.nn.Linear(5, 3)
    fn = tmp_path / "test.pt"
    torch.save(m.state_dict(), fn)
    with lazy_load_from_utils(fn) as sd_lazy:
        assert "LazyTensor" in str(next(iter(sd_lazy.values())))   # (the reference prints NotYetLoadedTensor here)
        m2 = torch.nn.Linear(5, 3)
        m2.load_state_dict(sd_lazy)
    x = torch.randn(2, 5)
    torch.testing.assert_close(m2(x), m(x))

    t = torch.randn(2, 3)[:, 1:]
    sd = {1: t, 2: torch.nn.Parameter(t), 3: torch.Tensor._make_subclass(ATensor, t)}
    fn2 = tmp_path / "sub.pt"
    torch.save(sd, fn2)
    with lazy_load_from_utils(fn2) as sd_lazy:
        for k in sd:
            torch.testing.assert_close(sd_lazy[k]._load_tensor(), sd[k])
        assert isinstance(sd_lazy[2]._load_tensor(), torch.nn.Parameter)
        assert type(sd_lazy[3]._load_tensor()) is ATensor


def test_lazy_row_shards_read_only_their_byte_range(tmp_path):
    w = torch.arange(64 * 16, dtype=torch.float32).view(64, 16)
    path = tmp_path / "w.pth"
    torch.save({"w": w}, path)
    with lazy_load(path) as ck:
        lt = ck["w"]
        reader = lt._storage.reader
        piece = lt[16:32]                       # stays lazy
        assert isinstance(piece, LazyTensor) and piece.shape == (16, 16) and reader.bytes_mapped == 0
        assert torch.equal(piece.materialize(), w[16:32])
        assert reader.bytes_mapped == 16 * 16 * 4, "a row shard is a byte sub-range of the member"
        assert torch.equal(lt.narrow(1, 4, 8).materialize(), w[:, 4:12])
        with pytest.raises(IndexError):
            lt.narrow(0, 60, 8)


def test_module_load_state_dict_consumes_lazy_tensors(tmp_path):
    cfg = LLaMAConfig(block_size=32, vocab_size=64, n_layer=2, n_head=4, n_embd=64)
    sd = synth.make_state_dict(cfg, seed=3, mode=None)
    path = tmp_path / "lit-llama.pth"
    torch.save(sd, path)
    model = LLaMA(cfg)
    with lazy_load_from_utils(path) as ck:  # the name generate.py imports (lit_llama/utils.py)
        missing = model.load_state_dict(ck, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_tp_shard_on_load_from_a_lazy_checkpoint(tmp_path):
    """scripts/convert_checkpoint.py:57-65 shard map applied to lazy tensors: identical shards, and the column-parallel
    fp weights of a rank are read as 1 / world of their bytes."""
    cfg = LLaMAConfig(block_size=32, vocab_size=64, n_layer=1, n_head=4, n_embd=64)
    sd = synth.make_state_dict(cfg, seed=5, mode=None)
    path = tmp_path / "full.pth"
    torch.save(sd, path)
    want = tp.shard_state_dict(sd, cfg, 1, 2)
    with lazy_load(path) as ck:
        reader = next(iter(ck.values()))._storage.reader
        got = tp.shard_state_dict(ck, cfg, 1, 2)
        got = {k: (v.materialize() if isinstance(v, LazyTensor) else v) for k, v in got.items()}
        total = sum(v.numel() * v.element_size() for v in sd.values())
        assert reader.bytes_mapped < total, "sharding must not read the whole checkpoint"
    assert got.keys() == want.keys()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_rejects_non_checkpoint_pickles(tmp_path):
    import pickle
    import zipfile

    path = tmp_path / "evil.pth"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("archive/data.pkl", pickle.dumps({"f": print}))
        z.writestr("archive/version", "3")
    with pytest.raises(pickle.UnpicklingError):
        lazy_load(path)


@pytest.mark.parametrize("seed", range(6))
def test_lazy_load_random_views_and_dtypes(tmp_path, seed):
    """Random tensors, transposes, slices and expanded views of shared storages: metadata and values as torch.load."""
    gen = torch.Generator().manual_seed(seed)
    dtypes = [torch.float32, torch.bfloat16, torch.float16, torch.uint8, torch.int8, torch.int64, torch.bool]
    sd = {}
    for i in range(8):
        dt = dtypes[int(torch.randint(0, len(dtypes), (1,), generator=gen))]
        shape = [int(v) for v in torch.randint(1, 9, (int(torch.randint(1, 4, (1,), generator=gen)),), generator=gen)]
        if dt.is_floating_point:
            base = (torch.randn(shape, generator=gen) * 8).to(dt)
        else:
            base = torch.randint(0, 2 if dt == torch.bool else 100, shape, generator=gen).to(dt)
        sd[f"t{i}"] = base
        if base.dim() >= 2:
            sd[f"t{i}.T"] = base.transpose(0, 1)                 # shares the storage, other strides
            sd[f"t{i}.rows"] = base[1:]                          # offset into the storage
        sd[f"t{i}.x"] = base.reshape(-1)[:1].expand(3)           # stride 0
    path = tmp_path / "r.pth"
    torch.save(sd, path)
    ref = torch.load(path, weights_only=True)
    with lazy_load(path) as ck:
        assert ck.keys() == ref.keys()
        for k in ref:
            lt = ck[k]
            assert lt.shape == ref[k].shape and lt.dtype == ref[k].dtype and lt.stride() == ref[k].stride(), k
            assert torch.equal(lt.materialize(), ref[k]), k
            assert torch.equal(lt.to(torch.float32), ref[k].to(torch.float32)), k


@pytest.mark.gpu
def test_lazy_load_to_decode_end_to_end(tmp_path):
    """SURVEY.md 8 f2 on the GPU: a gptq.int4 checkpoint written with torch.save is opened with `lazy_load` (nothing
    read), `load_state_dict` streams every tensor to HBM, the engine repacks it into the weight arena and decodes —
    token for token what the same state dict loaded from memory decodes; and the tensor-parallel path (rank-local
    row shards cut from the LAZY tensors, so a rank reads only its part) reproduces the loop-back TP run."""
    import lit_llama_amd
    from lit_llama_amd import synth, tp
    from lit_llama_amd.checkpoint import lazy_load
    from lit_llama_amd.model import LLaMA, LLaMAConfig
    from lit_llama_amd.utils import EmptyInitOnDevice

    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=2, n_head=4, n_embd=256)
    sd = synth.make_state_dict(cfg, seed=0, mode="gptq.int4", dtype=torch.bfloat16)
    path = tmp_path / "lit-llama.pth"
    torch.save(sd, path)
    prompt = synth.make_prompt(9).to(dev)

    def fresh():
        with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
            return LLaMA(cfg)

    ref_model = fresh()
    ref_model.load_state_dict(sd)
    ref = lit_llama_amd.generate(ref_model, prompt, 12, top_k=1)
    with lazy_load(path) as ckpt:
        model = fresh()
        model.load_state_dict(ckpt)
    assert model.engine() is not None, model._engine_failed
    out = lit_llama_amd.generate(model, prompt, 12, top_k=1)
    assert torch.equal(out, ref)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), ref_model.state_dict()[k].cpu()), k
    # tensor-parallel: shards cut from the lazy checkpoint
    world = 2
    with lazy_load(path) as ckpt:
        lazy_shards, eager_shards = [], []
        for r in range(world):
            m = tp.build_local_model(cfg, world, device=dev, mode="gptq.int4")
            m.load_state_dict(tp.shard_state_dict(ckpt, cfg, r, world))
            lazy_shards.append(tp.EngineShard(m, world))
            m2 = tp.build_local_model(cfg, world, device=dev, mode="gptq.int4")
            m2.load_state_dict(tp.shard_state_dict(sd, cfg, r, world))
            eager_shards.append(tp.EngineShard(m2, world))
    a = tp.TPDecoder(lazy_shards, tp.LoopbackComm(world), cfg).generate(prompt, 8)
    b = tp.TPDecoder(eager_shards, tp.LoopbackComm(world), cfg).generate(prompt, 8)
    assert torch.equal(a, b)
