"""GPTQ weight quantiser on the MI355X against the oracle (oracle/gptq.py, pinned bit for bit to
the reference by tests/golden/gptq_*.npz) and against the reference's own golden results."""
import numpy as np
import pytest
import torch

from lit_llama_amd import ops
from lit_llama_amd.gptq import GPTQQuantizer, llama_blockwise_quantization
from lit_llama_amd.quantization import ColBlockQuantizedLinear
from oracle import gptq as ogptq
from oracle import oracle

pytestmark = pytest.mark.gpu


def _hinv(K, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn((4 * K, K), generator=gen) * (1.0 + 3.0 * (torch.rand(K, generator=gen) > 0.9).float())
    H = 2.0 / x.shape[0] * x.t() @ x
    H += 0.01 * torch.mean(torch.diag(H)) * torch.eye(K)
    return torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)


@pytest.mark.parametrize("N,count,per_col", [(64, 128, False), (200, 128, True), (37, 50, False), (4096, 128, False)])
def test_gptq_block_kernel_is_bit_exact(dev, N, count, per_col):
    """The block loop (quantization.py:573-592) given identical inputs: every output bit-identical to the f32 CPU
    arithmetic (the kernel uses no fused multiply-add)."""
    gen = torch.Generator().manual_seed(N + count)
    Hinv1 = _hinv(count, seed=count)
    W1 = torch.randn((N, count), generator=gen) * count**-0.5
    s, z = ogptq.row_params(W1, 15)
    if per_col:
        sc = (s[:, None] * (1 + 0.25 * torch.rand((N, count), generator=gen))).contiguous()
        zc = torch.round(z[:, None] + torch.randint(-1, 2, (N, count), generator=gen)).clamp(0, 15).contiguous()
    else:
        sc, zc = s[:, None].expand(N, count).contiguous(), z[:, None].expand(N, count).contiguous()
    Q_o, E_o, L_o = ogptq.block_loop(W1.clone(), Hinv1, sc, zc, 15)
    # W1 handed over as a column slice of a wider matrix (the way the quantiser calls it)
    wide = torch.zeros((N, count + 40))
    wide[:, 8:8 + count] = W1
    Q, E, L = ops.gptq_block(wide.to(dev)[:, 8:8 + count], Hinv1.to(dev), (sc if per_col else s).to(dev),
                             (zc if per_col else z).to(dev), 15)
    assert torch.equal(Q.cpu(), Q_o), f"{int((Q.cpu() != Q_o).sum())} quantised values differ"
    assert torch.equal(E.cpu(), E_o)
    assert torch.equal(L.cpu(), L_o)


@pytest.mark.parametrize("name", ["gptq_actorder", "gptq_plain"])
def test_gptq_quantizer_against_reference_golden(dev, golden, name):
    """Whole quantiser on the GPU vs the reference's CPU run.  Hessian accumulation and row parameters are the same
    f32 elementwise / GEMM arithmetic; the Cholesky factor comes from rocSOLVER instead of LAPACK, so a few levels
    may land on the neighbouring level — bounded here, and the reported error must agree."""
    g = golden(name)
    W = torch.from_numpy(g["weight"])
    lin = torch.nn.Linear(W.shape[1], W.shape[0], bias=False)
    lin.weight.data.copy_(W)
    lin = lin.to(dev)
    q = GPTQQuantizer(lin, bits=int(g["bits"]), groupsize=int(g["groupsize"]), actorder=bool(g["actorder"]))
    for b in torch.from_numpy(g["batches"]):
        q.collect_input_stats(None, (b.to(dev),), None)
    qmod, err = q.quantize()
    assert isinstance(qmod, ColBlockQuantizedLinear) and qmod.quant_weight.shape == g["quant_weight"].shape
    assert torch.equal(qmod.scales.cpu(), torch.from_numpy(g["scales"]))
    assert torch.equal(qmod.zeros.cpu(), torch.from_numpy(g["zeros"]))
    got = oracle.colblock_get_weight(qmod.quant_weight.cpu(), qmod.scales.cpu(), qmod.zeros.cpu(), 4, W.shape[1])
    ref = oracle.colblock_get_weight(torch.from_numpy(g["quant_weight"]), torch.from_numpy(g["scales"]),
                                     torch.from_numpy(g["zeros"]), 4, W.shape[1])
    step = torch.from_numpy(g["scales"])  # one level = one scale
    moved = (got - ref).abs() > 0.5 * step
    assert moved.float().mean().item() <= 0.01, f"{int(moved.sum())} of {moved.numel()} levels differ"
    assert ((got - ref).abs() <= 1.001 * step).all(), "a weight moved by more than one level"
    assert abs(err - float(g["error"])) <= 0.02 * float(g["error"])
    # and the quantised module runs through the product path
    x = torch.randn((3, W.shape[1]), device=dev)
    y = qmod(x)
    assert (y.cpu() - x.cpu() @ got.t()).abs().max().item() <= 1e-3 * (x.cpu() @ got.t()).abs().max().item()


def test_gptq_quantizer_equals_oracle_given_the_same_hinv(dev, golden):
    """Without act-order and with the Cholesky factor taken from the GPU run, every block of the oracle's walk fed
    with the same Hinv must give the same quantised weights: isolates the kernel + update GEMM from rocSOLVER."""
    g = golden("gptq_plain")
    W = torch.from_numpy(g["weight"])
    hs = ogptq.Hessian(W.shape[1])
    for b in torch.from_numpy(g["batches"]):
        hs.add(b)
    N, K = W.shape
    Hd = hs.H.clone()
    Hd[torch.arange(K), torch.arange(K)] += 0.01 * torch.mean(torch.diag(Hd))
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    s, z = ogptq.row_params(W, 15)
    Wc, Wg = W.clone(), W.clone().to(dev)
    Hg = Hinv.to(dev)
    for i1 in range(0, K, 128):
        i2 = min(i1 + 128, K)
        sc, zc = s[:, None].expand(N, i2 - i1).contiguous(), z[:, None].expand(N, i2 - i1).contiguous()
        Q_o, E_o, _ = ogptq.block_loop(Wc[:, i1:i2].clone(), Hinv[i1:i2, i1:i2], sc, zc, 15)
        Q, E, _ = ops.gptq_block(Wg[:, i1:i2], Hg[i1:i2, i1:i2].contiguous(), s.to(dev), z.to(dev), 15)
        assert torch.equal(Q.cpu(), Q_o) and torch.equal(E.cpu(), E_o)
        Wc[:, i2:] -= E_o.matmul(Hinv[i1:i2, i2:])
        Wg[:, i2:] -= E.matmul(Hg[i1:i2, i2:])
        # the update GEMM may differ in summation order: keep the two walks on the same inputs
        Wg[:, i2:] = Wc[:, i2:].to(dev)


def test_blockwise_quantization_of_a_small_model(dev):
    """quantize/gptq.py:37-135 end to end on a 2-layer model: every linear becomes a ColBlockQuantizedLinear with
    the reference's state-dict layout, and the quantised model stays close to the fp one on the calibration data."""
    from lit_llama_amd.model import LLaMA, LLaMAConfig
    from lit_llama_amd import synth

    cfg = LLaMAConfig(block_size=32, vocab_size=64, n_layer=2, n_head=4, n_embd=128)
    sd = synth.make_state_dict(cfg, seed=0, mode=None)
    model = LLaMA(cfg).to(dev)
    model.load_state_dict(sd)
    model.eval()
    gen = torch.Generator().manual_seed(5)
    samples = torch.randint(0, cfg.vocab_size, (6, 32), generator=gen).to(dev)
    with torch.no_grad():
        ref_logits = model(samples[:2]).float().cpu()
    errors = llama_blockwise_quantization(model, samples, dev, bits=4, log=lambda *_: None)
    assert len(errors) == 2 * 5 + 1 and all(np.isfinite(v) for v in errors.values())
    for name, mod in model.named_modules():
        if name.endswith(("c_attn", "c_proj", "c_fc1", "c_fc2", "lm_head")):
            assert isinstance(mod, ColBlockQuantizedLinear), name
            assert mod.quant_weight.dtype == torch.uint8 and mod.quant_weight.stride() == (1, mod.out_features)
    keys = set(model.state_dict().keys())
    assert {"lm_head.quant_weight", "lm_head.scales", "lm_head.zeros",
            "transformer.h.0.attn.c_attn.quant_weight", "transformer.h.1.mlp.c_proj.zeros"} <= keys
    with torch.no_grad():
        q_logits = model(samples[:2]).float().cpu()
    rel = (q_logits - ref_logits).norm() / ref_logits.norm()
    assert rel.item() <= 0.25, f"int4 model drifted by {rel.item():.3f}"
