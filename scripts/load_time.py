#!/usr/bin/env python
"""Checkpoint -> HBM load time of a synthetic LLaMA-7B gptq.int4 checkpoint: `torch.load` + load_state_dict vs
`lazy_load` (lit_llama_amd/checkpoint.py) + load_state_dict, then the one-time native repack (DecodeEngine)."""
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import build_model  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice, lazy_load  # noqa: E402


class A:
    model, quantize, tune = "7B", "gptq.int4", None


def fresh(dev):
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        m = LLaMA(LLaMAConfig.from_name("7B"))
    return m.eval()


def main():
    dev = torch.device("cuda:0")
    model, cfg = build_model(A, dev)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    del model
    torch.cuda.empty_cache()
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    path = Path(d) / "lit-llama-7b-int4.pth"
    t0 = time.perf_counter()
    torch.save(sd, path)
    gb = path.stat().st_size / 1e9
    print(f"checkpoint {gb:.2f} GB written in {time.perf_counter() - t0:.1f} s")
    del sd

    for name in ("torch.load", "lazy_load"):
        m = fresh(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "torch.load":
            ck = torch.load(path, map_location="cpu", weights_only=True)
            m.load_state_dict(ck)
            del ck
        else:
            with lazy_load(path) as ck:
                m.load_state_dict(ck)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng = m.engine()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert eng is not None, m._engine_failed
        print(f"{name:10s}: file -> HBM {t1 - t0:6.2f} s ({gb / (t1 - t0):5.2f} GB/s), native repack {t2 - t1:5.2f} s")
        del m, eng
        torch.cuda.empty_cache()
    path.unlink()


if __name__ == "__main__":
    main()
