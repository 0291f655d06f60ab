"""Build libmi355llama.so (gfx950) in-tree with hipcc.

No torch linkage: the library is a plain C-ABI shared object (include/mi355_llama.h) that only needs the
HIP runtime, which `_native.py` makes sure is the instance PyTorch already loaded.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libmi355llama.so"
OBJ_DIR = PKG / "csrc" / "_obj"
SOURCES = ["generic.hip", "gemv.hip", "int8.hip", "int8_gemm.hip", "attention.hip", "engine.hip", "gptq.hip", "fused_step.hip", "fused_step_ring.hip", "fused_step_wide.hip", "tp_comm.hip", "gemm.hip", "flash_prefill.hip", "sample.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    headers = [CSRC / "common.h", CSRC / "fused_step_common.h", CSRC / "gemm_fuse.h", PKG.parent / "include" / "mi355_llama.h"]
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]

    def compile_one(name: str):
        src = CSRC / name
        obj = OBJ_DIR / (name + ".o")
        if force or _stale(obj, [src, *headers]):
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {name}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return obj

    compiled = []

    def compile_tracked(name: str):
        obj = OBJ_DIR / (name + ".o")
        before = obj.stat().st_mtime if obj.exists() else None
        out = compile_one(name)
        if before is None or out.stat().st_mtime != before:
            compiled.append(name)
        return out

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_tracked, SOURCES))
    relink = force or _stale(LIB, objs)
    if verbose:  # say what this call did: an up-to-date tree compiles nothing, and the caller should see that
        print(f"[build] {len(SOURCES)} sources for {ARCH}: compiled {sorted(compiled) if compiled else 'none (objects up to date)'}; "
              f"{'linking' if relink else 'library up to date:'} {LIB}", file=sys.stderr)
    if relink:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
