"""The C-ABI shared library: loads without a GPU, exports every symbol include/mi355_llama.h declares, struct
layouts agree with the ctypes mirror, argument errors come back as codes + messages (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from lit_llama_amd import _native as nat

HEADER = Path(__file__).resolve().parents[1] / "include" / "mi355_llama.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|size_t|char\s*\*|const char\s*\*)\s+\**\s*(mi355_\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    lib = nat.lib()
    names = declared_functions()
    assert len(names) >= 30, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mi355_llama.h but not exported"
    missing_proto = [n for n in names if n not in nat.PROTOTYPES]
    assert not missing_proto, f"ctypes prototypes missing for {missing_proto}"
    extra = [n for n in nat.PROTOTYPES if n not in names]
    assert not extra, f"ctypes binds undeclared symbols {extra}"


def test_struct_layouts_match():
    lib = nat.lib()
    for i, st in enumerate(nat.ABI_STRUCTS):
        assert lib.mi355_sizeof(i) == C.sizeof(st), st.__name__
    assert lib.mi355_sizeof(99) == -1
    assert lib.mi355_version() == 1


def test_every_field_of_every_struct_sits_where_the_header_puts_it(tmp_path):
    """The ctypes mirrors (lit_llama_amd/_native.py) against include/mi355_llama.h compiled as C99 by gcc: same field names,
    same offsets, same sizes — mi355_sizeof only covers the sizes, and a drop-in binding lives or dies by the offsets."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("mi355_linear_args", nat.LinearArgs), ("mi355_attn_args", nat.AttnArgs), ("mi355_adapter_args", nat.AdapterArgs),
             ("mi355_int8_args", nat.Int8Args), ("mi355_weight", nat.Weight), ("mi355_layer", nat.Layer),
             ("mi355_model", nat.Model), ("mi355_fused_step_args", nat.FusedStepArgs), ("mi355_tp_comm", nat.TpComm)]
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "mi355_llama.h"', "int main(void) {"]
    for cname, st in pairs:
        src.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in st._fields_:
            src.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    src.append("return 0; }")
    (tmp_path / "o.c").write_text("\n".join(src))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", f"-I{HEADER.parent}", str(tmp_path / "o.c"), "-o", str(tmp_path / "o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr  # (also: the header is plain C and every ctypes field name exists in it)
    got = dict(line.split() for line in subprocess.run([str(tmp_path / "o")], capture_output=True, text=True).stdout.splitlines())
    for cname, st in pairs:
        assert int(got[cname]) == C.sizeof(st), cname
        for f, _ in st._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(st, f).offset, f"{cname}.{f}"
    assert len(got) > 200


def test_the_binding_stubs_in_integration_md_follow_the_structs():
    """INTEGRATION.md shows the ctypes stubs a lit-llama maintainer would paste; a struct that grew must grow there too."""
    text = (HEADER.parents[1] / "INTEGRATION.md").read_text()
    blk = text[text.index("class _LinearArgs"):text.index("def _check(rc)")]
    assert re.findall(r'\("(\w+)", C\.c_', blk) == [f for f, _ in nat.LinearArgs._fields_]
    blk = text[text.index("class _FusedArgs"):text.index("# once per model")]
    names = re.findall(r'\("(\w+)", C\.c_', blk)
    ints = [x.strip().strip('"') for x in re.search(r"\(n, C\.c_int32\) for n in\s*\(([^)]*)\)", blk).group(1).split(",")]
    k = names.index("eps")
    assert names[:k] + ints + names[k:] == [f for f, _ in nat.FusedStepArgs._fields_]


def test_packed_bytes_accounting():
    lib = nat.lib()
    # 7B shapes: one byte per two int4 weights, no padding
    assert lib.mi355_packed_bytes(nat.W_Q4, 12288, 4096, 2, 0) == 12288 * 4096 // 2
    assert lib.mi355_packed_bytes(nat.W_Q4, 4096, 11008, 1, 0) == 4096 * 11008 // 2
    assert lib.mi355_packed_bytes(nat.W_Q4, 11008, 4096, 2, 1) == 2 * 11008 * 4096 // 2  # c_fc1 + c_fc2 interleaved
    assert lib.mi355_packed_bytes(nat.W_BF16, 32000, 4096, 2, 0) == 32000 * 4096 * 2
    assert lib.mi355_packed_bytes(nat.W_I8, 4096, 4096, 1, 0) == 4096 * 4096
    # padding: rows to whole tiles, K to whole 128-column units (65B TP=8 mlp.c_proj shard: K = 2752)
    assert lib.mi355_packed_bytes(nat.W_Q4, 8192, 2752, 1, 0) == 8192 * 2816 // 2
    assert lib.mi355_packed_bytes(nat.W_Q4, 40, 256, 1, 0) == 48 * 256 // 2
    assert lib.mi355_packed_bytes(nat.W_Q4, 0, 256, 1, 0) == 0
    assert lib.mi355_packed_bytes(nat.W_Q4, 64, 256, 3, 0) == 0


def test_argument_errors_are_reported_not_crashed():
    lib = nat.lib()
    assert lib.mi355_linear_fast(None, None) == -1
    assert b"null" in lib.mi355_last_error()
    a = nat.LinearArgs()
    a.fmt, a.R, a.M, a.N, a.K = nat.W_Q4, 1, 17, 64, 256
    a.w = a.x = a.y = 0x1000  # never dereferenced: validation fails first
    assert lib.mi355_linear_fast(C.byref(a), None) == -2
    assert b"M=17" in lib.mi355_last_error()
    a.M = 1
    a.epi = nat.EPI_SWIGLU  # needs the interleaved stream
    assert lib.mi355_linear_fast(C.byref(a), None) == -1
    a.epi = nat.EPI_STORE
    assert lib.mi355_linear_fast(C.byref(a), None) == -1  # Q4 without scales / zeros
    assert b"scales" in lib.mi355_last_error()
    a.fmt = nat.W_I8
    assert lib.mi355_linear_fast(C.byref(a), None) == -1
    assert lib.mi355_attention(None, None) == -1
    assert lib.mi355_argmax(None, 10, None, None, None, None) == -1
    with pytest.raises(nat.NativeError, match="rc=-1"):
        nat.check(lib.mi355_forward(None, 1, 1, 0, None), "mi355_forward")
    m = nat.Model()
    assert lib.mi355_graph_capture(C.byref(m), 0, None, C.byref(C.c_void_p())) == -1
    assert b"default stream" in lib.mi355_last_error()
