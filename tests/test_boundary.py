"""The drop-in boundary as a C ABI (include/sgfhe_hip.h): the header is valid C, the parameter
struct has the layout both bindings assume (ctypes `SgfheParams`, Julia `CParams`), the Julia shim
binds only exported symbols, and a library records which sources it was built from.  No GPU."""

import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "sgfhe_hip.h")
JL = os.path.join(ROOT, "sgfhe.jl_amd", "julia", "SGFHEHip.jl")
FIELDS = ["n", "r", "m", "ell", "Q", "B", "DQ_tilde"]


def test_header_is_plain_c():
    """`extern "C"`, plain pointers and sizes: the header compiles as C99 and as C++ with warnings
    as errors."""
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror",
                           "-fsyntax-only", HDR])
    subprocess.check_call(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Wextra", "-Werror",
                           "-fsyntax-only", HDR])


def test_params_struct_layout_matches_bindings(S, tmp_path):
    """sizeof(sgfhe_params) == 80 and every field offset, as the C compiler lays it out, equals
    the ctypes mirror and the field order / types of the Julia struct."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sgfhe_hip.h"\n'
                   'int main(void) { printf("%zu", sizeof(sgfhe_params));\n'
                   + "".join('printf(" %%zu", offsetof(sgfhe_params, %s));\n' % f for f in FIELDS)
                   + 'printf(" %u", (unsigned)SGFHE_ABI_VERSION); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.dirname(HDR), "-o", str(exe), str(src)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    size, offs, abi = nums[0], nums[1:-1], nums[-1]
    P = S._lib.SgfheParams
    assert size == 80 == ctypes.sizeof(P)
    assert [f for f, _ in P._fields_] == FIELDS
    assert offs == [getattr(P, f).offset for f in FIELDS] == [0, 8, 16, 24, 32, 48, 64]
    assert abi == S.ABI_VERSION
    # Julia: struct CParams, fields in the same order, UInt64 / NTuple{2,UInt64}
    jl = open(JL).read()
    body = re.search(r"struct CParams\n(.*?)\nend", jl, re.S).group(1)
    fields = [tuple(x.strip() for x in line.split("::")) for line in body.strip().splitlines()]
    assert [f for f, _ in fields] == FIELDS
    assert [t for _, t in fields] == ["UInt64"] * 4 + ["NTuple{2,UInt64}"] * 3
    assert int(re.search(r"const ABI_VERSION = UInt32\((\d+)\)", jl).group(1)) == abi


def test_julia_shim_binds_only_exported_symbols(S):
    """Every `ccall((:name, libsgfhe_hip), ...)` of SGFHEHip.jl names a symbol the header declares
    and the library exports, with the argument count of the C prototype."""
    jl = open(JL).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    protos = {m.group(1): m.group(2) for m in
              re.finditer(r"\b(sgfhe_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, re.S)}
    L = ctypes.CDLL(S.build())
    calls = re.findall(r"ccall\(\(:(\w+), libsgfhe_hip\),\s*\w+,\s*\(([^()]*)\)", jl, re.S)
    assert len(calls) >= 8
    for name, argt in calls:
        assert name in S.EXPORTED_SYMBOLS and name in protos and hasattr(L, name), name
        n_jl = len([a for a in argt.replace("\n", " ").split(",") if a.strip()])
        c_args = protos[name].strip()
        n_c = 0 if c_args in ("", "void") else len(c_args.split(","))
        assert n_jl == n_c, (name, argt, c_args)
    assert "sgfhe_abi_version" in [c[0] for c in calls]     # a stale library is refused at load


def test_library_records_its_sources(S):
    """sgfhe_build_id() is the hash of csrc/ the library was compiled from: readable without
    loading the file, equal to the hash of the sources for the in-tree build."""
    path = S.build()
    assert S.embedded_build_id(path) == S.source_hash()
    assert S.lib().sgfhe_build_id().decode() == S.source_hash()
    assert S.lib().sgfhe_abi_version() == S.ABI_VERSION


def test_header_documents_what_the_engine_accepts():
    """The two places round 2's header had drifted from the engine."""
    hdr = open(HDR).read()
    eng = open(os.path.join(ROOT, "sgfhe.jl_amd", "csrc", "engine.hip")).read()
    assert "B < 2^47" in hdr and "(c->B >> 47)" in eng
    assert "Always the deterministic flatten" not in hdr


def test_c_example_compiles_against_the_header(S, tmp_path):
    """examples/gate_demo.c is plain C99 over the C ABI alone; it builds and links here (running it
    needs a GPU: tests/test_gpu_c_example.py) and fails loudly without a device."""
    lib_dir = os.path.dirname(S.build())
    exe = str(tmp_path / "gate_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.dirname(HDR),
                           os.path.join(ROOT, "examples", "gate_demo.c"), "-L", lib_dir, "-lsgfhe_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "-> -3" in r.stderr          # SGFHE_ERR_NO_DEVICE, no fallback
