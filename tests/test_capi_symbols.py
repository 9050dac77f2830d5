"""CPU: libadt_hip.so loads, exports every symbol include/adt_hip.h declares, and the ctypes signatures in
adt_amd/_lib.py agree with the header argument by argument.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from adt_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes():
    src = open(os.path.join(REPO, "include", "adt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"(const char\*|int64_t|int)\s+(adt_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [a.strip() for a in args.replace("\n", " ").split(",")]
        if args == ["void"]:
            args = []
        protos[name] = (ret, args)
    return protos


def code_of(arg):
    if "adt_sasrec_cfg" in arg:
        return "CP"
    if "adt_enc_layer_ptrs" in arg:
        return "EP"
    if "adt_dec_layer_ptrs" in arg:
        return "DP"
    if "*" in arg:
        return "P"
    t = arg.split()[0] if not arg.startswith("const") else arg.split()[1]
    return {"float": "F", "uint32_t": "U", "int64_t": "L", "int": "I", "int32_t": "I"}[t]


CT = {"P": ctypes.c_void_p, "I": ctypes.c_int, "F": ctypes.c_float, "U": ctypes.c_uint32, "L": ctypes.c_int64}


def test_header_and_ctypes_signatures_agree():
    protos = header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, args) in protos.items():
        res, argtypes = _lib.SIGNATURES[name]
        assert len(args) == len(argtypes), (name, len(args), len(argtypes))
        for i, (a, t) in enumerate(zip(args, argtypes)):
            c = code_of(a)
            if c in ("CP", "EP", "DP"):
                assert t is ctypes.POINTER({"CP": _lib.SasrecCfg, "EP": _lib.EncLayerPtrs, "DP": _lib.DecLayerPtrs}[c]), (name, i, a)
            else:
                assert t is CT[c], (name, i, a, t)
        want = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "const char*": ctypes.c_char_p}[ret]
        assert res is want, name


def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()
    for name in header_prototypes():
        assert hasattr(lib, name), name
    assert lib.adt_version() >= 1


def test_host_side_helpers_without_gpu():
    """Pure host entry points: hash RNG known answers (same as oracle/rng.py) and the flat layouts."""
    import numpy as np
    from oracle import rng
    lib = _lib.load()
    idx = np.arange(64)
    for seed, site, p in ((7, 17, 0.5), (123456789, 130, 0.2), (0, 1, 0.9)):
        want = rng.keep_mask(seed, site, idx, p).astype(int)
        got = [lib.adt_rng_keep(seed, site, int(i), p) for i in idx]
        assert list(want) == got
    cfg = _lib.SasrecCfg(3416, 200, 64, 2, 2, 0.5, 1)
    offs = (ctypes.c_int64 * (4 + 30 * 2))()
    total = lib.adt_sasrec_param_layout(ctypes.byref(cfg), offs)
    assert offs[0] == 0 and offs[1] >= 3417 * 64 and total > 365892
    assert all(o % 64 == 0 for o in offs)
    ws = lib.adt_sasrec_workspace_floats(ctypes.byref(cfg), 256)
    assert ws > 0 and ws * 4 < 4 << 30
    assert lib.adt_sasrec_ws_offset(ctypes.byref(cfg), 256, 0, 1) == 256 * 200 * 64


def test_library_issues_no_memset_nodes():
    """The entry points are captured into HIP graphs by the trainers.  A captured hipMemsetAsync node was observed on the MI355X box
    to start writing a stale non-zero pattern after a few hundred replays (gradient-norm accumulators reading ~4e30 / NaN for the rest
    of the process, which silently disables or saturates clipping; DESIGN.md "graph memset"), so the library zero-fills with its own
    kernel and must not import any hipMemset* entry point."""
    import subprocess
    so = os.path.join(REPO, "adt_amd", "csrc", "libadt_hip.so")
    out = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in out
    assert not [l for l in out.splitlines() if "hipMemset" in l], out
