"""Build libadt_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["adt_capi.hip", "adt_sasrec.hip", "adt_wide.hip", "adt_seq.hip", "adt_lce.hip"]
HEADERS = ["adt_common.cuh", "adt_itemgrad.cuh", "adt_rowops.cuh", "adt_attn.cuh", "adt_attn_bf16.cuh", "adt_misc.cuh", "adt_wave.cuh", "adt_bwdchain.cuh", "adt_bwdchain_args.h", "adt_fwdchain.cuh", "adt_fwdchain_args.h", "adt_seq_args.h", "adt_seqfwd.cuh", "adt_seqfwd_tt.cuh", "adt_tt.cuh", "adt_seqattn.cuh", "adt_seqbwd_tt.cuh", "adt_seqbwd_args.h", "adt_seqpost_tt.cuh", "adt_host.h", "adt_gemm.cuh", "adt_attn_gen.cuh", "adt_wide.cuh", "adt_stosa.cuh", "adt_wattn_mfma.cuh", "adt_lce.cuh", "../../include/adt_hip.h"]
OUT = os.path.join(HERE, "libadt_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


HOST_SRC = "adt_hostdata.cpp"
HOST_OUT = os.path.join(HERE, "libadt_host.so")


def build_host(force=False, verbose=True):
    """Host-side batch sampler (plain g++, OpenMP): no GPU code."""
    if not force and os.path.exists(HOST_OUT) and os.path.getmtime(HOST_OUT) >= os.path.getmtime(os.path.join(HERE, HOST_SRC)):
        return HOST_OUT
    cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-o", HOST_OUT, HOST_SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return HOST_OUT


def _includes(src, seen=None):
    """Local headers a source file includes, transitively (so that only the objects a change touches are rebuilt)."""
    import re
    seen = set() if seen is None else seen
    for m in re.finditer(r'#include "([^"]+)"', open(os.path.join(HERE, src)).read()):
        h = os.path.normpath(m.group(1))
        if h not in seen and os.path.exists(os.path.join(HERE, h)):
            seen.add(h)
            _includes(h, seen)
    return seen


def build(force=False, verbose=True):
    """One object per translation unit (compiled in parallel, rebuilt only when the unit or a header it includes
    changed), then one link."""
    build_host(force, verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(obj)
        deps = [src] + sorted(_includes(src))
        stale = force or not os.path.exists(obj) or any(os.path.getmtime(os.path.join(HERE, d)) > os.path.getmtime(obj) for d in deps)
        if stale:
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if procs or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [hipcc] + flags + ["-shared", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
