"""Build libadt_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["adt_capi.hip", "adt_sasrec.hip"]
HEADERS = ["adt_common.cuh", "adt_rowops.cuh", "adt_attn.cuh", "adt_attn_bf16.cuh", "adt_misc.cuh", "adt_chain.cuh", "adt_wave.cuh", "adt_bwdchain.cuh", "adt_bwdchain_args.h", "adt_fwdchain.cuh", "adt_fwdchain_args.h", "adt_chain_args.h", "adt_host.h", "../../include/adt_hip.h"]
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


def build(force=False, verbose=True):
    build_host(force, verbose)
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
