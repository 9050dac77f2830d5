"""ctypes binding of libadt_host.so (C ABI: include/adt_host.h): the host-side batch pipeline of the SASRec-ADT trainers -- native sampler
(WarpDataset.sample_data / random_neq of the reference, sasrec/utils.py:73-77,288-307), packing of an id batch into the trainer's pinned
ring, and the wait on the GPU's "slot consumed" counter.  Plain g++ / OpenMP code, no GPU calls."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadt_host.so")

_P, _I, _L, _F, _U64, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint32

SIGNATURES = {
    "adt_host_version": (_I, []),
    "adt_host_sample_batch": (_I, [_P, _P, _P, _I, _I, _I, _U64, _P, _P, _P, _P, _I]),
    "adt_host_sample_rows": (_I, [_P, _P, _P, _I, _I, _I, _I, _U64, _P, _P, _P, _P, _I]),
    "adt_host_count_targets": (_L, [_P, _P, _I, _I]),
    "adt_host_pack_batch": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F]),
    "adt_host_wait_ge": (_I, [_P, _U, _L]),
    "adt_host_store_release": (_I, [_P, _U]),
}

_lib = None


def load():
    """The library, or an error: the trainers' step() needs it (there is no Python fallback for the ring protocol)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libadt_host.so is missing (%s): run `python -m adt_amd.csrc.build`" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.adt_host_version() < 3:
            raise RuntimeError("libadt_host.so is stale (version %d < 3): rebuild it" % lib.adt_host_version())
        _lib = lib
    return _lib


def available():
    return os.path.exists(LIB_PATH)
