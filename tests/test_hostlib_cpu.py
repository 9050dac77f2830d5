"""CPU: libadt_host.so (include/adt_host.h) -- exports, ctypes signatures, and the host half of the trainer's id ring: the packed block
layout, the sharded native sampler (the union of the ranks' rows is the batch one process draws; layout of WarpDataset.sample_data,
sasrec/utils.py:288-307), the BCE normaliser taken from history lengths, and the wrap-safe wait on the "consumed" counter."""
import ctypes
import os
import re
import threading
import time

import numpy as np

from adt_amd import _hostlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_ctypes_signatures_agree():
    src = open(os.path.join(REPO, "include", "adt_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"(int64_t|int)\s+(adt_host_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        protos[m.group(2)] = (m.group(1), [] if args == ["void"] else args)
    assert set(protos) == set(_hostlib.SIGNATURES), set(protos) ^ set(_hostlib.SIGNATURES)
    code = {"float": ctypes.c_float, "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64, "int": ctypes.c_int}
    lib = _hostlib.load()
    for name, (ret, args) in protos.items():
        res, argtypes = _hostlib.SIGNATURES[name]
        assert hasattr(lib, name) and len(args) == len(argtypes), name
        assert res is {"int": ctypes.c_int, "int64_t": ctypes.c_int64}[ret], name
        for a, t in zip(args, argtypes):
            want = ctypes.c_void_p if "*" in a else code[[w for w in a.split() if w not in ("const", "volatile")][0]]
            assert t is want, (name, a)


def _warp(L=12):
    from adt_amd.sasrec import synth, utils as U
    hist, nu, ni = synth.generate("tiny", 23)
    train = {u: v[:-2] for u, v in hist.items()}
    return U.WarpDataset(train, nu, ni, L), train, ni


def test_sharded_rows_equal_the_single_process_batch_and_follow_the_reference_layout():
    w, train, ni = _warp()
    r = np.random.RandomState(1)
    users = next(iter(w.epoch_users(16, r)))
    seed = w.next_seed(r)
    full = [np.full((16, 12), -1, np.int32) for _ in range(4)]
    w.sample_rows_into(users, seed, full)
    parts = []
    for lo, hi in ((0, 6), (6, 11), (11, 16)):          # three "ranks", uneven
        out = [np.full((hi - lo, 12), -1, np.int32) for _ in range(4)]
        w.sample_rows_into(users[lo:hi], seed, out, b0=lo)
        parts.append(out)
    for k in range(4):
        assert (np.concatenate([p[k] for p in parts]) == full[k]).all()
    seq, dec, pos, neg = full
    for b, u in enumerate(users):                          # WarpDataset.sample_data, sasrec/utils.py:288-307
        h = train[u]
        n = min(len(h) - 1, 12)
        assert list(seq[b, 12 - n:]) == h[-(n + 1):-1] and (seq[b, :12 - n] == 0).all()
        assert list(pos[b, 12 - n:]) == h[-n:] and (pos[b, :12 - n] == 0).all()
        assert dec[b, 0] == 0 and (dec[b, 1:] == seq[b, :-1]).all()
        assert ((neg[b] == 0) == (pos[b] == 0)).all()
        assert all(1 <= t <= ni and t not in set(h) for t in neg[b, 12 - n:])
    assert w.count_targets(users) == np.count_nonzero(pos)
    # same stream as sample_batch() with the same rng state
    w2, _, _ = _warp()
    r2 = np.random.RandomState(1)
    users2 = next(iter(w2.epoch_users(16, r2)))
    _, s2, d2, p2, n2 = w2.sample_batch(users2, r2)
    assert users2 == users and (s2 == seq).all() and (n2 == neg).all()


def test_pack_batch_layout_and_in_place_sources():
    lib = _hostlib.load()
    T = 40
    r = np.random.RandomState(0)
    arrs = [r.randint(0, 100, size=T).astype(np.int32) for _ in range(4)]
    dst = np.full(4 * T + 4, -7, np.int32)
    assert lib.adt_host_pack_batch(dst.ctypes.data, *[a.ctypes.data for a in arrs], T, 3.0, 2560.0, 80.0) == 0
    for k in range(4):
        assert (dst[k * T:(k + 1) * T] == arrs[k]).all()
    assert list(dst[4 * T:4 * T + 3].view(np.float32)) == [3.0, 2560.0, 80.0] and dst[4 * T + 3] == 0
    # sources that already are their part of dst (sampled in place) are left alone
    dst2 = dst.copy()
    ptrs = [dst2[k * T:(k + 1) * T].ctypes.data for k in range(4)]
    assert lib.adt_host_pack_batch(dst2.ctypes.data, *ptrs, T, 1.0, 2.0, 3.0) == 0
    assert (dst2[:4 * T] == dst[:4 * T]).all() and list(dst2[4 * T:4 * T + 3].view(np.float32)) == [1.0, 2.0, 3.0]


def test_wait_ge_wraps_and_times_out():
    lib = _hostlib.load()
    c = np.zeros(4, np.uint32)
    assert lib.adt_host_wait_ge(c.ctypes.data, 0, 1000) == 0
    t0 = time.time()
    assert lib.adt_host_wait_ge(c.ctypes.data, 1, 20000) == -1 and time.time() - t0 >= 0.015
    c[0] = 0xFFFFFFFE
    assert lib.adt_host_wait_ge(c.ctypes.data, 0xFFFFFFFD, 1000) == 0
    assert lib.adt_host_wait_ge(c.ctypes.data, 2, 5000) == -1          # 2 is "ahead" of 0xFFFFFFFE across the wrap
    threading.Timer(0.05, lambda: c.__setitem__(0, 3)).start()           # the "GPU" publishes a new count
    assert lib.adt_host_wait_ge(c.ctypes.data, 2, 2000000) == 0
