"""TEST INFRASTRUCTURE ONLY -- counter-based dropout RNG shared by the oracle and the HIP kernels.

The reference draws dropout masks from ATen's CPU generator (F.dropout at
sasrec/modules.py:60-61,629 and nn.Dropout at sasrec/model.py:20,38), a stream
that cannot be reproduced inside a GPU kernel.  Both the HIP path
(adt_amd/csrc/adt_common.cuh: adt_hash32 / adt_keep) and this oracle instead
derive every keep/drop decision from a stateless integer hash of
(seed, site, element index), so the two sides agree bit-for-bit on masks and a
training-mode step can be compared exactly.  Parity with the *reference* is
therefore exact only with dropout == 0 (golden vectors) and statistical
otherwise (NDCG tolerance), as DESIGN.md states.
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_GOLD = np.uint32(0x9E3779B9)


def hash32(x):
    """lowbias32 integer hash (public-domain finaliser by C. Wellons), vectorised over uint32."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(16)
    return x


def site_key(seed, site):
    """Per-(seed, site) key: hash32(seed ^ site*golden)."""
    with np.errstate(over="ignore"):
        s = np.uint32(seed) ^ (np.uint32(site) * _GOLD)
    return hash32(np.array([s], dtype=np.uint32))[0]


def threshold(p):
    """8-bit drop threshold: an element is dropped iff its random byte < threshold(p) = round(p * 256), at most 255."""
    t = int(float(p) * 256.0 + 0.5)
    return np.uint32(min(max(t, 0), 255))


def drop_prob(p):
    """The drop probability actually applied, threshold(p) / 256 (0.5 -> 0.5, 0.2 -> 0.19922, 0.3 -> 0.30078): survivors are
    scaled by 1 / (1 - drop_prob(p)), so the mask stays unbiased at the quantised rate."""
    return float(threshold(p)) / 256.0


def keep_mask(seed, site, idx, p):
    """Boolean keep mask for element indices `idx` (any shape, values < 2^32).  One 32-bit hash serves the four elements
    4k .. 4k+3, one byte each (element idx reads byte idx & 3 of hash32((idx >> 2) ^ key)): a kernel whose lane holds four
    consecutive elements -- an MFMA accumulator register quad -- hashes once per quad instead of four times."""
    idx = np.asarray(idx)
    assert idx.size == 0 or int(idx.max()) < (1 << 32)
    key = site_key(seed, site)
    idx = idx.astype(np.uint32)
    r = hash32((idx >> np.uint32(2)) ^ key)
    byte = (r >> (np.uint32(8) * (idx & np.uint32(3)))) & np.uint32(0xFF)
    return byte >= threshold(p)
