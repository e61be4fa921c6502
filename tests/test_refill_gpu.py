"""Slot refill under the parity stop (round 5; nrldpc_decode_z64p.h): the workgroups of the interleaved and packed geometries hold
NCW codewords, and under the reference's only mode -- 'Parity check satisfied', NRLDPCDecoder.m:120 -- they used to live until the
LAST of them stopped.  Now a finished codeword's lanes write their result and take the next codeword of the batch (a counter in
device memory), the others keep their messages; launches are persistent.  Which slot decodes which codeword depends on timing;
what every codeword gets must not: hard decisions and iteration counts are the oracle's, bit for bit.

The refill path needs more codewords than the resident workgroups start with (tens of thousands on an MI355X); NRLDPC_REFILL_GRID
caps the workgroups of a launch instead, so that every kernel goes through it with a few hundred codewords."""
import os

import numpy as np
import pytest

from conftest import ALL_Z, BG_DIMS, awgn_llr, rule_kw
from test_decode_gpu import _waterfall_esn0

pytestmark = pytest.mark.gpu


def check(pkg, orc, rng, bg, Z, B, esn0, iters, nl, dt):
    kb = BG_DIMS[bg][2]
    info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
    llr = awgn_llr(rng, orc.encode(bg, Z, info), esn0, dt, Z)
    llr[rng.integers(0, B, max(1, B // 16))] *= 0.05   # a few hopeless codewords: they run to the cap while their neighbours come and go
    c = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=dt)
    try:
        h, it = c.decode(llr, want_iters=True)
        h2, it2 = c.decode(llr, want_iters=True)         # the counters of the handle's ring are reused
    finally:
        c.close()
    ref = orc.decode_nmsq(bg, Z, llr.astype(np.float64), iters, n_layers=nl, early_term=True, **rule_kw(c))
    assert (h == ref[0]).all() and (it == ref[1]).all(), (bg, Z, nl, B)
    assert (h2 == h).all() and (it2 == it).all()
    assert it.min() < iters and it.max() == iters        # early leavers and stragglers in one launch


@pytest.mark.parametrize("bg", [1, 2])
@pytest.mark.parametrize("grid", [1, 3])
def test_every_small_lifting_size_with_two_or_three_workgroups(pkg, orc, bg, grid):
    """Every lifting size up to 192 (whatever kernel serves its parity stop: interleaved, packed, block geometry, run-time-Z --
    the cap only binds the refilling ones), all rows and a pruned count, a batch of several rounds per slot, not a multiple of
    anything."""
    rng = np.random.default_rng(5100 + 10 * bg + grid)
    rows = BG_DIMS[bg][0]
    os.environ["NRLDPC_REFILL_GRID"] = str(grid)
    try:
        for Z in [z for z in ALL_Z if z <= 192]:
            B = 5 * (512 // Z + 1) + 3
            for nl in (0, 13 if bg == 1 else 9):
                check(pkg, orc, rng, bg, Z, B, _waterfall_esn0(bg, nl or rows) + 0.3, 12, nl, np.float16 if (Z + grid) % 2 else np.float32)
    finally:
        del os.environ["NRLDPC_REFILL_GRID"]


def test_a_batch_larger_than_the_resident_workgroups(pkg, orc):
    """No cap: the launch is as wide as the device holds (occupancy x compute units) and the batch is several times what those
    workgroups start with -- the shape of a BLER sweep over a short code (BASELINE configs[0]: Z = 20)."""
    rng = np.random.default_rng(77)
    for bg, Z, nl, B in ((2, 20, 12, 60000), (1, 8, 0, 60000)):
        check(pkg, orc, rng, bg, Z, B, _waterfall_esn0(bg, nl or BG_DIMS[bg][0]) + 0.3, 15, nl, np.float16)
