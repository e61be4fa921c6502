"""GPU encoder (replacement of comm.LDPCEncoder at NRLDPCEncoder.m:49,158): H*c = 0, systematic,
identical to the oracle's encoder, linear."""
import numpy as np
import pytest

from conftest import ALL_Z, BG_DIMS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bg", [1, 2])
def test_encoder_all_lifting_sizes(pkg, orc, bg):
    rng = np.random.default_rng(300 + bg)
    kb = BG_DIMS[bg][2]
    for Z in ALL_Z:
        info = rng.integers(0, 2, (3, kb * Z), dtype=np.uint8)
        c = pkg.Codec(bg, Z, max_iter=1)
        cw = c.encode(info)
        c.close()
        assert (cw[:, : kb * Z] == info).all()
        assert (cw == orc.encode(bg, Z, info)).all()
        assert all(orc.syndrome_weight(bg, Z, cw[b]) == 0 for b in range(3))


def test_encoder_large_batch_linearity(pkg, orc):
    rng = np.random.default_rng(8)
    c = pkg.Codec(1, 384, max_iter=1)
    info = rng.integers(0, 2, (512, c.K), dtype=np.uint8)
    cw = c.encode(info)
    x = c.encode(info[:256] ^ info[256:])
    c.close()
    assert ((cw[:256] ^ cw[256:]) == x).all()
    assert orc.syndrome_weight(1, 384, cw[511]) == 0 and orc.syndrome_weight(1, 384, cw[0]) == 0
