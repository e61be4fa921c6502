import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NRLDPC_TEST_HOOKS", "1")  # the library reads its test-only knobs (NRLDPC_REFILL_GRID) only under this
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

ALL_Z = sorted(a * 2 ** j for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * 2 ** j <= 384)
BG_DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has hyphens, so import by string)."""
    m = importlib.import_module("ldpc-3gpp-matlab_amd")
    sys.modules.setdefault("nrldpc_amd", m)
    return m


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import oracle
    oracle.lib()
    return oracle


def awgn_llr(rng, cw, esn0_db, dtype, Z, E=None):
    """QPSK/AWGN LLRs as in plot_BLER_vs_SNR.m:105-106 / NRModulator.m:75; first 2Z punctured."""
    import numpy as np
    mu = 2.0 * 10.0 ** (esn0_db / 10.0)
    llr = (1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * rng.standard_normal(cw.shape)
    llr[:, : 2 * Z] = 0
    if E is not None:
        llr[:, 2 * Z + E:] = 0
    return llr.astype(dtype)


def rule_kw(codec, scale=8):
    """The check-node rule a Codec resolved (cfg.alpha == 0 -> nrldpc_default_rule) as oracle keyword arguments:
    the C ABI takes beta in LLR units, the oracle in grid units (LLR * scale)."""
    return {"alpha": codec.alpha, "beta": codec.beta * scale}
