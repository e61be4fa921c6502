"""tests/golden/bler_ref.npz pinned on the CPU suite: the committed sum-product outcomes the GPU suite's BLER comparisons use
(tests/test_bler_gap_gpu.py) are what oracle/orc_decode_bp_flood gives on the seeded inputs of tests/bler_cases.py.  Every stored
run is re-computed for its first blocks (the whole file takes twenty minutes on eight cores: tests/golden/make_bler_ref.py)."""
import os

import numpy as np

import bler_cases as BC

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bler_ref.npz")
FIRST = 12


def test_every_run_of_the_suite_is_stored_and_reproduces(orc):
    d = np.load(PATH)
    runs = BC.runs()
    assert len(runs) * 3 == len(d.files)
    cache = {}
    nerr = 0
    for key, factory, case, bg, Z, nl, cap, snr in runs:
        assert key + "/err" in d.files, key
        ck = (factory.__name__, case[0])
        if ck not in cache:
            cache.clear()
            cache[ck] = factory(case, orc.encode, first=FIRST)
        inp = cache[ck]
        llr = inp.llr_at(snr)
        assert int(d[key + "/llr_crc"]) == BC.llr_crc(llr), "%s was computed from other LLRs: re-run tests/golden/make_bler_ref.py" % key
        hb, ib = orc.decode_bp_flood(bg, Z, llr, cap, n_layers=nl, nthreads=min(8, os.cpu_count() or 1))
        err = (hb[:, :inp.Kp] != inp.info[:, :inp.Kp]).any(1)
        nblk = case[-1]
        stored = np.unpackbits(d[key + "/err"])[:nblk].astype(bool)
        assert (stored[:FIRST] == err).all(), key
        assert 1.0 <= float(d[key + "/sweeps"]) <= cap
        nerr += int(stored.sum())
    assert nerr > 1000  # the grids sit in the waterfalls: thousands of block errors over the file
