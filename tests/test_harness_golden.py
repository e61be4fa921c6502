"""CPU side of the harness fixture: tests/golden/harness_golden.json is what the oracle-backed harness gives
(the GPU side, tests/test_harness_gpu.py::test_result_files_match_committed_fixture, reproduces it on the MI355X)."""
import json
import os
import sys

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)


def test_fixture_is_what_the_oracle_backed_harness_gives(pkg, orc):
    import make_harness_golden as M
    want = json.load(open(os.path.join(GOLD, "harness_golden.json")))
    assert M.run_all(M.oracle_encoder, M.oracle_decoder) == want
    for files in want.values():            # reference line format '%f\t%e' (plot_BLER_vs_SNR.m:165), BLER < 1 only (:164)
        for txt in files.values():
            for ln in txt.splitlines():
                e, b = ln.split("\t")
                assert len(e.split(".")[1]) == 6 and "e" in b and 0.0 < float(b) < 1.0
