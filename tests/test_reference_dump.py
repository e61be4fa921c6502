"""The door to pinned parity.

The reference's decoder core is MathWorks' closed-source comm.LDPCDecoder (NRLDPCDecoder.m:120, :265); no MATLAB exists in the
build image, so the decoder oracle is unpinned.  matlab/dump_reference_vectors.m runs that decoder on the committed fixture LLRs
(tests/golden/reference_inputs.mat) wherever MATLAB exists and writes tests/golden/reference_outputs.mat; this module compares
the reference-semantics oracle (oracle/nrldpc_oracle.c: orc_decode_bp_flood -- flooding sum-product, double, parity-check stop,
every row of H) with that file WHEN IT IS PRESENT and is skipped otherwise.  The comparison code itself is exercised either way:
on a stand-in file written from the oracle's own output in the MATLAB script's format."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUTS = os.path.join(GOLD, "reference_inputs.mat")
OUTPUTS = os.path.join(GOLD, "reference_outputs.mat")


def load_cases(path):
    from scipy.io import loadmat
    m = loadmat(path, squeeze_me=False, struct_as_record=False)
    key = "cases" if "cases" in m else "results"
    return [c for c in m[key].ravel()], m


def compare_with_dump(orc, path):
    """Every case, every iteration cap: the oracle's hard decisions and sweep counts against the dump's.  Returns a list of
    (case, cap, codewords, bit mismatches, sweep-count mismatches)."""
    ins, _ = load_cases(INPUTS)
    outs, _ = load_cases(path)
    by_name = {str(o.name.ravel()[0]) if o.name.dtype.kind in "OU" else str(o.name): o for o in outs}
    report = []
    for c in ins:
        name = str(c.name.ravel()[0])
        assert name in by_name, "the dump holds no case %r: re-run matlab/dump_reference_vectors.m on the committed inputs" % name
        o = by_name[name]
        bg, Z = int(c.BG.ravel()[0]), int(c.Z.ravel()[0])
        caps = [int(v) for v in c.iterations.ravel()]
        assert [int(v) for v in o.iterations.ravel()] == caps and int(o.BG.ravel()[0]) == bg and int(o.Z.ravel()[0]) == Z
        llr = np.ascontiguousarray(c.llr.T, np.float64)  # [batch][ncols*Z]
        B = llr.shape[0]
        hard = np.asarray(o.hard).reshape(-1, B, len(caps))        # K x batch x caps
        its = np.asarray(o.num_iterations).reshape(B, len(caps))
        for t, cap in enumerate(caps):
            h, n = orc.decode_bp_flood(bg, Z, llr, cap)            # every row of H, parity-check stop: NRLDPCDecoder.m:120
            report.append((name, cap, B, int((h != hard[:, :, t].T).sum()), int((n != its[:, t]).sum())))
    return report


def test_reference_inputs_are_the_golden_llrs():
    """reference_inputs.mat is the committed export of nmsq_golden.npz (tests/golden/export_reference_inputs.py): same LLRs,
    one codeword per column, the reference's two iteration caps (plot_BLER_vs_SNR.m:37 -> 8, NRLDPCDecoder.m:41 -> 50)."""
    d = np.load(os.path.join(GOLD, "nmsq_golden.npz"))
    cases, _ = load_cases(INPUTS)
    names = sorted({k.split("/")[0] for k in d.files})
    assert sorted(str(c.name.ravel()[0]) for c in cases) == names
    for c in cases:
        n = str(c.name.ravel()[0])
        assert (c.llr.T == d[n + "/llr"].astype(np.float64)).all()
        assert int(c.BG.ravel()[0]) == int(d[n + "/cfg"][0]) and int(c.Z.ravel()[0]) == int(d[n + "/cfg"][1])
        assert [int(v) for v in c.iterations.ravel()] == [8, 50]
    src = open(os.path.join(os.path.dirname(GOLD), "..", "matlab", "dump_reference_vectors.m")).read()
    for must in ("comm.LDPCDecoder('ParityCheckMatrix', H, 'MaximumIterationCount'", "'Parity check satisfied'", "get_pcm(get_3gpp_base_graph(BG, i_LS), Z)",
                 "reference_inputs.mat", "reference_outputs.mat"):
        assert must in src, must


def test_the_comparison_runs_on_a_dump_in_the_scripts_format(orc, tmp_path):
    """The consumer of the dump, exercised without MATLAB: a file in exactly the layout dump_reference_vectors.m saves (struct
    array `results`: name, BG, Z, iterations, hard K x batch x caps uint8, num_iterations batch x caps int32), filled from the
    oracle itself, compares clean; with one bit and one count flipped the comparison reports exactly those."""
    from scipy.io import savemat
    ins, _ = load_cases(INPUTS)
    small = [c for c in ins if int(c.Z.ravel()[0]) <= 208]  # the Z = 384 cases are what the real dump is for; keep this test short
    arr = np.zeros(len(ins), dtype=[("name", "O"), ("BG", "O"), ("Z", "O"), ("iterations", "O"), ("hard", "O"), ("num_iterations", "O")])
    for i, c in enumerate(ins):
        bg, Z = int(c.BG.ravel()[0]), int(c.Z.ravel()[0])
        caps = [int(v) for v in c.iterations.ravel()]
        llr = np.ascontiguousarray(c.llr.T, np.float64)
        K = (22 if bg == 1 else 10) * Z
        hard = np.zeros((K, llr.shape[0], len(caps)), np.uint8)
        its = np.zeros((llr.shape[0], len(caps)), np.int32)
        for t, cap in enumerate(caps):
            h, n = orc.decode_bp_flood(bg, Z, llr, cap, nthreads=4)
            hard[:, :, t], its[:, t] = h.T, n
        arr[i] = (str(c.name.ravel()[0]), float(bg), float(Z), np.array(caps, np.float64)[None, :], hard, its)
    path = str(tmp_path / "reference_outputs.mat")
    savemat(path, {"results": arr, "toolbox_version": "stand-in", "matlab_release": "none"}, do_compression=True)
    rep = compare_with_dump(orc, path)
    assert len(rep) == 2 * len(ins) and all(r[3] == 0 and r[4] == 0 for r in rep), rep
    arr[0]["hard"][3, 0, 1] ^= 1
    arr[0]["num_iterations"][0, 0] += 1
    savemat(path, {"results": arr}, do_compression=True)
    rep = compare_with_dump(orc, path)
    assert sum(r[3] for r in rep) == 1 and sum(r[4] for r in rep) == 1
    assert small  # (the fixture holds small cases too)


@pytest.mark.skipif(not os.path.exists(OUTPUTS), reason="tests/golden/reference_outputs.mat is absent: nobody has run "
                    "matlab/dump_reference_vectors.m (needs MATLAB + Communications Toolbox) -- decoder parity stays UNPINNED")
def test_oracle_against_the_reference_dump(orc):
    """PINNED PARITY, when the dump exists: comm.LDPCDecoder's hard decisions and NumIterations on the committed LLRs against
    orc_decode_bp_flood's, every case, both caps.  Bit-exact is the expectation (the same documented algorithm in double); a
    codeword whose a-posteriori values sit within rounding of zero could differ by an ulp-induced flip, so the bar is stated:
    at most 1e-4 of the bits, and sweep counts equal for at least 95 % of the codewords."""
    rep = compare_with_dump(orc, OUTPUTS)
    bits = sum(r[2] * 1 for r in rep)
    print(rep)
    total_bits = 0
    ins, _ = load_cases(INPUTS)
    for c in ins:
        total_bits += (22 if int(c.BG.ravel()[0]) == 1 else 10) * int(c.Z.ravel()[0]) * c.llr.shape[1] * len(c.iterations.ravel())
    assert sum(r[3] for r in rep) <= 1e-4 * total_bits, rep
    assert sum(r[4] for r in rep) <= 0.05 * bits, rep
