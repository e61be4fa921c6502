"""testbench.m:19-73 as a test (CPU half): the reference's randomized conformance sweep -- R, I_LBRM, A = ceil(1e5^rand),
TBS_LBRM, Q_m, N_L, G, rv_id drawn as :21-36, BG by the rule of :26-30, parameter sets the objects refuse
('UnsupportedParameters', :48-56) skipped.  The reference compares its encoder with nrULSCH of the 5G Toolbox (:59-73),
which does not exist here; what the draws are checked against instead:
  * every Dependent property of the product's NRLDPC mirror (nrldpc.py) vs a second transcription of NRLDPC.m:297-543
    that shares no code with it (tests/testbench_ref.py::derive);
  * the host mirror of NRLDPCEncoder.step (CRC, segmentation, oracle-encoded code blocks, rate matching) vs literal
    per-element loops of NRLDPCEncoder.m:168-256 on the same code blocks."""
import numpy as np
import pytest

import testbench_ref as TB

N_DRAWS = 260  # accepted parameter sets (the skip rate is ~35 %: B' not a multiple of C)


def accepted_draws(seed, n, a_max=None):
    rng = np.random.default_rng(seed)
    out, skipped = [], 0
    while len(out) < n:
        kw = TB.draw(rng)
        if a_max and kw["A"] > a_max:
            continue
        try:
            ref = TB.derive(**kw)
        except TB.Unsupported:
            skipped += 1
            continue
        out.append((kw, ref))
    return out, skipped


def test_parameter_chain_on_testbench_draws(pkg):
    draws, skipped = accepted_draws(20240929, N_DRAWS)
    assert skipped > 0  # the sweep does hit parameter sets the reference refuses
    names = ("transport_block_L", "B", "K_cb", "code_block_L", "C", "B_prime", "K_prime", "K_b", "Z_c", "K", "i_LS", "N",
             "N_ref", "N_cb", "C_prime", "E_r", "k_0")
    seen_bg, seen_multi, seen_lbrm = set(), 0, 0
    for kw, ref in draws:
        p = pkg.NRLDPC(**kw)
        p.validate()
        got = {n: getattr(p, n) for n in names}
        assert got == ref, (kw, {n: (got[n], ref[n]) for n in names if got[n] != ref[n]})
        assert sum(ref["E_r"]) == kw["G"]
        seen_bg.add(kw["BG"]); seen_multi += ref["C"] > 1; seen_lbrm += ref["N_cb"] < ref["N"]
    assert seen_bg == {1, 2} and seen_multi > 10 and seen_lbrm > 10  # the sweep reaches segmentation and LBRM


def test_refused_draws_are_refused_by_the_mirror_too(pkg):
    rng = np.random.default_rng(7)
    n = 0
    while n < 60:
        kw = TB.draw(rng)
        try:
            TB.derive(**kw)
        except TB.Unsupported:
            with pytest.raises(pkg.UnsupportedParameters):
                pkg.NRLDPC(**kw).validate()
            n += 1


def test_host_encoder_tail_vs_literal_loops_on_testbench_draws(pkg, orc):
    """Rate matching of the host mirror (vectorised index maps) vs the reference's while / for loops, on the code blocks
    the oracle encoder produces for random payloads; A capped so that the pure-Python loops stay in seconds."""
    draws, _ = accepted_draws(99, 40, a_max=6000)
    for kw, ref in draws:
        if kw["G"] > 60000:
            continue
        enc = pkg.NRLDPCEncoder(**kw)
        rng = np.random.default_rng(kw["A"])
        a = rng.integers(0, 2, (1, kw["A"]), dtype=np.uint8)
        c = enc.code_block_segmentation(enc.crc_calculation(a)).reshape(ref["C"], ref["K"])
        cw = orc.encode(kw["BG"], ref["Z_c"], c)
        g = enc.rate_match(cw.reshape(1, ref["C"], -1)[:, :, 2 * ref["Z_c"]:])[0]
        d = cw[:, 2 * ref["Z_c"]:].copy()
        d[:, max(ref["K_prime"] - 2 * ref["Z_c"], 0): ref["K"] - 2 * ref["Z_c"]] = 2  # fillers are NaN in d (NRLDPCEncoder.m:160)
        lit = TB.literal_encoder_tail(d, dict(ref, Q_m=kw["Q_m"]))
        assert g.shape == (kw["G"],) and (g == lit).all(), kw
