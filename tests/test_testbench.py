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


def matlab_active_layers(p):
    """`active_layers` of matlab/ldpc-3gpp-matlab.patch, statement by statement (MATLAB's 0-based `pos` values kept; only the
    indexing pos(1:n) becomes pos[:n]) -- written from the patch text, sharing no code with nrldpc.py::active_layers."""
    Z_c_, N_cb_ = p.Z_c, p.N_cb
    E_ = max(p.E_r)
    rows, core_cols = (46, 22) if p.BG == 1 else (42, 10)
    n_layers = 4
    if E_ > 0:
        pos = np.mod(p.k_0 + np.arange(0, N_cb_), N_cb_)
        pos = pos[(pos < max(p.K_prime - 2 * Z_c_, 0)) | (pos >= p.K - 2 * Z_c_)]
        top = int(pos[: min(E_, pos.size)].max())
        n_layers = min(max((top + 2 * Z_c_) // Z_c_ - core_cols + 1, 4), rows)
    return n_layers


def test_the_matlab_patch_passes_the_exact_row_count(pkg):
    """the patched NRLDPCDecoder.LDPC_coding hands `nrldpc_mex('decode', ...)` the active row count it derives from
    E_r, k_0 and N_cb (sticky while HARQ state is pending, cleared by resetImpl) instead of leaving the library to scan cw_tilde.
    Over testbench.m's draws -- rv_id 0..3, LBRM, repetition, segmentation -- that count equals what NRLDPC_LAYERS_AUTO reads off
    the very cw_tilde the reference's chain builds (nrldpc_count_layers, the definition of AUTO: host function, no device), for a
    first transmission and along the reference's HARQ sequence [0 2 3 1] (plot_BLER_vs_SNR.m:36)."""
    import os
    patch = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matlab", "ldpc-3gpp-matlab.patch")).read()
    for needle in ("function n_layers = active_layers(obj)", "pos = mod(obj.k_0 + (0:N_cb_-1), N_cb_);",
                   "pos = pos(pos < max(obj.K_prime-2*Z_c_,0) | pos >= obj.K-2*Z_c_);", "top = max(pos(1:min(E_,numel(pos))));",
                   "n_layers = min(max(floor((top + 2*Z_c_)/Z_c_) - core_cols + 1, 4), rows);",
                   "n_layers = max(n_layers, obj.layers_seen);", "nrldpc_mex('decode', obj.hLDPCDecoder, cw_tilde, n_layers)",
                   "obj.layers_seen = 0;"):
        assert needle in patch, needle
    C = pkg._capi
    draws, _ = accepted_draws(606, 90, a_max=30000)
    rng = np.random.default_rng(9)
    seen = {"pruned": 0, "full": 0, "lbrm": 0, "rep": 0, "rv": set(), "harq_grew": 0}
    for kw, ref in draws:
        for harq in (False, True):
            dec = pkg.NRLDPCDecoder(I_HARQ=int(harq), **kw)
            dec._nb = 1
            dec._setup()
            sticky = 0
            for rv in ((0, 2, 3, 1) if harq else (kw["rv_id"],)):
                dec.rv_id = rv
                g = rng.standard_normal((1, dec.G)) + 0.25  # no exact zeros: every transmitted position shows in cw_tilde
                g[g == 0] = 1.0
                d = dec.rate_recover(g)[0]  # [C][N], NaN at the fillers (NRLDPCDecoder.m:143-242)
                cw = np.concatenate([np.zeros((dec.C, 2 * dec.Z_c)), d], axis=1)  # :262
                cw[np.isnan(cw)] = np.inf  # :264
                want = C.count_layers(dec.BG, dec.Z_c, cw)
                n = matlab_active_layers(dec)
                assert n == dec.active_layers()
                if harq:
                    seen["harq_grew"] += n > sticky > 0
                    n = sticky = max(n, sticky)
                assert n == want, (kw, rv, harq, n, want)
                rows = 46 if dec.BG == 1 else 42
                seen["pruned" if n < rows else "full"] += 1
                seen["lbrm"] += dec.N_cb < dec.N
                seen["rep"] += max(dec.E_r) > dec.N_cb
                seen["rv"].add(rv)
    assert seen["pruned"] > 20 and seen["full"] > 20 and seen["lbrm"] > 10 and seen["rep"] > 5 and seen["rv"] == {0, 1, 2, 3}
    assert seen["harq_grew"] > 5  # a later redundancy version reached higher than the first: the sticky maximum mattered
