"""The oracle itself: encoder pinned by H*c = 0, decoders by round trips, and the C restatement of
the build's algorithm cross-checked against an independent numpy restatement and committed fixtures."""
import os

import numpy as np
import pytest

from conftest import ALL_Z, BG_DIMS, awgn_llr

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("bg", [1, 2])
def test_encoder_parity_all_Z(orc, bg):
    rng = np.random.default_rng(100 + bg)
    kb = BG_DIMS[bg][2]
    for Z in ALL_Z:
        info = rng.integers(0, 2, (2, kb * Z), dtype=np.uint8)
        cw = orc.encode(bg, Z, info)
        assert (cw[:, : kb * Z] == info).all()
        for b in range(2):
            assert orc.syndrome_weight(bg, Z, cw[b]) == 0
        # linearity: enc(a) ^ enc(b) = enc(a ^ b)
        assert ((cw[0] ^ cw[1]) == orc.encode(bg, Z, info[0] ^ info[1])[0]).all()
    assert orc.syndrome_weight(bg, 8, np.zeros(BG_DIMS[bg][1] * 8, np.uint8)) == 0


def nmsq_numpy(bg, Z, llr, max_iter, n_layers, early_term, alpha, scale, orc, beta=0.0):
    """Independent restatement of NMS-Q (vectorised over z, written from the algorithm statement in
    DESIGN.md, not from the C code)."""
    nrows, ncols, kb = BG_DIMS[bg]
    n_layers = n_layers or nrows
    r_, c_, s_ = orc.graph_edges(bg, Z)
    x = np.asarray(llr, np.float64)
    q = np.where(np.isnan(x), 0.0, np.rint(np.clip(np.float32(x) * np.float32(scale), -127, 127))).astype(np.int64)
    core = (np.arange(ncols * Z) // Z) < kb + 4
    q = np.where(np.isinf(x), np.where(core, np.sign(x) * 2 ** 20, np.sign(x) * 127), q).astype(np.int64)
    APP = q.copy()
    msg = {}
    zz = np.arange(Z)
    it_done = max_iter
    for it in range(1, max_iter + 1):
        for l in range(n_layers):
            es = np.nonzero(r_ == l)[0]
            vidx = [c_[e] * Z + (zz + s_[e]) % Z for e in es]
            t = np.stack([APP[v] - msg.get(e, 0) for e, v in zip(es, vidx)])
            a = np.abs(t)
            srt = np.sort(a, axis=0)
            m1, m2 = srt[0], srt[1]
            S = (t < 0).sum(0) & 1
            # alpha*m - beta exactly (float64 holds it: 24-bit alpha x 21-bit m), rounded once to the nearest integer,
            # ties to even, then clamped to [0, 127]
            fm = lambda m: np.clip(np.rint(np.float64(np.float32(alpha)) * m - np.float64(np.float32(beta))), 0, 127).astype(np.int64)
            M1, M2 = fm(m1), fm(m2)
            for k, (e, v) in enumerate(zip(es, vidx)):
                mag = np.where(a[k] == m1, M2, M1)
                rr = np.where(((t[k] < 0) ^ (S == 1)), -mag, mag)
                APP[v] = t[k] + rr
                msg[e] = rr
        if early_term:
            bad = 0
            for l in range(n_layers):
                p = np.zeros(Z, np.int64)
                for e in np.nonzero(r_ == l)[0]:
                    p ^= (APP[c_[e] * Z + (zz + s_[e]) % Z] < 0)
                bad += p.sum()
            if bad == 0:
                it_done = it
                break
    return (APP[: kb * Z] < 0).astype(np.uint8), it_done, (APP / scale).astype(np.float32)


@pytest.mark.parametrize("bg,Z,nl,et,alpha,scale,esn0,beta", [
    (1, 8, 0, False, 0.75, 8, 1.0, 0.0), (1, 24, 10, True, 0.625, 16, 3.0, 0.0), (2, 20, 0, True, 0.75, 8, 0.0, 0.0),
    (2, 6, 6, False, 0.8125, 4, 4.0, 0.0), (1, 36, 46, True, 0.6875, 8, -1.0, 0.0),
    (1, 36, 46, True, 0.875, 8, -1.0, 3.0), (2, 20, 0, True, 0.8125, 8, 0.0, 2.0), (1, 8, 0, False, 0.7, 16, 1.0, 2.5),
    (2, 20, 0, True, 0.9, 4, 0.0, 1.0), (1, 24, 10, False, 0.3, 8, 2.0, 0.5)])
def test_nmsq_c_vs_numpy(orc, bg, Z, nl, et, alpha, scale, esn0, beta):
    rng = np.random.default_rng(7)
    kb = BG_DIMS[bg][2]
    info = rng.integers(0, 2, (3, kb * Z), dtype=np.uint8)
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, esn0, np.float32, Z).astype(np.float64)
    llr[0, 5 * Z + 1] = np.inf
    llr[1, 3 * Z] = np.nan
    llr[2, (kb + 5) * Z] = -np.inf
    h, it, app = orc.decode_nmsq(bg, Z, llr, 6, n_layers=nl, early_term=et, alpha=alpha, scale=scale, want_app=True,
                                 beta=beta)
    for b in range(3):
        hn, itn, appn = nmsq_numpy(bg, Z, llr[b], 6, nl, et, alpha, scale, orc, beta)
        assert (h[b] == hn).all() and it[b] == itn
        assert (app[b] == appn).all()


def bp_flood_numpy(bg, Z, llr, max_iter, n_layers, orc):
    """Independent restatement of the reference-semantics decoder, written from SURVEY.md appendix B1 (not from
    the C code): flooding sum-product in double, r_cv = 2 atanh(prod_{v' != v} tanh(q_v'c / 2)) with the product
    clipped to +-(1 - 1e-15), q_vc = lambda_v + sum_{c' != c} r_c'v, stop when every parity check holds
    ('Parity check satisfied', NRLDPCDecoder.m:120).  Leave-one-out products are formed directly per edge."""
    nrows, ncols, kb = BG_DIMS[bg]
    n_layers = n_layers or nrows
    r_, c_, s_ = orc.graph_edges(bg, Z)
    zz = np.arange(Z)
    lam = np.where(np.isnan(llr), 0.0, np.asarray(llr, np.float64))
    rows = [[(e, c_[e] * Z + (zz + s_[e]) % Z) for e in np.nonzero(r_ == l)[0]] for l in range(n_layers)]
    r = {e: np.zeros(Z) for row in rows for e, _ in row}

    def totals():
        a = lam.copy()
        for row in rows:
            for e, v in row:
                a[v] += r[e]
        return a

    it_done = max_iter
    with np.errstate(invalid="ignore"):
        for it in range(1, max_iter + 1):
            app = totals()
            new = {}
            for row in rows:
                q = [np.where(np.isinf(app[v]), app[v], app[v] - r[e]) for e, v in row]
                th = [np.tanh(0.5 * x) for x in q]
                for k, (e, v) in enumerate(row):
                    p = np.ones(Z)
                    for k2 in range(len(row)):
                        if k2 != k:
                            p = p * th[k2]
                    new[e] = 2.0 * np.arctanh(np.clip(p, -(1 - 1e-15), 1 - 1e-15))
            r = new
            app = totals()
            bad = 0
            for row in rows:
                par = np.zeros(Z, np.int64)
                for e, v in row:
                    par ^= (app[v] < 0)
                bad += int(par.sum())
            if bad == 0:
                it_done = it
                break
    return (app[: kb * Z] < 0).astype(np.uint8), it_done, app


@pytest.mark.parametrize("bg,Z,nl,esn0,iters", [(2, 20, 12, 1.0, 10), (1, 8, 0, 0.0, 6), (2, 6, 0, 3.0, 8), (1, 24, 10, 2.5, 5),
                                                 (1, 36, 46, -1.0, 7)])
def test_bp_flood_c_vs_numpy(orc, bg, Z, nl, esn0, iters):
    """The stand-in for comm.LDPCDecoder (CPU baseline, BLER yardstick) against an independent restatement:
    hard bits and sweep counts equal at the full sweep count; a-posteriori LLRs within 1e-8 after 2 sweeps (later,
    converged messages sit at the clip 1 - 1e-15 where one ulp of the product moves r = ln(2/eps) by up to 0.1, so
    soft values are order-of-multiplication dependent there and are compared at 1e-2 only)."""
    rng = np.random.default_rng(17)
    kb = BG_DIMS[bg][2]
    info = rng.integers(0, 2, (4, kb * Z), dtype=np.uint8)
    info[:, kb * Z - 7:] = 0
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, esn0, np.float64, Z)
    llr[:, kb * Z - 7: kb * Z] = np.inf     # fillers (NRLDPCDecoder.m:264)
    llr[1, 3 * Z + 1] = np.nan             # treated as 0
    llr[2, (kb + 6) * Z:] = 0.0            # untransmitted tail
    llr[3] *= 1.5                          # larger LLRs, short of saturating tanh in double
    h, it, app = orc.decode_bp_flood(bg, Z, llr, iters, n_layers=nl, want_app=True)
    for b in range(4):
        hn, itn, appn = bp_flood_numpy(bg, Z, llr[b], iters, nl, orc)
        assert it[b] == itn and (h[b] == hn).all()
        fin = np.isfinite(appn)
        assert (np.isinf(app[b]) == ~fin).all() and (app[b][~fin] == appn[~fin]).all()
        assert np.allclose(app[b][fin], appn[fin], rtol=1e-2, atol=1e-2)
    h2, it2, app2 = orc.decode_bp_flood(bg, Z, llr, 2, n_layers=nl, want_app=True)
    for b in range(4):
        hn, itn, appn = bp_flood_numpy(bg, Z, llr[b], 2, nl, orc)
        fin = np.isfinite(appn)
        assert it2[b] == itn and (h2[b] == hn).all()
        assert np.allclose(app2[b][fin], appn[fin], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("bg,Z", [(1, 384), (2, 384), (1, 2), (2, 3), (1, 208), (2, 20)])
def test_noise_free_round_trip(orc, bg, Z):
    rng = np.random.default_rng(11)
    kb = BG_DIMS[bg][2]
    info = rng.integers(0, 2, (2, kb * Z), dtype=np.uint8)
    cw = orc.encode(bg, Z, info)
    llr = 10.0 * (1 - 2.0 * cw)
    llr[:, : 2 * Z] = 0  # punctured columns are recovered by the decoder
    h, it = orc.decode_nmsq(bg, Z, llr, 5, early_term=True)
    assert (h == info).all() and (it <= 2).all()
    h, it = orc.decode_bp_flood(bg, Z, llr, 5)
    assert (h == info).all() and (it <= 3).all()


def test_bp_and_nmsq_decode_awgn(orc):
    """Both algorithms clear a comfortable SNR: BG2 Z=20 (cfg1: A=100 -> K'=116, 84 fillers), 10 iterations."""
    rng = np.random.default_rng(5)
    bg, Z, kb, Kp = 2, 20, 10, 116
    info = rng.integers(0, 2, (40, kb * Z), dtype=np.uint8)
    info[:, Kp:] = 0
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, 4.0, np.float64, Z, E=300)
    llr[:, Kp: kb * Z] = np.inf  # fillers (NRLDPCDecoder.m:264)
    h1, _ = orc.decode_nmsq(bg, Z, llr, 10, n_layers=12, early_term=True)
    h2, _ = orc.decode_bp_flood(bg, Z, llr, 10)
    assert ((h1 != info).any(1)).mean() < 0.2 and ((h2 != info).any(1)).mean() < 0.2


def test_golden_fixture(orc):
    """Committed input/output vectors (generated by tests/golden/make_golden.py from the oracle; the
    reference holds no decoder vectors -- 'parity unpinned')."""
    g = np.load(os.path.join(GOLD, "nmsq_golden.npz"))
    for name in sorted(set(k.split("/")[0] for k in g.files)):
        bg, Z, nl, it, et, alpha, scale = (g[name + "/cfg"][i] for i in range(7))
        h, iters, app = orc.decode_nmsq(int(bg), int(Z), g[name + "/llr"].astype(np.float64), int(it),
                                        n_layers=int(nl), early_term=bool(et), alpha=float(alpha),
                                        scale=int(scale), want_app=True)
        assert (np.packbits(h, axis=1) == g[name + "/hard_packed"]).all()
        assert (iters == g[name + "/iters"]).all()
        assert (app.astype(np.float16) == g[name + "/app_f16"]).all()
        cw = orc.encode(int(bg), int(Z), g[name + "/info"])
        assert (np.packbits(cw, axis=1) == g[name + "/cw_packed"]).all()


@pytest.mark.parametrize("bg,Z,nl,Kp,poly,L,esn0", [(2, 20, 12, 116, 0x11021, 16, 1.6), (1, 24, 0, 400, 0x1800063, 24, -0.3), (2, 36, 22, 360, 0x1864CFB, 24, -0.9)])
def test_crc_aided_stop_variant_of_the_oracle(orc, bg, Z, nl, Kp, poly, L, esn0):
    """orc_decode_onmsq_crc (the checker of nrldpc_cfg.early_term = 2) pinned against its own definition, built from pieces that
    are pinned elsewhere: run the fixed-iteration decoder for 1, 2, 3, ... iterations; the CRC-aided stop must end at the first
    count whose hard decisions satisfy every active parity check OR leave CRC remainder 0 on the first K' bits with a bit set
    (bit-serial CRC: orc_crc, itself pinned against the chain's literal loops), with exactly those hard decisions."""
    rng = np.random.default_rng(31 * Z + Kp)
    rows, cols, kb = BG_DIMS[bg]
    K, B, cap = kb * Z, 10, 12
    nla = nl or rows
    info = np.zeros((B, K), np.uint8)
    info[:, : Kp - L] = rng.integers(0, 2, (B, Kp - L), dtype=np.uint8)
    for b in range(B):
        r = orc.crc(poly, L, info[b, : Kp - L])
        info[b, Kp - L: Kp] = (r >> np.arange(L - 1, -1, -1)) & 1
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, esn0, np.float32, Z).astype(np.float64)
    llr[:, Kp:K] = np.inf
    llr[3, : 4 * Z] = 0  # punctured systematic bits: all-zero decisions early on must not count as a CRC match
    hard, iters = orc.decode_nmsq_crc(bg, Z, llr, cap, (poly, L, Kp), n_layers=nl, alpha=0.875, beta=2.0)
    _, it_parity = orc.decode_nmsq(bg, Z, llr, cap, n_layers=nl, early_term=True, alpha=0.875, beta=2.0)
    assert (iters <= it_parity).all()
    stopped_by_crc = 0
    for b in range(B):
        want = None
        for it in range(1, cap + 1):
            h, _, app = orc.decode_nmsq(bg, Z, llr[b:b + 1], it, n_layers=nl, early_term=False, alpha=0.875, beta=2.0, want_app=True)
            word = (app[0] < 0).astype(np.uint8)
            parity_ok = orc.syndrome_weight(bg, Z, word, n_layers=nla) == 0
            crc_ok = orc.crc(poly, L, h[0, :Kp]) == 0 and h[0, :Kp].any()
            if parity_ok or crc_ok or it == cap:
                want = (it, h[0])
                stopped_by_crc += int(crc_ok and not parity_ok)
                break
        assert iters[b] == want[0] and (hard[b] == want[1]).all(), b
    assert stopped_by_crc > 0  # the case exercises the CRC branch, not only the parity one


def test_wide_grid_variant_equals_the_build_algorithm_at_127(orc):
    """orc_decode_onmsq_wide with the kernels' own saturation (+-127) IS orc_decode_onmsq; with a wider grid it differs only
    where a value saturated."""
    rng = np.random.default_rng(8)
    bg, Z = 1, 16
    info = rng.integers(0, 2, (6, 22 * Z), dtype=np.uint8)
    llr = awgn_llr(rng, orc.encode(bg, Z, info), 1.0, np.float32, Z).astype(np.float64)
    llr[0] *= 40.0  # this block saturates the 8-bit grid
    a = orc.decode_nmsq(bg, Z, llr, 8, early_term=True, alpha=0.875, beta=3.0)
    b = orc.decode_nmsq_wide(bg, Z, llr, 8, early_term=True, alpha=0.875, beta=3.0, qmax=127)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    c = orc.decode_nmsq_wide(bg, Z, llr, 8, early_term=True, alpha=0.875, beta=3.0, qmax=32767)
    assert (c[0][1:] == a[0][1:]).all() and (c[1][1:] == a[1][1:]).all()
