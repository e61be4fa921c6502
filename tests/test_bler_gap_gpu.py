"""'BLER match vs MATLAB ref' as a tested number (BASELINE.json metric, second half).

The reference's decoder is comm.LDPCDecoder = flooding sum-product with a parity-check stop (NRLDPCDecoder.m:120);
its arithmetic is closed source, so its stand-in is oracle/orc_decode_bp_flood (cross-checked against an independent
numpy restatement in tests/test_oracle.py).  For every BASELINE configuration the GPU decoder -- created with
cfg.alpha = 0, i.e. the rule the C ABI itself picks, what a MEX gateway would get -- and the sum-product oracle decode
IDENTICAL noise realisations at the same iteration cap over an Es/N0 grid; the test asserts

    gap = EsN0(GPU decoder reaches BLER 0.1) - EsN0(sum-product reaches BLER 0.1)  <=  BOUND_DB

with both crossings interpolated on log10(BLER).  Round 1 measured +0.20 dB at the headline code with plain
normalised min-sum; the offset rule of round 2 brings it to about +0.03 dB.  Results go to gpurun_out/bler_gap.json
(DESIGN.md section 6 is regenerated from that file).

Round 3: where the curve is steep.  For the headline code and cfg3 R = 1/3 a second crossing at
BLER 1e-2 on 4096 blocks with its own stated bound (BOUND_DB_1E2), and -- recorded, NOT bounded -- the gap to the
reference's DEFAULT of 50 sum-product sweeps (NRLDPCDecoder.m:41) at that BLER.  PARITY STATUS: all of this compares with a
restatement of the reference's documented algorithm; the reference's own arithmetic is closed source (parity unpinned).
"""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import BG_DIMS, rule_kw

pytestmark = pytest.mark.gpu

BOUND_DB = 0.10  # stated bound at equal iteration caps, every BASELINE configuration
TARGET = 0.1
BOUND_DB_1E2 = 0.10  # stated bound at BLER 1e-2, equal iteration caps (headline, cfg3 R = 1/3, cfg5 R = 8/9)
BOUND_DB_1E3 = 0.10  # stated bound at BLER 1e-3 (the reference sweep's stopping point, plot_BLER_vs_SNR.m:38), equal caps: headline, cfg3 R = 1/3


def _threads():
    """CPU threads for the sum-product oracle: the cgroup quota, not the 256 CPUs a GPU box shows."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return n

import bler_cases as BC  # noqa: E402  (the cases, their seeded inputs and the committed sum-product outcomes)
from bler_cases import CASES, CASES_1E2, CASES_1E3, CASES_50  # noqa: E402

REF = BC.Ref(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bler_ref.npz")
             if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bler_ref.npz")) else None)


def sum_product(orc, key, inp, llr, bg, Z, nl, cap):
    """Block errors and mean sweeps of the reference-semantics decoder on these LLRs: the committed outcome of
    tests/golden/make_bler_ref.py when it was computed from exactly these LLRs (tests/test_bler_ref.py re-computes slices of it
    on the CPU suite), else orc_decode_bp_flood here and now."""
    got = REF.get(key, llr, inp.Kp, inp.info, llr.shape[0])
    if got is not None:
        return got
    hb, ib = orc.decode_bp_flood(bg, Z, llr, cap, n_layers=nl, nthreads=_threads())
    return (hb[:, :inp.Kp] != inp.info[:, :inp.Kp]).any(1), float(ib.mean())


def crossing(snrs, blers, nblk, target=TARGET):
    """Es/N0 where the curve crosses `target`, linear in log10(BLER) between the bracketing grid points."""
    lb = [np.log10(max(b, 0.5 / nblk)) for b in blers]
    lt = np.log10(target)
    for i in range(1, len(snrs)):
        if lb[i - 1] > lt >= lb[i]:
            return snrs[i - 1] + (snrs[i] - snrs[i - 1]) * (lb[i - 1] - lt) / (lb[i - 1] - lb[i])
    return None


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_db_gap_to_flooding_sum_product(pkg, orc, case):
    name, bg, Z, Kp, E, nl, iters, snrs, nblk = case
    codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float32)  # alpha = 0: library rule
    assert (codec.alpha, codec.beta) == pkg.default_rule(bg, nl)
    inp = BC.inputs_gap(case, orc.encode)
    info = inp.info
    assert (codec.encode(info[:16]) == inp.cw[:16]).all()
    b_gpu, b_bp, it_gpu, it_bp = [], [], [], []
    for snr in snrs:
        llr = inp.llr_at(snr)
        hg, ig = codec.decode(llr.astype(np.float32), want_iters=True)
        eb, sb = sum_product(orc, "gap/%s/%g" % (name, snr), inp, llr, bg, Z, nl, iters)
        b_gpu.append(float((hg[:, :Kp] != info[:, :Kp]).any(1).mean()))
        b_bp.append(float(eb.mean()))
        it_gpu.append(float(ig.mean())); it_bp.append(sb)
        if snr == snrs[len(snrs) // 2]:             # and the GPU result is the oracle's, bit for bit, in the waterfall
            ho, io = orc.decode_nmsq(bg, Z, llr[:8].astype(np.float32).astype(np.float64), iters, n_layers=nl,
                                     early_term=True, **rule_kw(codec))
            assert (hg[:8] == ho).all() and (ig[:8] == io).all()
    codec.close()
    x_gpu, x_bp = crossing(snrs, b_gpu, nblk), crossing(snrs, b_bp, nblk)
    rec = {"case": name, "blocks": nblk, "iterations": iters, "alpha": codec.alpha, "beta_llr": codec.beta, "EsN0_dB": snrs,
           "bler_gpu": b_gpu, "bler_sum_product": b_bp, "mean_iters_gpu": it_gpu, "mean_sweeps_sum_product": it_bp,
           "EsN0_at_bler_0.1_gpu": x_gpu, "EsN0_at_bler_0.1_sum_product": x_bp,
           "gap_dB": None if x_gpu is None or x_bp is None else x_gpu - x_bp, "bound_dB": BOUND_DB}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        p = os.path.join(out, "bler_gap.json")
        allr = json.load(open(p)) if os.path.exists(p) else {}
        allr[name] = rec
        json.dump(allr, open(p, "w"), indent=1)
    except OSError:
        pass
    print(rec)
    assert x_gpu is not None and x_bp is not None, "grid does not bracket BLER 0.1: %s %s" % (b_gpu, b_bp)
    assert x_gpu - x_bp <= BOUND_DB, rec
    assert all(b_gpu[i] >= b_gpu[i + 1] - 0.02 for i in range(len(snrs) - 1))  # monotone up to sampling noise


@pytest.mark.parametrize("case", CASES_1E2, ids=[c[0] for c in CASES_1E2])
def test_db_gap_at_bler_1e2(pkg, orc, case):
    name, bg, Z, Kp, E, nl, iters, snrs, snrs50, nblk = case
    codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float32)
    inp = BC.inputs_1e2(case, orc.encode)
    info = inp.info
    b_gpu, b_bp, b_bp50 = [], [], []
    for snr in snrs:
        llr = inp.llr_at(snr)
        hg = codec.decode(llr.astype(np.float32))
        b_gpu.append(float((hg[:, :Kp] != info[:, :Kp]).any(1).mean()))
        b_bp.append(float(sum_product(orc, "1e2/%s/%g" % (name, snr), inp, llr, bg, Z, nl, iters)[0].mean()))
    for snr in snrs50:  # the reference's default iteration count (NRLDPCDecoder.m:41)
        b_bp50.append(float(sum_product(orc, "1e2_50/%s/%g" % (name, snr), inp, inp.llr_at(snr), bg, Z, nl, 50)[0].mean()))
    codec.close()
    x_gpu, x_bp, x_bp50 = crossing(snrs, b_gpu, nblk, 1e-2), crossing(snrs, b_bp, nblk, 1e-2), crossing(snrs50, b_bp50, nblk, 1e-2)
    extra = {"blocks_at_bler_0.01": nblk, "EsN0_dB_at_bler_0.01_grid": snrs, "bler_gpu_fine": b_gpu, "bler_sum_product_fine": b_bp,
             "EsN0_at_bler_0.01_gpu": x_gpu, "EsN0_at_bler_0.01_sum_product": x_bp,
             "gap_dB_at_bler_0.01": None if x_gpu is None or x_bp is None else x_gpu - x_bp, "bound_dB_at_bler_0.01": BOUND_DB_1E2,
             "EsN0_dB_grid_50_sweeps": snrs50, "bler_sum_product_50_sweeps": b_bp50, "EsN0_at_bler_0.01_sum_product_50_sweeps": x_bp50,
             "gap_dB_vs_50_sum_product_sweeps_at_bler_0.01": None if x_gpu is None or x_bp50 is None else x_gpu - x_bp50,
             "note_50_sweeps": "recorded, not bounded: %d layered min-sum iterations against the reference's default of 50 "
                               "flooding sum-product sweeps (NRLDPCDecoder.m:41)" % iters}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        p = os.path.join(out, "bler_gap.json")
        allr = json.load(open(p)) if os.path.exists(p) else {}
        allr.setdefault(name, {"case": name}).update(extra)
        json.dump(allr, open(p, "w"), indent=1)
    except OSError:
        pass
    print(extra)
    assert x_gpu is not None and x_bp is not None, "grid does not bracket BLER 1e-2: %s %s" % (b_gpu, b_bp)
    assert x_gpu - x_bp <= BOUND_DB_1E2, extra


@pytest.mark.parametrize("case", CASES_1E3, ids=[c[0] for c in CASES_1E3])
def test_db_gap_at_bler_1e3(pkg, orc, case):
    """Where the reference's sweep stops -- target_BLER = 1e-3 (plot_BLER_vs_SNR.m:38) -- on 16384 blocks with identical payloads and
    noise: Es/N0 at which the GPU decoder and flooding sum-product (committed outcomes, tests/golden/bler_ref.npz) cross BLER 1e-3 at
    equal iteration caps.  Bounded by BOUND_DB_1E3."""
    name, bg, Z, Kp, E, nl, iters, snrs, snrs_bp, nblk = case
    codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float32)
    inp = BC.inputs_1e3(case, orc.encode)
    info = inp.info
    b_gpu, b_bp = [], []
    for snr in snrs:
        hg = codec.decode(inp.llr_at(snr).astype(np.float32))
        b_gpu.append(float((hg[:, :Kp] != info[:, :Kp]).any(1).mean()))
    for snr in snrs_bp:  # (its own grid: the two waterfalls need not overlap)
        b_bp.append(float(sum_product(orc, "1e3/%s/%g" % (name, snr), inp, inp.llr_at(snr), bg, Z, nl, iters)[0].mean()))
    codec.close()
    x_gpu, x_bp = crossing(snrs, b_gpu, nblk, 1e-3), crossing(snrs_bp, b_bp, nblk, 1e-3)
    extra = {"blocks_at_bler_0.001": nblk, "EsN0_dB_at_bler_0.001_grid_gpu": snrs, "EsN0_dB_at_bler_0.001_grid_sum_product": snrs_bp,
             "bler_gpu_at_0.001": b_gpu, "bler_sum_product_at_0.001": b_bp, "EsN0_at_bler_0.001_gpu": x_gpu, "EsN0_at_bler_0.001_sum_product": x_bp,
             "gap_dB_at_bler_0.001": None if x_gpu is None or x_bp is None else x_gpu - x_bp, "bound_dB_at_bler_0.001": BOUND_DB_1E3}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        p = os.path.join(out, "bler_gap.json")
        allr = json.load(open(p)) if os.path.exists(p) else {}
        allr.setdefault(name, {"case": name}).update(extra)
        json.dump(allr, open(p, "w"), indent=1)
    except OSError:
        pass
    print(extra)
    assert x_gpu is not None and x_bp is not None, "grids do not bracket BLER 1e-3: %s %s" % (b_gpu, b_bp)
    assert x_gpu - x_bp <= BOUND_DB_1E3, extra


# ---- round 4: the reference's own operating points ----------------------------------------------------
def _record(name, extra):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        p = os.path.join(out, "bler_gap.json")
        allr = json.load(open(p)) if os.path.exists(p) else {}
        allr.setdefault(name, {"case": name}).update(extra)
        json.dump(allr, open(p, "w"), indent=1)
    except OSError:
        pass
    print(name, extra)


# Stated bound at the reference's default cap, 50 iterations against 50 sweeps (NRLDPCDecoder.m:41), BLER 1e-2.  Wider than at 25:
# layered min-sum has converged by then (25 -> 50 iterations moves its BLER 1e-2 point by 0.08 dB at the headline code), flooding
# sum-product has not (25 -> 50 sweeps: 0.25 dB), so at 50 the gap is the min-sum approximation itself.
BOUND_DB_50 = 0.25
@pytest.mark.parametrize("case", CASES_50, ids=[c[0] for c in CASES_50])
def test_db_gap_at_the_reference_default_of_50_iterations(pkg, orc, case):
    """Equal caps at the reference's DEFAULT `iterations = 50` (NRLDPCDecoder.m:41): 50 layered offset-min-sum iterations on
    the GPU against 50 flooding sum-product sweeps, identical noise, crossing of BLER 1e-2 on 4096 blocks."""
    name, bg, Z, Kp, E, nl, snrs, snrs_bp, nblk = case
    codec = pkg.Codec(bg, Z, max_iter=50, n_layers=nl, early_term=True, llr_dtype=np.float32)
    inp = BC.inputs_50(case, orc.encode)
    info = inp.info
    b_gpu, it_gpu, b_bp, it_bp = [], [], [], []
    for snr in snrs:
        hg, ig = codec.decode(inp.llr_at(snr).astype(np.float32), want_iters=True)
        b_gpu.append(float((hg[:, :Kp] != info[:, :Kp]).any(1).mean())); it_gpu.append(float(ig.mean()))
    for snr in snrs_bp:
        eb, sb = sum_product(orc, "50/%s/%g" % (name, snr), inp, inp.llr_at(snr), bg, Z, nl, 50)
        b_bp.append(float(eb.mean())); it_bp.append(sb)
    codec.close()
    x_gpu, x_bp = crossing(snrs, b_gpu, nblk, 1e-2), crossing(snrs_bp, b_bp, nblk, 1e-2)
    rec = {"blocks": nblk, "iterations": 50, "EsN0_dB_gpu": snrs, "bler_gpu": b_gpu, "mean_iters_gpu": it_gpu,
           "EsN0_dB_sum_product": snrs_bp, "bler_sum_product": b_bp, "mean_sweeps_sum_product": it_bp,
           "EsN0_at_bler_0.01_gpu": x_gpu, "EsN0_at_bler_0.01_sum_product": x_bp,
           "gap_dB_at_bler_0.01": None if x_gpu is None or x_bp is None else x_gpu - x_bp, "bound_dB": BOUND_DB_50}
    _record(name, rec)
    assert x_gpu is not None and x_bp is not None, "grids do not bracket BLER 1e-2: %s %s" % (b_gpu, b_bp)
    assert x_gpu - x_bp <= BOUND_DB_50, rec


def _reference_decoder_class(pkg, orc):
    """The reference's receiver exactly as it is built: the System object's chain (NRLDPCDecoder.m:133-140) around
    comm.LDPCDecoder(H, iterations, 'Parity check satisfied') (:120, :265) -- here the product's host-side mirror of that chain
    with its stage 4 handed to the sum-product oracle on the full parity-check matrix (test infrastructure; no GPU involved)."""
    class ReferenceDecoder(pkg.NRLDPCDecoder):
        def LDPC_coding(self, d_tilde):  # NRLDPCDecoder.m:245-268
            Z, K_ = self.Z_c, self.K
            nb, C_ = d_tilde.shape[0], self.C
            cw = np.concatenate([np.zeros((nb, C_, 2 * Z)), d_tilde], axis=2)  # :262
            filler = np.isnan(cw[0, 0, :K_])
            cw[np.isnan(cw)] = np.inf  # :264
            hard, iters = orc.decode_bp_flood(self.BG, Z, cw.reshape(nb * C_, -1), self._setup_iterations, nthreads=_threads())  # :265
            self.last_iterations = iters.reshape(nb, C_)
            c_hat = hard.reshape(nb, C_, K_).astype(np.float64)
            c_hat[:, :, filler] = np.nan  # :266
            return c_hat
    return ReferenceDecoder


BOUND_DB_DEMO = 0.10
# name, plot_BLER_vs_SNR.m arguments (A, R, BG, Modulation, rv_id_sequence, iterations), Es/N0 grid, transport blocks
CASES_DEMO = [
    # the script's own defaults (plot_BLER_vs_SNR.m:29-41): two code blocks of BG2 Z = 208, 8 iterations
    ("demo A=3842 BG2 R=1/3 QPSK 8it rv[0]", 3842, 1 / 3, 2, "QPSK", (0,), 8, [-1.0, -0.5, 0.0, 0.5, 1.0], 512),
    # the same point sent at R = 2/3 with up to four HARQ transmissions (:38's rv_id_sequence option, I_HARQ = 1 as :99)
    ("demo A=3842 BG2 R=2/3 QPSK 8it rv[0 2 3 1]", 3842, 2 / 3, 2, "QPSK", (0, 2, 3, 1), 8, [-4.75, -4.25, -3.75, -3.25, -2.75, -2.25], 512),
    # one higher-order point: exact LLRs of the 64QAM demapper (NRDemodulator.m:80)
    ("demo A=3842 BG2 R=1/2 64QAM 8it rv[0]", 3842, 1 / 2, 2, "64QAM", (0,), 8, [10.75, 11.25, 11.75, 12.25, 12.75, 13.25, 13.75, 14.25], 512),
]


@pytest.mark.parametrize("case", CASES_DEMO, ids=[c[0] for c in CASES_DEMO])
def test_db_gap_through_the_harness_at_the_reference_defaults(pkg, orc, case):
    """plot_BLER_vs_SNR.m's own loop body (harness.simulate_point = :118-137: payload, NRLDPCEncoder, modulation, AWGN, exact
    LLRs, NRLDPCDecoder with I_HARQ = 1 and the HARQ loop over rv_id_sequence) run twice on identical payloads and noise: with
    the product's decoder object (GPU) and with the reference's receiver (same chain, sum-product core), both at the script's
    `iterations = 8`.  Gap of the BLER 0.1 crossings."""
    name, A, R, BG, mod, rvs, iters, snrs, nblk = case
    import importlib
    H = importlib.import_module(pkg.__name__ + ".harness")
    Q_m = H.Q_M[mod]
    G = int(round(A / R / Q_m) * Q_m)  # plot_BLER_vs_SNR.m:94
    Ref = _reference_decoder_class(pkg, orc)
    blers = {}
    for tag, cls in (("gpu", pkg.NRLDPCDecoder), ("ref", Ref)):
        hEnc = pkg.NRLDPCEncoder(A=A, BG=BG, G=G, Q_m=Q_m)
        hDec = cls(A=A, BG=BG, G=G, Q_m=Q_m, I_HARQ=1, iterations=iters)
        out = []
        for snr in snrs:
            rng = np.random.default_rng(zlib.crc32((name + str(snr)).encode()))  # the same payloads and noise for both receivers
            ok = H.simulate_point(hEnc, hDec, Q_m, snr, rvs, nblk, rng)
            out.append(float(1.0 - ok.mean()))
        hEnc.release(); hDec.release()
        blers[tag] = out
    x_gpu, x_ref = crossing(snrs, blers["gpu"], nblk), crossing(snrs, blers["ref"], nblk)
    rec = {"blocks": nblk, "iterations": iters, "A": A, "R": R, "BG": BG, "Modulation": mod, "rv_id_sequence": list(rvs), "G": G,
           "EsN0_dB": snrs, "bler_gpu": blers["gpu"], "bler_reference_chain_sum_product": blers["ref"],
           "EsN0_at_bler_0.1_gpu": x_gpu, "EsN0_at_bler_0.1_reference": x_ref,
           "gap_dB": None if x_gpu is None or x_ref is None else x_gpu - x_ref, "bound_dB": BOUND_DB_DEMO}
    _record(name, rec)
    assert x_gpu is not None and x_ref is not None, "grid does not bracket BLER 0.1: %s" % blers
    assert x_gpu - x_ref <= BOUND_DB_DEMO, rec


def test_cost_of_the_8_bit_grid(pkg, orc):
    """What the int8 grid costs: the SAME layered offset-min-sum rule on a wide grid -- channel values and
    messages saturating at +/-32767 grid units instead of +/-127, i.e. no +/-15.9 LLR ingest clamp and 16-bit messages
    (oracle/orc_decode_onmsq_wide; no kernel computes it) -- against the product on identical noise, headline code, 25
    iterations, crossing of BLER 1e-2 on 4096 blocks.  Recorded and bounded: the 8-bit grid may cost at most 0.05 dB."""
    name, bg, Z, Kp, E, nl, iters = "cfg2 headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25
    snrs, nblk = [-1.40, -1.35, -1.30, -1.25, -1.20], 4096
    K = 22 * Z
    rng = np.random.default_rng(zlib.crc32(b"8-bit grid"))
    codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float32)
    info = rng.integers(0, 2, (nblk, K), dtype=np.uint8)
    cw = codec.encode(info)
    noise = rng.standard_normal(cw.shape).astype(np.float32)
    b8, bw = [], []
    for snr in snrs:
        mu = 2 * 10 ** (snr / 10)
        llr = ((1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * noise).astype(np.float32)
        llr[:, : 2 * Z] = 0
        llr[:, 2 * Z + E:] = 0
        hg = codec.decode(llr)
        hw, _ = orc.decode_nmsq_wide(bg, Z, llr.astype(np.float64), iters, n_layers=nl, early_term=True, **rule_kw(codec))
        b8.append(float((hg != info).any(1).mean())); bw.append(float((hw != info).any(1).mean()))
    codec.close()
    x8, xw = crossing(snrs, b8, nblk, 1e-2), crossing(snrs, bw, nblk, 1e-2)
    rec = {"grid_cost": {"blocks": nblk, "EsN0_dB": snrs, "bler_int8_grid_gpu": b8, "bler_wide_grid_oracle": bw,
                         "EsN0_at_bler_0.01_int8": x8, "EsN0_at_bler_0.01_wide": xw,
                         "cost_dB_of_the_8_bit_grid_at_bler_0.01": None if x8 is None or xw is None else x8 - xw,
                         "note": "same rule (alpha, beta), same noise; wide = values and messages saturate at 32767 grid units"}}
    _record(name, rec)
    assert x8 is not None and xw is not None, (b8, bw)
    assert x8 - xw <= 0.05, rec
