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
BOUND_DB_1E2 = 0.10  # stated bound at BLER 1e-2, equal iteration caps (headline, cfg3 R = 1/3)


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

# name, bg, Z, K' (payload + CRC bits), E (transmitted bits, rv0), active layers, iteration cap, Es/N0 grid, blocks
CASES = [
    ("cfg1 BG2 A=100 R=1/3 QPSK 10it", 2, 20, 116, 300, 12, 10, [0.0, 0.5, 1.0, 1.5, 2.0], 4096),
    ("cfg2 headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.6, -1.5, -1.4, -1.3, -1.2], 1024),
    ("cfg3 BG2 Z=384 R=1/5 25it", 2, 384, 3840, 19120, 42, 25, [-4.2, -4.1, -4.0, -3.9, -3.8], 512),
    ("cfg3 BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.5, -1.4, -1.3, -1.2, -1.1], 512),
    ("cfg3 BG2 Z=384 R=2/3 25it", 2, 384, 3840, 5736, 7, 25, [2.6, 2.8, 3.0, 3.2, 3.4], 512),
    ("cfg5 BG1 Z=384 R=8/9 25it", 1, 384, 8448, 9478, 5, 25, [5.8, 6.0, 6.2, 6.4, 6.6], 512),
]


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
    rows, cols, kb = BG_DIMS[bg]
    K = kb * Z
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float32)  # alpha = 0: library rule
    assert (codec.alpha, codec.beta) == pkg.default_rule(bg, nl)
    info = rng.integers(0, 2, (nblk, K), dtype=np.uint8)
    info[:, Kp:] = 0
    cw = codec.encode(info)
    noise = rng.standard_normal(cw.shape)
    b_gpu, b_bp, it_gpu, it_bp = [], [], [], []
    for snr in snrs:
        mu = 2 * 10 ** (snr / 10)                   # QPSK, N0 = 10^(-EsN0/10)  (plot_BLER_vs_SNR.m:105-106)
        llr = (1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * noise
        llr[:, : 2 * Z] = 0                         # punctured systematic columns (NRLDPCDecoder.m:262)
        llr[:, 2 * Z + E + (K - Kp):] = 0           # beyond the E transmitted non-filler bits (k0 = 0)
        llr[:, Kp:K] = np.inf                       # fillers (NRLDPCDecoder.m:264)
        hg, ig = codec.decode(llr.astype(np.float32), want_iters=True)
        hb, ib = orc.decode_bp_flood(bg, Z, llr, iters, n_layers=nl, nthreads=_threads())
        b_gpu.append(float((hg[:, :Kp] != info[:, :Kp]).any(1).mean()))
        b_bp.append(float((hb[:, :Kp] != info[:, :Kp]).any(1).mean()))
        it_gpu.append(float(ig.mean())); it_bp.append(float(ib.mean()))
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


# name (as in CASES), bg, Z, K', E, layers, iteration cap, grid at equal caps, grid of the 50-sweep sum-product reference, blocks
CASES_1E2 = [
    # (three grid points each, bracketing the crossings found with five -- profiles/r03_bler_gap.json of the first run: the
    # sum-product oracle on 4096 blocks is what the GPU suite's wall time is made of)
    ("cfg2 headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.35, -1.30, -1.25], [-1.65, -1.60, -1.55], 4096),
    ("cfg3 BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.30, -1.20, -1.10], [-1.60, -1.50, -1.40], 4096),
]


@pytest.mark.parametrize("case", CASES_1E2, ids=[c[0] for c in CASES_1E2])
def test_db_gap_at_bler_1e2(pkg, orc, case):
    name, bg, Z, Kp, E, nl, iters, snrs, snrs50, nblk = case
    rows, cols, kb = BG_DIMS[bg]
    K = kb * Z
    rng = np.random.default_rng(zlib.crc32((name + " 1e-2").encode()))
    codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float32)
    info = rng.integers(0, 2, (nblk, K), dtype=np.uint8)
    info[:, Kp:] = 0
    cw = codec.encode(info)
    noise = rng.standard_normal(cw.shape).astype(np.float32)
    nth = _threads()

    def llr_at(snr):
        mu = 2 * 10 ** (snr / 10)
        llr = ((1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * noise).astype(np.float64)
        llr[:, : 2 * Z] = 0
        llr[:, 2 * Z + E + (K - Kp):] = 0
        llr[:, Kp:K] = np.inf
        return llr
    b_gpu, b_bp, b_bp50 = [], [], []
    for snr in snrs:
        llr = llr_at(snr)
        hg = codec.decode(llr.astype(np.float32))
        hb, _ = orc.decode_bp_flood(bg, Z, llr, iters, n_layers=nl, nthreads=nth)
        b_gpu.append(float((hg[:, :Kp] != info[:, :Kp]).any(1).mean()))
        b_bp.append(float((hb[:, :Kp] != info[:, :Kp]).any(1).mean()))
    for snr in snrs50:  # the reference's default iteration count (NRLDPCDecoder.m:41)
        hb, _ = orc.decode_bp_flood(bg, Z, llr_at(snr), 50, n_layers=nl, nthreads=nth)
        b_bp50.append(float((hb[:, :Kp] != info[:, :Kp]).any(1).mean()))
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
