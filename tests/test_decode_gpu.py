"""GPU parity tests proper: the HIP decoder behind the C ABI vs the CPU oracle, bit-exact
(integer algorithm: hard bits, iteration counts and soft outputs must all be identical)."""
import os

import numpy as np
import pytest

from conftest import ALL_Z, BG_DIMS, awgn_llr, rule_kw

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_case(pkg, orc, rng, bg, Z, B, esn0, iters, nl=0, et=True, dt=np.float16, alpha=None, scale=8, app=True, beta=0.0):
    """alpha None: the library's own rule for (bg, nl) (cfg.alpha = 0); beta in LLR units as in nrldpc_cfg."""
    kb = BG_DIMS[bg][2]
    info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, esn0, dt, Z)
    c = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=et, alpha=alpha or 0.0, beta=beta, llr_scale=scale,
                  llr_dtype=dt)
    try:
        out = c.decode(llr, want_iters=True, want_app=app)
    finally:
        c.close()
    if alpha is not None:  # the offset is kept on a grid of half fixed-point units (nrldpc_cfg.beta)
        assert c.alpha == np.float32(alpha) and c.beta == np.float32(np.rint(2 * np.float32(beta) * scale) / (2 * scale))
    ref = orc.decode_nmsq(bg, Z, llr.astype(np.float64), iters, n_layers=nl, early_term=et, scale=scale, want_app=app,
                          **rule_kw(c, scale))
    assert (out[0] == ref[0]).all(), "hard decisions differ"
    assert (out[1] == ref[1]).all(), "iteration counts differ"
    if app:
        assert (out[2] == ref[2]).all(), "soft outputs differ"
    return out, info


def test_loaded_library_is_the_in_tree_hip_build(pkg):
    path = pkg._capi.lib_path()
    assert path.endswith("libnrldpc_hip.so") and os.path.exists(path)
    maps = open("/proc/self/maps").read()
    pkg.load()
    assert "libnrldpc_hip.so" in open("/proc/self/maps").read() or "libnrldpc_hip.so" in maps


@pytest.mark.parametrize("bg", [1, 2])
def test_every_lifting_size(pkg, orc, bg):
    """All 51 Z of Table 5.3.2-1, both base graphs, ragged batches (not a multiple of codewords/workgroup)."""
    rng = np.random.default_rng(1000 + bg)
    for Z in ALL_Z:
        B = 1 + int(rng.integers(0, 5)) + (768 // Z if Z < 64 else 0)
        run_case(pkg, orc, rng, bg, Z, B, 2.0, 6, et=bool(Z & 2), dt=np.float32 if Z % 3 else np.float16)


@pytest.mark.parametrize("bg", [1, 2])
def test_hard_output_kernels_iteration_by_iteration(pkg, orc, bg):
    """The kernels a hard-output call runs (pipelined row form, split form, packed geometry -- soft output, as in the test above,
    is always the general kernel's) for every lifting size, at an SNR where the decoder does NOT converge in the iterations
    given: after 1, 2 and 3 iterations every hard decision must equal the oracle's, so one stale LDS word shows (at a
    comfortable SNR later iterations repair it and the final decisions agree anyway -- how a wrong exec mask in the packed
    kernels' twin writes once passed the test above).  Then the parity-check stop: decisions and iteration counts."""
    rng = np.random.default_rng(4000 + bg)
    for Z in ALL_Z:
        B = 2 + int(rng.integers(0, 4)) + (400 // Z if Z < 64 else 0)
        for iters in (1, 2, 3):
            run_case(pkg, orc, rng, bg, Z, B, -2.0 if bg == 1 else -3.0, iters, et=False, app=False,
                     dt=np.float16 if (Z + iters) % 2 else np.float32)
        run_case(pkg, orc, rng, bg, Z, B, 0.3 if bg == 1 else -0.5, 12, et=True, app=False)
    if bg == 2:  # pruned layer counts with pipelined builds of their own: the reference's default point (Z = 208, 21 rows) ...
        for iters in (1, 2, 3):
            run_case(pkg, orc, rng, 2, 208, 5, -1.0, iters, nl=21, et=False, app=False)
        run_case(pkg, orc, rng, 2, 208, 5, 1.0, 12, nl=21, et=True, app=False)
    if bg == 2:  # ... and NRLDPC_Z64P_NL_LIST (BASELINE configs[0]: Z = 20, 12 rows), packed geometry
        for iters in (1, 2, 3):
            run_case(pkg, orc, rng, 2, 20, 37, 1.0, iters, nl=12, et=False, app=False)
        run_case(pkg, orc, rng, 2, 20, 37, 3.0, 12, nl=12, et=True, app=False)


@pytest.mark.parametrize("bg,Z,nl,esn0", [(1, 384, 46, -0.8), (1, 384, 5, 6.2), (1, 384, 20, 1.0), (2, 384, 42, -1.0),
                                          (2, 384, 7, 4.5), (2, 384, 22, 0.5), (1, 352, 30, 0.0), (2, 208, 21, 1.0),
                                          (1, 64, 4, 8.0), (2, 20, 12, 2.0),
                                          # every count of NRLDPC_Z64_NL_LIST: pipelined kernels of their own
                                          (1, 384, 13, 3.0), (1, 384, 24, 1.0), (2, 384, 32, -2.8), (2, 384, 17, 0.0),
                                          (2, 384, 12, 1.0), (2, 384, 9, 2.2),
                                          # neighbours of those counts: the general (run-time layer count) kernel
                                          (1, 384, 6, 6.0), (1, 384, 12, 3.0), (2, 384, 8, 3.5), (2, 384, 23, -1.0)])
def test_rate_pruned_layers(pkg, orc, bg, Z, nl, esn0):
    rng = np.random.default_rng(Z * 100 + nl)
    run_case(pkg, orc, rng, bg, Z, 5, esn0, 12, nl=nl, et=True)
    run_case(pkg, orc, rng, bg, Z, 3, esn0, 7, nl=nl, et=False)
    run_case(pkg, orc, rng, bg, Z, 4, esn0, 9, nl=nl, et=False, app=False)  # the fixed-iteration throughput build
    run_case(pkg, orc, rng, bg, Z, 5, esn0, 14, nl=nl, et=True, app=False)  # ... and its early-termination twin


def _waterfall_esn0(bg, nl):
    """Rough Es/N0 (dB) at which (bg, nl active rows) converges within a dozen iterations for about half the blocks."""
    pts = {1: [(4, 8.0), (5, 6.2), (8, 4.6), (13, 3.0), (17, 2.0), (24, 1.0), (30, 0.2), (46, -0.8)],
           2: [(4, 7.0), (7, 4.5), (9, 2.6), (12, 1.3), (17, 0.0), (22, -0.7), (32, -1.8), (42, -2.5)]}[bg]
    return float(np.interp(nl, [p[0] for p in pts], [p[1] for p in pts]))


# every kernel family a hard-output call with pruned rows can land on: split form (block geometry; 384 and 288 with dual rows),
# one-thread-per-row pipelined form (144, 192, 320, 352; BG2 384), packed geometry (8 ... 80), run-time-Z kernel (the rest)
RT_GRID_Z = (8, 20, 32, 56, 64, 80, 96, 128, 144, 192, 208, 256, 288, 320, 352, 384)
RT_GRID_NL = (4, 5, 8, 13, 17, 24, 30)


@pytest.mark.parametrize("bg", [1, 2])
@pytest.mark.parametrize("Z", RT_GRID_Z)
def test_run_time_layer_count_grid(pkg, orc, bg, Z):
    """Pruned layer counts that have no build of their own run the pipelined / split / packed kernels with the count as a
    run-time prefix of the all-rows tables (NL_RT, nrldpc_device.h): after 1, 2 and 3 iterations at an SNR where nothing
    converges (one stale copy of a ring word shows), then 25 fixed iterations and the parity-check stop in the waterfall."""
    rng = np.random.default_rng(7000 + 100 * bg + Z)
    rows = BG_DIMS[bg][0]
    B = 3 + (300 // Z if Z < 64 else 0)
    for nl in RT_GRID_NL + (rows - 1,):
        w = _waterfall_esn0(bg, nl)
        for iters in (1, 2, 3):
            run_case(pkg, orc, rng, bg, Z, B, w - 2.5, iters, nl=nl, et=False, app=False, dt=np.float16 if (nl + iters) % 2 else np.float32)
        run_case(pkg, orc, rng, bg, Z, B, w, 25, nl=nl, et=False, app=False)
        run_case(pkg, orc, rng, bg, Z, B + 2, w + 0.2, 14, nl=nl, et=True, app=False)


@pytest.mark.parametrize("bg", [1, 2])
def test_interleaved_block_geometry(pkg, orc, bg):
    """NRLDPC_Z64I_LIST: NCW codewords of a small lifting size Zr interleaved (ring position z * NCW + c) in one workgroup of the
    block geometry of the virtual size Zr * NCW.  Every entry: a batch that leaves the last workgroup one codeword (the other
    lanes decode zeros) and one smaller than a workgroup; every row active and two pruned layer counts (the run-time-prefix
    builds); after 1, 2, 3 iterations where nothing converges, 25 fixed iterations, and the parity stop in the waterfall for
    the modes the entry serves (the others run the kernels of the tests above)."""
    import importlib
    bld = importlib.import_module("ldpc-3gpp-matlab_amd.build")
    rng = np.random.default_rng(9100 + bg)
    rows = BG_DIMS[bg][0]
    entries = [(z, n, m) for b, z, n, m in bld.Z64I if b == bg]
    assert entries
    for Z, ncw, mode in entries:
        for nl in (0, 7 if bg == 2 else 13, rows - 3):
            w = _waterfall_esn0(bg, nl or rows)
            for B in (ncw + 1, max(1, ncw - 1)):
                if mode & 1:
                    for iters in (1, 2, 3):
                        run_case(pkg, orc, rng, bg, Z, B, w - 2.5, iters, nl=nl, et=False, app=False,
                                 dt=np.float16 if (Z + iters) % 2 else np.float32)
                if mode & (2 if nl == 0 else 4):
                    run_case(pkg, orc, rng, bg, Z, B, w + 0.2, 14, nl=nl, et=True, app=False)
            if mode & 1:
                run_case(pkg, orc, rng, bg, Z, 2 * ncw + 1, w, 25, nl=nl, et=False, app=False)


@pytest.mark.parametrize("bg", [1, 2])
def test_z384_kernel_variants(pkg, orc, bg):
    """The compile-time Z=384 kernel has three builds (plain fixed-iteration, full-H with early
    termination / soft output, pruned layers); each against the oracle, with odd batch sizes so that a
    workgroup holds one live and one dead codeword."""
    rng = np.random.default_rng(384 + bg)
    rows = BG_DIMS[bg][0]
    for B in (1, 5):
        run_case(pkg, orc, rng, bg, 384, B, 0.0, 6, nl=0, et=False, app=False)       # PLAIN
        run_case(pkg, orc, rng, bg, 384, B, 0.0, 12, nl=0, et=True, app=False)       # FULL + early termination
        run_case(pkg, orc, rng, bg, 384, B, 0.0, 5, nl=0, et=False, app=True)        # FULL + soft output
        run_case(pkg, orc, rng, bg, 384, B, 2.0, 9, nl=rows - 1, et=True, app=True)  # pruned (cuts a barrier group)
        run_case(pkg, orc, rng, bg, 384, B, 3.0, 7, nl=17, et=False, app=False)      # pruned, fixed iterations


@pytest.mark.parametrize("bg,esn0", [(1, -1.6), (1, -1.2), (2, -1.4)])
def test_headline_kernel_full_depth(pkg, orc, bg, esn0):
    """The pipelined fixed-iteration kernel at the headline depth (25 iterations, every layer) in the waterfall,
    where many codewords never converge and a-posteriori values grow large: every codeword vs the oracle."""
    rng = np.random.default_rng(2500 + bg)
    run_case(pkg, orc, rng, bg, 384, 97, esn0, 25, nl=0, et=False, app=False)
    run_case(pkg, orc, rng, bg, 384, 33, esn0, 25, nl=0, et=True, app=True)


def test_per_iteration_soft_llrs(pkg, orc):
    """Soft a-posteriori LLRs after 1, 2, ..., 8 iterations (tolerance: none, values are k/scale exactly)."""
    rng = np.random.default_rng(77)
    for bg, Z in ((1, 384), (2, 96)):
        kb = BG_DIMS[bg][2]
        info = rng.integers(0, 2, (2, kb * Z), dtype=np.uint8)
        llr = awgn_llr(rng, orc.encode(bg, Z, info), -0.5, np.float32, Z)
        for it in range(1, 9):
            c = pkg.Codec(bg, Z, max_iter=it, early_term=False, llr_dtype=np.float32)
            h, iters, app = c.decode(llr, want_iters=True, want_app=True)
            c.close()
            ho, io, ao = orc.decode_nmsq(bg, Z, llr.astype(np.float64), it, early_term=False, want_app=True, **rule_kw(c))
            assert (app == ao).all() and (h == ho).all() and (iters == it).all()


@pytest.mark.parametrize("alpha,scale,beta", [(0.625, 8, 0.0), (0.6875, 16, 0.0), (0.8, 4, 0.0), (1.0, 8, 0.0), (0.75, 32, 0.0),
                                              (0.5, 1, 0.0), (0.875, 8, 0.375), (0.8125, 8, 0.25), (1.0, 16, 0.5),
                                              (0.9, 4, 0.3), (0.75, 8, 4.0)])
def test_alpha_beta_and_scale(pkg, orc, alpha, scale, beta):
    """Explicit check-node rules: normalisation alpha, offset beta (LLR units), grid scale."""
    rng = np.random.default_rng(int(alpha * 1000) + scale)
    run_case(pkg, orc, rng, 1, 96, 4, 0.5, 10, alpha=alpha, scale=scale, beta=beta)
    run_case(pkg, orc, rng, 2, 384, 2, 0.0, 6, alpha=alpha, scale=scale, et=False, beta=beta)
    run_case(pkg, orc, rng, 1, 384, 3, -1.0, 8, alpha=alpha, scale=scale, et=False, app=False, beta=beta)  # pipelined build


def test_default_rule_lives_in_the_c_abi(pkg):
    """cfg.alpha = 0 resolves to nrldpc_default_rule(bg, n_layers) inside nrldpc_create: a MEX gateway that passes
    no rule gets the same decoder as the Python mirror (the ABI used to fall back to alpha = 0.75)."""
    for bg, rows in ((1, 46), (2, 42)):
        for nl in (0, 4, 5, 7, 12, 13, 22, 24, 32, rows):
            a, b = pkg.default_rule(bg, nl)
            c = pkg.Codec(bg, 384, max_iter=2, n_layers=nl)
            assert (c.alpha, c.beta) == (a, b) and 0.5 <= a <= 1.0 and 0.0 <= b <= 1.0
            c.close()
    c = pkg.Codec(1, 384, max_iter=2, alpha=0.75)
    assert (c.alpha, c.beta) == (0.75, 0.0)  # an explicit alpha keeps plain normalised min-sum unless beta is given
    c.close()
    for bad in (dict(alpha=1.5), dict(alpha=-0.5), dict(alpha=0.75, beta=-1.0), dict(alpha=0.75, beta=9.0)):
        with pytest.raises(pkg.UnsupportedParameters):
            pkg.Codec(1, 384, max_iter=2, **bad)


def test_special_llr_values(pkg, orc):
    """+inf fillers (NRLDPCDecoder.m:264), -inf, NaN, zeros, saturating magnitudes, all-zero input."""
    rng = np.random.default_rng(5)
    bg, Z, kb, Kp = 2, 20, 10, 116
    info = rng.integers(0, 2, (6, kb * Z), dtype=np.uint8)
    info[:, Kp:] = 0
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, 3.0, np.float32, Z, E=300)
    llr[:, Kp: kb * Z] = np.inf
    llr[1, 50] = np.nan
    llr[2, 70] = -np.inf
    llr[2, (kb + 6) * Z + 3] = np.inf
    llr[3] *= 1000.0
    llr[4] = 0.0
    llr[5, ::3] = -0.0
    for dt in (np.float32, np.float16, np.float64):
        c = pkg.Codec(bg, Z, max_iter=10, n_layers=12, early_term=True, llr_dtype=dt)
        h, it, app = c.decode(llr.astype(dt), want_iters=True, want_app=True)
        c.close()
        ho, io, ao = orc.decode_nmsq(bg, Z, llr.astype(dt).astype(np.float64), 10, n_layers=12, early_term=True,
                                     want_app=True, **rule_kw(c))
        assert (h == ho).all() and (it == io).all() and (app == ao).all()
        assert (h[0] == info[0]).all() and it[4] == 1 and not h[4].any()


def test_randomised_configurations(pkg, orc):
    """Differential fuzz: random (BG, Z, active layers, iterations, alpha, scale, early termination, dtype, SNR,
    batch) against the oracle -- hard bits, iteration counts and soft outputs."""
    rng = np.random.default_rng(20260929)
    for _ in range(60):
        bg = int(rng.integers(1, 3))
        Z = int(rng.choice(ALL_Z)) if rng.random() < 0.7 else 384
        rows = BG_DIMS[bg][0]
        nl = 0 if rng.random() < 0.4 else int(rng.integers(4, rows + 1))
        run_case(pkg, orc, rng, bg, Z, int(rng.integers(1, 7)), float(rng.uniform(-2.0, 8.0)), int(rng.integers(1, 13)),
                 nl=nl, et=bool(rng.integers(0, 2)), dt=[np.float16, np.float32][int(rng.integers(0, 2))],
                 alpha=float(rng.choice([0.5, 0.625, 0.6875, 0.75, 0.8, 0.875, 1.0])),
                 scale=int(rng.choice([2, 4, 8, 16])), app=bool(rng.integers(0, 2)),
                 beta=float(rng.choice([0.0, 0.0, 0.125, 0.25, 0.375, 0.5, 0.7])))


def test_golden_fixture_on_gpu(pkg):
    g = np.load(os.path.join(GOLD, "nmsq_golden.npz"))
    for name in sorted(set(k.split("/")[0] for k in g.files)):
        bg, Z, nl, it, et, alpha, scale = (g[name + "/cfg"][i] for i in range(7))
        c = pkg.Codec(int(bg), int(Z), max_iter=int(it), n_layers=int(nl), early_term=bool(et), alpha=float(alpha),
                      llr_scale=int(scale), llr_dtype=np.float16)
        h, iters, app = c.decode(g[name + "/llr"], want_iters=True, want_app=True)
        cw = c.encode(g[name + "/info"])
        c.close()
        assert (np.packbits(h, axis=1) == g[name + "/hard_packed"]).all(), name
        assert (iters == g[name + "/iters"]).all(), name
        assert (app.astype(np.float16) == g[name + "/app_f16"]).all(), name
        assert (np.packbits(cw, axis=1) == g[name + "/cw_packed"]).all(), name


CRC24A, CRC24B, CRC16 = (0x1864CFB, 24), (0x1800063, 24), (0x11021, 16)  # get_3gpp_crc_polynomial.m:3-14


# bg, Z, active rows, K' (payload + CRC), generator, Es/N0 of the waterfall: split kernels with dual rows (1/384), pruned builds of
# their own (5, 22 rows), run-time layer counts, the one-thread-per-row form (BG2 384, BG1 320), merged groups (1/256), a packed
# size (-> run-time-Z kernel), retiring lanes (BG1 Z = 104: blocks of 52), a K' that ends inside a column, short CRC16 blocks
@pytest.mark.parametrize("bg,Z,nl,Kp,crc,esn0", [
    (1, 384, 0, 8448, CRC24A, -0.9), (1, 384, 5, 8448, CRC24A, 6.3), (1, 384, 17, 8000, CRC24B, 2.1), (2, 384, 22, 3840, CRC16, -0.6),
    (2, 384, 0, 3000, CRC24B, -2.4), (1, 320, 0, 7040, CRC24B, -0.8), (1, 256, 30, 5632, CRC24A, 0.3), (2, 208, 21, 1957, CRC24B, -0.2),
    (2, 20, 12, 116, CRC16, 1.8), (1, 104, 0, 2288, CRC24A, -0.6), (1, 88, 9, 1900, CRC24B, 4.2), (2, 7, 0, 70, CRC16, -1.5),
    # the one-thread-per-row form with blocks of 48 rows and FOUR codewords per workgroup -- 80 slot words cleared by
    # lanes of which 48..63 of every wave have retired; all rows and pruned
    (2, 144, 0, 1440, CRC24B, -2.4), (2, 144, 17, 1300, CRC16, 0.1)])
def test_crc_aided_stop(pkg, orc, bg, Z, nl, Kp, crc, esn0):
    """nrldpc_cfg.early_term = 2: a codeword stops when its parity checks hold OR the CRC over its first K'
    hard decisions does (and they are not all zero).  Hard decisions and iteration counts against oracle/orc_decode_onmsq_crc;
    the stop must never come later than the parity-check stop, and must come earlier for some codewords in the waterfall.
    Second half: the systematic bits punctured (rv_id 2 / 3 of a HARQ retransmission) -- all-zero hard decisions in the first
    iterations must not pass for a CRC match."""
    rng = np.random.default_rng(9000 + 7 * Z + nl)
    rows, cols, kb = BG_DIMS[bg]
    K, B = kb * Z, 24 + (200 // Z)
    poly, L = crc
    info = np.zeros((B, K), np.uint8)
    info[:, : Kp - L] = rng.integers(0, 2, (B, Kp - L), dtype=np.uint8)
    for b in range(B):  # payload followed by its CRC (NRLDPCEncoder.m:70-89, 92-124), fillers = 0 behind
        r = orc.crc(poly, L, info[b, : Kp - L])
        info[b, Kp - L: Kp] = (r >> np.arange(L - 1, -1, -1)) & 1
        assert orc.crc(poly, L, info[b, :Kp]) == 0
    cw = orc.encode(bg, Z, info)
    for punct in (False, True):
        llr = awgn_llr(rng, cw, esn0 + (2.5 if punct else 0.0), np.float32, Z)
        llr[:, Kp:K] = np.inf  # fillers (NRLDPCDecoder.m:264)
        if punct:
            llr[:, : (kb // 2) * Z] = 0
        c = pkg.Codec(bg, Z, max_iter=20, n_layers=nl, llr_dtype=np.float32, crc=(poly, L, Kp))
        hard, it = c.decode(llr, want_iters=True)
        c1 = pkg.Codec(bg, Z, max_iter=20, n_layers=nl, early_term=True, llr_dtype=np.float32)
        hard1, it1 = c1.decode(llr, want_iters=True)
        c.close(); c1.close()
        ho, io = orc.decode_nmsq_crc(bg, Z, llr.astype(np.float64), 20, (poly, L, Kp), n_layers=nl, **rule_kw(c))
        assert (hard == ho).all(), "hard decisions differ"
        assert (it == io).all(), "iteration counts differ"
        assert (it <= it1).all()
        if not punct:
            assert (it < it1).any(), "no codeword stopped on its CRC before its parity checks held"


def test_crc_aided_stop_in_a_mixed_call_and_without_soft_output(pkg, orc):
    """A handle with early_term = 2 inside nrldpc_decode_multi_dev gets a launch of its own (the shared kernel is built without
    the CRC fold) and gives what it gives alone; soft output together with the CRC-aided stop is refused, not silently downgraded."""
    import torch
    rng = np.random.default_rng(2024)
    poly, L = CRC16
    work = []
    for bg, Z, Kp, crc in ((2, 36, 300, True), (1, 24, 500, False), (2, 384, 3840, True)):
        kb = BG_DIMS[bg][2]
        K, B = kb * Z, 9
        info = np.zeros((B, K), np.uint8)
        info[:, : Kp - L] = rng.integers(0, 2, (B, Kp - L), dtype=np.uint8)
        for b in range(B):
            r = orc.crc(poly, L, info[b, : Kp - L])
            info[b, Kp - L: Kp] = (r >> np.arange(L - 1, -1, -1)) & 1
        llr = awgn_llr(rng, orc.encode(bg, Z, info), 0.5 if bg == 1 else -0.8, np.float16, Z)
        c = pkg.Codec(bg, Z, max_iter=15, early_term=True, llr_dtype=np.float16, crc=(poly, L, Kp) if crc else None)
        ref_h, ref_i = c.decode(llr, want_iters=True)
        work.append((c, torch.from_numpy(llr).cuda(), torch.zeros((B, K), dtype=torch.uint8, device="cuda"),
                     torch.zeros(B, dtype=torch.int32, device="cuda"), ref_h, ref_i))
    pkg.decode_multi_dev([w[0] for w in work], [w[1].data_ptr() for w in work], [w[1].shape[0] for w in work],
                         [w[2].data_ptr() for w in work], [w[3].data_ptr() for w in work], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for c, _, h, it, ref_h, ref_i in work:
        assert (h.cpu().numpy() == ref_h).all() and (it.cpu().numpy() == ref_i).all()
    with pytest.raises(pkg.UnsupportedParameters):
        work[0][0].decode(work[0][1].cpu().numpy(), want_app=True)
    for w in work:
        w[0].close()


@pytest.mark.parametrize("bg,Z,B", [(1, 384, 700), (2, 7, 33), (1, 3, 5), (2, 96, 9000), (1, 64, 1)])
def test_bit_packed_hard_output(pkg, orc, bg, Z, B):
    """nrldpc_decode_packed (ABI revision 4): the same decisions as nrldpc_decode, bit k of a codeword in byte k // 8 at bit
    k % 8, unused bits of the last byte zero -- lifting sizes whose K is not a multiple of 8 (K = 66, 70), a batch large
    enough for the pipelined host path (> 8 MB of LLRs), and one codeword."""
    rng = np.random.default_rng(800 + Z)
    kb = BG_DIMS[bg][2]
    K = kb * Z
    info = rng.integers(0, 2, (min(B, 64), K), dtype=np.uint8)
    cw = np.tile(orc.encode(bg, Z, info), (-(-B // info.shape[0]), 1))[:B]
    llr = awgn_llr(rng, cw, 1.0, np.float16, Z)
    for dt in (np.float16, np.float64):
        c = pkg.Codec(bg, Z, max_iter=8, early_term=True, llr_dtype=dt)
        hard, it = c.decode(llr.astype(dt), want_iters=True)
        packed, itp = c.decode_packed(llr.astype(dt), want_iters=True)
        c.close()
        assert packed.shape == (B, (K + 7) // 8) and (itp == it).all()
        bits = np.unpackbits(packed, axis=1, bitorder="little")
        assert (bits[:, :K] == hard).all() and not bits[:, K:].any()


def test_empty_and_argument_errors(pkg):
    c = pkg.Codec(1, 8, max_iter=3)
    assert c.decode(np.zeros((0, c.N_cw), np.float32)).shape == (0, c.K)
    with pytest.raises(pkg.NRLDPCError):
        c.decode(np.zeros(c.N_cw + 1, np.float32))
    with pytest.raises(pkg.NRLDPCError):
        c.encode(np.zeros(c.K - 1, np.uint8))
    c.close()
    with pytest.raises(pkg.NRLDPCError):
        pkg.Codec(1, 8, device_id=99)


def test_device_pointer_entry_and_timing(pkg, orc):
    import torch
    rng = np.random.default_rng(9)
    bg, Z, B = 1, 384, 64
    info = rng.integers(0, 2, (B, 22 * Z), dtype=np.uint8)
    llr = awgn_llr(rng, orc.encode(bg, Z, info), 0.0, np.float16, Z)
    c = pkg.Codec(bg, Z, max_iter=10, early_term=True, llr_dtype=np.float16)
    d_llr = torch.from_numpy(llr).cuda()
    d_hard = torch.zeros((B, 22 * Z), dtype=torch.uint8, device="cuda")
    d_it = torch.zeros(B, dtype=torch.int32, device="cuda")
    c.set_timing(True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c.decode_dev(d_llr.data_ptr(), B, d_hard.data_ptr(), d_it.data_ptr(), None, s.cuda_stream)
    ms = c.last_kernel_ms()
    s.synchronize()
    ho, io = orc.decode_nmsq(bg, Z, llr.astype(np.float64), 10, early_term=True, **rule_kw(c))
    assert (d_hard.cpu().numpy() == ho).all() and (d_it.cpu().numpy() == io).all()
    assert 0.0 < ms < 1000.0
    c.close()


@pytest.mark.parametrize("bg,Z,dt", [(1, 384, np.float16), (2, 256, np.float32), (1, 320, np.float16), (1, 240, np.float16), (2, 104, np.float32)])
def test_unaligned_device_pointers(pkg, orc, bg, Z, dt):
    """The compile-time-Z kernels move LLRs and hard bits in 8/16-byte and 4-byte pieces when the caller's
    pointers allow it; pointers at odd element offsets must take the narrow path and give the same answer."""
    import torch
    rng = np.random.default_rng(Z)
    kb, B = BG_DIMS[bg][2], 9
    info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
    llr = awgn_llr(rng, orc.encode(bg, Z, info), 0.5, dt, Z)
    tdt = torch.float16 if dt == np.float16 else torch.float32
    for early in (True, False):
        c = pkg.Codec(bg, Z, max_iter=6, early_term=early, llr_dtype=dt)
        ref, ref_it = c.decode(llr, want_iters=True)
        buf = torch.zeros(llr.size + 8, dtype=tdt, device="cuda")
        buf[1:1 + llr.size] = torch.from_numpy(llr).cuda().flatten()          # one element off any 16-byte boundary
        hard = torch.zeros(B * kb * Z + 8, dtype=torch.uint8, device="cuda")
        it = torch.zeros(B, dtype=torch.int32, device="cuda")
        c.decode_dev(buf.data_ptr() + buf.element_size(), B, hard.data_ptr() + 1, it.data_ptr(), None,
                     torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        c.close()
        assert (hard[1:1 + B * kb * Z].cpu().numpy().reshape(B, -1) == ref).all() and (it.cpu().numpy() == ref_it).all()
        assert int(hard[0]) == 0 and int(hard[1 + B * kb * Z:].sum()) == 0
