"""Monte-Carlo BLER harness: the control flow of the reference's plot_BLER_vs_SNR.m:104-171 with
transport blocks simulated in batches so that the decoder sees full GPU launches.

Per (BG, R, A) point it sweeps Es/N0 upward from EsN0_start in EsN0_delta steps until BLER <=
target_BLER, collecting target_block_errors errors per SNR (the "found_start" restart rule of :139-144
included), runs the HARQ loop over rv_id_sequence with I_HARQ = 1 and a reset per block (:120-137), and
writes `results/BLER_vs_SNR_<A>_<R>_<BG>_<Mod>_<iters>_<errs>_<EsN0start>_<seed>.txt` with one
`%f\\t%e` line per finished SNR (:79, :165).  MATLAB's RNG streams cannot be reproduced, so curves
compare statistically, never per realisation.

Modulation / demodulation restate NRModulator.m:73-81 / NRDemodulator.m:76-84: TS 38.211 Gray maps
(the reference's custom symbol tables), unit average power, exact log-likelihood ratios with the noise
variance N0 = 10^(-EsN0/10) (plot_BLER_vs_SNR.m:105-106).
"""
import os

import numpy as np

from ._capi import UnsupportedParameters
from .decoder import NRLDPCDecoder
from .encoder import NRLDPCEncoder
from .shard import shard_range

Q_M = {"BPSK": 1, "QPSK": 2, "16QAM": 4, "64QAM": 6, "256QAM": 8}


def _pam_levels(nb):
    """TS 38.211 5.1: amplitude of one I/Q rail from its nb bits (first bit = sign, then nested Gray):
    16QAM (1-2b0)(2-(1-2b2)); 64QAM (1-2b0)(4-(1-2b2)(2-(1-2b4))); 256QAM one level more."""
    pts = {}
    for code in range(1 << nb):
        bits = [(code >> (nb - 1 - i)) & 1 for i in range(nb)]
        x = 1.0
        for j, b in enumerate(reversed(bits[1:]), start=1):  # innermost bit first
            x = float(1 << j) - (1 - 2 * b) * x
        pts[code] = (1 - 2 * bits[0]) * x
    return pts


def _rail(nb):
    lv = _pam_levels(nb)
    codes = np.arange(1 << nb)
    amps = np.array([lv[c] for c in codes], np.float64)
    bits = ((codes[:, None] >> np.arange(nb - 1, -1, -1)[None, :]) & 1).astype(np.uint8)
    return amps, bits


def modulate(g, Q_m):
    """bits [..., G] -> complex symbols [..., G/Q_m], unit average power."""
    g = np.asarray(g, np.uint8)
    if Q_m == 1:  # comm.PSKModulator order 2, phase offset pi/4 (NRModulator.m:73)
        return (1 - 2.0 * g) * np.exp(1j * np.pi / 4)
    b = g.reshape(g.shape[:-1] + (-1, Q_m))
    nb = Q_m // 2
    amps, _ = _rail(nb)
    wi = sum(b[..., 2 * k].astype(np.int64) << (nb - 1 - k) for k in range(nb))
    wq = sum(b[..., 2 * k + 1].astype(np.int64) << (nb - 1 - k) for k in range(nb))
    norm = np.sqrt(2.0 * np.mean(amps ** 2))
    return (amps[wi] + 1j * amps[wq]) / norm


def demodulate_llr(rx, Q_m, N0):
    """Exact LLRs (positive = bit 0) for the maps of `modulate`, complex noise variance N0."""
    if Q_m == 1:
        y = np.real(rx * np.exp(-1j * np.pi / 4))
        return 4.0 * y / N0
    nb = Q_m // 2
    amps, bits = _rail(nb)
    norm = np.sqrt(2.0 * np.mean(amps ** 2))
    pts = amps / norm
    out = np.empty(rx.shape + (Q_m,), np.float64)
    for rail, y in ((0, np.real(rx)), (1, np.imag(rx))):
        metric = -((y[..., None] - pts) ** 2) / N0            # [..., 2^nb]
        for k in range(nb):
            m0 = np.where(bits[:, k] == 0, metric, -np.inf)
            m1 = np.where(bits[:, k] == 1, metric, -np.inf)
            out[..., 2 * k + rail] = np.logaddexp.reduce(m0, axis=-1) - np.logaddexp.reduce(m1, axis=-1)
    return out.reshape(rx.shape[:-1] + (-1,))


def modulate_t(g, Q_m):
    """torch version of `modulate` (device tensors)."""
    import torch
    if Q_m == 1:
        return (1 - 2.0 * g.double()) * complex(np.cos(np.pi / 4), np.sin(np.pi / 4))
    nb = Q_m // 2
    amps, _ = _rail(nb)
    norm = float(np.sqrt(2.0 * np.mean(amps ** 2)))
    lut = torch.tensor(amps / norm, dtype=torch.float64, device=g.device)
    b = g.reshape(g.shape[:-1] + (-1, Q_m)).long()
    wi = sum(b[..., 2 * k] << (nb - 1 - k) for k in range(nb))
    wq = sum(b[..., 2 * k + 1] << (nb - 1 - k) for k in range(nb))
    return torch.complex(lut[wi], lut[wq])


def demodulate_llr_t(rx, Q_m, N0):
    """torch version of `demodulate_llr` (device tensors), exact log-MAP LLRs."""
    import torch
    if Q_m == 1:
        rot = complex(np.cos(np.pi / 4), -np.sin(np.pi / 4))
        return 4.0 * (rx * rot).real / N0
    nb = Q_m // 2
    amps, bits = _rail(nb)
    norm = float(np.sqrt(2.0 * np.mean(amps ** 2)))
    pts = torch.tensor(amps / norm, dtype=torch.float64, device=rx.device)
    bt = torch.tensor(bits, device=rx.device)
    out = torch.empty(rx.shape + (Q_m,), dtype=torch.float64, device=rx.device)
    ninf = torch.tensor(float("-inf"), dtype=torch.float64, device=rx.device)
    for rail, y in ((0, rx.real), (1, rx.imag)):
        metric = -((y[..., None] - pts) ** 2) / N0
        for k in range(nb):
            m0 = torch.where(bt[:, k] == 0, metric, ninf)
            m1 = torch.where(bt[:, k] == 1, metric, ninf)
            out[..., 2 * k + rail] = torch.logsumexp(m0, dim=-1) - torch.logsumexp(m1, dim=-1)
    return out.reshape(rx.shape[:-1] + (-1,))


_SM64 = (0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB)


def payload_bits_np(seed, first_block, n, A):
    """Payload of transport blocks first_block .. first_block+n-1 (numpy restatement of payload_bits): bit i of block b is
    bit (i mod 64) of splitmix64(seed + (b*W + i div 64 + 1) * golden), W = ceil(A/64) -- a function of the GLOBAL block
    index only, so any split of a batch over devices draws the same payloads (the reference draws round(rand(A,1)) per
    block, plot_BLER_vs_SNR.m:118)."""
    W = (A + 63) // 64
    idx = (np.arange(first_block, first_block + n, dtype=np.uint64)[:, None] * np.uint64(W) + np.arange(W, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        x = np.uint64(seed % (1 << 64)) + (idx + np.uint64(1)) * np.uint64(_SM64[0])
        x = (x ^ (x >> np.uint64(30))) * np.uint64(_SM64[1])
        x = (x ^ (x >> np.uint64(27))) * np.uint64(_SM64[2])
        x = x ^ (x >> np.uint64(31))
    bits = (x[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)
    return bits.reshape(n, W * 64)[:, :A].astype(np.uint8)


def payload_bits(seed, first_block, n, A, dev):
    """The same on the device: one kernel of the library (nrldpc_payload_bits_dev; through round 5 fifteen small tensor operations,
    a fifth of a demo-sized step of the loop)."""
    import torch
    from ._capi import payload_bits_dev
    out = torch.empty((n, A), dtype=torch.uint8, device=dev)
    if n:
        with torch.cuda.device(dev):
            payload_bits_dev(seed, first_block, n, A, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    return out


ATTEMPT_STRIDE = 1 << 40  # Philox symbol counter = attempt * 2^40 + global block index * symbols per block + symbol


def simulate_point_device(chains, Q_m, EsN0, rv_id_sequence, batch, seed, first_block):
    """simulate_point with every stage on the GPU(s) (rows N1-N4 of SURVEY.md section 8f): payload, CRC attachment,
    encoding, rate matching, modulation + AWGN + exact LLRs (one HIP kernel, nrldpc_awgn_llr_dev), rate recovery, decoding,
    CRC check, error count.

    chains: one (DeviceEncodeChain, DeviceDecodeChain) pair per shard -- one per GPU of the node, or several on one GPU.
    The batch of `batch` transport blocks with GLOBAL indices first_block .. first_block+batch-1 is dealt to the shards
    in contiguous slices; payload bits and channel noise are functions of (seed, global block index, attempt) alone, so
    the outcome vector -- and the result file plot_BLER_vs_SNR writes from it -- is identical for 1, 2, 4 or 8 shards:
    the reference's "parallel instances ... aggregated together" (plot_BLER_vs_SNR.m:23-27) without the manual step,
    and without any collective (transport blocks are independent).  Work of all shards is launched before the first
    synchronisation, so GPUs run concurrently under one host thread."""
    import torch
    from ._capi import awgn_llr_dev
    D = len(chains)
    cuts = [shard_range(batch, d, D)[0] for d in range(D)] + [batch]   # contiguous slices (shard.py), as every multi-GPU path here
    N0 = 1.0 / 10.0 ** (EsN0 / 10.0)
    st = []
    for d, (enc_chain, dec_chain) in enumerate(chains):
        n = cuts[d + 1] - cuts[d]
        p, dev = enc_chain.p, enc_chain.dev
        with torch.cuda.device(dev):
            a = payload_bits(seed, first_block + cuts[d], n, p.A, dev)                     # :118
            st.append(dict(a=a, ok=torch.zeros(n, dtype=torch.bool, device=dev), n=n, live=n > 0,
                           a_hat=torch.zeros((n, p.A), dtype=torch.uint8, device=dev)))
            if n:
                dec_chain.reset()                                                          # :122
    for n_rv, rv in enumerate(rv_id_sequence):                                             # :124-137
        for d, (enc_chain, dec_chain) in enumerate(chains):
            s = st[d]
            if not s["live"]:
                continue
            p, dev = enc_chain.p, enc_chain.dev
            p.rv_id = rv
            with torch.cuda.device(dev):
                g = enc_chain.step(s["a"])
                g_tilde = torch.empty(g.shape, dtype=torch.float32, device=dev)           # :130-132 in one kernel
                first_symbol = n_rv * ATTEMPT_STRIDE + (first_block + cuts[d]) * (p.G // Q_m)
                awgn_llr_dev(g.data_ptr(), g.numel(), Q_m, EsN0, seed, first_symbol, g_tilde.data_ptr(),
                             torch.cuda.current_stream(dev).cuda_stream)
                dec, good, _ = dec_chain.step(g_tilde)
                newly = good & ~s["ok"]
                s["a_hat"] = torch.where(newly[:, None], dec, s["a_hat"])  # (no boolean indexing: a host sync per batch)
                s["ok"] = s["ok"] | good
        if n_rv + 1 < len(rv_id_sequence):  # a shard whose blocks are all in stops retransmitting (:136); one sync per attempt
            for s in st:
                s["live"] = s["live"] and not bool(s["ok"].all())
            if not any(s["live"] for s in st):
                break
    out = [(s["ok"] & (s["a_hat"] == s["a"]).all(dim=1)).cpu().numpy() for s in st]
    return np.concatenate(out) if out else np.zeros(0, bool)


def _count_outcomes(outcomes, found_start, errors, blocks, BLER, target_block_errors):
    """The per-block bookkeeping of plot_BLER_vs_SNR.m:139-155 over one batch of outcomes (True = block decoded), in order:
    before the curve has had its first success a failure ends the point with BLER = 1 (:139-141); afterwards blocks and errors
    are counted until the error target is met.  Returns (found_start, keep_going, errors, blocks, BLER).  Whole-array arithmetic:
    as a Python loop over millions of outcomes this was a fifth of a device sweep's wall time; tests/test_harness_golden.py holds
    the loop and compares."""
    ok = np.asarray(outcomes, dtype=bool)
    if ok.size == 0:
        return found_start, True, errors, blocks, BLER
    if not found_start and not ok[0]:
        return found_start, False, errors, blocks, 1.0
    bad = np.cumsum(~ok) + errors
    hit = np.nonzero(bad >= target_block_errors)[0]
    n = int(hit[0]) + 1 if hit.size else ok.size          # blocks of this batch that are counted
    errors = int(bad[n - 1])
    blocks += n
    return True, True, errors, blocks, errors / blocks


def _num2str(x):
    """MATLAB num2str for the values used in the result file names: integers plainly, otherwise
    %.<d>g with d = max(1, floor(log10|x|) + 1) + 4 significant digits ('0.33333', '0.01', '0.001', '1234.5')."""
    if float(x) == int(x):
        return str(int(x))
    d = max(1, int(np.floor(np.log10(abs(float(x))))) + 1) + 4
    return "%.*g" % (d, x)


def simulate_point(hEnc, hDec, Q_m, EsN0, rv_id_sequence, batch, rng):
    """One batch of transport blocks at one SNR through the HARQ loop of plot_BLER_vs_SNR.m:118-137.
    Returns per-block success flags (a == a_hat) in simulation order."""
    A = hEnc.A
    a = rng.integers(0, 2, (batch, A), dtype=np.uint8)            # :118
    N0 = 1.0 / 10.0 ** (EsN0 / 10.0)                                 # :106
    ok = np.zeros(batch, bool)
    a_hat = np.zeros((batch, A), np.uint8)
    hDec._nb = batch
    hDec.reset()                                                     # :122
    for rv in rv_id_sequence:                                        # :124-137
        hEnc.rv_id = rv
        hDec.rv_id = rv
        g = hEnc.step_batch(a)                                       # :129
        tx = modulate(g, Q_m)                                        # :130
        noise = np.sqrt(N0 / 2.0) * (rng.standard_normal(tx.shape) + 1j * rng.standard_normal(tx.shape))
        g_tilde = demodulate_llr(tx + noise, Q_m, N0)                # :131-132
        dec, good = hDec.step_batch(g_tilde)                         # :133
        newly = good & ~ok                                           # a block stops retransmitting once decoded
        a_hat[newly] = dec[newly]
        ok |= good
        if ok.all():
            break
    return ok & (a_hat == a).all(axis=1)


def plot_BLER_vs_SNR(A=3842, R=1 / 3, BG=2, Modulation="QPSK", rv_id_sequence=(0,), iterations=8,
                     target_block_errors=3, target_BLER=1e-3, EsN0_start=0.0, EsN0_delta=0.5, seed=0,
                     results_dir="results", batch=256, max_points=200, decoder_kwargs=None, device=False, devices=None):
    """Same positional parameters and defaults as plot_BLER_vs_SNR.m:1,30-42 (no figure is drawn).
    device=True keeps every stage on the GPU (simulate_point_device); devices = HIP ordinals of the shards (default [0];
    an ordinal may repeat: logical shards on one GPU) -- the result file does not depend on their number.
    Returns {(A, R, BG): [(EsN0, BLER, blocks), ...]}."""
    rng = np.random.default_rng(seed)                                # :45
    Q_m = Q_M.get(Modulation)
    if Q_m is None:
        raise UnsupportedParameters("Unsupported modulation")
    A_list, R_list, BG_list = np.atleast_1d(A), np.atleast_1d(R), np.atleast_1d(BG)
    os.makedirs(results_dir, exist_ok=True)
    curves = {}
    for bg in BG_list:                                               # :53
        for r in R_list:                                             # :54
            for a_len in A_list:                                     # :67
                a_len, bg = int(a_len), int(bg)
                name = "BLER_vs_SNR_%s_%s_%s_%s_%s_%s_%s_%s.txt" % (
                    _num2str(a_len), _num2str(r), _num2str(bg), Modulation, _num2str(iterations),
                    _num2str(target_block_errors), _num2str(EsN0_start), _num2str(seed))  # :79
                points = []
                try:
                    G = int(round(a_len / r / Q_m) * Q_m)            # :94
                    hEnc = NRLDPCEncoder(A=a_len, BG=bg, G=G, Q_m=Q_m)                                   # :98
                    hDec = NRLDPCDecoder(A=a_len, BG=bg, G=G, Q_m=Q_m, I_HARQ=1, iterations=iterations,
                                         **(decoder_kwargs or {}))                                       # :99
                    hEnc.validate()
                    if device:
                        from .device_chain import DeviceDecodeChain, DeviceEncodeChain
                        from .nrldpc import NRLDPC
                        chains = []
                        for ordinal in (devices or [0]):  # one parameter object and one chain pair per shard
                            shared = NRLDPC(A=a_len, BG=bg, G=G, Q_m=Q_m)
                            chains.append((DeviceEncodeChain(shared, device_id=int(ordinal)),
                                           DeviceDecodeChain(shared, iterations=iterations, I_HARQ=1, device_id=int(ordinal),
                                                             **(decoder_kwargs or {}))))
                        curve_seed = int(seed) * 0x9E3779B97F4A7C15 % (1 << 64)  # payload / noise streams of this curve
                        first_block = 0                                          # global index of the next transport block
                    with open(os.path.join(results_dir, name), "w") as fid:
                        BLER, EsN0, found_start = 1.0, float(EsN0_start), False                          # :84-88
                        while BLER > target_BLER and len(points) < max_points:                           # :104
                            blocks = errors = 0
                            keep_going = True
                            while keep_going and errors < target_block_errors:                           # :116
                                if device:
                                    outcomes = simulate_point_device(chains, Q_m, EsN0, rv_id_sequence, batch, curve_seed, first_block)
                                    first_block += batch
                                else:
                                    outcomes = simulate_point(hEnc, hDec, Q_m, EsN0, rv_id_sequence, batch, rng)
                                found_start, keep_going, errors, blocks, BLER = _count_outcomes(
                                    outcomes, found_start, errors, blocks, BLER, target_block_errors)
                            if BLER < 1:                                                                 # :164-166
                                fid.write("%f\t%e\n" % (EsN0, BLER))
                                fid.flush()
                                points.append((EsN0, BLER, blocks))
                            EsN0 += EsN0_delta                                                           # :169
                    hEnc.release()
                    hDec.release()
                    if device:
                        for tx_chain, rx_chain in chains:
                            tx_chain.close()
                            rx_chain.close()
                except UnsupportedParameters:                        # :172-176: skip this (A, R, BG)
                    continue
                curves[(a_len, float(r), bg)] = points
    return curves


def plot_SNR_vs_A(A=tuple(range(1000, 9000, 1000)), R=1 / 3, BG=1, Modulation="QPSK", rv_id_sequence=(0,), iterations=50,
                  target_block_errors=100, target_BLER=1e-2, EsN0_start=-2.0, EsN0_delta=0.1, seed=0,
                  results_dir="results", batch=256, max_points=400, decoder_kwargs=None, device=False, devices=None,
                  simulate=None):
    """The reference's second harness, plot_SNR_vs_A.m:1-194 (same positional parameters and defaults, :38-48; no figure): for every
    coding rate R and information block length A, the Es/N0 at which the BLER crosses target_BLER -- the SNR is stepped up from
    EsN0_start in EsN0_delta steps (:103-108), target_block_errors errors are collected per SNR with the same found_start rule as
    plot_BLER_vs_SNR (:145-161), and the crossing is interpolated between the last two SNRs in the log10-BLER domain (:175).  One
    result file per rate, `results/SNR_vs_A_<target_BLER>_<R>_<BG>_<Mod>_<iters>_<errs>_<seed>.txt` (:80), one `%d\t%f` line per A
    (:186); parameter sets the objects refuse are skipped (:165-172).  Blocks are simulated in batches, on the host mirror or with
    every stage on the GPU(s) (device=True), exactly as in plot_BLER_vs_SNR.
    simulate: test hook -- a function (A, EsN0, n, first_block) -> per-block outcomes that replaces the simulation.
    Returns {R: [(A, EsN0 or nan), ...]}."""
    rng = np.random.default_rng(seed)                                # :51
    Q_m = Q_M.get(Modulation)
    if Q_m is None:
        raise UnsupportedParameters("Unsupported modulation")
    A_list, R_list = [int(a) for a in np.atleast_1d(A)], [float(r) for r in np.atleast_1d(R)]
    os.makedirs(results_dir, exist_ok=True)
    out = {}
    for r in R_list:                                                 # :69
        name = "SNR_vs_A_%s_%s_%s_%s_%s_%s_%s.txt" % (_num2str(target_BLER), _num2str(r), _num2str(BG), Modulation,
                                                      _num2str(iterations), _num2str(target_block_errors), _num2str(seed))  # :80
        rows = []
        with open(os.path.join(results_dir, name), "w") as fid:
            for a_len in A_list:                                     # :88
                found_start = False                                  # :90
                chains = hEnc = hDec = None
                try:
                    BLER, prev_BLER = 1.0, float("nan")              # :95-96
                    EsN0 = float(EsN0_start) - float(EsN0_delta)     # :97
                    prev_EsN0 = EsN0
                    G = int(round(a_len / r / Q_m) * Q_m)            # :98
                    if simulate is None:
                        hEnc = NRLDPCEncoder(A=a_len, BG=int(BG), G=G, Q_m=Q_m)                                   # :101
                        hDec = NRLDPCDecoder(A=a_len, BG=int(BG), G=G, Q_m=Q_m, I_HARQ=1, iterations=iterations,
                                             **(decoder_kwargs or {}))                                            # :102
                        hEnc.validate()
                        if device:
                            from .device_chain import DeviceDecodeChain, DeviceEncodeChain
                            from .nrldpc import NRLDPC
                            chains = []
                            for ordinal in (devices or [0]):
                                shared = NRLDPC(A=a_len, BG=int(BG), G=G, Q_m=Q_m)
                                chains.append((DeviceEncodeChain(shared, device_id=int(ordinal)),
                                               DeviceDecodeChain(shared, iterations=iterations, I_HARQ=1, device_id=int(ordinal),
                                                                 **(decoder_kwargs or {}))))
                    curve_seed = (int(seed) * 0x9E3779B97F4A7C15 + a_len) % (1 << 64)  # payload / noise streams of this (seed, A)
                    first_block, n_points = 0, 0
                    while BLER > target_BLER and n_points < max_points:                 # :105
                        prev_EsN0 = EsN0                                                 # :106
                        EsN0 = EsN0 + float(EsN0_delta)                                  # :107
                        blocks = errors = 0                                              # :114-115
                        keep_going = True
                        bler_now = 1.0
                        while keep_going and errors < target_block_errors:               # :120
                            if simulate is not None:
                                outcomes = simulate(a_len, EsN0, batch, first_block)
                            elif device:
                                outcomes = simulate_point_device(chains, Q_m, EsN0, rv_id_sequence, batch, curve_seed, first_block)
                            else:
                                outcomes = simulate_point(hEnc, hDec, Q_m, EsN0, rv_id_sequence, batch, rng)
                            first_block += batch
                            found_start, keep_going, errors, blocks, bler_now = _count_outcomes(
                                outcomes, found_start, errors, blocks, bler_now, target_block_errors)
                        prev_BLER = BLER                                                 # :162
                        BLER = bler_now                                                  # :163 (block_error_count / block_count; 1 / 1 before the first success)
                        n_points += 1
                except UnsupportedParameters:                        # :165-172: skip this (A, R)
                    continue
                finally:
                    if hEnc is not None:
                        hEnc.release()
                    if hDec is not None:
                        hDec.release()
                    for tx_chain, rx_chain in (chains or []):
                        tx_chain.close()
                        rx_chain.close()
                # :175 interp1(log10([prev_BLER, BLER]), [prev_EsN0, EsN0], log10(target_BLER)).  prev_BLER is 1 at EsN0_start -
                # EsN0_delta when the first SNR already meets the target (:95-97,162); NaN, as interp1 gives out of range, when the
                # sweep was cut by max_points before the target was met
                x0, x1, xt = np.log10(prev_BLER) if prev_BLER == prev_BLER else float("nan"), np.log10(BLER), np.log10(target_BLER)
                if x0 != x0 or x0 == x1 or not (min(x0, x1) <= xt <= max(x0, x1)):
                    es = float("nan")
                else:
                    es = prev_EsN0 + (EsN0 - prev_EsN0) * (xt - x0) / (x1 - x0)
                fid.write("%d\t%s\n" % (a_len, "NaN" if es != es else "%f" % es))      # :186
                fid.flush()
                rows.append((a_len, es))
        out[r] = rows
    return out
