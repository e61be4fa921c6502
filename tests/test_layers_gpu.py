"""ABI revision 5: the active layer count is a property of the CALL (nrldpc_set_layers) and can be read off the data
(NRLDPC_LAYERS_AUTO).  The reference decodes the full H whatever the rate (NRLDPCDecoder.m:120); a row whose extension-parity
column holds LLR 0 in every codeword contributes nothing.  Everything here is bit-exact: a call under
AUTO must give what a call with the same explicit count gives (hard decisions AND iteration counts), through every entry point."""
import numpy as np
import pytest

from conftest import BG_DIMS, awgn_llr, rule_kw
from test_decode_gpu import _waterfall_esn0

pytestmark = pytest.mark.gpu
AUTO = -1


def count_ref(bg, Z, llr):
    """numpy restatement of nrldpc.h "Active layers": highest column holding anything but +-0 / NaN -> rows."""
    rows, cols, kb = BG_DIMS[bg]
    x = np.asarray(llr, np.float64).reshape(-1, cols, Z)
    nz = ((x != 0) & ~np.isnan(x)).any(axis=(0, 2))
    top = int(np.nonzero(nz)[0].max()) if nz.any() else 0
    return max(4, top - kb + 1)


def dev_decode(codec, llr, dt):
    import torch
    B = llr.shape[0]
    d_llr = torch.from_numpy(np.ascontiguousarray(llr.astype(dt))).cuda()
    d_h = torch.empty((B, codec.K), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(B, dtype=torch.int32, device="cuda")
    codec.decode_dev(d_llr.data_ptr(), B, d_h.data_ptr(), d_it.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_h.cpu().numpy(), d_it.cpu().numpy()


# one size per kernel family of test_run_time_layer_count_grid's grid (interleaved / packed / split / row / run-time-Z), and the
# counts around which the check-node rule changes (nrldpc_default_rule) plus the call whose count one codeword lifts (8)
LAYER_Z = (8, 20, 64, 96, 144, 208, 352, 384)
LAYER_NL = (4, 8, 13, 24)


@pytest.mark.parametrize("bg", [1, 2])
@pytest.mark.parametrize("Z", LAYER_Z)
def test_auto_equals_explicit_on_the_layer_count_grid(pkg, orc, bg, Z):
    """test_run_time_layer_count_grid's grid: ONE handle created under AUTO; per count, LLRs whose columns above the count are
    zero (what rate matching leaves, NRLDPCDecoder.m:216-234); the AUTO call through host and device pointers, then
    nrldpc_set_layers(count) on the same handle, then a handle CREATED with that count, then the oracle -- all identical,
    with the parity stop (iteration counts are the sensitive part: a spare row adds parity checks) and at fixed iterations."""
    rng = np.random.default_rng(8800 + 100 * bg + Z)
    rows, cols, kb = BG_DIMS[bg]
    B = 3 + (300 // Z if Z < 64 else 0)
    dt = np.float16 if Z % 3 else np.float32
    c = pkg.Codec(bg, Z, max_iter=14, n_layers=AUTO, early_term=True, llr_dtype=dt)
    cf = pkg.Codec(bg, Z, max_iter=6, n_layers=AUTO, early_term=False, llr_dtype=dt)
    assert c.n_layers == AUTO
    try:
        for nl in LAYER_NL + (rows - 1, rows):
            info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
            llr = awgn_llr(rng, orc.encode(bg, Z, info), _waterfall_esn0(bg, nl) + 0.2, dt, Z, E=(kb + nl - 2) * Z)
            if nl == 8:  # the count is that of the WHOLE call: one codeword reaching higher lifts everybody; NaN and -0 do not count
                llr[B - 1, (kb + 9) * Z + Z // 2] = 0.25
                llr[0, (kb + 12) * Z] = np.nan
                llr[0, (kb + 13) * Z + 1] = -0.0
                want = 10
            else:
                want = nl
            assert pkg._capi.count_layers(bg, Z, llr) == want == count_ref(bg, Z, llr)
            c.set_layers(AUTO)
            h_auto, it_auto = c.decode(llr, want_iters=True)
            assert c.last_layers() == want
            d_auto = dev_decode(c, llr, dt)
            assert c.last_layers() == want
            c.set_layers(want)
            assert c.n_layers == want
            h_exp, it_exp = c.decode(llr, want_iters=True)
            d_exp = dev_decode(c, llr, dt)
            c2 = pkg.Codec(bg, Z, max_iter=14, n_layers=want, early_term=True, llr_dtype=dt)
            assert (c2.alpha, c2.beta) == (c.alpha, c.beta)  # the rule follows the count in use
            h_new, it_new = c2.decode(llr, want_iters=True)
            ref = orc.decode_nmsq(bg, Z, llr.astype(np.float64), 14, n_layers=want, early_term=True, **rule_kw(c2))
            c2.close()
            for h, it in ((h_auto, it_auto), d_auto, (h_exp, it_exp), d_exp, (h_new, it_new)):
                assert (h == ref[0]).all() and (it == ref[1]).all(), (bg, Z, nl)
            f_auto = cf.decode(llr)
            ref = orc.decode_nmsq(bg, Z, llr.astype(np.float64), 6, n_layers=want, early_term=False,
                                  alpha=pkg._capi.default_rule(bg, want)[0], beta=pkg._capi.default_rule(bg, want)[1] * 8)
            assert (f_auto == ref[0]).all() and cf.last_layers() == want
    finally:
        c.close(); cf.close()
    with pytest.raises(pkg.UnsupportedParameters):
        pkg.Codec(bg, Z, n_layers=3)
    c = pkg.Codec(bg, Z)
    with pytest.raises(pkg.UnsupportedParameters):
        c.set_layers(rows + 1)
    c.set_layers(0)
    assert c.n_layers == rows
    c.close()


@pytest.mark.parametrize("kw,rvs,esn0", [
    (dict(BG=1, A=4000, G=4800, Q_m=2), (0, 2, 3, 1), 2.0),      # R = 0.83 per transmission: rv 2 / 3 start high in the buffer and wrap
    (dict(BG=2, A=3842, G=11526, Q_m=2), (0,), 0.0),             # plot_BLER_vs_SNR.m's defaults: 21 of 42 rows
    (dict(BG=2, A=1000, G=1400, Q_m=4), (3, 0, 2), 6.0),
    (dict(BG=1, A=8424, G=9478, Q_m=2), (0, 1), 7.0),            # BASELINE configs[4]: 5 rows at rv 0
])
def test_auto_through_the_system_object_with_harq(pkg, kw, rvs, esn0):
    """The decoder System object with prune_layers="auto" (what the MEX gateway does: it sees cw_tilde only) over a HARQ
    sequence with I_HARQ = 1 (plot_BLER_vs_SNR.m:120-137): the count found in each step's combined buffer (a) equals the numpy
    restatement on the cw_tilde the object built, (b) never exceeds what the parameters say is active, (c) gives the a_hat and
    the iteration counts of an object that is told that count explicitly (same handle state otherwise: HARQ buffers included)."""
    from test_system_objects_gpu import qpsk_awgn_llr
    rng = np.random.default_rng(kw["A"] + len(rvs))
    enc = pkg.NRLDPCEncoder(**kw)
    da = pkg.NRLDPCDecoder(I_HARQ=1, iterations=20, prune_layers="auto", **kw)
    dp = pkg.NRLDPCDecoder(I_HARQ=1, iterations=20, prune_layers=True, **kw)
    for _ in range(3):
        a = rng.integers(0, 2, kw["A"], dtype=np.uint8)
        da.reset(); dp.reset()
        for rv in rvs:
            enc.rv_id = da.rv_id = dp.rv_id = rv
            y = qpsk_awgn_llr(rng, enc.step(a), esn0)
            ra = da.step(y)
            # the combined buffer the object just decoded (:236-239), as cw_tilde (:262-264)
            d = da.d_tilde_buffer if da.d_tilde_buffer is not None else None
            cw = np.concatenate([np.zeros((da.C, 2 * da.Z_c)), np.pad(d[0], ((0, 0), (0, da.N - da.N_cb)))], axis=1)
            want = count_ref(kw["BG"], da.Z_c, cw)
            assert da.last_layers == want, (rv, da.last_layers, want)
            rp = dp.step(y)
            assert da.last_layers <= dp.last_layers
            if da.last_layers == dp.last_layers:
                assert ra.size == rp.size and (ra == rp).all() and (da.last_iterations == dp.last_iterations).all()
            if ra.size:
                assert (ra == a).all()
    da.release(); dp.release()


@pytest.mark.parametrize("dt", [np.float16, np.float64])
def test_auto_on_the_pipelined_host_path(pkg, orc, dt):
    """A batch above 8 MB takes the chunked host path: under AUTO (and with any explicit count) the copy threads neither
    quantise nor send what no active row reads -- 27 of 68 columns at R = 8/9 (BASELINE configs[4]).  Same hard decisions and
    iteration counts as one device launch with the explicit count; and an explicit count really does ignore the rest of the
    row: junk above the active columns changes nothing."""
    rng = np.random.default_rng(55)
    bg, Z, B = 1, 384, 700
    c = pkg.Codec(bg, Z, max_iter=25, n_layers=AUTO, early_term=True, llr_dtype=dt)
    info = rng.integers(0, 2, (B, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), 7.5, dt, Z, E=9478)
    llr[B // 2:] *= 0.05                                       # second half never converges: uneven iteration counts
    llr[3, 5 * Z: 5 * Z + 9] = np.inf                          # fillers, and one chunk with a -inf (travels in its own format)
    llr[B - 5, 2 * Z + 3] = -np.inf
    assert llr.nbytes >= (8 << 20) * (4 if dt == np.float64 else 1)
    h1, it1 = c.decode(llr, want_iters=True)
    assert c.last_layers() == 5 == pkg._capi.count_layers(bg, Z, llr)
    p1 = np.unpackbits(c.decode_packed(llr), axis=1, bitorder="little")[:, :c.K]
    dev_dt = np.float32 if dt == np.float64 else dt
    cd = pkg.Codec(bg, Z, max_iter=25, n_layers=5, early_term=True, llr_dtype=dev_dt)
    h2, it2 = dev_decode(cd, llr, dev_dt)
    assert (h1 == h2).all() and (it1 == it2).all() and (p1 == h1).all()
    assert 0.3 < (h1 == info).all(1).mean() < 0.9 and it1.min() < 6 and it1.max() == 25
    junk = llr.copy()
    junk[:, 27 * Z:] = rng.standard_normal((B, 41 * Z)).astype(dt) * 5
    c.set_layers(5)
    h3, it3 = c.decode(junk, want_iters=True)
    h4, it4 = dev_decode(cd, junk, dev_dt)
    assert (h3 == h1).all() and (it3 == it1).all() and (h4 == h1).all() and (it4 == it1).all()
    c.set_layers(AUTO)                                         # ... which AUTO would not have ignored
    c.decode(junk[:8])
    assert c.last_layers() == 46
    # the scan goes chunk by chunk (the count of the first chunk, each later chunk checked before it is sent): ONE value above that
    # count in the last codeword of the batch makes the call start again with the count of the whole batch
    late = llr.copy()
    late[B - 1, 40 * Z + 7] = 0.5
    assert pkg._capi.count_layers(bg, Z, late) == 19
    h5, it5 = c.decode(late, want_iters=True)
    assert c.last_layers() == 19
    cd.set_layers(19)
    h6, it6 = dev_decode(cd, late, dev_dt)
    assert (h5 == h6).all() and (it5 == it6).all()
    c.close(); cd.close()


def test_auto_in_the_pool_and_in_the_mixed_batch_call(pkg, orc):
    """nrldpc_pool_set_layers(AUTO): the count is found once per call over the WHOLE batch -- one codeword in the last chunk
    that reaches higher lifts every shard -- for host pointers (byte-per-bit and bit-packed) and for device pointers (every shard
    scans its slice, the maximum is taken, then all launch).  nrldpc_decode_multi_dev: every handle under AUTO gets its own
    count from its own codewords."""
    import torch
    rng = np.random.default_rng(77)
    bg, Z, B = 2, 384, 900
    c = pkg.Codec(bg, Z, max_iter=20, n_layers=9, early_term=True, llr_dtype=np.float16)
    info = rng.integers(0, 2, (B, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), 3.0, np.float16, Z, E=(10 + 7 - 2) * Z)
    llr[B - 2, (10 + 8) * Z + 5] = 1.5                         # row 8's extension column, in the last codewords only
    assert pkg._capi.count_layers(bg, Z, llr) == 9 and pkg._capi.count_layers(bg, Z, llr[:B - 2]) == 7
    ref_h, ref_i = c.decode(llr, want_iters=True)
    pool = pkg.CodecPool(bg, Z, [0, 0, 0], chunks_per_device=2, max_iter=20, n_layers=AUTO, early_term=True, llr_dtype=np.float16)
    h, it = pool.decode(llr, want_iters=True)
    assert (h == ref_h).all() and (it == ref_i).all()
    pk, it = pool.decode_packed(llr, want_iters=True)
    assert (np.unpackbits(pk, axis=1, bitorder="little")[:, :c.K] == ref_h).all() and (it == ref_i).all()
    cuts = [0, 300, 650, 900]
    d_llr = [torch.from_numpy(llr[cuts[i]:cuts[i + 1]]).cuda() for i in range(3)]
    d_hard = [torch.full((cuts[i + 1] - cuts[i], c.K), 7, dtype=torch.uint8, device="cuda") for i in range(3)]
    d_it = [torch.zeros(cuts[i + 1] - cuts[i], dtype=torch.int32, device="cuda") for i in range(3)]
    torch.cuda.synchronize()
    pool.decode_dev([t.data_ptr() for t in d_llr], [t.shape[0] for t in d_llr], [t.data_ptr() for t in d_hard], [t.data_ptr() for t in d_it])
    assert (torch.cat(d_hard).cpu().numpy() == ref_h).all() and (torch.cat(d_it).cpu().numpy() == ref_i).all()
    pool.set_layers(7)                                         # explicit again: shard 2's stray value is ignored, the others' results are the 7-row ones
    c.set_layers(7)
    ref7_h, ref7_i = c.decode(llr, want_iters=True)
    h, it = pool.decode(llr, want_iters=True)
    assert (h == ref7_h).all() and (it == ref7_i).all()
    pool.close()
    # mixed batch: three handles under AUTO, three different counts, one call
    cs = [pkg.Codec(b, z, max_iter=15, n_layers=AUTO, early_term=True, llr_dtype=np.float16) for b, z in ((1, 64), (2, 20), (1, 384))]
    nls, llrs, refs = (13, 12, 46), [], []
    for k, n in zip(cs, nls):
        kb = BG_DIMS[k.bg][2]
        inf = rng.integers(0, 2, (40, k.K), dtype=np.uint8)
        x = awgn_llr(rng, k.encode(inf), _waterfall_esn0(k.bg, n) + 0.5, np.float16, k.Z, E=(kb + n - 2) * k.Z)
        e = pkg.Codec(k.bg, k.Z, max_iter=15, n_layers=n, early_term=True, llr_dtype=np.float16)
        refs.append(e.decode(x, want_iters=True))
        e.close()
        llrs.append(torch.from_numpy(x).cuda())
    outs = [torch.empty((40, k.K), dtype=torch.uint8, device="cuda") for k in cs]
    its = [torch.empty(40, dtype=torch.int32, device="cuda") for _ in cs]
    pkg._capi.decode_multi_dev(cs, [t.data_ptr() for t in llrs], [40] * 3, [t.data_ptr() for t in outs], [t.data_ptr() for t in its],
                               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k, n, o, i, r in zip(cs, nls, outs, its, refs):
        assert k.last_layers() == n and (o.cpu().numpy() == r[0]).all() and (i.cpu().numpy() == r[1]).all()
        k.close()
    c.close()
