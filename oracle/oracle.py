"""ctypes binding of the CPU oracle (oracle/nrldpc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BG_DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}  # rows, cols, kb


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libnrldpc_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        i32, f32 = C.c_int, C.c_float
        P = C.c_void_p
        L.orc_set_index.argtypes = [i32]
        L.orc_lifting_size.argtypes = [i32, i32]
        L.orc_graph_edges.argtypes = [i32, i32, P, P, P]
        L.orc_syndrome_weight.argtypes = [i32, i32, i32, P]
        L.orc_encode.argtypes = [i32, i32, P, i32, P]
        L.orc_decode_nmsq.argtypes = [i32, i32, i32, i32, i32, f32, i32, P, i32, P, P, P]
        L.orc_decode_onmsq.argtypes = [i32, i32, i32, i32, i32, f32, f32, i32, P, i32, P, P, P]
        L.orc_decode_onmsq_crc.argtypes = [i32, i32, i32, i32, f32, f32, i32, C.c_uint32, i32, i32, P, i32, P, P, P]
        L.orc_decode_onmsq_wide.argtypes = [i32, i32, i32, i32, i32, f32, f32, i32, i32, P, i32, P, P, P]
        L.orc_decode_bp_flood.argtypes = [i32, i32, i32, i32, P, i32, P, P, i32]
        L.orc_decode_bp_flood_app.argtypes = [i32, i32, i32, i32, P, i32, P, P, i32, P]
        L.orc_set_threads.argtypes = [i32]
        L.orc_rate_recover.argtypes = [i32] * 9 + [P, P, i32, P, P]
        L.orc_rate_match.argtypes = [i32] * 9 + [P, P, i32, P]
        L.orc_crc.argtypes = [C.c_uint32, i32, P, i32]
        L.orc_crc.restype = C.c_uint32
        # OpenMP's default team is every CPU the box SHOWS (256 on the MI355X boxes, of which the cgroup grants 16): a parallel
        # region over a handful of codewords then costs 0.1-0.2 s in thread start-up and throttling -- test files run on their own
        # took ten times as long as inside the whole suite, where an earlier test happened to set the count.  Bound it once, here.
        n = os.cpu_count() or 1
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(q) // int(per)))
        except (OSError, ValueError):
            pass
        L.orc_set_threads(n)
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_index(Z):
    return lib().orc_set_index(Z)


def lifting_size(kb, kprime):
    return lib().orc_lifting_size(kb, kprime)


def graph_edges(bg, Z):
    nnz = 316 if bg == 1 else 197
    r = np.zeros(nnz, np.int32); c = np.zeros(nnz, np.int32); s = np.zeros(nnz, np.int32)
    n = lib().orc_graph_edges(bg, Z, _p(r), _p(c), _p(s))
    assert n == nnz
    return r, c, s


def syndrome_weight(bg, Z, cw, n_layers=0):
    cw = np.ascontiguousarray(cw, np.uint8)
    return lib().orc_syndrome_weight(bg, Z, n_layers, _p(cw))


def encode(bg, Z, info):
    rows, cols, kb = BG_DIMS[bg]
    info = np.ascontiguousarray(info, np.uint8).reshape(-1, kb * Z)
    cw = np.zeros((info.shape[0], cols * Z), np.uint8)
    rc = lib().orc_encode(bg, Z, _p(info), info.shape[0], _p(cw))
    assert rc == 0, rc
    return cw


def decode_nmsq(bg, Z, llr, max_iter, n_layers=0, early_term=False, alpha=0.75, scale=8, want_app=False, beta=0.0):
    """Layered offset-normalised min-sum on the fixed-point grid; beta is in grid units (LLR * scale)."""
    rows, cols, kb = BG_DIMS[bg]
    llr = np.ascontiguousarray(llr, np.float64).reshape(-1, cols * Z)
    B = llr.shape[0]
    hard = np.zeros((B, kb * Z), np.uint8)
    iters = np.zeros(B, np.int32)
    app = np.zeros((B, cols * Z), np.float32) if want_app else None
    rc = lib().orc_decode_onmsq(bg, Z, n_layers, max_iter, int(early_term), alpha, beta, scale, _p(llr), B,
                                _p(hard), _p(iters), _p(app))
    assert rc == 0, rc
    return (hard, iters, app) if want_app else (hard, iters)


def decode_nmsq_crc(bg, Z, llr, max_iter, crc, n_layers=0, alpha=0.75, scale=8, want_app=False, beta=0.0):
    """decode_nmsq with the CRC-aided stop (nrldpc_cfg.early_term = 2): crc = (poly with its x^L term, L, K')."""
    rows, cols, kb = BG_DIMS[bg]
    llr = np.ascontiguousarray(llr, np.float64).reshape(-1, cols * Z)
    B = llr.shape[0]
    hard = np.zeros((B, kb * Z), np.uint8)
    iters = np.zeros(B, np.int32)
    app = np.zeros((B, cols * Z), np.float32) if want_app else None
    rc = lib().orc_decode_onmsq_crc(bg, Z, n_layers, max_iter, alpha, beta, scale, int(crc[0]), int(crc[1]), int(crc[2]), _p(llr), B,
                                    _p(hard), _p(iters), _p(app))
    assert rc == 0, rc
    return (hard, iters, app) if want_app else (hard, iters)


def decode_nmsq_wide(bg, Z, llr, max_iter, n_layers=0, early_term=False, alpha=0.75, scale=8, beta=0.0, qmax=32767):
    """decode_nmsq on a wide grid (values and messages saturate at +/-qmax instead of +/-127): not what the kernels compute;
    used to measure what the 8-bit grid costs."""
    rows, cols, kb = BG_DIMS[bg]
    llr = np.ascontiguousarray(llr, np.float64).reshape(-1, cols * Z)
    B = llr.shape[0]
    hard = np.zeros((B, kb * Z), np.uint8)
    iters = np.zeros(B, np.int32)
    rc = lib().orc_decode_onmsq_wide(bg, Z, n_layers, max_iter, int(early_term), alpha, beta, scale, qmax, _p(llr), B,
                                     _p(hard), _p(iters), None)
    assert rc == 0, rc
    return hard, iters


def decode_bp_flood(bg, Z, llr, max_iter, n_layers=0, nthreads=0, want_app=False):
    rows, cols, kb = BG_DIMS[bg]
    llr = np.ascontiguousarray(llr, np.float64).reshape(-1, cols * Z)
    B = llr.shape[0]
    hard = np.zeros((B, kb * Z), np.uint8)
    iters = np.zeros(B, np.int32)
    app = np.zeros((B, cols * Z), np.float64) if want_app else None
    rc = lib().orc_decode_bp_flood_app(bg, Z, n_layers, max_iter, _p(llr), B, _p(hard), _p(iters), nthreads, _p(app))
    assert rc == 0, rc
    return (hard, iters, app) if want_app else (hard, iters)


def rate_recover(Z, C_, K, K_prime, N, N_cb, k_0, Q_m, G, E_r, g_tilde, harq=None):
    """Literal NRLDPCDecoder.m:143-242,262-264 in fp32; returns [n_tb*C][2Z+N] (harq updated in place)."""
    g_tilde = np.ascontiguousarray(g_tilde, np.float32)
    n_tb = g_tilde.shape[0] if g_tilde.ndim == 2 else g_tilde.size // max(G, 1)  # G == 0 (testbench.m can draw it): shape [n_tb][0]
    g_tilde = g_tilde.reshape(n_tb, G)
    E = np.ascontiguousarray(E_r, np.int32)
    out = np.zeros((n_tb * C_, 2 * Z + N), np.float32)
    if harq is not None:
        assert harq.dtype == np.float32 and harq.shape == (n_tb, C_, N_cb) and harq.flags.c_contiguous
    rc = lib().orc_rate_recover(Z, C_, K, K_prime, N, N_cb, k_0, Q_m, G, _p(E), _p(g_tilde), n_tb, _p(harq), _p(out))
    assert rc == 0
    return out


def rate_match(Z, C_, K, K_prime, N, N_cb, k_0, Q_m, G, E_r, cw):
    """Literal NRLDPCEncoder.m:168-256; cw: [n_tb*C][2Z+N] bytes -> [n_tb][G] bytes."""
    cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, 2 * Z + N)
    n_tb = cw.shape[0] // C_
    E = np.ascontiguousarray(E_r, np.int32)
    g = np.zeros((n_tb, G), np.uint8)
    rc = lib().orc_rate_match(Z, C_, K, K_prime, N, N_cb, k_0, Q_m, G, _p(E), _p(cw), n_tb, _p(g))
    assert rc == 0
    return g


def crc(poly, L, bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    return int(lib().orc_crc(poly, L, _p(bits), bits.size))
