"""ctypes binding of the C ABI declared in include/nrldpc.h (libnrldpc_hip.so).

There is no CPU implementation behind this module: if the shared object is missing it is built
with hipcc; if that is impossible, import fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK, ERR_UNSUPPORTED, ERR_ARG, ERR_HIP, ERR_NOMEM = 0, 1, 2, 3, 4
LLR_F32, LLR_F16, LLR_F64 = 0, 1, 2


class UnsupportedParameters(ValueError):
    """Mirror of the MATLAB identifier 'ldpc_3gpp_matlab:UnsupportedParameters'
    (e.g. NRLDPC.m:240-294, get_3gpp_set_index.m:10)."""
    identifier = "ldpc_3gpp_matlab:UnsupportedParameters"


class NRLDPCError(RuntimeError):
    """Mirror of the MATLAB identifier 'ldpc_3gpp_matlab:Error' (e.g. NRLDPCDecoder.m:149)."""
    identifier = "ldpc_3gpp_matlab:Error"


ABI_VERSION = 6  # NRLDPC_ABI_VERSION of include/nrldpc.h
LAYERS_ALL, LAYERS_AUTO = 0, -1  # NRLDPC_LAYERS_*


class Cfg(C.Structure):
    """nrldpc_cfg; struct_size is filled in by the constructor (positional arguments start at bg)."""
    _fields_ = [("struct_size", C.c_uint32), ("bg", C.c_int32), ("Z", C.c_int32), ("n_layers", C.c_int32), ("max_iter", C.c_int32),
                ("early_term", C.c_int32), ("alpha", C.c_float), ("llr_scale", C.c_int32),
                ("llr_dtype", C.c_int32), ("device_id", C.c_int32), ("max_batch", C.c_int32), ("beta", C.c_float),
                ("crc_poly", C.c_uint32), ("crc_len", C.c_int32), ("crc_bits", C.c_int32)]


def _cfg_init(self, *args, **kw):
    C.Structure.__init__(self, C.sizeof(Cfg), *args, **kw)


Cfg.__init__ = _cfg_init


class Dims(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("nrows", C.c_int32), ("ncols", C.c_int32), ("kb", C.c_int32), ("i_ls", C.c_int32),
                ("K", C.c_int32), ("N_cw", C.c_int32), ("n_layers", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float)]


MAX_C = 160


class TbParams(C.Structure):
    _fields_ = [("bg", C.c_int32), ("Z", C.c_int32), ("A", C.c_int32), ("B", C.c_int32), ("C", C.c_int32),
                ("K", C.c_int32), ("K_prime", C.c_int32), ("N", C.c_int32), ("N_cb", C.c_int32), ("k_0", C.c_int32),
                ("Q_m", C.c_int32), ("G", C.c_int32), ("tb_crc_len", C.c_int32), ("cb_crc_len", C.c_int32),
                ("E_r", C.c_int32 * MAX_C)]


def tb_params(p):
    """nrldpc_tb_params from an NRLDPC parameter object (NRLDPC.m:297-543)."""
    if p.C > MAX_C:
        raise UnsupportedParameters("more than %d code blocks" % MAX_C)
    t = TbParams(p.BG, p.Z_c, p.A, p.B, p.C, p.K, int(p.K_prime), p.N, p.N_cb, p.k_0, p.Q_m, p.G,
                 p.transport_block_L, p.code_block_L)
    for r, e in enumerate(p.E_r):
        t.E_r[r] = e
    return t


EXPORTS = ["nrldpc_awgn_llr_dev", "nrldpc_rate_recover_dev",  "nrldpc_crc_check_dev", "nrldpc_crc_check_harq_dev", "nrldpc_crc_attach_dev", "nrldpc_rate_match_dev", "nrldpc_create", "nrldpc_destroy", "nrldpc_get_dims", "nrldpc_decode", "nrldpc_decode_dev",
           "nrldpc_decode_multi_dev", "nrldpc_quantise_llr", "nrldpc_encode", "nrldpc_encode_dev", "nrldpc_set_timing", "nrldpc_last_kernel_ms",
           "nrldpc_set_index", "nrldpc_lifting_size", "nrldpc_default_rule", "nrldpc_strerror", "nrldpc_last_error",
           "nrldpc_version", "nrldpc_build_id", "nrldpc_kernel_id", "nrldpc_pool_create", "nrldpc_pool_decode", "nrldpc_pool_last_split",
           "nrldpc_pool_destroy", "nrldpc_pool_decode_dev", "nrldpc_pool_size", "nrldpc_abi_version", "nrldpc_decode_packed",
           "nrldpc_set_layers", "nrldpc_set_llr_dtype", "nrldpc_last_layers", "nrldpc_count_layers", "nrldpc_pool_set_layers", "nrldpc_pool_decode_packed",
           "nrldpc_decode_packed_layers", "nrldpc_pool_set_timing", "nrldpc_pool_last_kernel_ms", "nrldpc_last_host_phases", "nrldpc_payload_bits_dev"]

_lib = None


def lib_path():
    return _build.LIB


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if this
    library pulled in /opt/rocm's copy first, torch would later find "No HIP GPUs".  Importing torch
    first (when it is installed) makes both use the same runtime; without torch nothing happens."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load():
    """Load (building first if needed) libnrldpc_hip.so.  Raises if it cannot be produced."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    path = _build.LIB
    if os.environ.get("NRLDPC_LIB"):
        pass  # an explicitly selected library (kernel experiments): used as it is
    elif _build._stale():  # missing, or built from other sources than the tree holds now (content hash)
        import fcntl
        with open(path + ".lock", "w") as lk:  # several ranks of one node may get here at once: one builds, the rest wait
            fcntl.flock(lk, fcntl.LOCK_EX)
            if _build._stale():
                path = _build.build_lib()
    L = C.CDLL(path)
    # the ABI revision first, before any symbol that only this revision exports is touched: a stale library (NRLDPC_LIB skips
    # the build-id check) must fail with this message, not with a ctypes AttributeError on a missing symbol
    abi = getattr(L, "nrldpc_abi_version", None)
    have = abi() if abi is not None else 0
    if have != ABI_VERSION:
        raise RuntimeError("libnrldpc_hip.so speaks ABI revision %s, this binding %d"
                           % (have if abi is not None else "< 3 (no nrldpc_abi_version)", ABI_VERSION))
    L.nrldpc_build_id.restype = C.c_char_p
    L.nrldpc_kernel_id.restype = C.c_char_p
    if not os.environ.get("NRLDPC_LIB") and L.nrldpc_build_id().decode() != _build.source_id():
        raise RuntimeError("libnrldpc_hip.so (build %s) does not match the sources in the tree (%s) and could not be "
                           "rebuilt" % (L.nrldpc_build_id().decode(), _build.source_id()))
    vp, i32 = C.c_void_p, C.c_int32
    L.nrldpc_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
    L.nrldpc_destroy.argtypes = [vp]
    L.nrldpc_destroy.restype = None
    L.nrldpc_get_dims.argtypes = [vp, C.POINTER(Dims)]
    L.nrldpc_decode.argtypes = [vp, vp, i32, vp, vp, vp]
    L.nrldpc_decode_packed.argtypes = [vp, vp, i32, vp, vp]
    L.nrldpc_decode_packed_layers.argtypes = [vp, vp, i32, vp, vp, i32]
    L.nrldpc_pool_set_timing.argtypes = [vp, i32]
    L.nrldpc_last_host_phases.argtypes = [vp, C.POINTER(C.c_double)]
    L.nrldpc_payload_bits_dev.argtypes = [C.c_uint64, C.c_uint64, i32, i32, vp, vp]
    L.nrldpc_pool_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.nrldpc_decode_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.nrldpc_quantise_llr.argtypes = [vp, vp, C.c_int64, i32, i32]
    L.nrldpc_decode_multi_dev.argtypes = [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), vp]
    L.nrldpc_encode.argtypes = [vp, vp, i32, vp]
    L.nrldpc_encode_dev.argtypes = [vp, vp, i32, vp, vp]
    L.nrldpc_rate_recover_dev.argtypes = [C.POINTER(TbParams), vp, i32, vp, vp, i32, vp]
    L.nrldpc_crc_check_dev.argtypes = [C.POINTER(TbParams), vp, i32, vp, vp, vp, vp]
    L.nrldpc_crc_check_harq_dev.argtypes = [C.POINTER(TbParams), vp, i32, vp, vp, vp, vp, i32, vp]
    L.nrldpc_awgn_llr_dev.argtypes = [vp, C.c_int64, i32, C.c_float, C.c_uint64, C.c_uint64, vp, vp]
    L.nrldpc_crc_attach_dev.argtypes = [C.POINTER(TbParams), vp, i32, vp, vp]
    L.nrldpc_rate_match_dev.argtypes = [C.POINTER(TbParams), vp, i32, vp, vp]
    L.nrldpc_pool_create.argtypes = [C.POINTER(Cfg), C.POINTER(i32), i32, i32, C.POINTER(vp)]
    L.nrldpc_pool_decode.argtypes = [vp, vp, i32, vp, vp]
    L.nrldpc_pool_decode_packed.argtypes = [vp, vp, i32, vp, vp]
    L.nrldpc_pool_set_layers.argtypes = [vp, i32]
    L.nrldpc_set_layers.argtypes = [vp, i32]
    L.nrldpc_set_llr_dtype.argtypes = [vp, i32]
    L.nrldpc_last_layers.argtypes = [vp, C.POINTER(i32)]
    L.nrldpc_count_layers.argtypes = [i32, i32, vp, i32, i32]
    L.nrldpc_pool_last_split.argtypes = [vp, C.POINTER(i32)]
    L.nrldpc_pool_decode_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp)]
    L.nrldpc_pool_size.argtypes = [vp]
    L.nrldpc_pool_destroy.argtypes = [vp]
    L.nrldpc_pool_destroy.restype = None
    L.nrldpc_set_timing.argtypes = [vp, i32]
    L.nrldpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.nrldpc_set_index.argtypes = [i32]
    L.nrldpc_lifting_size.argtypes = [i32, i32]
    L.nrldpc_default_rule.argtypes = [i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for f in ("nrldpc_strerror", "nrldpc_last_error", "nrldpc_version", "nrldpc_build_id"):
        getattr(L, f).restype = C.c_char_p
    L.nrldpc_strerror.argtypes = [i32]
    _lib = L
    return L


def check(rc):
    if rc == OK:
        return
    L = load()
    msg = (L.nrldpc_last_error() or b"").decode() or L.nrldpc_strerror(rc).decode()
    if rc == ERR_UNSUPPORTED:
        raise UnsupportedParameters(msg)
    raise NRLDPCError(msg)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


_NP2DT = {np.dtype(np.float32): LLR_F32, np.dtype(np.float16): LLR_F16, np.dtype(np.float64): LLR_F64}


class Codec:
    """One (BG, Z) LDPC coding core on one GPU: the object that takes the place of
    comm.LDPCDecoder / comm.LDPCEncoder in the reference (NRLDPCDecoder.m:120, NRLDPCEncoder.m:49)."""

    def __init__(self, bg, Z, max_iter=50, n_layers=0, early_term=True, alpha=0.0, llr_scale=0,
                 llr_dtype=np.float32, device_id=0, max_batch=0, beta=0.0, crc=None):
        """alpha = 0: the library picks the check-node rule (alpha, beta) by rate (nrldpc_default_rule);
        otherwise message magnitude = max(alpha*min - beta, 0), beta in LLR units.
        n_layers: 0 = every row, 4..rows, or LAYERS_AUTO (-1) = read off each call's LLRs; set_layers() changes it between calls.
        crc = (poly with its x^L term, L, K'): the CRC-aided stop (nrldpc_cfg.early_term = 2) on the first K' information bits."""
        L = load()
        self._lib = L
        self._h = C.c_void_p()
        self.llr_dtype = np.dtype(llr_dtype)
        cfg = Cfg(int(bg), int(Z), int(n_layers), int(max_iter), 2 if crc else int(bool(early_term)), float(alpha),
                  int(llr_scale), _NP2DT[self.llr_dtype], int(device_id), int(max_batch), float(beta),
                  *((int(crc[0]), int(crc[1]), int(crc[2])) if crc else (0, 0, 0)))
        check(L.nrldpc_create(C.byref(cfg), C.byref(self._h)))
        d = Dims(C.sizeof(Dims))
        check(L.nrldpc_get_dims(self._h, C.byref(d)))
        self.bg, self.Z = int(bg), int(Z)
        self.K, self.N_cw, self.kb, self.ncols, self.nrows = d.K, d.N_cw, d.kb, d.ncols, d.nrows
        self.i_ls, self.n_layers = d.i_ls, d.n_layers
        self.alpha, self.beta = float(d.alpha), float(d.beta)  # resolved check-node rule

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.nrldpc_destroy(self._h)
            self._h = C.c_void_p()

    def set_layers(self, n_layers):
        """nrldpc_set_layers: active layer count of the calls that follow (0 all, 4..rows, LAYERS_AUTO); no device work."""
        check(self._lib.nrldpc_set_layers(self._h, int(n_layers)))
        d = Dims(C.sizeof(Dims))
        check(self._lib.nrldpc_get_dims(self._h, C.byref(d)))
        self.n_layers, self.alpha, self.beta = d.n_layers, float(d.alpha), float(d.beta)

    def set_llr_dtype(self, llr_dtype):
        """nrldpc_set_llr_dtype: element type of the arrays the calls that follow hand over (np.float32 / float16 / float64)."""
        dt = np.dtype(llr_dtype)
        check(self._lib.nrldpc_set_llr_dtype(self._h, _NP2DT[dt]))
        self.llr_dtype = dt

    def last_layers(self):
        """nrldpc_last_layers: the count the most recent decode call ran with (what LAYERS_AUTO found)."""
        n = C.c_int32()
        check(self._lib.nrldpc_last_layers(self._h, C.byref(n)))
        return int(n.value)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-pointer entry points (numpy arrays) -------------------------------------------------
    def decode(self, llr, want_iters=False, want_app=False, out=None):
        """out: a (B, K) uint8 array to decode into.  A caller that decodes batch after batch should pass one: a fresh 35 MB
        array per call is mmap'd, page-faulted by sixteen copy threads and munmap'd every time -- 3-5 ms on top of a 4 ms call and
        20-30 ms every few calls (profiles/r05_host_stall.txt; what round 4 recorded as a stall of this entry point)."""
        llr = np.ascontiguousarray(llr, self.llr_dtype)
        if llr.size % self.N_cw:
            raise NRLDPCError("llr should hold a whole number of codewords of length %d" % self.N_cw)
        B = llr.size // self.N_cw
        if out is not None and (out.shape != (B, self.K) or out.dtype != np.uint8 or not out.flags.c_contiguous):
            raise NRLDPCError("out should be a C-contiguous uint8 array of shape (%d, %d)" % (B, self.K))
        hard = out if out is not None else np.empty((B, self.K), np.uint8)
        iters = np.empty(B, np.int32) if want_iters else None
        app = np.empty((B, self.N_cw), np.float32) if want_app else None
        check(self._lib.nrldpc_decode(self._h, _ptr(llr), B, _ptr(hard), _ptr(iters), _ptr(app)))
        out = (hard,)
        if want_iters:
            out += (iters,)
        if want_app:
            out += (app,)
        return out[0] if len(out) == 1 else out

    def decode_packed(self, llr, want_iters=False, out=None, n_layers=None):
        """nrldpc_decode_packed: hard decisions as [B][ceil(K/8)] bytes, bit k of a codeword in byte k // 8 at bit k % 8
        (np.unpackbits(out, axis=1, bitorder="little")[:, :K] gives decode()'s array).  out: see decode().
        n_layers: the active layer count of THIS call only (nrldpc_decode_packed_layers: 0 all, 4..rows, LAYERS_AUTO); the
        handle's own count is untouched."""
        llr = np.ascontiguousarray(llr, self.llr_dtype)
        if llr.size % self.N_cw:
            raise NRLDPCError("llr should hold a whole number of codewords of length %d" % self.N_cw)
        B = llr.size // self.N_cw
        if out is not None and (out.shape != (B, (self.K + 7) // 8) or out.dtype != np.uint8 or not out.flags.c_contiguous):
            raise NRLDPCError("out should be a C-contiguous uint8 array of shape (%d, %d)" % (B, (self.K + 7) // 8))
        packed = out if out is not None else np.empty((B, (self.K + 7) // 8), np.uint8)
        iters = np.empty(B, np.int32) if want_iters else None
        if n_layers is None:
            check(self._lib.nrldpc_decode_packed(self._h, _ptr(llr), B, _ptr(packed), _ptr(iters)))
        else:
            check(self._lib.nrldpc_decode_packed_layers(self._h, _ptr(llr), B, _ptr(packed), _ptr(iters), int(n_layers)))
        return (packed, iters) if want_iters else packed

    def encode(self, info):
        info = np.ascontiguousarray(info, np.uint8)
        if info.size % self.K:
            raise NRLDPCError("info should hold a whole number of blocks of length %d" % self.K)
        B = info.size // self.K
        cw = np.empty((B, self.N_cw), np.uint8)
        check(self._lib.nrldpc_encode(self._h, _ptr(info), B, _ptr(cw)))
        return cw

    # -- device-pointer entry points (raw addresses, e.g. torch.Tensor.data_ptr()) ----------------
    def decode_dev(self, d_llr, batch, d_hard, d_iters=None, d_app=None, stream=0):
        check(self._lib.nrldpc_decode_dev(self._h, _ptr(d_llr), int(batch), _ptr(d_hard), _ptr(d_iters),
                                          _ptr(d_app), C.c_void_p(stream)))

    def encode_dev(self, d_info, batch, d_cw, stream=0):
        check(self._lib.nrldpc_encode_dev(self._h, _ptr(d_info), int(batch), _ptr(d_cw), C.c_void_p(stream)))

    def set_timing(self, on=True):
        check(self._lib.nrldpc_set_timing(self._h, int(on)))

    def last_host_phases(self):
        """nrldpc_last_host_phases as a dict (ms of the caller's thread in each phase of the last large host-pointer call), or None."""
        out = (C.c_double * 10)()
        if self._lib.nrldpc_last_host_phases(self._h, out) != 0:
            return None
        return {"chunks": int(out[0]), "codewords_per_chunk": int(out[1]), "layers": int(out[2]), "scan_ms": out[3],
                "copy_quantise_ms": out[4], "launch_enqueue_ms": out[5], "wait_device_ms": out[6], "copy_out_ms": out[7],
                "copy_threads_numa_node": int(out[8]), "caller_cpu": int(out[9])}

    def last_kernel_ms(self):
        ms = C.c_float()
        check(self._lib.nrldpc_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


class CodecPool:
    """One node, several GPUs (nrldpc_pool_*): one handle and one host thread per entry of device_ids, the batch cut
    into len(device_ids) * chunks_per_device chunks pulled from a queue -- no collective, results identical to one
    Codec.decode call.  A device ordinal may repeat (several logical shards on one GPU)."""

    def __init__(self, bg, Z, device_ids, chunks_per_device=3, max_iter=50, n_layers=0, early_term=True, alpha=0.0,
                 beta=0.0, llr_scale=0, llr_dtype=np.float32):
        L = load()
        self._lib = L
        self.llr_dtype = np.dtype(llr_dtype)
        self.device_ids = [int(d) for d in device_ids]
        cfg = Cfg(int(bg), int(Z), int(n_layers), int(max_iter), int(bool(early_term)), float(alpha), int(llr_scale),
                  _NP2DT[self.llr_dtype], 0, 0, float(beta))
        ids = (C.c_int32 * len(self.device_ids))(*self.device_ids)
        self._p = C.c_void_p()
        check(L.nrldpc_pool_create(C.byref(cfg), ids, len(self.device_ids), int(chunks_per_device), C.byref(self._p)))
        rows, cols, kb = {1: (46, 68, 22), 2: (42, 52, 10)}[int(bg)]
        self.K, self.N_cw = kb * int(Z), cols * int(Z)

    def decode(self, llr, want_iters=False):
        llr = np.ascontiguousarray(llr, self.llr_dtype)
        if llr.size % self.N_cw:
            raise NRLDPCError("llr should hold a whole number of codewords of length %d" % self.N_cw)
        B = llr.size // self.N_cw
        hard = np.empty((B, self.K), np.uint8)
        iters = np.empty(B, np.int32) if want_iters else None
        check(self._lib.nrldpc_pool_decode(self._p, _ptr(llr), B, _ptr(hard), _ptr(iters)))
        return (hard, iters) if want_iters else hard

    def decode_packed(self, llr, want_iters=False):
        """nrldpc_pool_decode_packed: as decode(), hard decisions bit-packed [B][ceil(K/8)] (Codec.decode_packed)."""
        llr = np.ascontiguousarray(llr, self.llr_dtype)
        if llr.size % self.N_cw:
            raise NRLDPCError("llr should hold a whole number of codewords of length %d" % self.N_cw)
        B = llr.size // self.N_cw
        packed = np.empty((B, (self.K + 7) // 8), np.uint8)
        iters = np.empty(B, np.int32) if want_iters else None
        check(self._lib.nrldpc_pool_decode_packed(self._p, _ptr(llr), B, _ptr(packed), _ptr(iters)))
        return (packed, iters) if want_iters else packed

    def set_layers(self, n_layers):
        """nrldpc_pool_set_layers (LAYERS_AUTO: found once per call over the whole batch)."""
        check(self._lib.nrldpc_pool_set_layers(self._p, int(n_layers)))

    def decode_dev(self, d_llr, batch, d_hard, d_iters=None):
        """nrldpc_pool_decode_dev: shard i decodes batch[i] codewords at device address d_llr[i] (memory of
        device_ids[i]) into d_hard[i]; returns when every shard's stream is idle.  No host copies.  Work already queued on a
        shard's device when the call is made (e.g. the torch kernel still producing d_llr[i]) completes before the decoder
        starts: each shard synchronizes its device first (nrldpc.h)."""
        n = len(self.device_ids)
        if not (len(d_llr) == len(batch) == len(d_hard) == n):
            raise NRLDPCError("one entry per shard expected")
        vp = C.c_void_p
        a_llr = (vp * n)(*[vp(int(x)) for x in d_llr])
        a_hard = (vp * n)(*[vp(int(x)) for x in d_hard])
        a_b = (C.c_int32 * n)(*[int(b) for b in batch])
        a_it = (vp * n)(*[vp(int(x) if x else None) for x in d_iters]) if d_iters is not None else None
        check(self._lib.nrldpc_pool_decode_dev(self._p, a_llr, a_b, a_hard, a_it))

    def set_timing(self, on=True):
        """nrldpc_pool_set_timing: event pairs around every shard's decode kernel, on the shard's own launch stream."""
        check(self._lib.nrldpc_pool_set_timing(self._p, int(on)))

    def last_kernel_ms(self):
        """Kernel time of every shard's last launch, ms (nrldpc_pool_last_kernel_ms; 0 for a shard without work)."""
        out = (C.c_float * len(self.device_ids))()
        check(self._lib.nrldpc_pool_last_kernel_ms(self._p, out))
        return [float(v) for v in out]

    def last_split(self):
        """Codewords each shard decoded in the last call (uneven under early termination: faster shards pull more)."""
        out = (C.c_int32 * len(self.device_ids))()
        check(self._lib.nrldpc_pool_last_split(self._p, out))
        return list(out)

    def close(self):
        if getattr(self, "_p", None) and self._p.value:
            self._lib.nrldpc_pool_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiCall:
    """The argument arrays of one nrldpc_decode_multi_dev call, built once: what a C caller holds anyway.  call(stream) is the
    library call and nothing else (marshalling 5 x 100 values through ctypes costs about as much as the call's host side)."""

    def __init__(self, codecs, d_llr, batch, d_hard, d_iters=None):
        n = self.n = len(codecs)
        vp = C.c_void_p
        self.codecs = list(codecs)  # keeps the handles alive
        self.hs = (vp * n)(*[c._h for c in codecs])
        self.llr = (vp * n)(*[int(x) for x in d_llr])
        self.hard = (vp * n)(*[int(x) for x in d_hard])
        self.its = (vp * n)(*[int(x) if x else None for x in d_iters]) if d_iters is not None else None
        self.bt = (C.c_int32 * n)(*[int(b) for b in batch])
        self.fn = load().nrldpc_decode_multi_dev

    def __call__(self, stream=0):
        check(self.fn(self.n, self.hs, self.llr, self.bt, self.hard, self.its, C.c_void_p(stream)))


def decode_multi_dev(codecs, d_llr, batch, d_hard, d_iters=None, stream=0):
    """One launch per base graph, LLR type and workgroup class for a mix of configurations (nrldpc_decode_multi_dev):
    codecs[i] decodes batch[i] codewords at device address d_llr[i] into d_hard[i] (and d_iters[i])."""
    MultiCall(codecs, d_llr, batch, d_hard, d_iters)(stream)


def rate_recover_dev(p, d_g_tilde, n_tb, d_harq, d_cw_llr, out_dtype=LLR_F32, stream=0):
    """nrldpc_rate_recover_dev on raw device addresses (p: NRLDPC parameter object or TbParams)."""
    t = p if isinstance(p, TbParams) else tb_params(p)
    check(load().nrldpc_rate_recover_dev(C.byref(t), _ptr(d_g_tilde), int(n_tb), _ptr(d_harq), _ptr(d_cw_llr),
                                         int(out_dtype), C.c_void_p(stream)))


def crc_check_dev(p, d_c_hat, n_tb, d_b_hat, d_ok, d_cb_pass=None, stream=0):
    """nrldpc_crc_check_dev on raw device addresses."""
    t = p if isinstance(p, TbParams) else tb_params(p)
    check(load().nrldpc_crc_check_dev(C.byref(t), _ptr(d_c_hat), int(n_tb), _ptr(d_b_hat), _ptr(d_ok),
                                      _ptr(d_cb_pass), C.c_void_p(stream)))


def crc_check_harq_dev(p, d_c_hat, n_tb, d_b_hat, d_ok, d_cb_pass, cbgti_flags=None, keep_b_hat=True, stream=0):
    """nrldpc_crc_check_harq_dev: the CRC stage with the reference's HARQ / CBGTI state (NRLDPCDecoder.m:283-316,337).
    d_b_hat and d_cb_pass are in/out device buffers owned by the caller; cbgti_flags is a host sequence of C flags."""
    t = p if isinstance(p, TbParams) else tb_params(p)
    flags = None
    if cbgti_flags is not None:
        flags = (C.c_uint8 * t.C)(*[1 if f else 0 for f in cbgti_flags])
    check(load().nrldpc_crc_check_harq_dev(C.byref(t), _ptr(d_c_hat), int(n_tb), _ptr(d_b_hat), _ptr(d_ok),
                                           _ptr(d_cb_pass), flags, int(bool(keep_b_hat)), C.c_void_p(stream)))


def awgn_llr_dev(d_g, n_bits, Q_m, EsN0_dB, seed, first_symbol, d_g_tilde, stream=0):
    """nrldpc_awgn_llr_dev: modulation + AWGN + exact LLRs in one kernel (plot_BLER_vs_SNR.m:130-132)."""
    check(load().nrldpc_awgn_llr_dev(_ptr(d_g), int(n_bits), int(Q_m), float(EsN0_dB), int(seed), int(first_symbol),
                                     _ptr(d_g_tilde), C.c_void_p(stream)))


def payload_bits_dev(seed, first_block, n_tb, A, d_a, stream=0):
    """nrldpc_payload_bits_dev: the Monte-Carlo loop's payload draw (plot_BLER_vs_SNR.m:118) for n_tb blocks in one kernel."""
    check(load().nrldpc_payload_bits_dev(int(seed) % (1 << 64), int(first_block), int(n_tb), int(A), _ptr(d_a), C.c_void_p(stream)))


def crc_attach_dev(p, d_a, n_tb, d_c, stream=0):
    t = p if isinstance(p, TbParams) else tb_params(p)
    check(load().nrldpc_crc_attach_dev(C.byref(t), _ptr(d_a), int(n_tb), _ptr(d_c), C.c_void_p(stream)))


def rate_match_dev(p, d_cw, n_tb, d_g, stream=0):
    t = p if isinstance(p, TbParams) else tb_params(p)
    check(load().nrldpc_rate_match_dev(C.byref(t), _ptr(d_cw), int(n_tb), _ptr(d_g), C.c_void_p(stream)))


def quantise_llr(llr, llr_scale=8):
    """The int8 form nrldpc_decode puts on the wire for large host batches (include/nrldpc.h): (q, saw_negative_infinity)."""
    llr = np.ascontiguousarray(llr)
    kind = {np.dtype(np.float32): LLR_F32, np.dtype(np.float16): LLR_F16, np.dtype(np.float64): LLR_F64}[llr.dtype]
    q = np.empty(llr.shape, np.int8)
    neg = load().nrldpc_quantise_llr(_ptr(q), _ptr(llr), llr.size, kind, int(llr_scale))
    return q, bool(neg)


def count_layers(bg, Z, llr):
    """nrldpc_count_layers: what LAYERS_AUTO finds for host LLRs [batch][ncols*Z] (f32 / f16 / f64).  No device needed."""
    llr = np.ascontiguousarray(llr)
    cols = {1: 68, 2: 52}[int(bg)]
    if llr.size % (cols * int(Z)):
        raise NRLDPCError("llr should hold a whole number of codewords")
    n = load().nrldpc_count_layers(int(bg), int(Z), _ptr(llr), llr.size // (cols * int(Z)), _NP2DT[llr.dtype])
    if n < 0:
        raise NRLDPCError((load().nrldpc_last_error() or b"").decode())
    return n


def default_rule(bg, n_layers=0):
    """(alpha, beta) nrldpc_create uses when cfg.alpha == 0 (beta in LLR units)."""
    a, b = C.c_float(), C.c_float()
    check(load().nrldpc_default_rule(int(bg), int(n_layers), C.byref(a), C.byref(b)))
    return float(a.value), float(b.value)


def set_index(Z):
    return load().nrldpc_set_index(int(Z))


def lifting_size(K_b, K_prime):
    return load().nrldpc_lifting_size(int(K_b), int(K_prime))
