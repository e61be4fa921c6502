"""Receive chain with every per-bit stage on the GPU: demodulator LLRs g_tilde in HBM -> rate recovery
(nrldpc_rate_recover_dev, row N1) -> LDPC decode (nrldpc_decode_dev, the hot path) -> CRC stages
(nrldpc_crc_check_dev, row N2) -> a_hat + ok flags.  The reference runs these as six interpreted stages
per transport block (NRLDPCDecoder.m:133-140); here one transport-block batch is three kernel launches
with no host loop and no PCIe traffic in between.  torch is used for device memory only.
"""
import numpy as np

from ._capi import LLR_F16, LLR_F32, Codec, NRLDPCError, crc_check_harq_dev, rate_recover_dev, tb_params
from .nrldpc import NRLDPC


class _Derived:
    """What a chain step needs of the parameter object's Dependent properties (NRLDPC.m:297-543), computed once per setting of the
    settable ones: the mirror re-derives every Dependent property on every access, as the reference does, and a step reads a dozen
    of them -- 1 ms of Python per step, a fifth of a small code's Monte-Carlo sweep."""
    __slots__ = ("t", "C", "Z_c", "N", "K", "B", "N_cb", "A", "G", "BG", "act", "flags")

    def __init__(self, p):
        self.t = tb_params(p)
        self.C, self.Z_c, self.N, self.K, self.B, self.N_cb, self.A, self.G, self.BG = p.C, p.Z_c, p.N, p.K, p.B, p.N_cb, p.A, p.G, p.BG
        self.act = p.active_layers()
        self.flags = list(p.CBGTI_flags)


def _derived(chain):
    p = chain.p
    key = (p.BG, p.A, p.I_LBRM, p.TBS_LBRM, p.rv_id, p.G, p.Q_m, p.N_L, tuple(p.CBGTI))  # every settable property (NRLDPC.m:28-84)
    if getattr(chain, "_dkey", None) != key:
        chain._dval, chain._dkey = _Derived(p), key
    return chain._dval


class DeviceDecodeChain:
    """Batched NRLDPCDecoder.step on device tensors.  `params` is an NRLDPC parameter object (or any
    NRLDPCDecoder).  The DiscreteState of the reference (NRLDPCDecoder.m:64-95) lives in HBM, one row per transport
    block of the batch: d_tilde_buffer (`harq`, only when I_HARQ), b_hat_buffer (`b_hat`) and the sticky
    code_block_CRC_passed flags (`cb_pass`); reset() clears them (:343-356).  CBGTI of `params` is honoured (:304)."""

    def __init__(self, params: NRLDPC, iterations=50, I_HARQ=0, alpha=None, llr_scale=0, prune_layers=True,
                 llr_dtype=np.float16, device_id=0, beta=0.0, crc_stop=False):
        import torch
        self.torch = torch
        params.validate()
        self.p = params
        self.crc_stop = bool(crc_stop)  # nrldpc_cfg.early_term = 2: a code block also stops when its CRC holds
        self.iterations, self.I_HARQ = int(iterations), int(I_HARQ)
        self.alpha, self.beta, self.llr_scale, self.prune = alpha, beta, llr_scale, prune_layers
        self.llr_dtype = np.dtype(llr_dtype)
        self.dev = torch.device("cuda", device_id)
        self.device_id = device_id
        self._codec, self._codec_layers, self._codec_code = None, None, None
        self._layers_seen = 4
        self.harq = self.b_hat = self.cb_pass = None

    def reset(self):
        """reset(hDec): clears d_tilde_buffer, b_hat_buffer and code_block_CRC_passed (NRLDPCDecoder.m:343-356)."""
        self.harq = self.b_hat = self.cb_pass = None  # re-allocated zeroed (for any batch size) by the next step
        self._layers_seen = 4

    def close(self):
        if self._codec is not None:
            self._codec.close()
            self._codec = None

    def _codec_for(self, n_layers):
        """One codec for the chain's lifetime; the layer count is a property of the call (nrldpc_set_layers, ABI revision 5)."""
        if self._codec is not None and self._codec_code != (self.p.BG, self.p.Z_c):
            # A changed on a plain parameter object between steps: Z_c (or BG) moved, and a codec built for the old pair would read
            # and write tensors sized for the new one out of bounds -- a new code is a new codec
            self._codec.close()
            self._codec = None
        if self._codec is None:
            self._codec_code = (self.p.BG, self.p.Z_c)
            self._codec_layers = None
            self._codec = Codec(self.p.BG, self.p.Z_c, max_iter=self.iterations, n_layers=n_layers, early_term=True,
                                alpha=self.alpha or 0.0, beta=self.beta, llr_scale=self.llr_scale, llr_dtype=self.llr_dtype, device_id=self.device_id,
                                crc=self.p.code_block_check() if self.crc_stop else None)
        elif self._codec_layers != n_layers:
            self._codec.set_layers(n_layers)
        self._codec_layers = n_layers
        return self._codec

    def step(self, g_tilde):
        """g_tilde: torch float32 tensor [n_tb][G] on the device (positive = bit 0).
        Returns (a_hat uint8 [n_tb][A], ok bool [n_tb], iters int32 [n_tb][C]) as device tensors."""
        torch, p = self.torch, self.p
        if g_tilde.dim() != 2 or g_tilde.shape[1] != p.G or g_tilde.dtype != torch.float32 or not g_tilde.is_cuda:
            raise NRLDPCError("g_tilde should be a float32 device tensor of shape [n_tb][G].")
        if g_tilde.device != self.dev:
            raise NRLDPCError("g_tilde lives on %s, this chain on %s." % (g_tilde.device, self.dev))
        with torch.cuda.device(self.dev):  # the stateless stage kernels launch on the current HIP device
            return self._step(g_tilde.contiguous())

    def _step(self, g_tilde):
        torch, d = self.torch, _derived(self)
        n_tb, C_ = g_tilde.shape[0], d.C
        t = d.t
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        ncwz = 2 * d.Z_c + d.N
        if self.cb_pass is not None and self.cb_pass.shape[0] != n_tb:
            # see NRLDPCDecoder.step_batch: only HARQ state that is really pending makes a new batch size an error
            # (the check synchronises, but only on this rare path); without it a new batch size starts a new set
            if self.I_HARQ and (bool(self.cb_pass.any()) or bool(self.harq.any())):
                raise NRLDPCError("batch size changed from %d to %d transport blocks with HARQ state pending "
                                  "(I_HARQ ~= 0); call reset() first." % (self.cb_pass.shape[0], n_tb))
            self.cb_pass = self.b_hat = self.harq = None
        if self.cb_pass is not None and (self.cb_pass.shape[1] != C_ or self.b_hat.shape[1] != d.B or
                                         (self.harq is not None and self.harq.shape[2] != d.N_cb)):
            # another code on the same parameter object (A changed between steps): buffers of the old one's sizes hold nothing that
            # belongs to these transport blocks -- a new set, as after reset() (the codec itself: _codec_for)
            self.cb_pass = self.b_hat = self.harq = None
            self._layers_seen = 4
        if self.cb_pass is None:
            self.cb_pass = torch.zeros((n_tb, C_), dtype=torch.int32, device=self.dev)
            self.b_hat = torch.zeros((n_tb, d.B), dtype=torch.uint8, device=self.dev)
            if self.I_HARQ:
                self.harq = torch.zeros((n_tb, C_, d.N_cb), dtype=torch.float32, device=self.dev)
        tdt = torch.float16 if self.llr_dtype == np.float16 else torch.float32
        cw_llr = torch.empty((n_tb * C_, ncwz), dtype=tdt, device=self.dev)
        rate_recover_dev(t, g_tilde.data_ptr(), n_tb, self.harq.data_ptr() if self.I_HARQ else None,
                         cw_llr.data_ptr(), LLR_F16 if tdt == torch.float16 else LLR_F32, stream)
        rows = 46 if d.BG == 1 else 42
        n_layers = rows
        if self.prune:
            act = d.act
            self._layers_seen = max(self._layers_seen, act) if self.I_HARQ else act
            n_layers = self._layers_seen
        codec = self._codec_for(n_layers)
        c_hat = torch.empty((n_tb * C_, d.K), dtype=torch.uint8, device=self.dev)
        iters = torch.empty(n_tb * C_, dtype=torch.int32, device=self.dev)
        codec.decode_dev(cw_llr.data_ptr(), n_tb * C_, c_hat.data_ptr(), iters.data_ptr(), None, stream)
        ok = torch.empty(n_tb, dtype=torch.int32, device=self.dev)
        crc_check_harq_dev(t, c_hat.data_ptr(), n_tb, self.b_hat.data_ptr(), ok.data_ptr(), self.cb_pass.data_ptr(),
                           d.flags, self.I_HARQ != 0, stream)
        return self.b_hat[:, : d.A].clone(), ok != 0, iters.view(n_tb, C_)


class DeviceEncodeChain:
    """Batched NRLDPCEncoder.step on device tensors: CRC attachment + segmentation
    (nrldpc_crc_attach_dev), LDPC encoding (nrldpc_encode_dev), rate matching (nrldpc_rate_match_dev).
    The reference runs these as six interpreted stages per transport block (NRLDPCEncoder.m:60-67)."""

    def __init__(self, params: NRLDPC, device_id=0):
        import torch
        from ._capi import crc_attach_dev, rate_match_dev
        self.torch, self._crc_attach, self._rate_match = torch, crc_attach_dev, rate_match_dev
        params.validate()
        self.p = params
        self.dev = torch.device("cuda", device_id)
        self._device_id = device_id
        self._codec_code = (params.BG, params.Z_c)
        self._codec = Codec(params.BG, params.Z_c, max_iter=1, llr_dtype=np.float32, device_id=device_id)

    def close(self):
        if self._codec is not None:
            self._codec.close()
            self._codec = None

    def step(self, a):
        """a: torch uint8 tensor [n_tb][A] on the device -> g: uint8 [n_tb][G]."""
        torch, p = self.torch, self.p
        if a.dim() != 2 or a.shape[1] != p.A or a.dtype != torch.uint8 or not a.is_cuda:
            raise NRLDPCError("a should be a uint8 device tensor of shape [n_tb][A].")
        if a.device != self.dev:
            raise NRLDPCError("a lives on %s, this chain on %s." % (a.device, self.dev))
        with torch.cuda.device(self.dev):
            return self._step(a.contiguous())

    def _step(self, a):
        torch, d = self.torch, _derived(self)
        n_tb = a.shape[0]
        t = d.t
        s = torch.cuda.current_stream(self.dev).cuda_stream
        c = torch.empty((n_tb * d.C, d.K), dtype=torch.uint8, device=self.dev)
        self._crc_attach(t, a.data_ptr(), n_tb, c.data_ptr(), s)
        cw = torch.empty((n_tb * d.C, 2 * d.Z_c + d.N), dtype=torch.uint8, device=self.dev)
        if self._codec_code != (d.BG, d.Z_c):  # A changed on the parameter object: another code, another codec (see DeviceDecodeChain)
            self._codec.close()
            self._codec_code = (d.BG, d.Z_c)
            self._codec = Codec(d.BG, d.Z_c, max_iter=1, llr_dtype=np.float32, device_id=self._device_id)
        self._codec.encode_dev(c.data_ptr(), n_tb * d.C, cw.data_ptr(), s)
        g = torch.empty((n_tb, d.G), dtype=torch.uint8, device=self.dev)
        self._rate_match(t, cw.data_ptr(), n_tb, g.data_ptr(), s)
        return g
