"""NRLDPCDecoder: host-side mirror of the reference's decoder System object (NRLDPCDecoder.m).

Same properties (I_HARQ Nontunable, iterations), same step()/reset() protocol, same HARQ state
(d_tilde_buffer, b_hat_buffer, code_block_CRC_passed; NRLDPCDecoder.m:64-95, 343-356) and the same
output convention: A decoded bits, or an empty array when a CRC fails (:336-339).

Stage 4 (LDPC_coding, :245-268) is the hot path: all C code blocks of the transport block go to the
GPU as one batch through nrldpc_decode (the reference decodes them one by one, :257).  The decoding
algorithm is the build's layered normalised min-sum, not comm.LDPCDecoder's flooding sum-product
(see DESIGN.md); `iterations` keeps the reference's meaning of 'MaximumIterationCount' and the
parity-check early stop of :120 stays on.

step_batch(g_tilde[B][G]) decodes B transport blocks at once (HARQ state is then kept per row); it
returns (a_hat[B][A], ok[B]) where ok[b] is False exactly when the reference would return [].
"""
import numpy as np

from . import chain
from ._capi import Codec, NRLDPCError
from .nrldpc import NRLDPC


def default_rule(bg, n_layers=0):
    """(alpha, beta) of the check-node rule the library applies when none is given: it lives in the C ABI
    (nrldpc_default_rule / nrldpc_create with cfg.alpha == 0), so that a MEX gateway gets the same decoder."""
    from ._capi import default_rule as _dr
    return _dr(bg, n_layers)


class NRLDPCDecoder(NRLDPC):
    _NONTUNABLE = NRLDPC._NONTUNABLE + ("I_HARQ",)
    _TUNABLE = NRLDPC._TUNABLE + ("iterations",)

    def __init__(self, device_id=0, alpha=None, llr_scale=0, prune_layers=True, beta=0.0, crc_stop=False, **kw):
        """prune_layers: True = the active rows from the object's parameters (active_layers(), kept at the maximum seen while
        HARQ state is pending); "auto" = NRLDPC_LAYERS_AUTO, read off every step's LLRs by the library (what the MEX gateway
        does: it sees cw_tilde only, NRLDPCDecoder.m:265); False = every row, as the reference."""
        self._I_HARQ = 0        # NRLDPCDecoder.m:34
        self._iterations = 50   # NRLDPCDecoder.m:41
        super().__init__(**kw)
        self._device_id, self._alpha, self._llr_scale = device_id, alpha, llr_scale
        self._beta = beta  # read only with an explicit alpha (nrldpc_cfg.beta)
        self._prune = prune_layers
        self._crc_stop = bool(crc_stop)  # also stop a code block when its CRC holds (nrldpc_cfg.early_term = 2); the reference: False
        self._codec = None
        self._codec_layers = None
        self._nb = 1
        self.d_tilde_buffer = None
        self.b_hat_buffer = None
        self.code_block_CRC_passed = None
        self._layers_seen = 4
        self.last_iterations = None

    I_HARQ = property(lambda s: s._I_HARQ)

    @I_HARQ.setter
    def I_HARQ(self, v):
        self._set_nontunable("I_HARQ", int(v))

    iterations = property(lambda s: s._iterations)

    @iterations.setter
    def iterations(self, v):
        # read at setup only, as in the reference (NRLDPCDecoder.m:120)
        self._iterations = int(v)

    # -- System-object protocol ------------------------------------------------------------------
    def _make_codec(self, n_layers):
        """One codec for the object's lifetime (setupImpl, NRLDPCDecoder.m:120); the active layer count is a property of the
        call (nrldpc_set_layers, ABI revision 5): a change of rate between steps -- G and rv_id are tunable, NRLDPC.m:51-85 --
        costs no device work."""
        if self._codec is None:
            self._codec = Codec(self.BG, self.Z_c, max_iter=self._setup_iterations, n_layers=n_layers,
                                early_term=True, alpha=self._alpha or 0.0, beta=self._beta, llr_scale=self._llr_scale,
                                llr_dtype=np.float32, device_id=self._device_id,
                                crc=self.code_block_check() if self._crc_stop else None)
        else:
            self._codec.set_layers(n_layers)
        self._codec_layers = n_layers

    def _setup(self):  # NRLDPCDecoder.m:107-130
        self.validate()
        self._setup_iterations = self._iterations
        object.__setattr__(self, "_locked", True)
        self.reset()

    def reset(self, rows=None):  # NRLDPCDecoder.m:343-356 (rows: only those transport blocks of a batch)
        if not self._locked:
            return
        nb = self._nb
        if rows is None or self.d_tilde_buffer is None or self.d_tilde_buffer.shape[0] != nb:
            self.d_tilde_buffer = np.zeros((nb, self.C, self.N_cb), np.float64)
            self.b_hat_buffer = np.zeros((nb, self.B), np.uint8)
            self.code_block_CRC_passed = np.zeros((nb, self.C), np.uint8)
        else:
            self.d_tilde_buffer[rows] = 0
            self.b_hat_buffer[rows] = 0
            self.code_block_CRC_passed[rows] = 0
        self._layers_seen = 4

    def release(self):
        if self._codec is not None:
            self._codec.close()
            self._codec = None
        super().release()

    def __call__(self, g_tilde):
        return self.step(g_tilde)

    def step(self, g_tilde):
        """g_tilde: G LLRs (positive = bit 0) -> a_hat: A bits, or an empty array on CRC failure."""
        g_tilde = np.asarray(g_tilde, np.float64)
        if g_tilde.ndim == 2 and g_tilde.shape[1] == 1:
            g_tilde = g_tilde[:, 0]
        if g_tilde.ndim != 1 or g_tilde.size != self.G:
            raise NRLDPCError("g_tilde should be a column vector of length G.")
        a_hat, ok = self.step_batch(g_tilde[None, :])
        return a_hat[0] if ok[0] else np.zeros(0, np.uint8)

    def step_batch(self, g_tilde):
        g_tilde = np.asarray(g_tilde, np.float64)
        if g_tilde.ndim != 2 or g_tilde.shape[1] != self.G:
            raise NRLDPCError("g_tilde should be a column vector of length G.")
        if not self._locked:
            self._nb = g_tilde.shape[0]
            self._setup()
        else:
            self.validate()
            if g_tilde.shape[0] != self._nb:
                # A batch of n transport blocks stands for n reference objects.  With I_HARQ ~= 0 those objects carry
                # soft buffers and CRC flags from step to step (NRLDPCDecoder.m:236-239, 286-314): a different batch size
                # with such state pending is an error, never a silent drop.  With I_HARQ == 0 nothing of a step is
                # used by the next one except the sticky pass flags, which say nothing about a different set of
                # transport blocks: a new batch size simply starts a new set.
                if self.I_HARQ != 0 and ((self.code_block_CRC_passed is not None and self.code_block_CRC_passed.any()) or
                                         (self.d_tilde_buffer is not None and self.d_tilde_buffer.any())):
                    raise NRLDPCError("batch size changed from %d to %d transport blocks with HARQ state pending "
                                      "(I_HARQ ~= 0); call reset() first." % (self._nb, g_tilde.shape[0]))
                self._nb = g_tilde.shape[0]
                self.reset()
        d_tilde = self.rate_recover(g_tilde)
        c_hat = self.LDPC_coding(d_tilde)
        b_hat = self.code_block_segmentation(c_hat)
        return self.crc_calculation(b_hat)

    # -- stages (all on [B][...] arrays) -----------------------------------------------------------
    def rate_recover(self, g_tilde):
        """Stages 1-3: de-concatenate, de-interleave, soft-combining bit de-selection and the HARQ
        buffer (NRLDPCDecoder.m:143-242).  Filler positions are marked NaN as in the reference."""
        g_tilde = np.asarray(g_tilde, np.float64)
        single = g_tilde.ndim == 1
        if single:
            g_tilde = g_tilde[None]
        nb = g_tilde.shape[0]
        N_, Z, K_, Kp, N_cb = self.N, self.Z_c, self.K, int(self.K_prime), self.N_cb
        d = np.zeros((nb, self.C, N_), np.float64)
        rows = np.arange(nb)[:, None]
        for r, (off, dpos, fpos) in enumerate(chain.g_to_d_maps(self)):
            if dpos.size:  # repetition soft-combines (:229-231); add.at keeps the reference's k order
                np.add.at(d, (rows, r, dpos[None, :]), g_tilde[:, off + fpos])
        if self.I_HARQ != 0 and self.d_tilde_buffer is not None and self.d_tilde_buffer.shape[0] == nb:  # :236-239
            d[:, :, :N_cb] += self.d_tilde_buffer
            self.d_tilde_buffer = d[:, :, :N_cb].copy()
        d[:, :, max(Kp - 2 * Z, 0): K_ - 2 * Z] = np.nan  # :224
        return d[0] if single else d

    def LDPC_coding(self, d_tilde):  # NRLDPCDecoder.m:245-268
        Z, K_ = self.Z_c, self.K
        nb, C_ = d_tilde.shape[0], self.C
        cw = np.concatenate([np.zeros((nb, C_, 2 * Z)), d_tilde], axis=2)  # :262
        filler = np.isnan(cw[0, 0, :K_])
        cw[np.isnan(cw)] = np.inf  # :264
        n_layers = 0
        if self._prune == "auto":
            n_layers = -1
        elif self._prune:
            act = self.active_layers()
            self._layers_seen = max(self._layers_seen, act) if self.I_HARQ else act
            n_layers = self._layers_seen
        want = n_layers if n_layers else (46 if self.BG == 1 else 42)
        if self._codec is None or self._codec_layers != want:
            self._make_codec(want)
        hard, iters = self._codec.decode(cw.astype(np.float32).reshape(nb * C_, -1), want_iters=True)  # :265
        self.last_layers = self._codec.last_layers()
        self.last_iterations = iters.reshape(nb, C_)
        c_hat = hard.reshape(nb, C_, K_).astype(np.float64)
        c_hat[:, :, filler] = np.nan  # :266
        return c_hat

    def code_block_segmentation(self, c_hat):  # NRLDPCDecoder.m:271-318
        C_, Kp, L = self.C, int(self.K_prime), self.code_block_L
        nb = c_hat.shape[0]
        flags = self.CBGTI_flags
        b_hat = self.b_hat_buffer.copy() if self.I_HARQ != 0 else np.zeros((nb, self.B), np.uint8)
        passed = self.code_block_CRC_passed.copy()
        s = 0
        for r in range(C_):
            blk = c_hat[:, r, :Kp].astype(np.uint8)
            failed = np.zeros(nb, bool)
            if C_ > 1:  # CB-CRC only when segmented (:298-301)
                failed = chain.crc_bits(blk, self.code_block_CRC_polynomial, L).any(axis=1)
            good = ~failed & (flags[r] == 1)
            b_hat[good, s: s + Kp - L] = blk[good, : Kp - L]
            passed[good, r] = 1
            s += Kp - L
        if self.I_HARQ != 0:
            self.b_hat_buffer = b_hat.copy()
        self.code_block_CRC_passed = passed
        return b_hat

    def crc_calculation(self, b_hat):  # NRLDPCDecoder.m:321-340
        failed = chain.crc_bits(b_hat, self.transport_block_CRC_polynomial, self.transport_block_L).any(axis=1)
        ok = ~failed & (self.code_block_CRC_passed != 0).all(axis=1)
        return b_hat[:, : self.A].copy(), ok
