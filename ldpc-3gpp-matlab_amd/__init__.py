"""MI355X-native NR LDPC decode engine (hot path of robmaunder/ldpc-3gpp-matlab).

The package directory name contains hyphens; import it with
    importlib.import_module("ldpc-3gpp-matlab_amd")
(tests/conftest.py and __graft_entry__.py do this and alias it as `nrldpc_amd`).
"""
from . import _capi, chain
from ._capi import (Codec, CodecPool, NRLDPCError, awgn_llr_dev, UnsupportedParameters, crc_attach_dev, crc_check_dev, crc_check_harq_dev, lifting_size, load,
                    rate_match_dev, rate_recover_dev, set_index, tb_params, decode_multi_dev, MultiCall)
from .decoder import NRLDPCDecoder, default_rule
from .encoder import NRLDPCEncoder
from .nrldpc import NRLDPC, get_3gpp_crc_polynomial

__all__ = ["Codec", "CodecPool", "NRLDPC", "NRLDPCDecoder", "NRLDPCEncoder", "NRLDPCError", "UnsupportedParameters",
           "awgn_llr_dev", "chain", "crc_attach_dev", "crc_check_dev", "crc_check_harq_dev", "decode_multi_dev", "MultiCall", "rate_match_dev", "default_rule", "rate_recover_dev", "tb_params", "get_3gpp_crc_polynomial", "lifting_size", "load", "set_index", "_capi"]
