"""MI355X-native NR LDPC decode engine (hot path of robmaunder/ldpc-3gpp-matlab).

The package directory name contains hyphens; import it with
    importlib.import_module("ldpc-3gpp-matlab_amd")
(tests/conftest.py and __graft_entry__.py do this and alias it as `nrldpc_amd`).
"""
from . import _capi, chain
from ._capi import Codec, NRLDPCError, UnsupportedParameters, lifting_size, load, set_index
from .decoder import NRLDPCDecoder, default_alpha
from .encoder import NRLDPCEncoder
from .nrldpc import NRLDPC, get_3gpp_crc_polynomial

__all__ = ["Codec", "NRLDPC", "NRLDPCDecoder", "NRLDPCEncoder", "NRLDPCError", "UnsupportedParameters",
           "chain", "default_alpha", "get_3gpp_crc_polynomial", "lifting_size", "load", "set_index", "_capi"]
