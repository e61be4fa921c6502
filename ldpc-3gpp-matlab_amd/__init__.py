"""MI355X-native NR LDPC decode engine (hot path of robmaunder/ldpc-3gpp-matlab).

The package directory name contains hyphens; import it with
    importlib.import_module("ldpc-3gpp-matlab_amd")
(tests/conftest.py and __graft_entry__.py do this and alias it as `nrldpc_amd`).
"""
from . import _capi
from ._capi import Codec, NRLDPCError, UnsupportedParameters, lifting_size, load, set_index

__all__ = ["Codec", "NRLDPCError", "UnsupportedParameters", "lifting_size", "load", "set_index", "_capi"]
