"""MI355X-native belief-propagation LDPC decoder behind LDPCDecoders.jl's BP API.

The directory is called ``ldpcdecoders.jl_amd``; because of the dot it is
imported through the loader module ``ldpcdecoders_jl_amd`` at the repo root::

    import ldpcdecoders_jl_amd as ldpc
    dec = ldpc.BeliefPropagationDecoder(H, 0.01, 50)
    guess, ok = ldpc.decode_(dec, syndrome)

Exports follow src/LDPCDecoders.jl:12-18 for the hot path (``!`` -> ``_``).
"""
from . import _capi, codes  # noqa: F401
from ._capi import LdpcError, build  # noqa: F401
from .codes import load_pcm, parity_check_matrix, save_pcm  # noqa: F401
from .decoder import (  # noqa: F401
    AbstractDecoder,
    BeliefPropagationDecoder,
    BeliefPropagationScratchSpace,
    batchdecode_,
    decode_,
    reset_,
    syndrome_bytes,
)

from .osd import BeliefPropagationOSDDecoder, OSDPostProcessor  # noqa: F401,E402
from .bpots import BPOTSDecoder  # noqa: F401,E402

__all__ = [
    "BeliefPropagationOSDDecoder", "OSDPostProcessor", "BPOTSDecoder",
    "decode_", "batchdecode_", "reset_", "AbstractDecoder", "BeliefPropagationDecoder",
    "BeliefPropagationScratchSpace", "parity_check_matrix", "save_pcm", "load_pcm",
    "LdpcError", "build", "codes", "syndrome_bytes",
]
