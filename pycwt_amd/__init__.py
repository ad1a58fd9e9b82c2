"""pycwt_amd -- MI355X-native engine for pycwt's FFT-convolution CWT hot path.

``cwt`` / ``icwt`` keep the signatures of regeirk/pycwt (pycwt/__init__.py:85-90 re-exports them
from pycwt/wavelet.py); the arithmetic runs in hand-written HIP kernels (pycwt_amd/csrc) reached
through the C ABI of include/cwt_hip.h.  Importing this package does not need a GPU; calling
``cwt`` / ``icwt`` does, and fails loudly without one.
"""
from . import helpers, mothers
from .helpers import ar1, ar1_spectrum, fft, fft_kwargs, find, get_cache_dir, rednoise
from .mothers import DOG, MexicanHat, Morlet, Paul
from .wavelet import (DeviceCoherence, DeviceTransform, cwt, cwt_batch, cwt_device, icwt, release_scratch, set_tolerance,
                      significance, wct, wct_device, wct_significance, xwt, xwt_device)

__version__ = "0.1.0"
__all__ = ["cwt", "cwt_batch", "cwt_device", "DeviceTransform", "DeviceCoherence", "wct_device", "xwt_device", "icwt", "set_tolerance", "release_scratch", "significance", "xwt", "wct", "wct_significance", "Morlet", "Paul", "DOG",
           "MexicanHat", "ar1", "ar1_spectrum", "rednoise", "find", "get_cache_dir", "helpers", "mothers", "fft", "fft_kwargs"]
