"""pycwt_amd -- MI355X-native engine for pycwt's FFT-convolution CWT hot path.

``cwt`` / ``icwt`` keep the signatures of regeirk/pycwt (pycwt/__init__.py:85-90 re-exports them
from pycwt/wavelet.py); the arithmetic runs in hand-written HIP kernels (pycwt_amd/csrc) reached
through the C ABI of include/cwt_hip.h.  Importing this package does not need a GPU; calling
``cwt`` / ``icwt`` does, and fails loudly without one.
"""
from .mothers import DOG, MexicanHat, Morlet, Paul
from .wavelet import cwt, icwt

__version__ = "0.1.0"
__all__ = ["cwt", "icwt", "Morlet", "Paul", "DOG", "MexicanHat"]
