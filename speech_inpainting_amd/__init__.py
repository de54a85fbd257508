"""MI355X-native I_ea speech-inpainting predict path (HuBERT encoder -> codeword splice -> HiFi-GAN).

Host side in Python (checkpoint loading, config, glue) over a C-ABI shared library of hand-written
HIP kernels for gfx950 (``csrc/`` -> ``libsi_hip.so``, declared in ``include/si_hip.h``).
There is no CPU fallback: any compute entry point raises if the library is missing.
"""
__version__ = "0.1.0"
