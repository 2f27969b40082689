"""MI355X-native subband-beamforming engine (btk2.0-compatible hot path).

Only what the hot path needs lives here: csrc/ (HIP kernels + C-ABI), engine (block-level
front-end over device tensors) and btk20 (host-side mirror of the reference's node /
iterator interface).  Importing the package does not load the HIP library; the first call
that needs it does, and fails loudly if it has not been built.
"""
__version__ = "0.1.0"
