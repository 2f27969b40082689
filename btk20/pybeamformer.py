"""btk20.pybeamformer -> distant_speech_recognition_amd.pybeamformer (the reference installs lib/pybeamformer.py under btk20)"""
from distant_speech_recognition_amd.pybeamformer import *      # noqa: F401,F403
from distant_speech_recognition_amd.pybeamformer import calc_delays, calc_la_delays, calc_array_manifold_f, calc_blocking_matrix      # noqa: F401
