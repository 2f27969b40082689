"""btk20.beamformer -> distant_speech_recognition_amd.btk20.beamformer"""
from distant_speech_recognition_amd.btk20.beamformer import *      # noqa: F401,F403
