"""btk20.modulated -> distant_speech_recognition_amd.btk20.modulated"""
from distant_speech_recognition_amd.btk20.modulated import *      # noqa: F401,F403
