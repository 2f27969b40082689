"""btk20.feature -> distant_speech_recognition_amd.btk20.feature"""
from distant_speech_recognition_amd.btk20.feature import *      # noqa: F401,F403
