"""btk20.postfilter -> distant_speech_recognition_amd.btk20.postfilter"""
from distant_speech_recognition_amd.btk20.postfilter import *      # noqa: F401,F403
