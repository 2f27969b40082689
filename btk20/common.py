"""btk20.common -> distant_speech_recognition_amd.btk20.common"""
from distant_speech_recognition_amd.btk20.common import *      # noqa: F401,F403
