"""btk20.stream -> distant_speech_recognition_amd.btk20.stream"""
from distant_speech_recognition_amd.btk20.stream import *      # noqa: F401,F403
