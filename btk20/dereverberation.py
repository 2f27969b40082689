"""btk20.dereverberation -> distant_speech_recognition_amd.btk20.dereverberation"""
from distant_speech_recognition_amd.btk20.dereverberation import *      # noqa: F401,F403
