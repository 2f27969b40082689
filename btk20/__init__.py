"""`btk20` -- the reference's import name.  Scripts written for the reference (`from btk20.beamformer import *`,
`import btk20.pybeamformer`) resolve to this repo's engine without edits: this package IS
`distant_speech_recognition_amd.btk20` under the reference's name -- its sub-modules are registered here as the very same module
objects (no second layer of re-exporting files), `btk20.pybeamformer` is `distant_speech_recognition_amd.pybeamformer`."""
import importlib
import sys

from distant_speech_recognition_amd.btk20 import *      # noqa: F401,F403
from distant_speech_recognition_amd.btk20 import __all__  # noqa: F401

for _name in ("stream", "feature", "modulated", "beamformer", "postfilter", "dereverberation", "common"):
    _mod = importlib.import_module("distant_speech_recognition_amd.btk20." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
pybeamformer = importlib.import_module("distant_speech_recognition_amd.pybeamformer")
sys.modules[__name__ + ".pybeamformer"] = pybeamformer
del _name, _mod
