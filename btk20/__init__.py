"""`btk20` import-name shim: scripts written for the reference (`from btk20.beamformer import *`, `import btk20.pybeamformer`)
resolve to this repo's MI355X engine mirror (distant_speech_recognition_amd.btk20 / .pybeamformer) without edits."""
from distant_speech_recognition_amd.btk20 import *      # noqa: F401,F403
