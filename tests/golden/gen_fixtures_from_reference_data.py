"""Convert DATA files the reference's own tests hold into small fixtures (dev container only).

Inputs (read-only, /root/reference/btk20_src/unit_test/...):
  prototype.ny/{h,g}-M256-m4-r1.pickle       -> prototype_M256_m4_r1.npz   (float64[1024] each)
  data/CMU/.../U1001_1M_16k_b16_c{1..4}.wav  -> kinect_4ch_16k.npz         (int16 [4][78064])
  confs/*.json values are quoted in the tests with file:line citations (not copied).
These are data (inputs), not source.  Run:  python tests/golden/gen_fixtures_from_reference_data.py
"""
import os, pickle, wave
import numpy as np

REF = "/root/reference/btk20_src/unit_test"
OUT = os.path.dirname(os.path.abspath(__file__))

def main():
    with open(f"{REF}/prototype.ny/h-M256-m4-r1.pickle", "rb") as fp:
        h = np.asarray(pickle.load(fp, encoding="latin1"), np.float64)
    with open(f"{REF}/prototype.ny/g-M256-m4-r1.pickle", "rb") as fp:
        g = np.asarray(pickle.load(fp, encoding="latin1"), np.float64)
    np.savez_compressed(f"{OUT}/prototype_M256_m4_r1.npz", h=h, g=g)
    chans = []
    for c in range(1, 5):
        w = wave.open(f"{REF}/data/CMU/R1/M1005/KINECT/RAW/segmented/U1001_1M_16k_b16_c{c}.wav", "rb")
        assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 16000
        chans.append(np.frombuffer(w.readframes(w.getnframes()), np.int16))
    pcm = np.stack(chans)
    np.savez_compressed(f"{OUT}/kinect_4ch_16k.npz", pcm=pcm)
    print(h.shape, g.shape, pcm.shape)

if __name__ == "__main__":
    main()
