#!/usr/bin/env python
"""Batch beamforming from second-order statistics on WAV files through the MI355X engine -- the application-level
counterpart of the reference's unit_test/test_sos_batch_beamforming.py on this repo's mirror.

Same command line (-a -s -M -m -r -i -o -c) and JSON schema as unit_test/confs/{smimvdr,bmvdr_*,gev_*}.json:
  beamformer.type in {smimvdr, bmvdr, gev} (+ energy_threshold, mu, gamma, ref_micx, offset),
  target.vad_label [[start, end], ...] or target.tfmask_path + noises[].tfmask_path,
  postfilter (zelinski / mccowan / lefkimmiatis) for the look-direction beamformer smimvdr.
TF-mask files: a stream of pickled rows (one per frame, the reference's format), or .npy / .npz (key "mask").
"""
import argparse
import json
import os
import pickle
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.online_beamforming import load_prototype, SSPEED      # noqa: E402


def load_tfmask(path):
    if path.endswith(".npy"):
        return np.load(path)
    if path.endswith(".npz"):
        return np.load(path)["mask"]
    rows = []
    with open(path, "rb") as fp:
        while True:
            try:
                rows.append(pickle.load(fp, encoding="latin1"))
            except EOFError:
                break
    return np.array(rows)


def load_tfmasks(ap_conf):
    mask_t = load_tfmask(ap_conf["target"]["tfmask_path"])
    mask_j = None
    for noise_conf in ap_conf["noises"]:
        if "tfmask_path" in noise_conf:
            mj = np.asarray(load_tfmask(noise_conf["tfmask_path"]), np.float64)
            mask_j = mj if mask_j is None else mask_j + mj
    return mask_t, mask_j / len(ap_conf["noises"])


def sos_batch_beamforming(h_fb, g_fb, D, M, m, r, input_audio_paths, out_path, ap_conf, samplerate, verbose=True):
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr,
                                                      OverSampledDFTSynthesisBankPtr, PyVectorComplexFeatureStreamPtr,
                                                      ZelinskiPostFilterPtr, McCowanPostFilterPtr, LefkimmiatisPostFilterPtr)
    from distant_speech_recognition_amd import pybeamformer as pb
    sample_feats, afbs = [], []
    for path in input_audio_paths:
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(path, samplerate)
        afbs.append(OverSampledDFTAnalysisBankPtr(sf, prototype=h_fb, M=M, m=m, r=r, delay_compensation_type=2))
        sample_feats.append(sf)
    bf_conf = ap_conf["beamformer"]
    bf_type = bf_conf["type"]
    if bf_type == "smimvdr":
        beamformer = pb.SubbandSMIMVDRBeamformer(afbs, Nc=1)
    elif bf_type == "bmvdr":
        beamformer = pb.SubbandBlindMVDRBeamformer(afbs)
    elif bf_type == "gev":
        beamformer = pb.SubbandGEVBeamformer(afbs)
    else:
        raise KeyError("Invalid batch-processing beamformer type: {}".format(bf_type))

    pybf = PyVectorComplexFeatureStreamPtr(beamformer)
    use_postfilter = False
    if "postfilter" not in ap_conf:
        spatial_filter = pybf
    else:
        if bf_type != "smimvdr":
            raise NotImplementedError("post-filters need a look direction: use them with smimvdr")
        pf_conf = ap_conf["postfilter"]
        if pf_conf["type"] == "zelinski":
            spatial_filter = ZelinskiPostFilterPtr(pybf, M, pf_conf.get("alpha", 0.6), pf_conf.get("subtype", 2))
        elif pf_conf["type"] == "mccowan":
            spatial_filter = McCowanPostFilterPtr(pybf, M, pf_conf.get("alpha", 0.6), pf_conf.get("subtype", 2))
            spatial_filter.set_diffuse_noise_model(ap_conf["microphone_positions"], samplerate, SSPEED)
            spatial_filter.set_all_diagonal_loading(bf_conf.get("diagonal_load", 0.01))
        elif pf_conf["type"] == "lefkimmiatis":
            spatial_filter = LefkimmiatisPostFilterPtr(pybf, M, pf_conf.get("min_sv", 1e-8), pf_conf.get("fbin_no1", 128),
                                                       pf_conf.get("alpha", 0.8), pf_conf.get("subtype", 2))
            spatial_filter.set_diffuse_noise_model(ap_conf["microphone_positions"], samplerate, SSPEED)
            spatial_filter.set_all_diagonal_loading(bf_conf.get("diagonal_load", 0.1))
            spatial_filter.calc_inverse_noise_spatial_spectral_matrix()
        else:
            raise KeyError("Invalid post-filter type: {}".format(pf_conf["type"]))
        use_postfilter = True
    sfb = OverSampledDFTSynthesisBankPtr(spatial_filter, prototype=g_fb, M=M, m=m, r=r, delay_compensation_type=2)

    energy_threshold = bf_conf.get("energy_threshold", 10)
    if bf_type == "smimvdr":
        delays_t = pb.calc_delays(ap_conf["array_type"], ap_conf["microphone_positions"], ap_conf["target"]["positions"][0][1],
                                  sspeed=SSPEED)
        beamformer.accu_stats_from_label(samplerate, target_labs=ap_conf["target"]["vad_label"], energy_threshold=energy_threshold)
        beamformer.finalize_stats()
        beamformer.calc_beamformer_weights(samplerate, delays_t, mu=bf_conf.get("mu", 1e-4))
    else:
        if "tfmask_path" in ap_conf["target"]:
            mask_t, mask_j = load_tfmasks(ap_conf)
            beamformer.accu_stats_from_tfmask(samplerate, mask_t, mask_j, energy_threshold=energy_threshold)
        else:
            beamformer.accu_stats_from_label(samplerate, target_labs=ap_conf["target"]["vad_label"], energy_threshold=energy_threshold)
        beamformer.finalize_stats(gamma=bf_conf.get("gamma", 1e-6))
        if bf_type == "bmvdr":
            beamformer.calc_beamformer_weights(ref_micx=bf_conf.get("ref_micx", 0), offset=bf_conf.get("offset", 0.0))
        else:
            beamformer.calc_beamformer_weights()
    if use_postfilter:
        spatial_filter.set_beamformer(beamformer.beamformer())
    for c, path in enumerate(input_audio_paths):                 # reload the test data (reset the feature pointer)
        sample_feats[c].read(path, samplerate)
    out_dir = os.path.dirname(out_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    wavefile = wave.open(out_path, "w")
    wavefile.setnchannels(1)
    wavefile.setsampwidth(2)
    wavefile.setframerate(int(samplerate))
    total_energy, frame_no = 0.0, -1
    for frame_no, buf in enumerate(sfb):
        buf = np.array(buf)
        if verbose and frame_no % 128 == 0:
            print("%0.2f sec. processed" % (frame_no * D / float(samplerate)))
        total_energy += float(np.inner(buf, buf))
        wavefile.writeframes(buf.astype(np.int16).tobytes())
    wavefile.close()
    return total_energy, frame_no


def main(argv=None):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proto = os.path.join(here, "tests", "golden", "prototype_M256_m4_r1.npz")
    p = argparse.ArgumentParser(description="batch SOS beamforming (SMI-MVDR, blind MVDR, GEV) on the MI355X engine")
    p.add_argument("-a", dest="analysis_filter_path", default=proto)
    p.add_argument("-s", dest="synthesis_filter_path", default=proto)
    p.add_argument("-M", dest="M", default=256, type=int)
    p.add_argument("-m", dest="m", default=4, type=int)
    p.add_argument("-r", dest="r", default=1, type=int)
    p.add_argument("-i", dest="input_audio_paths", nargs="+", required=True)
    p.add_argument("-o", dest="out_path", default="out/beamformed.wav")
    p.add_argument("-c", dest="ap_conf_path", required=True)
    p.add_argument("-q", dest="quiet", action="store_true")
    args = p.parse_args(argv)
    with open(args.ap_conf_path) as fp:
        ap_conf = json.load(fp)
    D = args.M // 2 ** args.r
    total_energy, frame_no = sos_batch_beamforming(load_prototype(args.analysis_filter_path, "h"),
                                                   load_prototype(args.synthesis_filter_path, "g"), D, args.M, args.m, args.r,
                                                   args.input_audio_paths, args.out_path, ap_conf, 16000, verbose=not args.quiet)
    print("Avg. output power: %f" % (total_energy / max(frame_no + 1, 1)))
    print("No. frames processed: %d" % (frame_no + 1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
