#!/usr/bin/env python
"""Online subband beamforming on WAV files through the MI355X engine -- the application-level counterpart of the
reference's unit_test/test_online_beamforming.py, written against this repo's btk20 / pybeamformer mirror.

Same command line (-a/-s prototype files, -M -m -r, -i input WAVs, -o output WAV, -c JSON configuration) and the same
JSON schema as the reference's unit_test/confs/*.json:
  array_type, microphone_positions, target.positions [[time, position], ...], noises[].positions,
  beamformer.type in {delay_and_sum, lcmv, super_directive, gsclms, gscrls} (+ its hyper-parameters),
  postfilter.type in {zelinski, mccowan, lefkimmiatis} (+ subtype, alpha, min_sv, fbin_no1).
Look directions change at the time stamps of target.positions: frames not yet served are recomputed with the new
weights, exactly like the per-frame loop of the reference script.  Prototype files may be the reference's pickles
(numpy arrays) or .npz files with keys h / g.
"""
import argparse
import json
import os
import pickle
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SSPEED = 343740.0


def load_prototype(path, key):
    if path.endswith(".npz"):
        return np.asarray(np.load(path)[key], np.float64)
    with open(path, "rb") as fp:
        return np.asarray(pickle.load(fp, encoding="latin1"), np.float64)


def check_position_data_format(ap_conf):
    need = {"linear": 1, "planar": 2, "circular": 2}.get(ap_conf["array_type"], 3)
    assert "positions" in ap_conf["target"], "No target position"
    for posx, (targ_t, pos) in enumerate(ap_conf["target"]["positions"]):
        assert len(pos) >= need, "Insufficient position info. at time %0.3f" % targ_t
        for noisex, noise in enumerate(ap_conf.get("noises", [])):
            noise_t, npos = noise["positions"][posx]
            assert targ_t == noise_t, "%d-th noise: Misaligned time stamp %0.4f != %0.4f" % (noisex, targ_t, noise_t)
            assert len(npos) >= need, "Insufficient position info. at time %0.3f" % noise_t


def online_beamforming(h_fb, g_fb, D, M, m, r, input_audio_paths, out_path, ap_conf, samplerate, verbose=True):
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr,
                                                      OverSampledDFTSynthesisBankPtr, PyVectorComplexFeatureStreamPtr,
                                                      ZelinskiPostFilterPtr, McCowanPostFilterPtr, LefkimmiatisPostFilterPtr)
    from distant_speech_recognition_amd import pybeamformer as pb

    sample_feats, afbs = [], []
    for path in input_audio_paths:
        sample_feat = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sample_feat.read(path, samplerate)
        afbs.append(OverSampledDFTAnalysisBankPtr(sample_feat, prototype=h_fb, M=M, m=m, r=r, delay_compensation_type=2))
        sample_feats.append(sample_feat)

    bf_conf = ap_conf["beamformer"]
    bf_type = bf_conf["type"]
    if bf_type == "delay_and_sum":
        beamformer = pb.SubbandGSCBeamformer(afbs, Nc=1)
    elif bf_type == "lcmv":
        beamformer = pb.SubbandGSCBeamformer(afbs, Nc=1 + len(ap_conf.get("noises", [0])))
    elif bf_type == "super_directive":
        beamformer = pb.SubbandMVDRBeamformer(afbs)
    elif bf_type == "gsclms":
        keys = ("beta", "gamma", "init_diagonal_load", "regularization_param", "energy_floor", "sil_thresh",
                "max_wa_l2norm", "min_frames", "slowdown_after")
        beamformer = pb.SubbandGSCLMSBeamformer(afbs, **{k: bf_conf[k] for k in keys if k in bf_conf})
    elif bf_type == "gscrls":
        keys = ("beta", "gamma", "mu", "init_diagonal_load", "regularization_param", "sil_thresh", "constraint_option",
                "alpha2", "max_wa_l2norm", "min_frames", "slowdown_after")
        beamformer = pb.SubbandGSCRLSBeamformer(afbs, **{k: bf_conf[k] for k in keys if k in bf_conf})
    else:
        raise KeyError("Invalid beamformer type: {}".format(bf_type))

    use_postfilter = False
    pybf = PyVectorComplexFeatureStreamPtr(beamformer)
    if "postfilter" not in ap_conf:
        spatial_filter = pybf
    elif bf_type in ("delay_and_sum", "lcmv", "super_directive"):
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
    else:
        raise NotImplementedError("Post-filter unsupported: {}".format(bf_type))

    sfb = OverSampledDFTSynthesisBankPtr(spatial_filter, prototype=g_fb, M=M, m=m, r=r, delay_compensation_type=2)

    def delays_at(posx):
        delays_t = pb.calc_delays(ap_conf["array_type"], ap_conf["microphone_positions"],
                                  ap_conf["target"]["positions"][posx][1], sspeed=SSPEED)
        delays_js = None
        if "noises" in ap_conf:
            delays_js = np.stack([pb.calc_delays(ap_conf["array_type"], ap_conf["microphone_positions"],
                                                 noise["positions"][posx][1], sspeed=SSPEED) for noise in ap_conf["noises"]])
        return delays_t, delays_js

    def calc_weights(delays_t, delays_js):
        if delays_js is not None and bf_type != "lcmv" and verbose:
            print("Noise information will be ignored")
        if bf_type == "super_directive":
            beamformer.calc_sd_beamformer_weights(samplerate, delays_t, ap_conf["microphone_positions"], sspeed=SSPEED,
                                                  mu=bf_conf.get("diagonal_load", 0.01))
        elif bf_type == "lcmv":
            assert delays_js is not None, "LCMV beamforming: missing noise source positions"
            beamformer.calc_beamformer_weights_n(samplerate, delays_t, delays_js)
        else:
            beamformer.calc_beamformer_weights(samplerate, delays_t)

    posx = 0
    calc_weights(*delays_at(posx))
    if use_postfilter:
        spatial_filter.set_beamformer(beamformer.beamformer())
    out_dir = os.path.dirname(out_path)
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir, exist_ok=True)
    wavefile = wave.open(out_path, "w")
    wavefile.setnchannels(1)
    wavefile.setsampwidth(2)
    wavefile.setframerate(int(samplerate))
    total_energy, elapsed_time, time_delta, frame_no = 0.0, 0.0, D / float(samplerate), -1
    positions = ap_conf["target"]["positions"]
    for frame_no, buf in enumerate(sfb):
        buf = np.array(buf)
        if verbose and frame_no % 128 == 0:
            print("%0.2f sec. processed" % (frame_no * time_delta))
        total_energy += float(np.inner(buf, buf))
        wavefile.writeframes(buf.astype(np.int16).tobytes())
        elapsed_time += time_delta
        if elapsed_time > positions[posx][0] and (posx + 1) < len(positions):
            posx += 1
            calc_weights(*delays_at(posx))
    wavefile.close()
    return total_energy, frame_no


def build_parser():
    M, m, r = 256, 4, 1
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proto = os.path.join(here, "tests", "golden", "prototype_M256_m4_r1.npz")
    parser = argparse.ArgumentParser(description="online subband beamforming on the MI355X engine")
    parser.add_argument("-a", dest="analysis_filter_path", default=proto, help="analysis filter prototype file (.pickle or .npz)")
    parser.add_argument("-s", dest="synthesis_filter_path", default=proto, help="synthesis filter prototype file")
    parser.add_argument("-M", dest="M", default=M, type=int, help="no. of subbands")
    parser.add_argument("-m", dest="m", default=m, type=int, help="Prototype filter length factor")
    parser.add_argument("-r", dest="r", default=r, type=int, help="Decimation factor")
    parser.add_argument("-i", dest="input_audio_paths", nargs="+", required=True, help="observation audio file(s)")
    parser.add_argument("-o", dest="out_path", default="out/beamformed.wav", help="output audio file")
    parser.add_argument("-c", dest="ap_conf_path", default=None, help="JSON path for array processing configuration")
    parser.add_argument("-q", dest="quiet", action="store_true", help="no progress output")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.ap_conf_path is None:
        ap_conf = {"array_type": "linear",
                   "microphone_positions": [[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]],
                   "target": {"positions": [[0.0, [-1.306379, None, None]]]},
                   "beamformer": {"type": "super_directive"},
                   "postfilter": {"type": "zelinski", "subtype": 2, "alpha": 0.7}}
    else:
        with open(args.ap_conf_path, "r") as fp:
            ap_conf = json.load(fp)
    check_position_data_format(ap_conf)
    D = args.M // 2 ** args.r
    h_fb = load_prototype(args.analysis_filter_path, "h")
    g_fb = load_prototype(args.synthesis_filter_path, "g")
    total_energy, frame_no = online_beamforming(h_fb, g_fb, D, args.M, args.m, args.r, args.input_audio_paths, args.out_path,
                                                ap_conf, 16000, verbose=not args.quiet)
    print("Avg. output power: %f" % (total_energy / max(frame_no + 1, 1)))
    print("No. frames processed: %d" % (frame_no + 1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
