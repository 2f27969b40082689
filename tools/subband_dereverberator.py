#!/usr/bin/env python
"""Subband WPE dereverberation of WAV files through the MI355X engine -- the application-level counterpart of the
reference's unit_test/test_subband_dereverberator.py on this repo's mirror.

Same command line (-a -s -M -m -r -i inputs -o outputs -c JSON -b start frame -e end frame) and JSON keys as
unit_test/confs/wpe.json: lower_num, upper_num, iterations_num, load_db, band_width, diagonal_bias.
One input -> single-channel WPE, several inputs -> multi-channel WPE (one output file per channel).
"""
import argparse
import json
import os
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.online_beamforming import load_prototype      # noqa: E402


def _open_wave(path, samplerate):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    w = wave.open(path, "w")
    w.setnchannels(1)
    w.setsampwidth(2)
    w.setframerate(int(samplerate))
    return w


def dereverberate(h_fb, g_fb, D, M, m, r, input_audio_paths, out_paths, wpe_conf, samplerate, start_frame_no, end_frame_no,
                  verbose=True):
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr,
                                                      OverSampledDFTSynthesisBankPtr, SingleChannelWPEDereverberationFeaturePtr,
                                                      MultiChannelWPEDereverberationPtr, MultiChannelWPEDereverberationFeaturePtr)
    C = len(input_audio_paths)
    assert len(out_paths) == C, "one output file per input channel"
    sample_feats, afbs = [], []
    for path in input_audio_paths:
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(path, samplerate)
        afbs.append(OverSampledDFTAnalysisBankPtr(sf, prototype=h_fb, M=M, m=m, r=r, delay_compensation_type=2))
        sample_feats.append(sf)
    kw = dict(lower_num=wpe_conf.get("lower_num", 0), upper_num=wpe_conf.get("upper_num", 32),
              iterations_num=wpe_conf.get("iterations_num", 2), load_db=wpe_conf.get("load_db", -20.0),
              band_width=wpe_conf.get("band_width", 0.0), samplerate=samplerate)
    if C == 1:
        dereverb = SingleChannelWPEDereverberationFeaturePtr(afbs[0], **kw)
        frame_num = dereverb.estimate_filter(start_frame_no, -1 if end_frame_no < 0 else end_frame_no - start_frame_no)
        sample_feats[0].read(input_audio_paths[0], samplerate)
        nodes = [dereverb]
    else:
        pre = MultiChannelWPEDereverberationPtr(subbands_num=M, channels_num=C, diagonal_bias=wpe_conf.get("diagonal_bias", 0.001), **kw)
        for a in afbs:
            pre.set_input(a)
        frame_num = pre.estimate_filter(start_frame_no, end_frame_no)
        for c in range(C):
            sample_feats[c].read(input_audio_paths[c], samplerate)
        nodes = [MultiChannelWPEDereverberationFeaturePtr(pre, channel_no=c) for c in range(C)]
    if verbose:
        print("%d frames are used for filter estimation" % frame_num)
    sfbs = [OverSampledDFTSynthesisBankPtr(n, prototype=g_fb, M=M, m=m, r=r, delay_compensation_type=2) for n in nodes]
    wavefiles = [_open_wave(p, samplerate) for p in out_paths]
    frame_no = 0
    while True:                                                    # lock-step pull over the channels
        try:
            for c in range(C):
                wavefiles[c].writeframes(np.array(sfbs[c].next()).astype(np.int16).tobytes())
        except StopIteration:
            break
        frame_no += 1
    for w in wavefiles:
        w.close()
    return frame_num, frame_no


def main(argv=None):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proto = os.path.join(here, "tests", "golden", "prototype_M256_m4_r1.npz")
    p = argparse.ArgumentParser(description="subband WPE dereverberation on the MI355X engine")
    p.add_argument("-a", dest="analysis_filter_path", default=proto)
    p.add_argument("-s", dest="synthesis_filter_path", default=proto)
    p.add_argument("-M", dest="M", default=256, type=int)
    p.add_argument("-m", dest="m", default=4, type=int)
    p.add_argument("-r", dest="r", default=1, type=int)
    p.add_argument("-i", dest="input_audio_paths", nargs="+", required=True)
    p.add_argument("-o", dest="out_paths", nargs="+", required=True)
    p.add_argument("-c", dest="wpe_conf_path", default=None)
    p.add_argument("-b", dest="start_frame_no", default=0, type=int)
    p.add_argument("-e", dest="end_frame_no", default=-1, type=int)
    p.add_argument("-q", dest="quiet", action="store_true")
    args = p.parse_args(argv)
    wpe_conf = {"lower_num": 0, "upper_num": 32, "iterations_num": 2, "load_db": -18.0, "band_width": 0.0, "diagonal_bias": 0.0001}
    if args.wpe_conf_path:
        with open(args.wpe_conf_path) as fp:
            wpe_conf = json.load(fp)
    D = args.M // 2 ** args.r
    frame_num, nblocks = dereverberate(load_prototype(args.analysis_filter_path, "h"), load_prototype(args.synthesis_filter_path, "g"),
                                       D, args.M, args.m, args.r, args.input_audio_paths, args.out_paths, wpe_conf, 16000,
                                       args.start_frame_no, args.end_frame_no, verbose=not args.quiet)
    print("No. frames used for estimation: %d" % frame_num)
    print("No. blocks written: %d" % nblocks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
