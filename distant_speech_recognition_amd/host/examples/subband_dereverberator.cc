// subband_dereverberator.cc -- WPE dereverberation through the C++ node layer, the flow of the reference's
// unit_test/test_subband_dereverberator.py:92-170:
//   SampleFeature -> OverSampledDFTAnalysisBank -> {Single,Multi}ChannelWPEDereverberation(Feature) -> OverSampledDFTSynthesisBank
// estimate_filter() on the whole utterance, then the dereverberated channels are pulled in lock step.
// usage: subband_dereverberator coeffs.f64 M m r lowerN upperN iterations load_db diagonal_bias out_prefix wav...
//        one wav: SingleChannelWPEDereverberationFeature (no diagonal bias); several: MultiChannelWPEDereverberation.
//        writes out_prefix.c<i>.f32, prints the number of frames used for the estimate.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dereverberation/dereverberation.h"
#include "feature/feature.h"
#include "modulated/modulated.h"

int main(int argc, char** argv)
{
  if (argc < 12) {
    fprintf(stderr, "usage: %s coeffs.f64 M m r lowerN upperN iterations load_db diagonal_bias out_prefix wav...\n", argv[0]);
    return 2;
  }
  const unsigned M = atoi(argv[2]), m = atoi(argv[3]), r = atoi(argv[4]);
  const unsigned lowerN = atoi(argv[5]), upperN = atoi(argv[6]), iters = atoi(argv[7]);
  const double load_db = atof(argv[8]), bias = atof(argv[9]);
  const std::string prefix(argv[10]);
  const unsigned D = M >> r;
  const int nchan = argc - 11;
  try {
    gsl_vector* h_fb = gsl_vector_calloc(m * M);
    gsl_vector* g_fb = gsl_vector_calloc(m * M);
    FILE* fc = fopen(argv[1], "rb");
    if (!fc || fread(h_fb->data, sizeof(double), m * M, fc) != m * M || fread(g_fb->data, sizeof(double), m * M, fc) != m * M) {
      fprintf(stderr, "cannot read %s\n", argv[1]); return 2;
    }
    fclose(fc);
    std::vector<SampleFeaturePtr> sampleFeatures;
    std::vector<OverSampledDFTAnalysisBankPtr> analysisFBs;
    for (int c = 0; c < nchan; c++) {
      SampleFeaturePtr sf = new SampleFeature("", D, D, true);
      sf->read(argv[11 + c], 16000);
      OverSampledDFTAnalysisBankPtr afb = new OverSampledDFTAnalysisBank((VectorFloatFeatureStreamPtr&)sf, h_fb, M, m, r);
      sampleFeatures.push_back(sf);
      analysisFBs.push_back(afb);
    }
    std::vector<OverSampledDFTSynthesisBankPtr> synth;
    unsigned nfr = 0;
    SingleChannelWPEDereverberationFeaturePtr single;
    MultiChannelWPEDereverberationPtr pre;
    std::vector<MultiChannelWPEDereverberationFeaturePtr> feats;
    if (nchan == 1) {
      single = new SingleChannelWPEDereverberationFeature((VectorComplexFeatureStreamPtr&)analysisFBs[0], lowerN, upperN, iters, load_db);
      nfr = single->estimate_filter();
      sampleFeatures[0]->read(argv[11], 16000);                              // the reference flow re-reads its inputs
      synth.push_back(new OverSampledDFTSynthesisBank((VectorComplexFeatureStreamPtr&)single, g_fb, M, m, r));
    } else {
      pre = new MultiChannelWPEDereverberation(M, nchan, lowerN, upperN, iters, load_db, 0.0, bias);
      for (int c = 0; c < nchan; c++) pre->set_input((VectorComplexFeatureStreamPtr&)analysisFBs[c]);
      nfr = pre->estimate_filter();
      for (int c = 0; c < nchan; c++) {
        sampleFeatures[c]->read(argv[11 + c], 16000);
        feats.push_back(new MultiChannelWPEDereverberationFeature(pre, c));
        synth.push_back(new OverSampledDFTSynthesisBank((VectorComplexFeatureStreamPtr&)feats[c], g_fb, M, m, r));
      }
    }
    std::vector<std::vector<float> > data(nchan);
    for (bool more = true; more;) {                                          // lock-step pull (:160-170)
      for (int c = 0; c < nchan; c++) {
        const gsl_vector_float* blk;
        try { blk = synth[c]->next(); } catch (jiterator_error& e) { more = false; break; }
        for (unsigned i = 0; i < D; i++) data[c].push_back(gsl_vector_float_get(blk, i));
      }
    }
    for (int c = 0; c < nchan; c++) {
      const std::string fn = prefix + ".c" + std::to_string(c) + ".f32";
      FILE* fo = fopen(fn.c_str(), "wb");
      fwrite(data[c].data(), sizeof(float), data[c].size(), fo);
      fclose(fo);
    }
    fprintf(stderr, "subband_dereverberator: %d channels, %u frames used for the estimate, %lu samples per channel\n", nchan, nfr,
            (unsigned long)data[0].size());
    gsl_vector_free(h_fb); gsl_vector_free(g_fb);
  } catch (j_error& e) {
    fprintf(stderr, "j_error: %s\n", e.what());
    return 1;
  }
  return 0;
}
