// beamformer_ds.cc -- the reference's canonical C++ caller of the path (src/beamformerDS.cc:144-223,
// doBeamforming) written against this repo's node layer: SampleFeature xN -> OverSampledDFTAnalysisBank xN
// -> SubbandGSC (+ ZelinskiPostFilter) -> OverSampledDFTSynthesisBank, pulled block by block until
// jiterator_error.  Output: raw float32 samples (un-normalised) to <out.f32>.
//
// usage: beamformer_ds <coeffs.f64 (h then g, m*M doubles each)> <M> <m> <r> <pf type or 0> <alpha> <out.f32> <delay_0> <wav_0> [<delay_1> <wav_1> ...]
#include <cstdio>
#include <cstdlib>
#include <list>
#include <vector>
#include "feature/feature.h"
#include "modulated/modulated.h"
#include "beamformer/beamformer.h"
#include "postfilter/postfilter.h"

int main(int argc, char** argv)
{
  if (argc < 10 || ((argc - 8) % 2)) { fprintf(stderr, "bad arguments\n"); return 2; }
  const unsigned M = atoi(argv[2]), m = atoi(argv[3]), r = atoi(argv[4]);
  const int pf = atoi(argv[5]);
  const double alpha = atof(argv[6]);
  const unsigned D = M >> r;
  const int nchan = (argc - 8) / 2;
  try {
    gsl_vector* h_fb = gsl_vector_calloc(m * M);
    gsl_vector* g_fb = gsl_vector_calloc(m * M);
    FILE* fc = fopen(argv[1], "rb");
    if (!fc || fread(h_fb->data, sizeof(double), m * M, fc) != m * M || fread(g_fb->data, sizeof(double), m * M, fc) != m * M) {
      fprintf(stderr, "cannot read %s\n", argv[1]); return 2;
    }
    fclose(fc);
    gsl_vector* delays = gsl_vector_calloc(nchan);
    std::list<SampleFeaturePtr> sampleFeaturePL;
    std::list<OverSampledDFTAnalysisBankPtr> analysisFBPL;
    SubbandGSCPtr beamformerP = new SubbandGSC(M, false);
    for (int c = 0; c < nchan; c++) {
      gsl_vector_set(delays, c, atof(argv[8 + 2 * c]));
      SampleFeaturePtr sampleFeatureP = new SampleFeature("", D, D, true);
      sampleFeatureP->read(argv[9 + 2 * c], 16000);
      OverSampledDFTAnalysisBankPtr analysisFBP =
          new OverSampledDFTAnalysisBank((VectorFloatFeatureStreamPtr&)sampleFeatureP, h_fb, M, m, r);
      beamformerP->setChannel((VectorComplexFeatureStreamPtr&)analysisFBP);
      sampleFeaturePL.push_back(sampleFeatureP);
      analysisFBPL.push_back(analysisFBP);
    }
    beamformerP->calcGSCWeights(16000.0, delays);
    VectorComplexFeatureStreamPtr tail;
    ZelinskiPostFilterPtr output;
    if (pf) {
      output = new ZelinskiPostFilter((VectorComplexFeatureStreamPtr&)beamformerP, M, alpha, pf);
      output->setBeamformer((SubbandDSPtr&)beamformerP);
      tail = (VectorComplexFeatureStreamPtr&)output;
    } else {
      tail = (VectorComplexFeatureStreamPtr&)beamformerP;
    }
    OverSampledDFTSynthesisBankPtr synthesisFBP = new OverSampledDFTSynthesisBank(tail, g_fb, M, m, r);
    std::vector<float> data;
    for (;;) {
      const gsl_vector_float* blk;
      try { blk = synthesisFBP->next(); } catch (jiterator_error& e) { break; }
      for (unsigned i = 0; i < D; i++) data.push_back(gsl_vector_float_get(blk, i));
    }
    FILE* fo = fopen(argv[7], "wb");
    fwrite(data.data(), sizeof(float), data.size(), fo);
    fclose(fo);
    fprintf(stderr, "beamformer_ds: %d channels, %lu samples written\n", nchan, (unsigned long)data.size());
    gsl_vector_free(h_fb); gsl_vector_free(g_fb); gsl_vector_free(delays);
  } catch (j_error& e) {
    fprintf(stderr, "j_error: %s\n", e.what());
    return 1;
  }
  return 0;
}
