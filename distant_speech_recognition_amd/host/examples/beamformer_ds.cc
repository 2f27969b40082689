// beamformer_ds.cc -- the reference's canonical C++ caller of the path (src/beamformerDS.cc:144-223,
// doBeamforming) written against this repo's node layer: SampleFeature xN -> OverSampledDFTAnalysisBank xN
// -> SubbandGSC (+ ZelinskiPostFilter) -> OverSampledDFTSynthesisBank, pulled block by block until
// jiterator_error.  Output: raw float32 samples (un-normalised) to <out.f32>.
//
// Environment switches select the other nodes of the layer (same pull graph):
//   BTK_EXAMPLE_BF=gscrls            SubbandGSCRLS (mu 0.97, sigma2 1e-3, init_precision_matrix(1e6), THRESHOLD_LIMITATION 0.1)
//   BTK_EXAMPLE_PF=mccowan|lefkimmiatis   McCowanPostFilter / LefkimmiatisPostFilter with the diffuse-noise model of
//                                    BTK_EXAMPLE_MPOS="x0,y0,z0;x1,y1,z1;..." (mm) and diagonal loading 0.01 / 0.1
//
// usage: beamformer_ds <coeffs.f64 (h then g, m*M doubles each)> <M> <m> <r> <pf type or 0> <alpha> <out.f32> <delay_0> <wav_0> [<delay_1> <wav_1> ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
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
    const char* bfkind = getenv("BTK_EXAMPLE_BF");
    const char* pfkind = getenv("BTK_EXAMPLE_PF");
    const bool rls = bfkind && !strcmp(bfkind, "gscrls");
    SubbandGSCRLSPtr rlsP;
    SubbandGSCPtr beamformerP;
    if (rls) { rlsP = new SubbandGSCRLS(M, false, 0.97f, 0.001f); beamformerP = (SubbandGSCPtr&)rlsP; }
    else beamformerP = new SubbandGSC(M, false);
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
    if (rls) { rlsP->init_precision_matrix(1.0e6f); rlsP->set_quadratic_constraint(0.1f, THRESHOLD_LIMITATION); }
    VectorComplexFeatureStreamPtr tail;
    ZelinskiPostFilterPtr output;
    McCowanPostFilterPtr mccowan;
    LefkimmiatisPostFilterPtr lefkimmiatis;
    if (pf && pfkind && (!strcmp(pfkind, "mccowan") || !strcmp(pfkind, "lefkimmiatis"))) {
      gsl_matrix* mpos = gsl_matrix_alloc(nchan, 3);
      const char* ms = getenv("BTK_EXAMPLE_MPOS");
      if (!ms) { fprintf(stderr, "BTK_EXAMPLE_MPOS is required\n"); return 2; }
      std::string str(ms);
      size_t pos = 0;
      for (int c = 0; c < nchan; c++)
        for (int j = 0; j < 3; j++) {
          size_t used = 0;
          gsl_matrix_set(mpos, c, j, std::stod(str.substr(pos), &used));
          pos += used + 1;
        }
      if (!strcmp(pfkind, "mccowan")) {
        mccowan = new McCowanPostFilter((VectorComplexFeatureStreamPtr&)beamformerP, M, alpha, pf);
        mccowan->set_diffuse_noise_model(mpos, 16000.0, SSPEED);
        mccowan->set_all_diagonal_loading(0.01f);
        mccowan->setBeamformer((SubbandDSPtr&)beamformerP);
        tail = (VectorComplexFeatureStreamPtr&)mccowan;
      } else {
        lefkimmiatis = new LefkimmiatisPostFilter((VectorComplexFeatureStreamPtr&)beamformerP, M, 1.0e-4, 100, alpha, pf);
        lefkimmiatis->set_diffuse_noise_model(mpos, 16000.0, SSPEED);
        lefkimmiatis->set_all_diagonal_loading(0.1f);
        lefkimmiatis->calc_inverse_noise_spatial_spectral_matrix();
        lefkimmiatis->setBeamformer((SubbandDSPtr&)beamformerP);
        tail = (VectorComplexFeatureStreamPtr&)lefkimmiatis;
      }
      gsl_matrix_free(mpos);
    } else if (pf) {
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
