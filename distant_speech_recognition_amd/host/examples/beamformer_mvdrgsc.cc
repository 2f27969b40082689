// beamformer_mvdrgsc.cc -- SubbandMVDRGSC through the C++ node layer (reference usage: beamformer.h:388-396):
//   SampleFeature -> OverSampledDFTAnalysisBank xN -> SubbandMVDRGSC -> OverSampledDFTSynthesisBank
//   1. set_channel  2. calc_array_manifold_vectors  3. set_diffuse_noise_model + set_all_diagonal_loading
//   4. calc_mvdr_weights  5. calc_blocking_matrix1 (BTK_EXAMPLE_BM=1, default) or calc_blocking_matrix2 (=2)
//   6. set_active_weights_f with the deterministic test vector wa_k[i] = 0.05 (cos(0.37 k + i), sin(0.11 k (i + 1)))
// usage: beamformer_mvdrgsc coeffs.f64 M m r diag_load out.f32 mpos "x,y,z;..." {delay wav}...
// With BTK_EXAMPLE_BMOUT=path the blocking-matrix output of column 0 (bins 0..M/2, frame by frame) is written too,
// after upgrade_blocking_matrix() when BTK_EXAMPLE_UPGRADE is set.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <vector>

#include "beamformer/beamformer.h"
#include "feature/feature.h"
#include "modulated/modulated.h"

int main(int argc, char** argv)
{
  if (argc < 10 || ((argc - 8) % 2) != 0) {
    fprintf(stderr, "usage: %s coeffs.f64 M m r diag_load out.f32 mpos {delay wav}...\n", argv[0]);
    return 2;
  }
  const unsigned M = atoi(argv[2]), m = atoi(argv[3]), r = atoi(argv[4]);
  const float diag = atof(argv[5]);
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
    gsl_matrix* mpos = gsl_matrix_alloc(nchan, 3);
    {
      std::string str(argv[7]);
      size_t pos = 0;
      for (int c = 0; c < nchan; c++)
        for (int j = 0; j < 3; j++) {
          size_t used = 0;
          gsl_matrix_set(mpos, c, j, std::stod(str.substr(pos), &used));
          pos += used + 1;
        }
    }
    gsl_vector* delays = gsl_vector_calloc(nchan);
    std::list<SampleFeaturePtr> sampleFeaturePL;
    std::list<OverSampledDFTAnalysisBankPtr> analysisFBPL;
    SubbandMVDRGSCPtr beamformerP = new SubbandMVDRGSC(M, false);
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
    beamformerP->calc_array_manifold_vectors(16000.0, delays);
    beamformerP->set_diffuse_noise_model(mpos, 16000.0, SSPEED);
    beamformerP->set_all_diagonal_loading(diag);
    beamformerP->calc_mvdr_weights(16000.0, 1.0e-8f);
    const char* bm = getenv("BTK_EXAMPLE_BM");
    if (bm && atoi(bm) == 2) { if (!beamformerP->calc_blocking_matrix2()) { fprintf(stderr, "calc_blocking_matrix2 failed\n"); return 1; } }
    else beamformerP->calc_blocking_matrix1(16000.0, delays);
    gsl_vector* packed = gsl_vector_calloc(2 * (nchan - 1));
    for (unsigned k = 1; k <= M / 2; k++) {
      for (int i = 0; i < nchan - 1; i++) {
        gsl_vector_set(packed, 2 * i, 0.05 * cos(0.37 * k + i));
        gsl_vector_set(packed, 2 * i + 1, 0.05 * sin(0.11 * k * (i + 1)));
      }
      beamformerP->set_active_weights_f(k, packed);
    }
    if (getenv("BTK_EXAMPLE_UPGRADE")) beamformerP->upgrade_blocking_matrix();
    const char* bmout = getenv("BTK_EXAMPLE_BMOUT");
    std::vector<float> data;
    if (bmout) {
      // frame by frame through the beamformer node itself: b_0^H x next to the beamformer output
      std::vector<double> bo;
      for (;;) {
        try { beamformerP->next(); } catch (jiterator_error& e) { break; }
        const gsl_vector_complex* v = beamformerP->blocking_matrix_output(0);
        for (unsigned k = 0; k <= M / 2; k++) { bo.push_back(v->data[2 * k]); bo.push_back(v->data[2 * k + 1]); }
      }
      FILE* fb = fopen(bmout, "wb");
      fwrite(bo.data(), sizeof(double), bo.size(), fb);
      fclose(fb);
    } else {
      VectorComplexFeatureStreamPtr tail = (VectorComplexFeatureStreamPtr&)beamformerP;
      OverSampledDFTSynthesisBankPtr synthesisFBP = new OverSampledDFTSynthesisBank(tail, g_fb, M, m, r);
      for (;;) {
        const gsl_vector_float* blk;
        try { blk = synthesisFBP->next(); } catch (jiterator_error& e) { break; }
        for (unsigned i = 0; i < D; i++) data.push_back(gsl_vector_float_get(blk, i));
      }
    }
    FILE* fo = fopen(argv[6], "wb");
    fwrite(data.data(), sizeof(float), data.size(), fo);
    fclose(fo);
    fprintf(stderr, "beamformer_mvdrgsc: %d channels, %lu samples written, %d identity fall-backs\n", nchan,
            (unsigned long)data.size(), beamformerP->identity_fallbacks());
    gsl_vector_free(h_fb); gsl_vector_free(g_fb); gsl_vector_free(delays); gsl_vector_free(packed); gsl_matrix_free(mpos);
  } catch (j_error& e) {
    fprintf(stderr, "j_error: %s\n", e.what());
    return 1;
  }
  return 0;
}
