// beamformer_sd.cc -- the flow of the reference's superdirective main (src/superdirectiveBeamformer.cc:151-248) through the
// C++ node layer, written against the ENABLE_LEGACY_BTK_API names that main uses:
//   SampleFeature -> OverSampledDFTAnalysisBank xN -> SubbandMVDR -> ZelinskiPostFilter -> OverSampledDFTSynthesisBank
//   setChannel; calcArrayManifoldVectors; setDiffuseNoiseModel; divideAllNonDiagonalElements(mu); calcMVDRWeights;
//   output->setBeamformer(beamformerP); pull synthesis blocks until jiterator_error.
// usage: beamformer_sd coeffs.f64 M m r pf alpha mu out.f32 mpos "x,y,z;..." {delay wav}...
// With BTK_EXAMPLE_NC=2|3 the quiescent vector is the LCMV one (calcArrayManifoldVectors2 / N): the interference delays
// are the look delays scaled by -0.5 (and +0.25 for the third constraint) -- deterministic test directions.
// With BTK_EXAMPLE_FIR=path the GSC form of the same chain writes its FIR taps (SubbandGSC::writeFIRCoeff) and exits.
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
#include "postfilter/postfilter.h"

int main(int argc, char** argv)
{
  if (argc < 12 || ((argc - 10) % 2) != 0) {
    fprintf(stderr, "usage: %s coeffs.f64 M m r pf alpha mu out.f32 mpos {delay wav}...\n", argv[0]);
    return 2;
  }
  const unsigned fftLen = atoi(argv[2]), m = atoi(argv[3]), r = atoi(argv[4]);
  const int pf = atoi(argv[5]);
  const double alpha = atof(argv[6]);
  const float mu = atof(argv[7]);
  const unsigned D = fftLen >> r;
  const double sampleRate = 16000.0;
  const int nchan = (argc - 10) / 2;
  try {
    gsl_vector* h_fb = gsl_vector_calloc(m * fftLen);
    gsl_vector* g_fb = gsl_vector_calloc(m * fftLen);
    FILE* fc = fopen(argv[1], "rb");
    if (!fc || fread(h_fb->data, sizeof(double), m * fftLen, fc) != m * fftLen || fread(g_fb->data, sizeof(double), m * fftLen, fc) != m * fftLen) {
      fprintf(stderr, "cannot read %s\n", argv[1]); return 2;
    }
    fclose(fc);
    gsl_matrix* micPos = gsl_matrix_alloc(nchan, 3);
    {
      std::string str(argv[9]);
      size_t pos = 0;
      for (int c = 0; c < nchan; c++)
        for (int j = 0; j < 3; j++) {
          size_t used = 0;
          gsl_matrix_set(micPos, c, j, std::stod(str.substr(pos), &used));
          pos += used + 1;
        }
    }
    gsl_vector* delays = gsl_vector_calloc(nchan);
    for (int c = 0; c < nchan; c++) gsl_vector_set(delays, c, atof(argv[10 + 2 * c]));

    if (const char* fir = getenv("BTK_EXAMPLE_FIR")) {          // SubbandGSC::writeFIRCoeff (beamformer.cc:775-828, 1364-1371)
      SubbandGSCPtr gscP = new SubbandGSC(fftLen, false);
      std::list<OverSampledDFTAnalysisBankPtr> keep;
      for (int c = 0; c < nchan; c++) {
        SampleFeaturePtr sampleFeatureP = new SampleFeature("", D, D, true);
        sampleFeatureP->read(argv[11 + 2 * c], sampleRate);
        OverSampledDFTAnalysisBankPtr analysisFBP = new OverSampledDFTAnalysisBank((VectorFloatFeatureStreamPtr&)sampleFeatureP, h_fb, fftLen, m, r);
        gscP->setChannel((VectorComplexFeatureStreamPtr&)analysisFBP);
        keep.push_back(analysisFBP);
      }
      gscP->calcGSCWeights(sampleRate, delays);
      gsl_vector* packed = gsl_vector_calloc(2 * (nchan - 1));
      for (unsigned k = 1; k <= fftLen / 2; k++) {
        for (int i = 0; i < nchan - 1; i++) {
          gsl_vector_set(packed, 2 * i, 0.05 * cos(0.37 * k + i));
          gsl_vector_set(packed, 2 * i + 1, 0.05 * sin(0.11 * k * (i + 1)));
        }
        gscP->setActiveWeights_f(k, packed);
      }
      const gsl_matrix_complex* B5 = gscP->getBlockingMatrix(0, 5);
      fprintf(stderr, "beamformer_sd: blocking matrix of bin 5 is %lu x %lu\n", (unsigned long)B5->size1, (unsigned long)B5->size2);
      return gscP->writeFIRCoeff(fir, 1) ? 0 : 1;
    }

    std::vector<SampleFeaturePtr> sampleFeaturePL;
    std::vector<OverSampledDFTAnalysisBankPtr> analysisFBPL;
    SubbandMVDRPtr beamformerP = new SubbandMVDR(fftLen, false);
    ZelinskiPostFilterPtr output = new ZelinskiPostFilter((VectorComplexFeatureStreamPtr&)beamformerP, fftLen, alpha, pf);
    OverSampledDFTSynthesisBankPtr synthesisFBP = new OverSampledDFTSynthesisBank((VectorComplexFeatureStreamPtr&)output, g_fb, fftLen, m, r);
    for (int c = 0; c < nchan; c++) {
      SampleFeaturePtr sampleFeatureP = new SampleFeature("", D, D, true);
      sampleFeatureP->read(argv[11 + 2 * c], sampleRate);
      OverSampledDFTAnalysisBankPtr analysisFBP = new OverSampledDFTAnalysisBank((VectorFloatFeatureStreamPtr&)sampleFeatureP, h_fb, fftLen, m, r);
      beamformerP->setChannel((VectorComplexFeatureStreamPtr&)analysisFBP);
      sampleFeaturePL.push_back(sampleFeatureP);
      analysisFBPL.push_back(analysisFBP);
    }
    const int NC = getenv("BTK_EXAMPLE_NC") ? atoi(getenv("BTK_EXAMPLE_NC")) : 1;
    if (NC == 1) {
      beamformerP->calcArrayManifoldVectors(sampleRate, delays);
    } else if (NC == 2) {
      gsl_vector* delaysJ = gsl_vector_calloc(nchan);
      for (int c = 0; c < nchan; c++) gsl_vector_set(delaysJ, c, -0.5 * gsl_vector_get(delays, c));
      beamformerP->calcArrayManifoldVectors2(sampleRate, delays, delaysJ);
      gsl_vector_free(delaysJ);
    } else {
      gsl_matrix* delaysJ = gsl_matrix_alloc(NC - 1, nchan);
      const double scale[3] = {-0.5, 0.25, 0.8};
      for (int n = 0; n < NC - 1; n++)
        for (int c = 0; c < nchan; c++) gsl_matrix_set(delaysJ, n, c, scale[n % 3] * gsl_vector_get(delays, c));
      beamformerP->calcArrayManifoldVectorsN(sampleRate, delays, delaysJ, NC);
      gsl_matrix_free(delaysJ);
    }
    beamformerP->setDiffuseNoiseModel(micPos, sampleRate, SSPEED);
    beamformerP->divideAllNonDiagonalElements(mu);
    beamformerP->calcMVDRWeights(sampleRate, 1.0E-8);

    std::list<float> dataFL;
    output->setBeamformer(beamformerP);
    for (;;) {
      const gsl_vector_float* data;
      try {
        data = synthesisFBP->next();
        if (true == analysisFBPL[0]->isEnd()) { /* the reference stops here; the block is still pulled below */ }
      } catch (jiterator_error& e) {
        break;
      }
      for (unsigned i = 0; i < D; i++) dataFL.push_back(gsl_vector_float_get(data, i));
    }
    std::vector<float> out(dataFL.begin(), dataFL.end());
    FILE* fo = fopen(argv[8], "wb");
    fwrite(out.data(), sizeof(float), out.size(), fo);
    fclose(fo);
    const gsl_matrix_complex* R10 = beamformerP->getNoiseSpatialSpectralMatrix(10);
    fprintf(stderr, "beamformer_sd: %d channels, %lu samples written, %d identity fall-backs, R_10[0][1] = %.9g\n", nchan,
            (unsigned long)out.size(), beamformerP->identity_fallbacks(), gsl_matrix_complex_get(R10, 0, 1).dat[0]);
    gsl_vector_free(h_fb); gsl_vector_free(g_fb); gsl_vector_free(delays); gsl_matrix_free(micPos);
  } catch (j_error& e) {
    fprintf(stderr, "j_error: %s\n", e.what());
    return 1;
  }
  return 0;
}
