// modulated/modulated.h -- OverSampledDFTAnalysisBank / OverSampledDFTSynthesisBank with the
// reference's constructors (reference modulated/modulated.h:268-340), computed through libbtkhip.
#pragma once
#include <vector>
#include "stream/stream.h"
#include "btkhip.h"

class OverSampledDFTAnalysisBank : public VectorComplexFeatureStream {
 public:
  OverSampledDFTAnalysisBank(VectorFloatFeatureStreamPtr& samp, gsl_vector* prototype, unsigned M, unsigned m, unsigned r,
                             unsigned delayCompensationType = 0, const String& nm = "OverSampledDFTAnalysisBank");
  ~OverSampledDFTAnalysisBank();
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual void reset();
  unsigned fftlen() const { return M_; }
  unsigned shiftlen() const { return D_; }
  bool isEnd() { return is_end(); }                     // src/superdirectiveBeamformer.cc:206 calls it (the reference's own header lost it)
  unsigned fftLen() const { return fftlen(); }          // ENABLE_LEGACY_BTK_API aliases
  unsigned nBlocks() const { return m_; }
  unsigned subSampRate() const { return r_; }
  // engine hooks used by the beamformer nodes to batch all channels into one launch
  const std::vector<float>& pcm();                      // drains the upstream node once
  const btk_fb_t* plan() const { return plan_; }
  unsigned delay_compensation_type() const { return dct_; }
  unsigned m() const { return m_; }
  unsigned r() const { return r_; }
 private:
  void prepare_();
  VectorFloatFeatureStreamPtr samp_;
  unsigned M_, m_, r_, D_, dct_;
  btk_fb_t* plan_;
  std::vector<float> pcm_;
  bool drained_;
  std::vector<double> frames_;                          // [T][2M]
  long nframes_;
  bool prepared_;
};
typedef Inherit<OverSampledDFTAnalysisBank, VectorComplexFeatureStreamPtr> OverSampledDFTAnalysisBankPtr;

class OverSampledDFTSynthesisBank : public VectorFloatFeatureStream {
 public:
  OverSampledDFTSynthesisBank(VectorComplexFeatureStreamPtr& samp, gsl_vector* prototype, unsigned M, unsigned m, unsigned r = 0,
                              unsigned delayCompensationType = 0, int gainFactor = 1,
                              const String& nm = "OverSampledDFTSynthesisBank");
  ~OverSampledDFTSynthesisBank();
  virtual const gsl_vector_float* next(int frame_no = -5);
  virtual void reset();
 private:
  void prepare_();
  VectorComplexFeatureStreamPtr samp_;
  unsigned M_, m_, r_, D_;
  int gain_;
  btk_fb_t* plan_;
  std::vector<float> blocks_;                           // [B][D]
  long nblocks_;
  bool prepared_;
};
typedef Inherit<OverSampledDFTSynthesisBank, VectorFloatFeatureStreamPtr> OverSampledDFTSynthesisBankPtr;
