// modulated/modulated.h -- OverSampledDFTAnalysisBank / OverSampledDFTSynthesisBank with the
// reference's constructors (reference modulated/modulated.h:268-340), computed through libbtkhip.
#pragma once
#include <vector>
#include "stream/stream.h"
#include <vector>

// Virtual-pull protocol between the engine's nodes (not part of the reference's interface, which stays in stream/stream.h).  A
// reference node hands over one frame per next(); an engine node computes the whole utterance in one launch.  To keep the
// per-frame MEANING when weights change between two next() calls (the moving look direction of
// unit_test/test_online_beamforming.py:209-226), a consumer that batches -- the synthesis bank -- asks a producer that
// implements this interface for its whole block instead of draining it through next(), tells it after every block it serves how
// far a per-frame graph would have pulled (advance_to), and re-fetches the block when block_version() changes: the producer
// recomputes only the frames beyond that mark with the new weights, everything already handed over keeps its value.
class BlockSource {
 public:
  virtual ~BlockSource() {}
  virtual unsigned long block_version() = 0;                 // changes whenever the frames not yet handed over may have changed
  virtual const std::vector<float>& block(long& T) = 0;      // complex64 [>= K rows][T], row k = bin k, frames <= the mark unchanged
  virtual void advance_to(long frame_idx) = 0;               // a per-frame graph would have pulled frames 0 .. frame_idx by now
  virtual bool has_block() { return true; }                  // a wrapper around a foreign object (pyStream) may have no block to offer: drain next()
};
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
  // source-less form (reference modulated/modulated.h:320-334, modulated.cc:500-518): the caller pushes one subband frame with
  // input_source_vector() and pulls one block with next(); the block is synthesised on the device from the ring of the last
  // m R + R frames.  A per-frame graph pushes exactly one frame per next(); any other pattern (whose ring the batch kernel's
  // frame sequence cannot express) raises jconsistency_error instead of returning a different signal.
  OverSampledDFTSynthesisBank(gsl_vector* prototype, unsigned M, unsigned m, unsigned r = 0, unsigned delayCompensationType = 0,
                              int gainFactor = 1, const String& nm = "OverSampledDFTSynthesisBank");
  ~OverSampledDFTSynthesisBank();
  virtual const gsl_vector_float* next(int frame_no = -5);
  virtual void reset();
  void input_source_vector(const gsl_vector_complex* block);
  void no_stream_feature(bool flag = true) { no_stream_feature_ = flag; }
  void inputSourceVector(const gsl_vector_complex* block) { input_source_vector(block); }          // ENABLE_LEGACY_BTK_API aliases
  void doNotUseStreamFeature(bool flag = true) { no_stream_feature(flag); }
 private:
  void init_(gsl_vector* prototype, unsigned dct);
  const gsl_vector_float* next_pushed_();
  void prepare_();
  void synthesize_(const std::vector<float>& Yk, long T, long keep_blocks);
  VectorComplexFeatureStreamPtr samp_;
  unsigned M_, m_, r_, D_;
  int gain_;
  btk_fb_t* plan_;
  std::vector<float> blocks_;                           // [B][D]
  long nblocks_;
  bool prepared_;
  BlockSource* bsrc_;                                   // samp_ seen as a block source (NULL: drained through next())
  unsigned long src_version_;
  bool no_stream_feature_;
  std::vector<float> ring_;                             // pushed frames, complex64 [W][K], oldest first (W = m R + R)
  long npushed_, npushed_at_next_;
  void *dWin_, *dBlk_;                                  // device window [K][W] complex64 and one output block
};
typedef Inherit<OverSampledDFTSynthesisBank, VectorFloatFeatureStreamPtr> OverSampledDFTSynthesisBankPtr;
