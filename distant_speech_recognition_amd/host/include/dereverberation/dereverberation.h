// dereverberation/dereverberation.h -- WPE nodes of the C++ layer with the reference's interface
// (reference dereverberation/dereverberation.h:31-190, dereverberation.cc:40-307, 312-760):
//   SingleChannelWPEDereverberationFeature, MultiChannelWPEDereverberation, MultiChannelWPEDereverberationFeature.
// estimate_filter() drains the (finite) inputs, runs btk_wpe_estimate over the chosen frame range and resets the
// inputs like the reference; the outputs of the whole utterance come from one btk_wpe_apply launch and are served
// frame by frame (node-owned buffer, consecutive frame numbers, jiterator_error at the end).
#pragma once
#include <vector>

#include "stream/stream.h"

class MultiChannelWPEDereverberation : public Countable {
 public:
  MultiChannelWPEDereverberation(unsigned subbandsN, unsigned channelsN, unsigned lowerN, unsigned upperN,
                                 unsigned iterationsN = 2, double loadDb = -20.0, double bandWidth = 0.0,
                                 double diagonal_bias = 0.0, double sampleRate = 16000.0);
  ~MultiChannelWPEDereverberation();
  unsigned size() const { return subbandsN_; }
  void reset();
  void set_input(VectorComplexFeatureStreamPtr& samples);
  const gsl_vector_complex* get_output(unsigned channelX);
  gsl_vector_complex** calc_every_channel_output(int frame_no = -5);
  unsigned estimate_filter(int start_frame_no = 0, int frame_num = -1);
  void reset_filter();
  void next_speaker();
  void print_objective_func(int subbandX) { (void)subbandX; }
  int frame_no() const { return frame_no_; }
  // whole-utterance output block, complex64 [K][C][T] (prepared on first use after estimate_filter / reset)
  const std::vector<float>& output_block(long* T);
  unsigned channelsN() const { return channelsN_; }
  void setInput(VectorComplexFeatureStreamPtr& samples) { set_input(samples); }
  const gsl_vector_complex* getOutput(unsigned channelX, int frame_no = -5) { (void)frame_no; return get_output(channelX); }
  void nextSpeaker() { next_speaker(); }
 private:
  long snapshots_(void** dX);                    // drains the inputs: X complex64 [K][C][T] on the device, returns T
  void prepare_output_();
  std::vector<VectorComplexFeatureStreamPtr> sources_;
  const unsigned subbandsN_, channelsN_, lowerN_, upperN_, iterationsN_;
  const double load_db_, diagonal_bias_;
  unsigned lower_bw_, upper_bw_;
  bool estimated_;
  unsigned framesN_;
  void* dG_;                                     // complex64 [C][K][C*L], zeros = a fresh object / next_speaker()
  std::vector<float> out_;                       // complex64 [K][C][T] host mirror of the dereverberated utterance
  long T_;
  bool have_out_;
  gsl_vector_complex** output_;
  int frame_no_;
};
typedef refcountable_ptr<MultiChannelWPEDereverberation> MultiChannelWPEDereverberationPtr;

class MultiChannelWPEDereverberationFeature : public VectorComplexFeatureStream {
 public:
  MultiChannelWPEDereverberationFeature(MultiChannelWPEDereverberationPtr& source, unsigned channelX, unsigned primaryChannelX = 0,
                                        const String& nm = "MultiChannelWPEDereverberationFeature");
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual void reset();
 private:
  MultiChannelWPEDereverberationPtr source_;
  const unsigned channelX_, primaryChannelX_;
};
typedef Inherit<MultiChannelWPEDereverberationFeature, VectorComplexFeatureStreamPtr> MultiChannelWPEDereverberationFeaturePtr;

// the C = 1 case of the same estimator, without a diagonal bias (dereverberation.cc:40-307)
class SingleChannelWPEDereverberationFeature : public VectorComplexFeatureStream {
 public:
  SingleChannelWPEDereverberationFeature(VectorComplexFeatureStreamPtr& samples, unsigned lowerN, unsigned upperN,
                                         unsigned iterationsN = 2, double loadDb = -20.0, double bandWidth = 0.0,
                                         double sampleRate = 16000.0, const String& nm = "SingleChannelWPEDereverberationFeature");
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual void reset();
  unsigned estimate_filter(int start_frame_no = 0, int frame_num = -1) { return core_->estimate_filter(start_frame_no, frame_num); }
  void reset_filter() { core_->reset_filter(); }
  void next_speaker();
  void print_objective_func(int subbandX) { (void)subbandX; }
  void nextSpeaker() { next_speaker(); }
 private:
  MultiChannelWPEDereverberationPtr core_;
};
typedef Inherit<SingleChannelWPEDereverberationFeature, VectorComplexFeatureStreamPtr> SingleChannelWPEDereverberationFeaturePtr;
