// feature/feature.h -- SampleFeature block reader (reference feature/feature.h:153-206,
// feature/feature.cc:238-389, 605-649): 16-bit PCM WAV -> un-normalised float blocks.
#pragma once
#include <vector>
#include "stream/stream.h"

class SampleFeature : public VectorFloatFeatureStream {
 public:
  SampleFeature(const String& fn = "", unsigned blockLen = 320, unsigned shiftLen = 160, bool padZeros = false,
                const String& nm = "Sample");
  virtual ~SampleFeature() {}
  // `format` is the reference's 2nd positional argument (callers pass the sample rate there)
  unsigned read(const String& fn, int format = 0, int samplerate = 16000, int chX = 1, int chN = 1,
                int cfrom = 0, int to = -1, int outsamplerate = -1, float norm = 0.0);
  void set_samples(const float* samples, size_t n);          // in-memory source (same state as after read())
  virtual const gsl_vector_float* next(int frame_no = -5);
  virtual void reset() { cur_ = 0; VectorFloatFeatureStream::reset(); is_end_ = false; }
  int getSampleRate() const { return samplerate_; }
  unsigned samplesN() const { return (unsigned)samples_.size(); }
 private:
  std::vector<float> samples_;
  bool have_samples_;
  unsigned shiftLen_;
  size_t cur_;
  bool pad_zeros_;
  int samplerate_;
};
typedef Inherit<SampleFeature, VectorFloatFeatureStreamPtr> SampleFeaturePtr;
