// feature/feature.h -- SampleFeature (reference feature/feature.h:153-206, feature/feature.cc:221-680): the utterance
// holder at the head of every beamforming graph -- 16-bit PCM WAV in (un-normalised float blocks out), the sample-level
// helpers the reference's scripts call on it, and the 16-bit WAV writer at the tail.
// Not carried over: sample-rate conversion (the reference's is behind #ifdef SRCONV, off in its build) and the non-WAV
// libsndfile containers (read()/write() raise jio_error for them).
#pragma once
#include <vector>
#include "stream/stream.h"
#include "common/devmem.h"

namespace sndfile {      // the two libsndfile constants callers of write() spell out (sndfile.h values)
enum { SF_FORMAT_WAV = 0x010000, SF_FORMAT_PCM_16 = 0x0002 };
}

class SampleFeature;
typedef Inherit<SampleFeature, VectorFloatFeatureStreamPtr> SampleFeaturePtr;

class SampleFeature : public VectorFloatFeatureStream {
 public:
  SampleFeature(const String& fn = "", unsigned blockLen = 320, unsigned shiftLen = 160, bool padZeros = false,
                const String& nm = "Sample");
  virtual ~SampleFeature();
  // `format` is the reference's 2nd positional argument (callers pass the sample rate there)
  unsigned read(const String& fn, int format = 0, int samplerate = 16000, int chX = 1, int chN = 1,
                int cfrom = 0, int to = -1, int outsamplerate = -1, float norm = 0.0);
  // 16-bit PCM WAV, one channel; with norm == 0 at read time the samples are taken as int16-scale values, otherwise as
  // [-1, 1) values divided by that norm first (feature.cc:429-518)
  void write(const String& fn, int format = sndfile::SF_FORMAT_WAV | sndfile::SF_FORMAT_PCM_16, int sampleRate = -1);
  void cut(unsigned cfrom, unsigned cto);
  void randomize(int startX, int endX, double sigma2);
  virtual const gsl_vector_float* next(int frame_no = -5);
  virtual void reset() { cur_ = 0; VectorFloatFeatureStream::reset(); is_end_ = false; }
  void exit() { reset(); throw jiterator_error("end of samples!"); }
  const gsl_vector_float* data();
  const gsl_vector* dataDouble();
  void copySamples(SampleFeaturePtr& src, unsigned cfrom, unsigned to);
  unsigned samplesN() const { return (unsigned)samples_.size(); }
  int getSampleRate() const { return samplerate_; }
  int getChanN() const { return nChan_; }
  void zeroMean();
  void addWhiteNoise(float snr);
  void setSamples(const gsl_vector* samples, unsigned sampleRate);
  void set_samples(const float* samples, size_t n);          // in-memory source (same state as after read())
  // engine hook: up to nmax consecutive next() calls at once, the blocks stored back to back in dst (nmax * size() floats);
  // returns how many there were -- fewer than nmax: the stream has ended exactly as the throwing next() would have ended it
  long next_blocks(float* dst, long nmax);
  // engine hooks (round 6): the utterance AS 16-BIT PCM.  A WAV holds int16 samples and read() with norm == 0 hands them out as
  // un-normalised floats (feature/feature.cc:265-269), so an analysis bank on the device may take the samples as they were stored:
  // half the bytes through host memory and PCIe, and -- the widening is exact -- the same bits out of the filter bank.
  // pcm16(): the loaded samples as int16 in pinned memory, zero-padded by at least two blocks behind the last sample; NULL when a
  // sample is not an integer of the int16 range (normalised reads, randomize(), noise that overflowed) or blocks overlap
  // (shiftLen != blockLen).  Built when first asked for after the samples changed; stays where it is until they change again.
  const short* pcm16();
  // the state transitions of up to nmax next() calls WITHOUT the copies (the caller reads the blocks out of pcm16()): returns how
  // many there were and the sample index the first of them starts at; fewer than nmax: the stream has ended as next() ends it
  long advance_blocks(long nmax, size_t* first_sample);
  unsigned shiftlen() const { return shiftLen_; }
  // counts the changes of the samples (a reader that keeps the pcm16() pointer compares it before every use)
  unsigned long samples_generation() const { return samples_gen_; }
 private:
  SampleFeature(const SampleFeature&);
  SampleFeature& operator=(const SampleFeature&);
  std::vector<float> samples_;
  bool have_samples_;
  float norm_;
  unsigned shiftLen_;
  size_t cur_;
  bool pad_zeros_;
  int samplerate_, nChan_, format_;
  gsl_vector_float* copy_fsamples_;
  gsl_vector* copy_dsamples_;
  PinnedBuffer pcm16_;           // int16 shadow of samples_ (pcm16())
  int pcm16_state_;              // 0: not built for the current samples, 1: valid, -1: the samples are not 16-bit PCM
  unsigned long samples_gen_;
  void samples_changed_() { pcm16_state_ = 0; samples_gen_++; }
};
