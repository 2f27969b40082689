// modulated/modulated.h -- OverSampledDFTAnalysisBank / OverSampledDFTSynthesisBank with the
// reference's constructors (reference modulated/modulated.h:268-340), computed through libbtkhip.
#pragma once
#include <vector>
#include "stream/stream.h"
#include "common/devmem.h"

// Block protocol between the engine's nodes (not part of the reference's interface, which stays in stream/stream.h).  A
// reference node hands over one frame per next(); an engine node computes a BLOCK of frames per launch: at most block_frames
// frames (btk_default_block_frames(), set_block_frames() on the source nodes; 0 = the whole utterance in one block).  The
// analysis banks pull at most that many input blocks per round and keep the m R blocks of sample history the next round's
// first frame reaches back to; every node downstream works on the block its source currently holds and carries its recursion
// state (post-filter densities, RLS precision matrix and weights, the synthesis bank's last m R + R - 1 frames) into the next.
// A live source therefore yields its first output after one block, and memory stays bounded however long the stream runs.
//
// To keep the per-frame MEANING when weights change between two next() calls (the moving look direction of
// unit_test/test_online_beamforming.py:209-226), a consumer that batches -- the synthesis bank -- asks a producer that
// implements this interface for its current block instead of draining it through next(), tells it after every output it serves
// how far a per-frame graph would have pulled (advance_to), and re-fetches the block when block_version() changes: the producer
// recomputes only the frames beyond that mark with the new weights, everything already handed over keeps its value.
class BlockSource {
 public:
  virtual ~BlockSource() {}
  virtual unsigned long block_version() = 0;                 // changes whenever the frames not yet handed over may have changed
  virtual const std::vector<float>& block(long& T) = 0;      // complex64 [>= K rows][T], row k = bin k, frames <= the mark unchanged
  virtual long block_base() { return 0; }                    // stream index of the block's first frame
  virtual bool next_block() { return false; }                // move on to the following block; false: the stream has ended
  virtual void advance_to(long frame_idx) = 0;               // a per-frame graph would have pulled frames 0 .. frame_idx (stream indices) by now
  virtual bool has_block() { return true; }                  // a wrapper around a foreign object (pyStream) may have no block to offer: drain next()
  // The same block where it was computed: complex64 [>= K rows][T_stride] on the device, T valid frames per row, written by
  // launches on btk_node_stream() (a consumer that launches on that stream needs no synchronisation; the pointer stays valid
  // until the producer is advanced, reset or asked again).  NULL: the producer has only the host view above.
  virtual const void* device_block(long& T, long& T_stride) { T = 0; T_stride = 0; return NULL; }
};
// frames per block: the environment variable BTK_BLOCK_FRAMES, 8192 when unset (0: unbounded, one block per utterance)
long btk_default_block_frames();
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
  // frames per round (see BlockSource above); takes effect at the next round
  void set_block_frames(long n) { block_frames_ = n < 0 ? 0 : n; }
  long block_frames() const { return block_frames_; }
  // engine hooks used by the beamformer nodes to batch all channels into one launch: the bank as a sliding window of samples
  bool pull_more();                                     // up to block_frames() more input blocks (all of them for 0); false: nothing came
  void reserve_round();                                 // the window sized for one more round of pull_more()
  bool parallel_pull_ok() const;                        // pull_more() of this bank may run next to other banks' (SampleFeature source, bounded rounds)
  bool at_end() const { return eos_; }                  // the upstream node has ended
  long blocks_pulled() const { return nblk_; }
  long frames_ready() const;                            // frames 0 .. frames_ready() - 1 can be computed from what was pulled
  long window_first_block() const { return win_b0_; }
  const float* window(long b0) const { return static_cast<const float*>(win_.get()) + (size_t)(b0 - win_b0_) * D_; }   // samples of blocks b0 .. blocks_pulled() - 1 (pinned host memory)
  long first_block_of_frame(long t) const;              // oldest input block frame t reads (>= 0)
  void release_before(long t);                          // frames < t are done: drop the samples only they needed
  // 16-bit streaming (round 6): when the source is a SampleFeature that holds 16-bit PCM (SampleFeature::pcm16), a beamformer
  // node may switch the bank -- at the start of a stream, before the first pull -- to reading the utterance where it lies: a
  // round then moves no samples on the host at all (pull_more() only advances the source's state), window16() points into the
  // source's pinned int16 copy, nothing is ever released, and the node uploads 2 bytes per sample.  reset() ends the mode.
  bool i16_source_ok() const;                           // ... and nothing has been pulled yet
  void begin_i16();
  bool i16_mode() const { return src16_ != NULL; }
  const short* window16(long b0) const { return src16_ + src16_pos0_ + (size_t)b0 * D_; }
  const btk_fb_t* plan() const { return plan_; }
  unsigned delay_compensation_type() const { return dct_; }
  unsigned m() const { return m_; }
  unsigned r() const { return r_; }
 private:
  bool load_chunk_();
  VectorFloatFeatureStreamPtr samp_;
  unsigned M_, m_, r_, D_, dct_;
  btk_fb_t* plan_;
  const short* src16_;                                  // 16-bit mode: the source's int16 samples; block b of the stream starts at
  size_t src16_pos0_;                                   //   src16_[src16_pos0_ + b D]
  unsigned long src16_gen_;                             //   (the source's samples_generation() when the stream began)
  long block_frames_;
  void append_(const float* blocks, long n);            // n more input blocks into the window
  PinnedBuffer win_;                                    // samples of input blocks win_b0_ .. nblk_ - 1 (pinned: uploaded as they lie)
  long win_b0_, nblk_;
  bool eos_;
  std::vector<double> frames_;                          // the current block of frames, [chunk_len_][2M]
  long chunk_base_, chunk_len_;
  DeviceBuffer dPcm_, dX_;                              // a bank that is pulled frame by frame: its window and its block of frames
  PinnedBuffer hX_;
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
  // Engine extension (not in the reference): the blocks the next calls of next() would hand out, as many as the round that is
  // resident still holds (at most max_blocks; 0: all of them) -- one call, no copy.  Returns their number (0: the stream has ended;
  // is_end() is then set, as after next()'s jiterator_error) and points *blocks at n x shiftlen() floats that stay valid until the
  // next call of either kind; current() is the last of them.  A Python caller pays its per-call cost once per round instead of
  // once per 16 ms block.
  long next_blocks(long max_blocks, const float** blocks);
  virtual void reset();
  void input_source_vector(const gsl_vector_complex* block);
  void no_stream_feature(bool flag = true) { no_stream_feature_ = flag; }
  void set_block_frames(long n) { block_frames_ = n < 0 ? 0 : n; }     // rounds of a source that is drained through next()
  long block_frames() const { return block_frames_; }
  // engine hooks (SubbandGraphPool synthesises the blocks of many graphs in one launch with this bank's plan)
  const btk_fb_t* plan() const { return plan_; }
  unsigned fftlen() const { return M_; }
  unsigned shiftlen() const { return D_; }
  unsigned m() const { return m_; }
  unsigned r() const { return r_; }
  int gain_factor() const { return gain_; }
  void inputSourceVector(const gsl_vector_complex* block) { input_source_vector(block); }          // ENABLE_LEGACY_BTK_API aliases
  void doNotUseStreamFeature(bool flag = true) { no_stream_feature(flag); }
 private:
  void init_(gsl_vector* prototype, unsigned dct);
  const gsl_vector_float* next_pushed_();
  void prepare_();
  void synthesize_(const std::vector<float>& Yk, long T, long base, bool last, long keep_blocks);
  void synthesize_dev_(const void* dYk, long T, long T_stride, long base, long keep_blocks);
  void run_window_(long Lw, long w0, long b_first, long keep_blocks);
  VectorComplexFeatureStreamPtr samp_;
  unsigned M_, m_, r_, D_;
  int gain_;
  btk_fb_t* plan_;
  bool fetch_round_();
  PinnedBuffer blocks_;                                 // output blocks blk_base_ .. blk_base_ + nblocks_ - 1, float32 [nblocks_][D]
  long nblocks_, blk_base_;
  bool prepared_, src_ended_;
  long block_frames_;                                   // frames per round when the source is drained through next()
  std::vector<float> hist_;                             // the last <= m R + R - 1 input frames before the current round, [K][hist_len_]
  long hist_len_, frames_in_;                           // frames_in_: input frames of the rounds before the current one
  std::vector<float> cur_;                              // the current round's input frames [K][cur_T_] (drained sources)
  long cur_T_;
  BlockSource* bsrc_;                                   // samp_ seen as a block source (NULL: drained through next())
  unsigned long src_version_;
  bool no_stream_feature_;
  std::vector<float> ring_;                             // pushed frames, complex64 [W][K], oldest first (W = m R + R)
  long npushed_, npushed_at_next_;
  void *dWin_, *dBlk_;                                  // source-less form: device window [K][W] complex64 and one output block
  // rounds: the input window [history | this round's frames] complex64 [K][win_len_] and the output blocks, on the device; the
  // history of a device-resident source never visits the host (dev_hist_: hist_ is then empty and the window itself is the record)
  DeviceBuffer dRound_, dRoundNext_, dOut_;
  PinnedBuffer hRound_;
  long win_len_, win_pitch_;                            // frames in the window / frames between its rows (even)
  bool dev_hist_;
  long carry_cols_;                                     // >= 0: the next device round starts a new window with that many frames of the current one as history
  long prev_nblocks_;
};
typedef Inherit<OverSampledDFTSynthesisBank, VectorFloatFeatureStreamPtr> OverSampledDFTSynthesisBankPtr;
