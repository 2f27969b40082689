// beamformer/beamformer.h -- SnapShotArray, BeamformerWeights, SubbandBeamformer, SubbandDS, SubbandGSC,
// SubbandMVDR with the reference's method names (reference beamformer/beamformer.h:28-383,
// beamformer/spectralinfoarray.h:6-36), computed through libbtkhip.
#pragma once
#include <complex>
#include <functional>
#include <list>
#include <vector>
#include "stream/stream.h"
#include "modulated/modulated.h"

#define SSPEED 343740.0

class SnapShotArray : public Countable {
 public:
  SnapShotArray(unsigned fftLn, unsigned nChn);
  virtual ~SnapShotArray();
  const gsl_vector_complex* snapshot(unsigned fbinX) const { return snapshots_[fbinX]; }
  void set_samples(const gsl_vector_complex* samp, unsigned chanX);
  // one bin's snapshot directly (reference beamformer.cc:79-93, used by the modal beamformers): the conjugate goes to bin
  // fftLen/2 - fbinX exactly as the reference writes it; do not call update() afterwards
  void set_snapshots(const gsl_vector_complex* snapshots, unsigned fbinX);
  const gsl_vector_complex* getSnapShot(unsigned fbinX) const { return snapshot(fbinX); }          // ENABLE_LEGACY_BTK_API aliases
  void newSample(const gsl_vector_complex* samp, unsigned chanX) { set_samples(samp, chanX); }     // (spectralinfoarray.h:26-27)
  virtual void update();
  virtual void zero();
  unsigned fftLen() const { return fftLen_; }
  unsigned nChan() const { return nChan_; }
  gsl_vector_complex** raw_snapshots() { return snapshots_; }
 protected:
  const unsigned fftLen_, nChan_;
  gsl_vector_complex** samples_;
  gsl_vector_complex** snapshots_;
};
typedef refcountable_ptr<SnapShotArray> SnapShotArrayPtr;

// SnapShotArray that also keeps, per bin, the recursively averaged outer product of the snapshot with ITSELF -- without
// conjugation, exactly as the reference writes it: R_k <- mu R_k + (1 - mu) x_k x_k^T (reference
// beamformer/spectralinfoarray.h:43-64, beamformer.cc:97-143).  Host container like SnapShotArray; the Hermitian covariance the
// beamformers need comes from btk_cov_accumulate on the device.  FBSpectralMatrixArray (:70-78, :147-173) indexes the
// per-channel sample vectors with a bin number (out of bounds whenever fftLen > nChan) and is not mirrored.
class SpectralMatrixArray : public SnapShotArray {
 public:
  SpectralMatrixArray(unsigned fftLn, unsigned nChn, float forgetFact = 0.95);
  virtual ~SpectralMatrixArray();
  gsl_matrix_complex* matrix_f(unsigned idx) const { return matrices_[idx]; }
  virtual void update();
  virtual void zero();
  gsl_matrix_complex* getSpecMatrix(unsigned idx) { return matrix_f(idx); }
 protected:
  const double mu_;
  gsl_matrix_complex** matrices_;
};
typedef refcountable_ptr<SpectralMatrixArray> SpectralMatrixArrayPtr;

// Host-side weights (float64): wq, B, wa, wl, ta  (reference beamformer.h:28-82)
class BeamformerWeights {
 public:
  BeamformerWeights(unsigned fftLen, unsigned chanN, bool halfBandShift, unsigned NC = 1);
  ~BeamformerWeights();
  void calcMainlobe(float samplerate, const gsl_vector* delays, bool isGSC);
  void calcMainlobe2(float samplerate, const gsl_vector* delaysT, const gsl_vector* delaysJ, bool isGSC);
  void calcMainlobeN(float samplerate, const gsl_vector* delaysT, const gsl_matrix* delaysIs, unsigned NC, bool isGSC);
  void calcSidelobeCancellerP_f(unsigned fbinX, const gsl_vector* packedWeight);
  void calcSidelobeCancellerU_f(unsigned fbinX, const std::complex<double>* wa);
  void calcSidelobeCancellerU_f(unsigned fbinX, const gsl_vector_complex* wa);
  void calcBlockingMatrix(unsigned fbinX);
  // FIR taps of the effective weights wq - B wa per channel (inverse FFT of the conjugated, half-length-shifted weights,
  // windowed): reference beamformer.cc:775-828.  winType: 0 rectangle, 2 Hanning, anything else Hamming (modulated.cc:47-72)
  bool write_fir_coeff(const String& fn, unsigned winType);
  bool writeFIRCoeff(const String& fn, unsigned winType) { return write_fir_coeff(fn, winType); }
  void setSidelobeCanceller_f(unsigned fbinX, gsl_vector_complex* wl_f);
  void setQuiescentVector(unsigned fbinX, gsl_vector_complex* wq_f, bool isGSC = false);
  void setQuiescentVectorAll(gsl_complex z, bool isGSC = false);
  void setTimeAlignment() { ta_v = wq_v; }
  // The reference's accessors (beamformer/beamformer.h:53-67), same names and return types.  The gsl objects ALIAS this object's
  // storage (owner = 0): what a caller writes through wq()[k], B()[k], wa()[k] is what the nodes compute with, as in the
  // reference.  CSDs(): the reference's post-filters keep the N x N auto / cross spectral densities of every bin here (upper triangle
  // i < j and the diagonal, postfilter.cc:77-116); this engine keeps only the recursively averaged SUMS the gains depend on, on the
  // device, and REBUILDS the matrices when CSDs() is called: the post-filter bound to this object installs
  // a provider that runs one exponentially weighted covariance launch over the frames it has served (ZelinskiPostFilter::fill_csds_).
  // Without a post-filter the vectors are what the reference's are at that point: zero.  wp1(): the post-filter gains of the last served frame, mirrored
  // from the post-filter node that is bound to this weight object (ZelinskiPostFilter::postfilter_weights()).
  bool isHalfBandShift() const { return halfBandShift_; }
  gsl_vector_complex** arrayManifold() const { return ta_views_; }
  gsl_vector_complex* wq_f(unsigned fbinX) const { return wq_views_[fbinX]; }
  gsl_vector_complex* wl_f(unsigned fbinX) const { return wl_views_[fbinX]; }
  gsl_matrix_complex* B_f(unsigned fbinX) const { return B_views_ ? B_views_[fbinX] : NULL; }
  gsl_vector_complex** wq() const { return wq_views_; }
  gsl_matrix_complex** B() const { return B_views_; }
  gsl_vector_complex** wa() const { return wa_views_; }
  gsl_vector_complex** CSDs() const { if (csd_provider_) csd_provider_(CSDs_); return CSDs_; }
  // owner: the post-filter that installs / removes the provider; a filter only removes the provider it installed itself (another
  // filter may have bound itself to the same weight object since)
  void set_csd_provider(const std::function<void(gsl_vector_complex**)>& f, const void* owner = NULL) { csd_provider_ = f; csd_owner_ = f ? owner : NULL; }
  void clear_csd_provider(const void* owner) { if (csd_owner_ == owner) { csd_provider_ = nullptr; csd_owner_ = NULL; } }
  gsl_vector_complex* wp1() const { return wp1_; }
  unsigned fftLen() const { return fftLen_; }
  unsigned chanN() const { return chanN_; }
  unsigned NC() const { return NC_; }
  // storage behind the views: [M][N], [M][N], [M][N], [M][N-NC], [M][N][N-NC]  (round 2 exposed these as wq / wl / ta / wa / B)
  std::vector<std::complex<double> > wq_v, wl_v, ta_v, wa_v, B_v;
 private:
  unsigned fftLen_, chanN_, NC_;
  bool halfBandShift_;
  gsl_vector_complex **wq_views_, **wl_views_, **ta_views_, **wa_views_, **CSDs_;
  gsl_matrix_complex** B_views_;
  gsl_vector_complex* wp1_;
  std::function<void(gsl_vector_complex**)> csd_provider_;
  const void* csd_owner_ = NULL;
};

class SubbandBeamformer : public VectorComplexFeatureStream {
 public:
  SubbandBeamformer(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandBeamformer");
  ~SubbandBeamformer();
  bool is_end() { return is_end_; }
  bool isEnd() { return is_end(); }
  unsigned fftLen() const { return fftLen_; }
  unsigned fftLen2() const { return fftLen2_; }
  unsigned chanN() const { return (unsigned)channelList_.size(); }
  virtual void reset();
  unsigned dim() const { return fftLen_; }
  void set_channel(VectorComplexFeatureStreamPtr& chan);
  virtual void clear_channel();
  const gsl_vector_complex* snapshot_array_f(unsigned fbinX) { return snapshot_array()->snapshot(fbinX); }
  virtual SnapShotArrayPtr snapshot_array();
  void setChannel(VectorComplexFeatureStreamPtr& chan) { set_channel(chan); }       // ENABLE_LEGACY_BTK_API aliases
  void clearChannel() { clear_channel(); }
  const gsl_vector_complex* snapShotArray_f(unsigned fbinX) { return snapshot_array_f(fbinX); }
  SnapShotArrayPtr getSnapShotArray() { return snapshot_array(); }
  bool is_half_band_shift() const { return halfBandShift_; }
  bool isHalfBandShift() const { return halfBandShift_; }
  // device hooks: the CURRENT BLOCK of the stream, frames chunk_base() .. chunk_base() + num_frames() - 1 (modulated/modulated.h,
  // BlockSource).  Over analysis banks the block's PCM windows are what is resident on the device; the snapshots are computed
  // from them when somebody asks -- a post-filter, an adaptive canceller, snapshot_array(), a Python beamformer --, and from
  // then on with every block.  A fixed-weight beamformer nobody asks runs the fused analysis -> apply kernel and the
  // N x K x T snapshots never exist.
  void* device_snapshots();       // complex64 [1][K][N][T] on the device, complete on return (the first block is loaded on demand)
  void* device_snapshots_all_bins() { device_snapshots(); return dXfull_; }   // halfBandShift over pulled sources: [1][M][N][T], else NULL
  long num_frames() { ensure_chunk_(); return T_; }
  long chunk_base() { ensure_chunk_(); return chunk_base_; }
  bool next_chunk();              // drop the current block and load the following one; false: the channels have ended
  bool snapshots_materialised() const { return snap_valid_; }     // (tests: which path a block took)
  void want_snapshots() { snapshots_wanted_ = true; }             // every block from now on brings its snapshots along (staged path)
  // frames per block for channels that are pulled through next(); analysis banks bring their own block_frames()
  void set_block_frames(long n) { block_frames_ = n < 0 ? 0 : n; }
  long block_frames() const { return block_frames_; }
  // Blocks end at stream indices that are multiples of q (the last one excepted).  The recursions that run as parallel scans over
  // 64-frame chunks (post-filter densities, the NLMS step-size control) round differently when a block boundary falls inside a
  // chunk; the nodes that own such a recursion ask for q = 64 so that their output does not depend on the block size at all.
  // A block then holds at least q frames; BTK_BLOCK_QUANTUM=1 in the environment trades that bit-equality for latency.
  void set_block_quantum(long q);
  long block_quantum() const { return quantum_; }
  // The next block of a beamformer whose channels are all analysis banks of one geometry, as SubbandGraphPool batches it with the
  // blocks of other graphs: which frames [f0, f0 + T), from which input block b0 on (L samples per channel), whether the stream
  // ends with it.  plan pulls the banks' input; commit makes the block the node's current one (without loading anything).
  struct BlockPlan { long f0, T, b0, L; bool ended; };
  bool banks_only();                                     // (re)scans the channel list
  void plan_bank_block(BlockPlan& p);
  void commit_bank_block(const BlockPlan& p);
  OverSampledDFTAnalysisBank* bank(unsigned c) { return banks_[c]; }
  // 16-bit streaming (modulated/modulated.h): at the start of a stream, when every bank's source holds 16-bit PCM, the node
  // switches its banks to it -- the block's samples then go up as int16 [N][pitch] and are widened on the device, inside the
  // fused kernel where it has an int16 entry (btk_fb_analysis_bf_i16), by btk_pcm_i16_to_f32 for every other consumer.
  // BTK_NODE_I16=0 in the environment keeps the float path (the two give the same bits: tests/test_gpu_node_i16.py).
  bool i16_stream_possible();
  void begin_i16_stream();
  bool i16_stream() const { return pcm_i16_; }
 protected:
  bool load_chunk_();
  void pull_bank_(size_t c);
  void pull_banks_(size_t c0, size_t c1);
  void plan_from_pulled_(BlockPlan& p, size_t nbanks) const;
  void ensure_chunk_() { if (!chunk_loaded_) load_chunk_(); }
  void* snapshots_();             // the block's snapshots, launched on the node stream (not waited for)
  void free_device_();
  long chunk_base_, block_frames_;
  bool chunk_loaded_, channels_ended_;
  long quantum_;
  bool halfBandShift_;
  void* dXfull_;
  typedef std::list<VectorComplexFeatureStreamPtr> ChannelList_;
  ChannelList_ channelList_;
  SnapShotArrayPtr snapshot_array_;
  unsigned fftLen_, fftLen2_;
  void* dX_;                      // device snapshots of the current block (NULL until somebody asked for them)
  long T_;
  std::vector<float> Xhost_;      // lazily fetched host copy for snapshot_array()
  std::vector<OverSampledDFTAnalysisBank*> banks_;   // the channels as analysis banks (banks_only_)
  bool banks_only_;
  DeviceBuffer dPcmBuf_, dPcm16Buf_, dXBuf_, dXfullBuf_;
  const float* pcm_f32_();        // the resident windows as float32 [N][pcm_pitch_] (widened once per block in a 16-bit stream)
  bool pcm_i16_, pcm_f32_valid_;  // the current stream's samples are resident as int16 (dPcm16Buf_); dPcmBuf_ holds their float copy
  long pcm_pitch_;                // samples between the rows of the resident windows
  // 16-bit streams: the block after the current one, planned and on its way up (prefetch_next_)
  void prefetch_next_(const BlockPlan& cur);
  DeviceBuffer dPcm16Next_;
  bool pre_valid_;
  BlockPlan pre_plan_;
  long pre_pitch_;
  PinnedBuffer hStage_;           // channels that are pulled frame by frame: the transposed block on its way up
  PinnedBuffer hRows_;            // where the channels' sample rows lie on the host: the table of the upload (btk_gather_rows)
  long pcm_L_, pcm_t0_;           // the resident PCM windows [N][pcm_L_]; stream frame chunk_base_ is frame pcm_t0_ of the window
  bool pcm_valid_, snap_valid_, snapshots_wanted_;
};

class SubbandDS : public SubbandBeamformer, public BlockSource {
 public:
  SubbandDS(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandDS");
  ~SubbandDS();
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual void reset();
  virtual void clear_channel();
  virtual const gsl_vector_complex* get_weights(unsigned fbinX);
  virtual BeamformerWeights* beamformer_weight_object(unsigned srcX = 0) const { return bfweight_; }
  virtual void calc_array_manifold_vectors(float samplerate, const gsl_vector* delays);
  // LCMV quiescent weights: look direction + one / NC-1 nulls (reference beamformer.h:142-156, beamformer.cc:1159-1206)
  virtual void calc_array_manifold_vectors_2(float samplerate, const gsl_vector* delaysT, const gsl_vector* delaysJ);
  virtual void calc_array_manifold_vectors_n(float samplerate, const gsl_vector* delaysT, const gsl_matrix* delaysJ, unsigned NC = 2);
  const gsl_vector_complex* getWeights(unsigned fbinX) { return get_weights(fbinX); }            // ENABLE_LEGACY_BTK_API aliases
  BeamformerWeights* getBeamformerWeightObject(unsigned srcX = 0) const { return beamformer_weight_object(srcX); }
  void calcArrayManifoldVectors(float samplerate, const gsl_vector* delays) { calc_array_manifold_vectors(samplerate, delays); }
  void calcArrayManifoldVectors2(float sampleRate, const gsl_vector* delaysT, const gsl_vector* delaysJ) { calc_array_manifold_vectors_2(sampleRate, delaysT, delaysJ); }
  void calcArrayManifoldVectorsN(float sampleRate, const gsl_vector* delaysT, const gsl_matrix* delaysJ, unsigned NC = 2) { calc_array_manifold_vectors_n(sampleRate, delaysT, delaysJ, NC); }
  // engine hooks for downstream GPU nodes (post-filter, synthesis)
  virtual void effective_weights(std::vector<float>& w);   // complex64 [K][N]
  // halfBandShift == true (reference beamformer.cc:1113-1128, 1276-1285): the weight vector of every one of the M bins, complex64 [M][N]
  virtual void effective_weights_all_bins(std::vector<float>& w);
  void alignment_vector(bool use_wq, std::vector<float>& d);
  unsigned long weights_version() const { return weights_version_; }
  // BlockSource (modulated/modulated.h): the whole beamformed utterance for a batching consumer, weight changes mid-stream
  virtual unsigned long block_version() { return weights_version_; }
  virtual const std::vector<float>& block(long& T);
  virtual const void* device_block(long& T, long& T_stride);
  virtual long block_base() { return chunk_base(); }
  virtual bool next_block() { return advance_chunk_(); }
  virtual void advance_to(long frame_idx);
  // true while the node's blocks come from the fused analysis -> apply kernel (channels = analysis banks of a geometry that has
  // one, no half-band shift, nobody has asked for the snapshots)
  bool fused_path();
 protected:
  void alloc_bfweight_(int NC);
  void compute_output_(long from_frame);
  void ensure_output_();          // the current block beamformed with the current weights (frames already handed over keep theirs)
  const float* host_output_();    // ... and mirrored on the host
  virtual bool advance_chunk_();  // the next block of snapshots; what was computed for this one is dropped
  virtual const char* need_weights_msg_() const { return "call calc_array_manifold_vectorsX() once\n"; }
  BeamformerWeights* bfweight_;
  unsigned long weights_version_, output_version_;
  long handed_;                   // block protocol: frames 0 .. handed_ were handed to a batching consumer (advance_to)
  std::vector<float> Yhost_;      // [K][T] complex64: host mirror of dYBuf_, fetched when a frame or the host block is asked for
  bool out_valid_, Yhost_valid_;
  DeviceBuffer dWBuf_, dYBuf_, dScratchBuf_;
  unsigned long w_dev_version_;   // weights_version_ dWBuf_ holds
  bool w_dev_valid_;
  gsl_vector_complex* wq_view_;
};
typedef Inherit<SubbandDS, VectorComplexFeatureStreamPtr> SubbandDSPtr;

class SubbandGSC : public SubbandDS {
 public:
  SubbandGSC(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandGSC")
      : SubbandDS(fftLen, halfBandShift, nm), normalize_weight_(false) {}
  void normalize_weight(bool flag) { normalize_weight_ = flag; weights_version_++; }
  void set_quiescent_weights_f(unsigned fbinX, const gsl_vector_complex* srcWq);
  void set_active_weights_f(unsigned fbinX, const gsl_vector* packedWeight);
  void zero_active_weights();
  void calc_gsc_weights(float samplerate, const gsl_vector* delaysT);
  void calc_gsc_weights_2(float samplerate, const gsl_vector* delaysT, const gsl_vector* delaysJ);
  void calc_gsc_weights_n(float samplerate, const gsl_vector* delaysT, const gsl_matrix* delaysJ, unsigned NC = 2);
  bool write_fir_coeff(const String& fn, unsigned winType = 1);
  gsl_matrix_complex* blocking_matrix(unsigned srcX, unsigned fbinX);
  void normalizeWeight(bool flag) { normalize_weight(flag); }                                  // ENABLE_LEGACY_BTK_API aliases
  void setQuiescentWeights_f(unsigned fbinX, const gsl_vector_complex* srcWq) { set_quiescent_weights_f(fbinX, srcWq); }
  void setActiveWeights_f(unsigned fbinX, const gsl_vector* packedWeight) { set_active_weights_f(fbinX, packedWeight); }
  void zeroActiveWeights() { zero_active_weights(); }
  void calcGSCWeights(float samplerate, const gsl_vector* delaysT) { calc_gsc_weights(samplerate, delaysT); }
  void calcGSCWeights2(float sampleRate, const gsl_vector* delaysT, const gsl_vector* delaysJ) { calc_gsc_weights_2(sampleRate, delaysT, delaysJ); }
  void calcGSCWeightsN(float sampleRate, const gsl_vector* delaysT, const gsl_matrix* delaysJ, unsigned NC = 2) { calc_gsc_weights_n(sampleRate, delaysT, delaysJ, NC); }
  bool writeFIRCoeff(const String& fn, unsigned winType = 1) { return write_fir_coeff(fn, winType); }
  gsl_matrix_complex* getBlockingMatrix(unsigned srcX, unsigned fbinX) { return blocking_matrix(srcX, fbinX); }
  virtual void effective_weights(std::vector<float>& w);
  virtual void effective_weights_all_bins(std::vector<float>& w);
 protected:
  virtual const char* need_weights_msg_() const { return "call calc_gsc_weights_X() once\n"; }
  bool normalize_weight_;
};
typedef Inherit<SubbandGSC, SubbandDSPtr> SubbandGSCPtr;

// RLS sidelobe canceller (reference beamformer.h:207-263, beamformer.cc:1447-1699): the recursion of the whole
// utterance runs block by block in btk_rls_process launches (mode 0), the state carried on the device; Pz_ and the active
// weights survive reset() as in the reference.
typedef enum { CONSTANT_NORM = 0x01, THRESHOLD_LIMITATION = 0x02, NO_QUADRATIC_CONSTRAINT = 0x00 } QuadraticConstraintType;

class SubbandGSCRLS : public SubbandGSC {
 public:
  SubbandGSCRLS(unsigned fftLen = 512, bool halfBandShift = false, float mu = 0.9, float sigma2 = 0.0,
                const String& nm = "SubbandGSCRLS");
  ~SubbandGSCRLS();
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual const std::vector<float>& block(long& T);          // the adaptive recursion over the current block (not the static apply)
  virtual const void* device_block(long& T, long& T_stride);
  virtual void reset() { SubbandGSC::reset(); block_ran_ = false; }
  // (the recursion needs the snapshots: an RLS node never takes the fused path)
  void init_precision_matrix(float sigma2 = 0.01);
  void set_precision_matrix(unsigned fbinX, gsl_matrix_complex* Pz);
  void update_active_weight_vecotrs(bool flag) { is_wa_updated_ = flag; }   // sic (reference spelling)
  void set_quadratic_constraint(float alpha, int qctype = 1) { alpha_ = alpha; qctype_ = (QuadraticConstraintType)qctype; }
  void initPrecisionMatrix(float sigma2 = 0.01) { init_precision_matrix(sigma2); }
  void setPrecisionMatrix(unsigned fbinX, gsl_matrix_complex* Pz) { set_precision_matrix(fbinX, Pz); }
  void updateActiveWeightVecotrs(bool flag) { update_active_weight_vecotrs(flag); }
  void setQuadraticConstraint(float alpha, int qctype = 1) { set_quadratic_constraint(alpha, qctype); }
 private:
  void alloc_state_();
  void upload_weights_();
  void change_basis_(void* dP, void* dW);
  void run_block_();
  void refresh_block_();
  const float* rls_host_output_();
  virtual bool advance_chunk_();
  float mu_, diagonal_weight_, alpha_;
  QuadraticConstraintType qctype_;
  unsigned long rls_version_;       // weights_version_ the cached block was computed with
  bool is_wa_updated_, have_P_;
  void *dP_, *dW_, *dV_, *dSS_;     // device: P complex128 [K][N][N], w complex128 [K][N], wq complex128 [K][N], stream state
  void* dCx_;                       // NC > 1: the further blocked directions, complex128 [K][NC-1][N] (btk_rls_*_nc)
  void *dP0_, *dW0_, *dSS0_;        // the state at the start of the current block (a weight change before its first frame is served reruns it)
  unsigned long uploaded_version_;  // weights_version_ dV_ / dCx_ were built from
  bool block_ran_;
  DeviceBuffer dWs_;                // workspace of the recursion
  std::vector<std::complex<double> > wq_uploaded_;   // the quiescent weights (bins 0..M/2) the device state was built with
};
typedef Inherit<SubbandGSCRLS, SubbandGSCPtr> SubbandGSCRLSPtr;

class SubbandMVDR : public SubbandDS {
 public:
  SubbandMVDR(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandMVDR");
  ~SubbandMVDR();
  virtual void clear_channel();
  bool calc_mvdr_weights(float samplerate, float dThreshold = 1.0E-8, bool calcInverseMatrix = true);
  const gsl_vector_complex* mvdr_weights(unsigned fbinX);
  bool set_noise_spatial_spectral_matrix(unsigned fbinX, gsl_matrix_complex* Rnn);
  bool set_diffuse_noise_model(const gsl_matrix* micPositions, float samplerate, float sspeed = 343740.0);
  void set_all_diagonal_loading(float diagonalWeight);
  void set_diagonal_looading(unsigned fbinX, float diagonalWeight);          // sic (reference spelling)
  // R_xy /= 1 + mu for x != y instead of diagonal loading (reference beamformer.h:353-362, beamformer.cc:2589-2599)
  void divide_all_nondiagonal_elements(float mu);
  void divide_nondiagonal_elements(unsigned fbinX, float mu);
  const gsl_matrix_complex* noise_spatial_spectral_matrix(unsigned fbinX);   // host copy of R_k (refreshed at every call)
  bool calcMVDRWeights(float sampleRate, float dThreshold = 1.0E-8, bool calcInverseMatrix = true) { return calc_mvdr_weights(sampleRate, dThreshold, calcInverseMatrix); }   // ENABLE_LEGACY_BTK_API aliases
  const gsl_vector_complex* getMVDRWeights(unsigned fbinX) { return mvdr_weights(fbinX); }
  const gsl_matrix_complex* getNoiseSpatialSpectralMatrix(unsigned fbinX) { return noise_spatial_spectral_matrix(fbinX); }
  bool setNoiseSpatialSpectralMatrix(unsigned fbinX, gsl_matrix_complex* Rnn) { return set_noise_spatial_spectral_matrix(fbinX, Rnn); }
  bool setDiffuseNoiseModel(const gsl_matrix* micPositions, float sampleRate, float sspeed = 343740.0) { return set_diffuse_noise_model(micPositions, sampleRate, sspeed); }
  void setAllLevelsOfDiagonalLoading(float diagonalWeight) { set_all_diagonal_loading(diagonalWeight); }
  void setLevelOfDiagonalLoading(unsigned fbinX, float diagonalWeight) { set_diagonal_looading(fbinX, diagonalWeight); }
  void divideAllNonDiagonalElements(float mu) { divide_all_nondiagonal_elements(mu); }
  void divideNonDiagonalElements(unsigned fbinX, float mu) { divide_nondiagonal_elements(fbinX, mu); }
  virtual void effective_weights(std::vector<float>& w);
  int identity_fallbacks() const { return fallbacks_; }
  // Which bins take the identity in place of inv(R_k) (reference beamformer.cc:253-270, 2379-2384): "linpack" (default; the
  // environment variable BTK_MVDR_SVD_RULE overrides) = exactly where the reference's float32 csvdc reports INFO != 0 or
  // leaves a singular value under the threshold; "exact" = only where a singular value really is under the threshold.
  void set_svd_rule(const String& rule);
  const String& svd_rule() const { return svd_rule_; }
  int csvdc_not_converged() const { return csvdc_not_converged_; }    // bins of the last design with INFO != 0
 protected:
  void alloc_R_();
  void* dR_;                        // device complex64 [K][N][N]
  std::vector<float> wmvdr_;        // complex64 [K][N]
  bool have_mvdr_;
  int fallbacks_;
  String svd_rule_;
  int csvdc_not_converged_;
  gsl_vector_complex* wm_view_;
  gsl_matrix_complex* R_view_;
};
typedef Inherit<SubbandMVDR, SubbandDSPtr> SubbandMVDRPtr;

// SubbandMVDRGSC (reference beamformer.h:385-437, beamformer.cc:2604-2773): MVDR quiescent vector + GSC lower branch.
//   1. set_channel()  2. calc_array_manifold_vectors()  3. set_noise_spatial_spectral_matrix() / set_diffuse_noise_model()
//   4. calc_mvdr_weights()  5. calc_blocking_matrix1() or calc_blocking_matrix2()  6. set_active_weights_f()
// calc_blocking_matrix1/2 re-create the weight object like the reference's alloc_bfweight_ (active weights start over).
class SubbandMVDRGSC : public SubbandMVDR {
 public:
  SubbandMVDRGSC(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandMVDR")
      : SubbandMVDR(fftLen, halfBandShift, nm), normalize_weight_(false) {}
  void normalize_weight(bool flag) { normalize_weight_ = flag; weights_version_++; }
  void set_active_weights_f(unsigned fbinX, const gsl_vector* packedWeight);
  void zero_active_weights();
  bool calc_blocking_matrix1(float samplerate, const gsl_vector* delaysT);
  bool calc_blocking_matrix2();
  void upgrade_blocking_matrix();
  const gsl_vector_complex* blocking_matrix_output(int outChanX = 0);
  virtual void effective_weights(std::vector<float>& w);
  void setActiveWeights_f(unsigned fbinX, const gsl_vector* packedWeight) { set_active_weights_f(fbinX, packedWeight); }
  void zeroActiveWeights() { zero_active_weights(); }
  bool calcBlockingMatrix1(float sampleRate, const gsl_vector* delaysT) { return calc_blocking_matrix1(sampleRate, delaysT); }
  bool calcBlockingMatrix2() { return calc_blocking_matrix2(); }
  void upgradeBlockingMatrix() { upgrade_blocking_matrix(); }
  const gsl_vector_complex* blockingMatrixOutput(int outChanX = 0) { return blocking_matrix_output(outChanX); }
 protected:
  bool normalize_weight_;
};
typedef Inherit<SubbandMVDRGSC, SubbandMVDRPtr> SubbandMVDRGSCPtr;

// Many utterance graphs advanced as ONE launch (not part of the reference's interface).  The reference's unit of work is one
// graph per utterance (unit_test/test_online_beamforming.py:80-88 builds SampleFeature x N -> OverSampledDFTAnalysisBank x N ->
// beamformer -> OverSampledDFTSynthesisBank for every file); pulled one by one, G utterances are G independent S = 1 launches
// per block.  A pool takes the tails of G such graphs -- a fixed-weight beamformer (SubbandDS / GSC / MVDR / MVDRGSC) whose
// channels are analysis banks, and the synthesis bank behind it -- and advances them in lock step: per round every graph's
// banks pull one block of input, the sample windows go up into one [G][N][L] block, ONE fused analysis -> apply launch with
// S = G and per-stream weights writes the beamformed frames straight into the synthesis banks' window block, ONE synthesis
// launch with S = G turns them into PCM.  next() then hands out one output block per graph and call, like G synthesis nodes
// pulled side by side.  The results are those of the graphs pulled on their own (tests/test_gpu_graph_pool.py).
// Rules: all graphs share the filter-bank geometry (which must have a fused kernel: btk_fb_analysis_bf_fused), the channel
// count and the banks' block_frames(); the member nodes are driven by the pool only (do not call their next()); weights are
// read once per round, so a weight change takes effect with the next round -- the per-frame meaning of a change inside a block
// (BlockSource::advance_to) belongs to the single-graph path.
class SubbandGraphPool : public Countable {
 public:
  SubbandGraphPool();
  ~SubbandGraphPool();
  void add(SubbandDSPtr& beamformer, OverSampledDFTSynthesisBankPtr& synthesis);
  unsigned size() const { return (unsigned)graphs_.size(); }
  bool next();                                              // one more output block of every graph that has one; false: all have ended
  const gsl_vector_float* output(unsigned g) const;         // graph g's block of the last next(); NULL once that graph has ended
  bool is_end(unsigned g) const;
  // The blocks the next calls of next() would hand out, a ROUND at a time: next_round() makes the blocks every graph still has in
  // the resident round (a new round when none has any) available at once -- false: all graphs have ended -- and round_blocks(g, &p)
  // says how many graph g got (0: that graph has ended or had no block in this round) and where they lie (n x shiftlen floats,
  // valid until the next call of next() / next_round()).  output(g) is then the last of them.
  bool next_round();
  long round_blocks(unsigned g, const float** blocks) const;
  long rounds() const { return rounds_; }                   // batched rounds (= launches of each kind) so far
  void reset();
 private:
  struct Graph {
    SubbandDSPtr bf;
    OverSampledDFTSynthesisBankPtr syn;
    bool live;                 // its channels have more input
    long T, nblocks, served;   // this round: valid frames, output blocks, blocks handed out
    gsl_vector_float* out;
    bool has_out;
    long round_first, round_n;   // next_round(): the blocks of the current round handed out at once
  };
  bool load_round_();
  // one round's input on its way to the device: every live graph's block plan and its sample windows in one block [G][N][Lmax]
  struct Stage {
    std::vector<SubbandBeamformer::BlockPlan> plans;
    std::vector<long> T;          // frames of graph g in the round (0: none)
    long Lmax, Tmax, t0, f0;
    bool valid;
    Stage() : Lmax(0), Tmax(0), t0(-1), f0(-1), valid(false) {}
  };
  void stage_round_(Stage& st, DeviceBuffer& buf, void* stream);   // pulls the banks, plans, issues the uploads on `stream`
  Stage pre_;                     // 16-bit streams: the round after the current one, staged while the current one is served
  DeviceBuffer dPcmNext_;
  std::vector<Graph> graphs_;
  long rounds_, base_, prev_T_, prev_Lw_, prev_Lp_, prev_hist_, blk_base_, out_stride_;
  bool first_round_;
  bool i16_;                      // the current streams go up as 16-bit PCM (every graph's sources hold it; decided in the first round)
  DeviceBuffer dPcm_, dW_, dWinA_, dWinB_, dOut_, dScratch_;
  PinnedBuffer hW_, hOut_, hRows_;
};
typedef refcountable_ptr<SubbandGraphPool> SubbandGraphPoolPtr;
