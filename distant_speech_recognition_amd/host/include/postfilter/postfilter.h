// postfilter/postfilter.h -- ZelinskiPostFilter, McCowanPostFilter, LefkimmiatisPostFilter
// (reference postfilter/postfilter.h:74-108, 123-162, 174-203).
#pragma once
#include "beamformer/beamformer.h"

typedef enum { TYPE_ZELINSKI1_REAL = 0x01, TYPE_ZELINSKI1_ABS = 0x02, TYPE_APAB = 0x04, TYPE_ZELINSKI2 = 0x08,
               NO_USE_POST_FILTER = 0x00 } PostfilterType;

class ZelinskiPostFilter : public VectorComplexFeatureStream, public BlockSource {
 public:
  ZelinskiPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha = 0.6, int type = 2,
                     int minFrames = 0, const String& nm = "ZelinskPostFilter");
  ~ZelinskiPostFilter();
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual void reset();
  void set_beamformer(SubbandDSPtr& beamformer);
  // without a beamformer object (reference postfilter.h:83-84, postfilter.cc:384-421): the caller keeps a SnapShotArray current
  // (set_samples() + update() per frame) and gives the alignment vector of every bin; next() then filters samp_'s frame with
  // the statistics of the snapshot of that moment -- one small launch per frame, state on the device
  void set_snapshot_array(SnapShotArrayPtr& snapShotArray);
  void set_array_manifold_vector(unsigned fbinX, gsl_vector_complex* arrayManifoldVector, bool halfBandShift, unsigned NC = 1);
  void setBeamformer(SubbandDSPtr& beamformer) { set_beamformer(beamformer); }                       // ENABLE_LEGACY_BTK_API aliases
  void setSnapShotArray(SnapShotArrayPtr& snapShotArray) { set_snapshot_array(snapShotArray); }
  void setArrayManifoldVector(unsigned fbinX, gsl_vector_complex* v, bool halfBandShift, unsigned NC = 1) { set_array_manifold_vector(fbinX, v, halfBandShift, NC); }
  const gsl_vector_complex* getPostFilterWeights() { return postfilter_weights(); }
  const gsl_vector_complex* postfilter_weights();
  // bf_weights_ of the reference (postfilter.h:104): the beamformer's weight object, or the one set_array_manifold_vector() made
  BeamformerWeights* weights_object() const { return has_bf_ptr_ ? bf_ptr_->beamformer_weight_object() : own_weights_; }
  // BlockSource: see modulated/modulated.h
  virtual unsigned long block_version() { return has_bf_ptr_ ? bf_ptr_->weights_version() : 0; }
  virtual const std::vector<float>& block(long& T);
  virtual const void* device_block(long& T, long& T_stride);
  virtual long block_base();
  virtual bool next_block();
  virtual void advance_to(long frame_idx);
 protected:
  virtual void compute_(long from_frame);
  bool advance_chunk_();                    // the beamformer's next block of snapshots; the recursions continue from this one's end
  void csd_state_(long t, std::vector<float>& R);
  const float* host_output_();              // the filtered block mirrored on the host (fetched on demand)
  const gsl_vector_complex* next_manual_(int frame_no);
  // the reference keeps the N x N spectral densities of every bin in BeamformerWeights::CSDs(); this engine keeps only their
  // sums on the device and rebuilds the matrices on demand: one weighted covariance launch over the frames served so far
  void bind_csd_provider_();
  void fill_csds_(gsl_vector_complex** out);
  virtual bool align_with_wq_() const { return (type_ & TYPE_ZELINSKI2) != 0; }     // postfilter.cc:452-457
  virtual bool lefkimmiatis_or_mccowan_() const { return false; }
  unsigned fftLen_;
  VectorComplexFeatureStreamPtr samp_;
  PostfilterType type_;
  double alpha_;
  int min_frames_;
  SubbandDSPtr bf_ptr_;
  bool has_bf_ptr_;
  std::vector<float> Yhost_, wlast_;
  bool Yhost_valid_;
  DeviceBuffer dWb_, dDb_, dYb_, dCb_, dEb_;   // weights, alignment vector, filtered block [K][T_], per-frame statistics (grow-only)
  long T_;
  bool prepared_;
  unsigned long bf_version_;
  void *dPhi_, *dPsi_, *dWl_;
  gsl_vector_complex* wp1_;
  long hist_start_;                         // stream index of the frame at which the density history last restarted (weights recomputed)
  SnapShotArrayPtr snapshot_array_;         // manual mode: multi-channel input kept current by the caller
  BeamformerWeights* own_weights_;          // manual mode: created by set_array_manifold_vector
  std::vector<std::complex<double> > csd_manual_;   // manual mode: the recursively averaged x x^H behind CSDs(), [K][N][N] (upper triangle)
  long manual_frames_;
  long base_;                               // stream index of the current block's first frame
  bool carry_state_;                        // the next compute_() continues the recursions (a new block of the same stream)
  std::vector<float> csd_carry_;            // CSD state after the block before, complex64 [K][N][N] (for CSDs())
  long handed_;                             // block protocol mark (see SubbandDS::advance_to)
};
typedef Inherit<ZelinskiPostFilter, VectorComplexFeatureStreamPtr> ZelinskiPostFilterPtr;

// McCowan post-filter: Zelinski's estimator with the microphone-pair terms weighted by a noise coherence matrix.
class McCowanPostFilter : public ZelinskiPostFilter {
 public:
  McCowanPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha = 0.6, int type = 2,
                    int minFrames = 0, float threshold = 0.99, const String& nm = "McCowanPostFilter");
  ~McCowanPostFilter();
  const gsl_matrix_complex* noise_spatial_spectral_matrix(unsigned fbinX);
  bool set_noise_spatial_spectral_matrix(unsigned fbinX, gsl_matrix_complex* Rnn);
  bool set_diffuse_noise_model(const gsl_matrix* micPositions, double sampleRate, double sspeed = 343740.0);
  void set_all_diagonal_loading(float diagonalWeight);
  void set_diagonal_looading(unsigned fbinX, float diagonalWeight);          // sic (reference spelling)
  void divide_all_nondiagonal_elements(float mu);
  void divide_nondiagonal_elements(unsigned fbinX, float mu);
  // ENABLE_LEGACY_BTK_API aliases (reference postfilter/postfilter.h:140-148)
  const gsl_matrix_complex* getNoiseSpatialSpectralMatrix(unsigned fbinX) { return noise_spatial_spectral_matrix(fbinX); }
  bool setNoiseSpatialSpectralMatrix(unsigned fbinX, gsl_matrix_complex* Rnn) { return set_noise_spatial_spectral_matrix(fbinX, Rnn); }
  bool setDiffuseNoiseModel(const gsl_matrix* mp, double fs, double c = 343740.0) { return set_diffuse_noise_model(mp, fs, c); }
  void setAllLevelsOfDiagonalLoading(float w) { set_all_diagonal_loading(w); }
  void setLevelOfDiagonalLoading(unsigned fbinX, float w) { set_diagonal_looading(fbinX, w); }
  void divideAllNonDiagonalElements(float mu) { divide_all_nondiagonal_elements(mu); }
  void divideNonDiagonalElements(unsigned fbinX, float mu) { divide_nondiagonal_elements(fbinX, mu); }
  // Lefkimmiatis: which bins take the identity in place of pinv(R_k) when Lambda is formed -- "linpack" (default; the
  // environment variable BTK_MVDR_SVD_RULE overrides) or "exact", as SubbandMVDR::set_svd_rule (beamformer/beamformer.h)
  void set_svd_rule(const String& rule);
  const String& svd_rule() const { return svd_rule_; }
 protected:
  virtual void compute_(long from_frame);
  virtual bool lefkimmiatis_() const { return false; }
  virtual bool lefkimmiatis_or_mccowan_() const { return true; }
  virtual bool align_with_wq_() const { return !lefkimmiatis_() && (type_ & TYPE_ZELINSKI2) != 0; }   // postfilter.cc:858-863 vs :1098
  virtual const char* no_R_msg_() const { return "McCowanPostFilter:  construct/set a noise coherence matrix\n"; }
  void fetch_R_();
  void push_R_();
  float threshold_of_Rij_;
  unsigned nChanR_;
  void* dR_;                        // device complex64 [K][N][N]
  std::vector<float> Rhost_;        // host mirror, complex64 [K][N][N]
  gsl_matrix_complex* Rview_;
  bool invR_computed_;
  double minSV_;
  unsigned fbinX1_;
  void *dU_, *dV_;
  DeviceBuffer dVsb_, dCsb_, dCvb_, dLamb_;
  String svd_rule_, lam_rule_;      // the rule in force / the one the kept Lambda was designed with
  bool lam_valid_;                  // dLamb_ matches R, the look direction (lam_version_) and the rule
  unsigned long lam_version_;
};
typedef Inherit<McCowanPostFilter, ZelinskiPostFilterPtr> McCowanPostFilterPtr;

// Lefkimmiatis post-filter: Wiener gain under the diffuse-noise-field assumption.
class LefkimmiatisPostFilter : public McCowanPostFilter {
 public:
  LefkimmiatisPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double minSV = 1.0E-8, unsigned fbinX1 = 0,
                         double alpha = 0.6, int type = 2, int minFrames = 0, float threshold = 0.99,
                         const String& nm = "LefkimmiatisPostFilter");
  void calc_inverse_noise_spatial_spectral_matrix();
  void calcInverseNoiseSpatialSpectralMatrix() { calc_inverse_noise_spatial_spectral_matrix(); }
 protected:
  virtual bool lefkimmiatis_() const { return true; }
  virtual const char* no_R_msg_() const { return "LefkimmiatisPostFilter:  construct/set a noise coherence matrix\n"; }
};
typedef Inherit<LefkimmiatisPostFilter, McCowanPostFilterPtr> LefkimmiatisPostFilterPtr;
