/*
 * btkhip.h -- thin C-ABI of the MI355X (gfx950) subband-beamforming engine.
 *
 * This is the drop-in boundary: plain C, opaque handles, explicit sizes, int status codes.
 * Bulk data pointers marked [dev] are DEVICE pointers (HBM) owned by the caller; pointers
 * marked [host] are host memory read during the call only.  `stream` is a hipStream_t passed
 * as void* (NULL = default stream).  Nothing here needs PyTorch.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * btk20_src/ of kkumatani/distant_speech_recognition).  The reference has no FFI of its own for
 * this path (it is a single-process C++/SWIG library); INTEGRATION.md shows the binding a
 * maintainer adds on the reference side.
 *
 * Data layout in HBM (K = M/2+1 computed bins):
 *   pcm   float32   [S][N][pcm_stride]     stream, channel, sample (un-normalised int16 scale)
 *   X     complex64 [S][K][N][T]           subband snapshots, FRAMES CONTIGUOUS
 *   W     complex64 [S or 1][K][N]         beamformer weights,  y = w^H x
 *   Y     complex64 [S][K][T]              beamformed subband frames
 *   out   float32   [S][nblocks*D]         synthesised samples
 *
 * All functions return BTK_OK (0) or a negative error code; btk_last_error() returns the
 * message of the calling thread's last failure (the C++ node layer turns codes into the
 * reference's j_error subclasses, common/jexception.h:26-161).
 */
#ifndef BTKHIP_H
#define BTKHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define BTK_OK              0
#define BTK_ERR_DIMENSION  -1   /* jdimension_error   */
#define BTK_ERR_CONSISTENCY -2  /* jconsistency_error */
#define BTK_ERR_ALLOCATION -3   /* jallocation_error  */
#define BTK_ERR_PARAMETER  -4   /* jparameter_error   */
#define BTK_ERR_HIP        -5   /* j_error (device runtime failure) */
#define BTK_ERR_NUMERIC    -6   /* jnumeric_error     */

typedef struct btk_fb btk_fb_t;          /* filter-bank plan (analysis or synthesis) */

const char* btk_last_error(void);
int  btk_version(void);
/* Number of HIP devices visible / select one for this thread (one process per GPU). */
int  btk_device_count(void);
int  btk_set_device(int device);
int  btk_synchronize(void* stream);

/* ---- Oversampled modulated-DFT filter bank --------------------------------------------
 * Replaces OverSampledDFTAnalysisBank / OverSampledDFTSynthesisBank
 * (modulated/modulated.h:268-340, modulated/modulated.cc:232-268 ctor delay logic).
 * prototype: [host] m*M float64 coefficients, copied at creation (modulated.cc:243-244).
 * M must be a power of two in [64, 2048]; r in [0, log2(M)]; delay_comp_type in {0,1,2}.   */
int  btk_fb_create(btk_fb_t** fb, int M, int m, int r, int delay_comp_type, int synthesis,
                   const double* prototype);
void btk_fb_destroy(btk_fb_t* fb);
int  btk_fb_processing_delay(const btk_fb_t* fb);   /* processing_delay_ */
int  btk_fb_lookahead(const btk_fb_t* fb);          /* laN_ */
/* frames an analysis bank emits for an nsamples-long channel before jiterator_error:
 * ceil(nsamples/D) - laN + processing_delay   (modulated.cc:419-469).                       */
long btk_fb_analysis_num_frames(const btk_fb_t* fb, long nsamples);
/* blocks a synthesis bank emits when its source ends after nframes: nframes - pd (>= 0)
 * (modulated.cc:569-612).                                                                   */
long btk_fb_synthesis_num_blocks(const btk_fb_t* fb, long nframes);

/* OverSampledDFTAnalysisBank::next for frames [t0, t0+tcount) of every (stream, channel):
 * modulated.cc:375-409.  pcm [dev] [S*N][pcm_stride] with nsamples valid samples per channel
 * (zero outside, like the zero-initialised ring and the end-of-stream padding).
 * X [dev] [S][K][N][T_stride]; frame t is stored at column t - t0.                         */
int  btk_fb_analysis(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride,
                     int S, int N, void* X, long T_stride, long t0, long tcount, void* stream);
/* A bin SHARD of the same transform (frequency-bin sharding): only bins [k0, k1) are written,
 * X [dev] complex64 [S][k1-k0][N][T_stride].  The FFT of a channel yields all bins, so a rank of a bin-sharded run
 * transforms every channel but stores -- the dominant cost of the analysis bank -- only 1/world of the snapshots. */
int  btk_fb_analysis_bins(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                          long T_stride, long t0, long tcount, int k0, int k1, void* stream);
/* Debug/parity entry: the M real polyphase sums of modulated.cc:384-391 (before the FFT),
 * P [dev] float32 [S*N][tcount][M].  Used to pin the integer polyphase indexing bit-exactly. */
int  btk_fb_analysis_polyphase(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride,
                               int S, int N, float* P, long t0, long tcount, void* stream);
/* OverSampledDFTSynthesisBank::next for blocks [b0, b0+bcount): modulated.cc:553-612.
 * Y [dev] [S][K][T_stride] holding nframes valid frames (frame index < 0 reads as zero == the
 * zeroed ring, modulated.cc:615-621); out [dev] [S][out_stride], block b at offset (b-b0)*D. */
int  btk_fb_synthesis(const btk_fb_t* fb, const void* Y, long nframes, long T_stride, int S,
                      float* out, long out_stride, long b0, long bcount, void* stream);
/* 1: this plan's geometry has a wider kernel form that btk_fb_synthesis takes for ALIGNED launches -- b0 + processing_delay even,
 * T_stride even, out_stride a multiple of 4, Y and out 16-byte aligned -- and whose results differ from the other form's by
 * <= 2 ulp (another FFT factorisation, the same summation order per sample).  A caller that cuts one stream into several launches
 * and needs the same bits for every partition keeps all of them aligned (b0 may be one less than the first block wanted, -1
 * included: blocks before the stream are zeros) or none; 0: one form, every partition gives the same bits anyway.        */
int  btk_fb_synthesis_aligned_form(const btk_fb_t* fb);

/* ---- Sample formats on either side of the path ------------------------------------------------
 * SampleFeature hands the analysis bank UN-NORMALISED floats of 16-bit PCM (feature/feature.cc:265-269) and the reference's
 * scripts write the synthesis output back as int16 (numpy.array(buf, numpy.int16): toward zero,
 * unit_test/test_online_beamforming.py:209).  Utterances cross PCIe as int16 and are widened / narrowed on the device.
 * in, out [dev], n samples.                                                                                            */
int  btk_pcm_i16_to_f32(const short* in, float* out, long n, void* stream);
int  btk_pcm_f32_to_i16(const float* in, short* out, long n, void* stream);
/* A multi-channel recording as stored (frame by frame): in [dev] int16 [L][N] -> out [dev] float32 [N][out_stride], all channels in
 * one pass (SampleFeature::read copies channel chX out of the interleaved frames, once per channel node: feature/feature.cc:333-334). */
int  btk_pcm_i16_deinterleave(const short* in, float* out, long L, int N, long out_stride, void* stream);

/* Rows of samples that lie in SEPARATE host allocations -> one device block, by ONE kernel that reads the host memory itself
 * (round 6; the node layer's 16-bit streams: every channel's SampleFeature owns its utterance, feature/feature.cc:238-389, and a
 * block of a 64-channel graph -- or of 32 graphs -- is 64 ... 2 048 rows; a hipMemcpyAsync per row reaches 27 GB/s at 0.5 MB per row,
 * the kernel the link's 57 GB/s at any row length).  table [PINNED host memory (hipHostMalloc: device-accessible) or dev]: nrows
 * entries; src [pinned host or dev], 16-byte aligned, `bytes` valid bytes (a multiple of 2) -- it must stay untouched until the
 * stream has passed the call, like the source of an asynchronous copy.  Row r goes to dst + r * dst_pitch_bytes [dev]; the bytes
 * from `bytes` up to dst_pitch_bytes are zeroed (bytes <= dst_pitch_bytes).  dst and dst_pitch_bytes multiples of 16.            */
typedef struct { const void* src; long bytes; } btk_row_t;
int  btk_gather_rows(const btk_row_t* table, void* dst, int nrows, long dst_pitch_bytes, void* stream);

/* ---- Fixed-weight beamformer apply ------------------------------------------------------
 * SubbandDS::next (beamformer/beamformer.cc:1095-1157), SubbandGSC::next + calc_gsc_output
 * (:1208-1316), SubbandMVDR::next (:2537-2587), SubbandMVDRGSC::next (:2720-2773):
 *   y_k[t] = w_k^H x_k[t], k = 0..M/2 (mirror bins are conjugates, formed at synthesis).
 * W [dev] complex64 [Sw][K][N] with Sw = S (per-stream weights) or 1 (shared).              */
int  btk_bf_apply(const void* W, int per_stream_weights, const void* X, void* Y,
                  int S, int K, int N, long T_stride, long T, void* stream);

/* Fused OverSampledDFTAnalysisBank x N -> fixed-weight beamformer (the chain SubbandGSC::next pulls per frame,
 * beamformer.cc:1267-1311): Y[s][k][t] = sum_n conj(W[k][n]) X_n[k][t] without materialising the snapshots in HBM
 * (fused kernels: M = 512 and M = 256 with m = 4, r <= 2, and M = 1024 / 2048 with m = 4, r = 1 -- there one stream of a large
 * array is split over channel groups whose partial sums are added in a fixed order; other geometries run btk_fb_analysis +
 * btk_bf_apply through `scratch`, which then needs T_stride == tcount).
 * scratch [dev] of btk_fb_analysis_bf_scratch_bytes(...) bytes, 16-byte aligned (the fused kernel stages its weight
 * pairs [Sw][N][320] float4 there and fetches them by LDS-DMA).  Y rows may be spaced T_stride >= tcount frames apart
 * (fused geometries); rows a multiple of 4 KiB apart are worth padding.  Use the staged calls when a post-filter, the
 * adaptive canceller or covariance accumulation needs the snapshots.                                          */
long btk_fb_analysis_bf_scratch_bytes(const btk_fb_t* fb, int S, int N, int per_stream_weights, long tcount);
/* 1: btk_fb_analysis_bf runs a fused kernel for this plan's geometry (any T_stride >= tcount); 0: the staged pair through
 * `scratch` (a caller that keeps the snapshots anyway then calls btk_fb_analysis + btk_bf_apply itself); < 0: not an analysis plan. */
int  btk_fb_analysis_bf_fused(const btk_fb_t* fb);
int  btk_fb_analysis_bf(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N,
                        const void* W, int per_stream_weights, void* Y, long T_stride, long t0, long tcount,
                        void* scratch, long scratch_bytes, void* stream);

/* The same operator on the samples AS THEY ARE STORED: 16-bit PCM (SampleFeature::read turns a WAV's int16 samples into
 * un-normalised floats, feature/feature.cc:265-269; over PCIe and in HBM they can stay int16).  pcm [dev] int16 [S*N][pcm_stride],
 * widened in registers inside the fused kernel: 2 D N + 8 K bytes per beamformed frame instead of 4 D N + 8 K, and -- the
 * conversion is exact -- the SAME BITS as btk_fb_analysis_bf on the float copies of the same samples.  Geometries:
 * btk_fb_analysis_bf_i16_fused() == 1 (M = 256 / 512 with m = 4, r <= 2, and M = 1024 / 2048 with m = 4, r = 1); for the others widen
 * with btk_pcm_i16_to_f32 and call btk_fb_analysis_bf (BTK_ERR_PARAMETER here).  scratch: btk_fb_analysis_bf_scratch_bytes.   */
int  btk_fb_analysis_bf_i16_fused(const btk_fb_t* fb);
/* The STAGED bank on 16-bit PCM (round 6): btk_fb_analysis with pcm [dev] int16 [S*N][pcm_stride], widened on the way into the
 * kernel's LDS span -- the same bits as btk_fb_analysis on the float copies, 2 D + 8 K instead of 4 D + 8 K bytes per frame and
 * channel, and no btk_pcm_i16_to_f32 pass in front of it (4 D + 2 D bytes per frame and channel more).  What the snapshot
 * consumers of a 16-bit stream run on: post-filters, adaptive cancellers, covariance, WPE (OverSampledDFTAnalysisBank::next over
 * SampleFeature::read, modulated/modulated.cc:375-409 over feature/feature.cc:265-269).  Geometries:
 * btk_fb_analysis_i16_direct() == 1 (M = 256 / 512 / 1024 / 2048, m = 4, r <= 2); BTK_ERR_PARAMETER for the others (widen, then btk_fb_analysis). */
int  btk_fb_analysis_i16_direct(const btk_fb_t* fb);
int  btk_fb_analysis_i16(const btk_fb_t* fb, const short* pcm, long nsamples, long pcm_stride,
                         int S, int N, void* X, long T_stride, long t0, long tcount, void* stream);
int  btk_fb_analysis_bf_i16(const btk_fb_t* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N,
                            const void* W, int per_stream_weights, void* Y, long T_stride, long t0, long tcount,
                            void* scratch, long scratch_bytes, void* stream);

/* ---- Adaptive GSC canceller: leaky power-normalised NLMS ----------------------------------
 * Replaces SubbandGSCLMSBeamformer.__iter__ / reset_stats (lib/pybeamformer.py:659-762), Nc = 1.
 * params [host] 8 floats: beta, gamma(init), regularization_param, energy_floor, sil_thresh,
 *   max_wa_l2norm, min_frames, slowdown_after (pybeamformer.py:597-607).
 * vs [dev] complex64 [K][N]: array manifold / N of bins 0..M/2 (calc_array_manifold_f :284-306;
 *   the upper branch is Yc = vs^H x, the blocking matrix is implied by vs -- see DESIGN.md).
 * State (all [dev], updated in place so consecutive blocks continue the recursion):
 *   u_state complex64 [S][K][N]  = wa^H B^T  (zeros == reset_stats; convert with btk_nlms_*_wa)
 *   sigma2  float32   [S][K]     = _subband_energy          (init_diagonal_load at reset)
 *   stream_state float64 [S][4]  = {_energy, _gamma, _isamp, _ttl_updates}
 * workspace [dev] btk_nlms_workspace_bytes(S,T) bytes.  Y [dev] complex64 [S][K][T_stride].    */
long btk_nlms_workspace_bytes(int S, long T);
int  btk_nlms_process(const float* params, const void* vs, const void* X, void* Y,
                      int S, int M, int N, long T_stride, long T,
                      void* u_state, float* sigma2, double* stream_state, void* workspace, void* stream);
/* Nc > 1 constraints (SubbandGSCLMSBeamformer(..., Nc), lib/pybeamformer.py:588-607, 742): the blocking matrix keeps the
 * first N - Nc Gram-Schmidt columns (calc_blocking_matrix, :309-341), so the canceller's projector loses Nc - 1 more
 * directions: cextra [dev] complex64 [K][NC-1][N], per bin from btk_nlms_constraint_vectors (host, complex128
 * [NC-1][N]; B [N][N-NC] from btk_weights_blocking_matrix).  NC = 1 is btk_nlms_process.  1 <= NC <= 8.              */
int  btk_nlms_process_nc(const float* params, const void* vs, const void* cextra, int NC, const void* X, void* Y,
                         int S, int M, int N, long T_stride, long T,
                         void* u_state, float* sigma2, double* stream_state, void* workspace, void* stream);
int  btk_nlms_constraint_vectors(const double* vs, const double* B, int N, int NC, double* cx);
int  btk_nlms_u_to_wa_nc(const double* u, const double* B, int N, int NC, double* waH);
/* Host-side change of basis between the reference's active weights wa^H (complex128 [N-1], as
 * kept in SubbandGSCLMSBeamformer._waH) and the engine state u = wa^H B^T (complex128 [N]);
 * B [host] complex128 [N][N-1] from btk_weights_blocking_matrix.                               */
int  btk_nlms_wa_to_u(const double* waH, const double* B, int N, double* u);
int  btk_nlms_u_to_wa(const double* u, const double* B, int N, double* waH);

/* ---- Adaptive GSC canceller: recursive least squares (float64 recursion) ---------------------
 * mode 0 replaces SubbandGSCRLS::next + update_active_weight_vector2_ (beamformer/beamformer.cc:1514-1645):
 *   params [host] 6 doubles: mu, diagonal_weight (sigma2 of the ctor, :1447-1467), qctype (0 none,
 *   1 CONSTANT_NORM, 2 THRESHOLD_LIMITATION, beamformer.h:215-219), alpha, normalize_weight, update flag
 *   (update_active_weight_vecotrs, beamformer.h:234);  v = wq (calcMainlobe), bins 1..M/2 adapt.
 * mode 1 replaces SubbandGSCRLSBeamformer.__iter__ / reset_stats (lib/pybeamformer.py:817-925), Nc = 1:
 *   params [host] 10 doubles: beta, gamma, mu, init_diagonal_load, regularization_param, sil_thresh,
 *   constraint_option, alpha2, max_wa_l2norm, min_frames (:776-787);  v = vs (calc_array_manifold_f :284-306).
 * v [dev] complex128 [Sv][K][N] (Sv = S if per_stream_v else 1); the upper branch is v^H x and the blocking
 * matrix is implied by v.  State [dev], updated in place so consecutive blocks continue:
 *   P_state complex128 [S][K][N][N] = B Pz B^H (mode 0) / conj(B) Pz B^T (mode 1)
 *   w_state complex128 [S][K][N]    = wl = B wa (mode 0) / u = wa^H B^T (mode 1)
 *   stream_state float64 [S][4]     = {_energy, -, _isamp, _ttl_updates} (mode 1; untouched in mode 0)
 * btk_rls_init writes P = p0 B B^H resp. p0 conj(B) B^T (closed form from v), w = 0  == init_precision_matrix (p0 = 1/sigma2,
 * beamformer.cc:1482-1494) resp. reset_stats (p0 = 1/init_diagonal_load, pybeamformer.py:921-925).
 * workspace [dev] btk_rls_workspace_bytes(S,T) bytes.  Y [dev] complex64 [S][K][T_stride].
 * N <= 64 with one constraint: P in registers (row and column copies, the reference's two products separately); otherwise, up to
 * N = 128: P as a packed Hermitian matrix in LDS; up to N = 256: P in place in the exported [N][N] state (rls_kernels.hip).
 * btk_rls_init_nc / btk_rls_process_nc: NC >= 1 constraints (SubbandGSCRLSBeamformer(..., Nc), pybeamformer.py:784-797;
 *   SubbandGSCRLS after calc_gsc_weights_2 / _n): cx [dev] complex128 [Sv][K][NC-1][N], the orthonormal directions the blocking
 *   matrix removes besides the one implied by v -- mode 1: btk_nlms_constraint_vectors(vs, B) (conj(B) B^T = I - vs vs^H/|vs|^2 -
 *   sum_j c_j c_j^H), mode 0: their complex conjugates taken with vs = wq (B B^H = I - conj(wq) wq^T/|wq|^2 - sum_j c_j c_j^H).
 *   NC = 1: cx may be NULL (== btk_rls_init / btk_rls_process).                                                               */
long btk_rls_workspace_bytes(int S, long T);
int  btk_rls_init(int mode, const void* v, int per_stream_v, double p0, int S, int K, int N,
                  void* P_state, void* w_state, void* stream);
int  btk_rls_process(int mode, const double* params, const void* v, int per_stream_v,
                     const void* X, void* Y, int S, int M, int N, long T_stride, long T,
                     void* P_state, void* w_state, double* stream_state, void* workspace, void* stream);
int  btk_rls_init_nc(int mode, const void* v, int per_stream_v, const void* cx, int NC, double p0, int S, int K, int N,
                     void* P_state, void* w_state, void* stream);
int  btk_rls_process_nc(int mode, const double* params, const void* v, int per_stream_v, const void* cx, int NC,
                        const void* X, void* Y, int S, int M, int N, long T_stride, long T,
                        void* P_state, void* w_state, double* stream_state, void* workspace, void* stream);

/* ---- Zelinski post-filter -------------------------------------------------------------------
 * Replaces ZelinskiFilter_f / ZelinskiFilter / ZelinskiPostFilter::next
 * (postfilter/postfilter.cc:57-219, 424-491).  Two calls per block:
 * btk_bf_apply_stats = btk_bf_apply plus, from the SAME pass over X, the per-frame statistics of
 *   the time-aligned snapshot x' = conj(d) x:  C = sum_{i<j} x'_i conj(x'_j) (complex64
 *   [S][K][T_stride]) and E = sum_i |x'_i|^2 (float32 [S][K][T_stride]).  D [dev] has W's layout
 *   and holds wq (type & 8, TYPE_ZELINSKI2) or ta_ (postfilter.cc:452-457).
 * btk_zelinski_process = the alpha-recursion on the summed cross/auto spectral densities
 *   (alpha = 0 for the post-filter's first two frames, :460-463), the gain
 *   clamp(f(Phi)/Psi * 2/(N-1), 1e-4, 1) with f = max(Re,0) (type & 1) or |.|, and y <- W y for
 *   frames >= min_frames.  frames_done = frames this post-filter produced before this block.
 *   State [dev]: phi complex64 [S][K], psi float32 [S][K], w_last float32 [S][K] (postfilter_weights()).
 *   Zero the state where the reference re-allocates BeamformerWeights (beamformer.cc:1082-1092).  */
int  btk_bf_apply_stats(const void* W, const void* D, int per_stream_weights, const void* X, void* Y,
                        void* C, float* E, int S, int K, int N, long T_stride, long T, void* stream);
int  btk_zelinski_process(void* Y, const void* C, const float* E, int S, int K, int N, long T_stride, long T,
                          double alpha, int type, int min_frames, long frames_done,
                          void* phi_state, float* psi_state, float* w_last, void* stream);

/* ---- McCowan / Lefkimmiatis post-filters ----------------------------------------------------------
 * Replace McCowanPostFilter::{estimate_average_clean_PSD_, post_filtering_, next} (postfilter/postfilter.cc:798-935)
 * and LefkimmiatisPostFilter::{calc_inverse_noise_spatial_spectral_matrix, calcLambda, estimate_average_noise_PSD_,
 * post_filtering_, next} (:967-1190).  R [dev] complex64 [K][N][N] is the noise coherence matrix R_ (build it with
 * btk_mvdr_diffuse_model == set_diffuse_noise_model :560-618, btk_mvdr_diagonal_loading == set_all_diagonal_loading
 * :620-632).  btk_pf_coherence_coeffs turns it (with threshold_of_Rij_) into the pair-weight matrices Cs (clean PSD)
 * and Cv (noise PSD, may be NULL for McCowan), complex64 [K][N][N].  btk_bf_apply_stats2 = btk_bf_apply plus, from
 * the same snapshots, the per-frame quadratic forms U (and V) complex64 [S][K][T_stride] and E (as btk_bf_apply_stats).
 *   McCowan:      btk_zelinski_process(Y, U, E, ...) -- the gain formula is Zelinski's with the weighted sum.
 *   Lefkimmiatis: btk_lefkimmiatis_process(Y, U, V, lambda, fbinX1, ...), lambda [dev] complex64 [K] = d^H pinv(R) d
 *                 from btk_mvdr_lambda (Cholesky with the identity fallback of :975-977), state u/v complex64 [S][K]. */
int  btk_pf_coherence_coeffs(const void* R, float threshold, int K, int N, void* Cs, void* Cv, void* stream);
int  btk_bf_apply_stats2(const void* W, const void* D, int per_stream_weights, const void* X, void* Y,
                         const void* Cs, const void* Cv, void* U, void* V, float* E,
                         int S, int K, int N, long T_stride, long T, void* stream);
int  btk_lefkimmiatis_process(void* Y, const void* U, const void* V, const void* lambda, int fbinX1,
                              int S, int K, int N, long T_stride, long T, double alpha, int type, int min_frames,
                              long frames_done, void* u_state, void* v_state, float* w_last, void* stream);
int  btk_mvdr_lambda(const void* R, const void* d, void* lambda, int K, int N, float threshold,
                     void* scratch, int* fallback_count, void* stream);

/* ---- Spatial covariance accumulation ----------------------------------------------------------
 * btk_frame_energy: |X_0^H X_0| / M of channel 0 over all M bins, per frame
 *   (MultiChannelSource.update_snapshot_array, lib/pybeamformer.py:263-277); energy [dev] [S][e_stride].
 * btk_cov_frame_gate: w[s][t] = (energy > threshold) * label[s][t] and count[s] += sum_t w
 *   (accu_stats_from_label, pybeamformer.py:967-985; label==NULL means all frames eligible).
 * btk_cov_accumulate: R[s][k] += sum_t tf[s][k][t] * fw[s][t] * x x^H  (either weight may be NULL;
 *   pybeamformer.py:979-981, 1087-1093, 1136-1147).  R [dev] complex64 [S][K][N][N].
 *   use_mfma != 0 selects the v_mfma_f32_32x32x2_f32 kernel, 0 the vector-ALU kernel.
 * btk_cov_finalize: R <- R/count, then (gamma > 0) improve_matrix_condition
 *   (pybeamformer.py:994-1000, 1200-1207, 1249-1263); count [dev] [S] or [S][K].                  */
int  btk_frame_energy(const void* X, int S, int M, int N, long T_stride, long T, float* energy,
                      long e_stride, void* stream);
int  btk_cov_frame_gate(const float* energy, const float* label, int S, long T_stride, long T,
                        float energy_threshold, float* frame_weights, float* frame_count, void* stream);
int  btk_cov_accumulate(const void* X, const float* tf_weights, const float* frame_weights, void* R,
                        int S, int K, int N, long T_stride, long T, int use_mfma, void* stream);
int  btk_cov_finalize(void* R, const float* count, int count_per_bin, int S, int K, int N, float gamma,
                      void* stream);

/* ---- Batch second-order-statistics beamformers: weight design from accumulated covariances -------
 * btk_cov_mask_count: count[s][k] += sum_t trunc(tf[s][k][t]) * fw[s][t] -- the per-bin integer frame counters of
 *   accu_stats_from_tfmask (lib/pybeamformer.py:1127-1147; the reference's counters are integer arrays, so fractional
 *   mask values weight the covariance but do not count).  count [dev] float32 [S][K].
 * btk_cov_trace_normalize: R <- R / (tr(R)/N) over nbins matrices (SubbandGEVBeamformer.finalize_stats :1326).
 * btk_bmvdr_weights: wqH_k = conj( inv(Rn_k) Rt_k u / (offset + tr(inv(Rn_k) Rt_k)) ), u = e_ref_mic
 *   (SubbandBlindMVDRBeamformer.calc_beamformer_weights :1225-1247).
 * btk_gev_weights: wqH_k = conj of the principal generalised eigenvector of (Rt_k, Rn_k), v^H Rn v = 1, phase of bin
 *   k aligned to bin k-1 (SubbandGEVBeamformer.calc_beamformer_weights :1280-1303).  scipy.linalg.eigh leaves the
 *   phase of each eigenvector to LAPACK; bin 0 is rotated so that its largest component is real positive, i.e. the
 *   result equals the reference's up to one global sign.
 * Rt, Rn [dev] complex64 [nbins][N][N]; WqH [dev] complex64 [nbins][N] (the beamformer output is wqH . x);
 * scratch [dev] btk_sos_scratch_bytes(nbins, N) bytes; *fail_count [dev] += bins whose Rn is not positive definite
 * (the reference raises ArithmeticError there).                                                               */
int  btk_cov_mask_count(const float* tf_weights, const float* frame_weights, int S, int K, long T_stride, long T,
                        float* count, void* stream);
int  btk_cov_trace_normalize(void* R, int nbins, int N, void* stream);
long btk_sos_scratch_bytes(int nbins, int N);
int  btk_bmvdr_weights(const void* Rt, const void* Rn, int nbins, int N, int ref_mic, double offset, void* WqH,
                       void* scratch, int* fail_count, void* stream);
int  btk_gev_weights(const void* Rt, const void* Rn, int K, int N, void* WqH, void* scratch, int* fail_count,
                     void* stream);

/* ---- SubbandMVDR weight design ------------------------------------------------------------------
 * btk_mvdr_diffuse_model: set_diffuse_noise_model (beamformer.cc:2442-2509); mpos [dev] float32 [N][3],
 *   R [dev] complex64 [K][N][N].
 * btk_mvdr_diagonal_loading: set_all_diagonal_loading (beamformer.cc:2511-2523) over nbins matrices.
 * btk_mvdr_weights: calc_mvdr_weights (beamformer.cc:2350-2402): w_k = R_k^-1 d / (N d^H R_k^-1 d),
 *   w_0 = ones; batched complex Cholesky, identity fall-back counted in *fallback_count [dev int]
 *   when a pivot <= threshold (the reference's pseudoinverse() failure path, :262-270, :2381-2383).
 *   wq [dev] complex64 [K][N] = d; W [dev] complex64 [K][N]; scratch [dev] [K][N][N] complex64 only
 *   needed for N > 271 (N <= 136: R_k in LDS; 136 < N <= 271: R_k in the matrix cores' accumulator
 *   registers, read once, never copied; above: panel solver on a copy of R_k); may be NULL otherwise. */
/* btk_mvdr_scratch_bytes: bytes of `scratch` the design entries below (btk_mvdr_weights*, btk_mvdr_lambda) need for K bins of
 *   N channels -- 0 when the solver that will run keeps R_k in LDS or in registers (scratch may then be NULL).            */
long btk_mvdr_scratch_bytes(int K, int N);
int  btk_mvdr_diffuse_model(const float* mpos, int N, int M, float samplerate, float sspeed, void* R, void* stream);
int  btk_mvdr_diagonal_loading(void* R, int nbins, int N, float weight, void* stream);
int  btk_mvdr_weights(const void* R, const void* wq, void* W, int K, int N, float threshold,
                      void* scratch, int* fallback_count, void* stream);
/* btk_mvdr_divide_nondiagonal: divide_nondiagonal_elements / divide_all_nondiagonal_elements (beamformer.cc:2589-2599,
 *   beamformer.h:357-362): R_xy /= (1 + mu), x != y, over nbins matrices.
 * btk_mvdr_weights_flags: btk_mvdr_weights(_shard) that also reports WHICH bins stopped in the Cholesky factorisation
 *   (fail_flags [dev] int32 [K]); btk_mvdr_pinv_fallback re-solves exactly those bins with the reference's own rule -- the
 *   float32-rounded matrix is pseudo-inverted through its SVD, singular values < threshold are zeroed and make the call
 *   "fail", upon which the identity is substituted (pseudoinverse(), beamformer.cc:232-289; :2381-2383) -- and patches W.
 *   It synchronises the stream; *identity_count [host] = bins that ended with the identity.  A Hermitian R_k that is
 *   indefinite but non-singular gets the pseudo-inverse weights the reference computes, not the identity.
 * btk_pinv: that pseudo-inverse for one host matrix A complex128 [M][N] -> invA [N][M]; *below_threshold = number of
 *   singular values zeroed (> 0 == the reference's `return false`).                                                   */
int  btk_mvdr_divide_nondiagonal(void* R, int nbins, int N, float mu, void* stream);
int  btk_mvdr_weights_flags(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                            void* scratch, int* fallback_count, int* fail_flags, void* stream);
/* btk_mvdr_weights_streams: the design of S independent streams (each with its own covariance matrices and look direction -- the
 *   SMI-MVDR of S utterances) in ONE launch: R [dev] complex64 [S][K][N][N], wq / W [dev] [S][K][N], fail_flags [dev] int [S*K];
 *   bin 0 of EVERY stream gets the all-ones weight.  The flagged bins go through btk_mvdr_pinv_fallback(R, wq, W, S*K, N, 0, ...)
 *   (a stream's bin 0 is never flagged).  scratch [dev] [S*K][N][N] complex64 for N > 271 only.                                   */
int  btk_mvdr_weights_streams(const void* R, const void* wq, void* W, int S, int K, int N, float threshold,
                              void* scratch, int* fallback_count, int* fail_flags, void* stream);
int  btk_mvdr_pinv_fallback(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                            const int* fail_flags, int* identity_count, void* stream);
/* The solve behind it (pinv_kernels.hip), one workgroup per flagged bin, float64 on the float32-rounded matrix: while the matrix
 *   fits in LDS (N <= 70) a one-sided Jacobi SVD with the reference's threshold / identity rule, weights formed without the
 *   inverse; above that the explicit inverse (in-place Gauss-Jordan with row pivoting in a slice of `scratch`) and
 *   1 / sigma_min = || R^-1 ||_2 by power iteration -- the rule only asks whether SOME singular value is below the threshold, and
 *   if none is the pseudo-inverse is the inverse.  btk_mvdr_pinv_fallback_async: on `stream`, no allocation, no host
 *   synchronisation; identity_count [dev int[2]]: [0] += bins that ended with the identity, [1] += bins whose Jacobi sweeps hit
 *   their limit of 60 without converging; scratch [dev] btk_mvdr_pinv_scratch_bytes(K, N) bytes (0 for the LDS form).
 *   btk_mvdr_pinv_not_converged: that second count for this thread's last (blocking) btk_mvdr_pinv_fallback call.
 *   btk_mvdr_pinv_fallback_host: the SVD solve bin by bin on one host thread (the round-2 form; checker / A-B timing only). */
long btk_mvdr_pinv_scratch_bytes(int K, int N);
int  btk_mvdr_pinv_fallback_async(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                                  const int* fail_flags, int* identity_count, void* scratch, void* stream);
int  btk_mvdr_pinv_fallback_host(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                                 const int* fail_flags, int* identity_count, void* stream);
int  btk_mvdr_pinv_not_converged(void);
int  btk_pinv(const double* A, int M, int N, float threshold, double* invA, int* below_threshold);
/* ---- The reference's float32 csvdc, decision for decision -------------------------------------------
 * pseudoinverse() (beamformer/beamformer.cc:232-289) returns false -- and calc_mvdr_weights (:2379-2384) /
 * LefkimmiatisPostFilter::calc_inverse_noise_spatial_spectral_matrix (postfilter/postfilter.cc:967-980) then use the identity --
 * when LINPACK's float32 csvdc (matrix/linpack_c.cc:9516) reports INFO != 0 (no convergence within 30 QR sweeps) or leaves a
 * singular value below the threshold.  On the 256-microphone diffuse model of BASELINE config C5 INFO != 0 on about half the
 * bins; the decision depends on the float32 roundings, so it is reproduced with the same arithmetic in the same order.
 * btk_csvdc_values: csvdc with job = 0 (no vectors: they never feed back into s, e or INFO) for K matrices A [dev] complex64
 *   [K][n][p] row-major (not modified); s, e [dev] float32 [K][min(n + 1, p)] (either may be NULL), info [dev] int32 [K];
 *   scratch [dev] btk_csvdc_scratch_bytes(K, n, p) bytes (0: the matrix lives in LDS, scratch may be NULL).  Bit-identical to
 *   the reference's compiled csvdc (tests/test_gpu_linpack_rule.py against oracle/_ref and tests/golden/c5_csvdc_info.npz).
 * btk_mvdr_linpack_rule: applies that decision to a design some solver has already written: R [K][N][N], wq [K][N]; every bin
 *   for which pseudoinverse() returns false gets W_k = d / (N d^H d) (W may be NULL) and lambda_k = d^H d (lambda may be NULL)
 *   and its fail_flags entry (may be NULL) is cleared, so btk_mvdr_pinv_fallback afterwards only visits bins that converged but
 *   are not positive definite.  skip_dc != 0: the DC bin (first_bin + k == 0, or k % kper == 0 for kper > 0 stacked streams)
 *   is left alone like calc_mvdr_weights does; the Lefkimmiatis filter decomposes bin 0 too (skip_dc = 0).
 *   counts [dev int[2]] (may be NULL): [0] += bins with INFO != 0, [1] += converged bins with a singular value < threshold.
 *   scratch [dev] btk_mvdr_linpack_rule_scratch_bytes(K, N) bytes.  All on `stream`, no allocation, no synchronisation.      */
long btk_csvdc_scratch_bytes(int K, int n, int p);
int  btk_csvdc_values(const void* A, int K, int n, int p, float* s, float* e, int* info, void* scratch, void* stream);
long btk_mvdr_linpack_rule_scratch_bytes(int K, int N);
int  btk_mvdr_linpack_rule(const void* R, const void* wq, void* W, void* lambda, int K, int N, int first_bin, int kper,
                           int skip_dc, float threshold, int* fail_flags, int* counts, void* scratch, void* stream);
/* The same solve for a bin SHARD [first_bin, first_bin + K) of a bin-sharded run: only global bin 0 gets the
 * all-ones weight of calc_mvdr_weights (beamformer.cc:2369-2371).                                                    */
int  btk_mvdr_weights_shard(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                            void* scratch, int* fallback_count, void* stream);

/* ---- Multi-channel WPE dereverberation ---------------------------------------------------------
 * Replaces MultiChannelWPEDereverberation::estimate_filter / calc_every_channel_output
 * (dereverberation/dereverberation.cc:312-698).  X [dev] complex64 [S][K][C][T_stride] (the snapshot
 * layout with C channels); L = upperN-lowerN+1 lags, prediction order P = C*L, lag vector ordered
 * channel-major then lag (:540-555).  lower_bw/upper_bw = lower_bandWidthN_/upper_bandWidthN_ (:361-369):
 * bins with lower_bw < k < upper_bw are left untouched.
 * btk_wpe_estimate: `iterations` rounds of { theta = max(|y - g^H ybar|,1e-3)^2 ; R = A diag(1/theta) A^H
 *   + bias I (fp32 MFMA HERK) ; r ; diagonal loading |R_ii| + max|R_ii| 10^(load_db/10) ; Cholesky solve }.
 *   G [dev] complex64 [S][C][K][P] in/out (zeros == a fresh object / next_speaker()).
 *   workspace [dev] btk_wpe_workspace_bytes(...) bytes; *fail_count [dev int] counts failed factorisations
 *   (the reference throws jnumeric_error, :676-680).
 * btk_wpe_apply: OUT[s][k][c][t] = y_c(t) - g_c^H ybar(t) for t >= lowerN using the L-frame ring rule of
 *   :471-480; other frames/bins are copied.                                                          */
long btk_wpe_workspace_bytes(int S, int K, int C, int lowerN, int upperN, long T_stride);
int  btk_wpe_estimate(const void* X, int S, int K, int C, long T_stride, long T, int lowerN, int upperN,
                      int iterations, double load_db, double diagonal_bias, int lower_bw, int upper_bw,
                      void* G, void* workspace, int* fail_count, void* stream);
int  btk_wpe_apply(const void* X, const void* G, void* OUT, int S, int K, int C, long T_stride, long T,
                   int lowerN, int upperN, int lower_bw, int upper_bw, void* stream);

/* ---- Multi-GPU: frequency-bin sharding ---------------------------------------------
 * One process per GPU.  Rank g owns the bins btk_bin_range(K, g, world) = [g ceil(K/world), ...) -- trailing ranks may
 * be short or empty -- for analysis storage (btk_fb_analysis_bins), weight design (btk_mvdr_weights_shard), covariance,
 * beamforming, post-filter; btk_allgather_bins is the ONE collective of the path: every rank contributes its
 * beamformed block Y_local [dev] complex64 [S][k1-k0][T_stride] and receives Y [dev] complex64 [S][K][T_stride] for the
 * synthesis bank.  nccl_comm: an initialised ncclComm_t of `world` ranks (RCCL over xGMI); the collective is enqueued on
 * `stream`.  RCCL is bound at run time, the library does not link it.  Stream sharding needs no entry point: streams are
 * independent (rank = stream mod world, no collective).                                                                 */
void btk_bin_range(int K, int rank, int world, int* k0, int* k1);
int  btk_allgather_bins(void* nccl_comm, const void* Y_local, void* Y, int S, int K, long T_stride, int rank, int world,
                        void* stream);
/* The even form -- north_star's "single RCCL all-gather": Y [dev] complex64 [S][Kp][T_stride] with
 * Kp = btk_bin_rows_padded(K, world) = world ceil(K / world) rows per stream (rows >= K are padding); every rank has written its
 * beamformed bins in place at rows [rank ceil(K/world), ...) (btk_bf_apply straight into that view), and ONE in-place
 * ncclAllGather per stream completes Y on every rank: no staging buffer, no copy.  btk_allgather_bins (grouped broadcasts) remains
 * for callers whose Y has exactly K rows.                                                                                  */
int  btk_bin_rows_padded(int K, int world);
int  btk_allgather_bins_inplace(void* nccl_comm, void* Y, int S, int K, long T_stride, int rank, int world, void* stream);

/* ---- Host-side weight design (double precision, one-off per look direction) -------------
 * BeamformerWeights::calcMainlobe (beamformer.cc:502-565): wq [host] complex128 [M][N].     */
int  btk_weights_mainlobe(int M, int N, float samplerate, const double* delays, double* wq);
/* the same with halfBandShift == true (beamformer.cc:515-527): bin k at (k + 0.5) fs / M, partner of bin k is bin M-1-k */
int  btk_weights_mainlobe_halfband(int M, int N, float samplerate, const double* delays, double* wq);
/* LCMV quiescent weights with two constraints (target + one null): calcMainlobe2 / calcMainlobeN +
 * calc_null_beamformer_ + calc_inverse_22mat_ (beamformer.cc:181-221, 299-363, 572-721); wq [M][N].   */
int  btk_weights_mainlobe_2(int M, int N, float samplerate, const double* delaysT, const double* delaysI, double* wq);
/* BeamformerWeights::calcMainlobeN (beamformer/beamformer.cc:600-721) with NC >= 2 constraints: delaysIs [host]
 * float64 [NC-1][N].  NC = 2 takes the closed-form path above; NC > 2 inverts the NC x NC Gram matrix in float64
 * (the reference: float32 SVD pseudoinverse, :352-355).                                                        */
int  btk_weights_mainlobe_n(int M, int N, float samplerate, const double* delaysT, const double* delaysIs, int NC,
                            double* wq);
/* calc_blocking_matrix_ (beamformer.cc:373-454): a [host] complex128 [N]; B [host] [N][N-NC] */
int  btk_weights_blocking_matrix(const double* a, int N, int NC, double* B);
/* calcSidelobeCancellerU_f (beamformer.cc:752-767): wl = B wa                               */
int  btk_weights_sidelobe(const double* B, const double* wa, int N, int NC, double* wl);
/* Effective GSC weights (wq - wl, optional normalisation of calc_gsc_output :1228-1237) for
 * bins 0..M/2 as complex64 [K][N] ready for btk_bf_apply; bin 0 is wq_0 alone (:1288-1291). */
int  btk_weights_gsc_effective(const double* wq, const double* wl, int M, int N,
                               int normalize, float* w_out);

#ifdef __cplusplus
}
#endif
#endif /* BTKHIP_H */
