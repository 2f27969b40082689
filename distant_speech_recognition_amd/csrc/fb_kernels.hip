// fb_kernels.hip -- oversampled modulated-DFT analysis / synthesis filter banks for gfx950.
//
// Replaces OverSampledDFTAnalysisBank::next and OverSampledDFTSynthesisBank::next
// (reference modulated/modulated.cc:375-409, 553-612) with whole-tile kernels:
//
//   analysis : one workgroup = one (stream,channel) x TT consecutive frames.
//              PCM span -> LDS (coalesced), polyphase FIR in registers (prototype taps held in
//              VGPRs), real-input FFT as an M/2-point complex FFT in LDS, Hermitian post-pass,
//              bins 0..M/2 written to X[S][K][N][T] as TT*8-byte contiguous runs.
//   synthesis: one workgroup = one stream x TB output blocks; Hermitian pre-pass + M/2-point
//              complex FFT (forward) gives the real sequence of modulated.cc:559-563, the
//              polyphase/overlap-add of :594-606 runs out of an LDS ring of m*R frames.
//
// The closed forms used here (validated against the literal ring-buffer restatement in
// oracle/btk_oracle.c):
//   analysis frame t: newest sample index n_t = (t + laN + 1) D - 1,
//       p[i] = sum_{k<m} h[i + M k] x[n_t - i - M k],  X_t[kappa] = sum_i p[i] e^{+j 2 pi kappa i / M}
//   synthesis block b: newest input frame f = b + pd,
//       v_f[i] = Re sum_kappa Y_f[kappa] e^{-j 2 pi kappa i / M}
//       s_f[i] = sum_{k<m} g[M-1-i+M k] v_{f-Rk}[i],   out_b[D-1-d] = sum_{j<R} s_{f-(R-1-j)}[d + j D]
#include "btk_internal.h"
#include "fft_lds.h"
#include <cstdlib>

namespace {

constexpr int NT = 256;                       // threads per workgroup (4 wavefronts)

template <int LOG2M> struct FbCfg {
  static constexpr int M = 1 << LOG2M;
  static constexpr int NF = M / 2;            // complex FFT length
  static constexpr int LOG2NF = LOG2M - 1;
  // frames per workgroup tile: FFT buffer stays at ~32 KB, capped at 32 frames
  static constexpr int TT = (8192 / M) > 32 ? 32 : (8192 / M);
  static constexpr int STRIDE = NF + 1;       // +1: slot for bin NF and bank de-phasing
};

// ------------------------------------------------------------------ analysis
// MT > 0: compile-time prototype length factor m (taps in registers); MT == 0: runtime m.
template <int LOG2M, int MT, bool POLYPHASE_ONLY>
__global__ __launch_bounds__(NT)
void analysis_kernel(const float* __restrict__ pcm, long nsamples, long pcm_stride,
                     const float* __restrict__ proto, const float2* __restrict__ twg,
                     int m_rt, int D, int laN, float gain,
                     int N, int K, float2* __restrict__ X, float* __restrict__ P,
                     long T_stride, long t0, long tcount, int ntiles, int nchan, int k0, int k1)
{
  // X [S][K][N][T_stride] holds the bins [k0, k1) of the plan (K = k1 - k0; the whole range for an unsharded plan)
  using C = FbCfg<LOG2M>;
  constexpr int M = C::M, NF = C::NF, TT = C::TT, STRIDE = C::STRIDE;
  const int m = MT > 0 ? MT : m_rt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* zbuf = reinterpret_cast<float2*>(smem);                 // [TT][STRIDE]
  float2* tw = zbuf + TT * STRIDE;                                // [M]
  float* xs = reinterpret_cast<float*>(tw + M);                   // [(TT-1) D + m M]

  const int tid = threadIdx.x;
  // XCD-aware mapping: consecutive tiles of one channel share an XCD's L2 (PCM halo reuse).
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int chan = (slot / ntiles) * 8 + xcd;
  const int tile = slot % ntiles;
  if (chan >= nchan) return;
  const long tt0 = (long)tile * TT;                               // first frame of tile, relative to t0
  const long tabs0 = t0 + tt0;                                    // absolute frame number

  for (int j = tid; j < M; j += NT) tw[j] = twg[j];

  // ---- PCM span -> LDS.  local index l <-> global sample g0 + l, g0 = (tabs0+laN+1) D - m M
  const int span = (TT - 1) * D + m * M;
  const long g0 = (tabs0 + laN + 1) * (long)D - (long)m * M;
  const float* src = pcm + (long)chan * pcm_stride;
  const bool vec_ok = ((pcm_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & 15) == 0) && ((g0 & 3) == 0) && ((D & 3) == 0);
  if (vec_ok && g0 >= 0 && g0 + span <= nsamples) {
    for (int l = tid * 4; l < span; l += NT * 4)
      *reinterpret_cast<float4*>(xs + l) = *reinterpret_cast<const float4*>(src + g0 + l);
  } else {
    for (int l = tid; l < span; l += NT) {
      const long g = g0 + l;
      xs[l] = (g >= 0 && g < nsamples) ? src[g] : 0.0f;
    }
  }
  __syncthreads();

  // ---- polyphase: z[f][n] = (p[2n], p[2n+1])
  constexpr int NOWN = NF > NT ? NF / NT : 1;        // pair-indices n owned by a thread
  constexpr int FPT = NF > NT ? TT : TT * NF / NT;   // frames handled per owned n
  constexpr int FSTEP = NF > NT ? 1 : NT / NF;
  constexpr int MTR = MT > 0 ? MT : 1;
  const int f_first = NF > NT ? 0 : tid / NF;
#pragma unroll
  for (int q = 0; q < NOWN; q++) {
    const int n = NF > NT ? tid + q * NT : tid % NF;
    float2 hreg[MTR];
    if (MT > 0) {
#pragma unroll
      for (int k = 0; k < MT; k++) hreg[k] = *reinterpret_cast<const float2*>(proto + 2 * n + M * k);
    }
#pragma unroll 4
    for (int fi = 0; fi < FPT; fi++) {
      const int f = f_first + fi * FSTEP;
      // x[n_t - (2n+1) - M k] and x[n_t - 2n - M k] sit at xs[base - M k], xs[base - M k + 1]
      const int base = f * D + m * M - 2 - 2 * n;
      float p0 = 0.0f, p1 = 0.0f;
      if (MT > 0) {
#pragma unroll
        for (int k = 0; k < MT; k++) {
          const float2 x = *reinterpret_cast<const float2*>(xs + base - M * k);
          p0 = fmaf(hreg[k].x, x.y, p0);
          p1 = fmaf(hreg[k].y, x.x, p1);
        }
      } else {
        for (int k = 0; k < m; k++) {
          const float2 h = *reinterpret_cast<const float2*>(proto + 2 * n + M * k);
          const float2 x = *reinterpret_cast<const float2*>(xs + base - M * k);
          p0 = fmaf(h.x, x.y, p0);
          p1 = fmaf(h.y, x.x, p1);
        }
      }
      if (POLYPHASE_ONLY) {
        if (tt0 + f < tcount)
          *reinterpret_cast<float2*>(P + (((long)chan * tcount + tt0 + f) * M + 2 * n)) = make_float2(p0, p1);
      } else {
        zbuf[f * STRIDE + n] = make_float2(p0, p1);
      }
    }
  }
  if (POLYPHASE_ONLY) return;
  __syncthreads();

  // ---- M/2-point complex FFT, backward sign (e^{+j...}), gsl_fft_complex_radix2_backward
  fft_lds<C::LOG2NF, TT, STRIDE, NT, +1>(zbuf, tw, tid);

  // ---- Hermitian post-pass, in place: X[k] = E[k] + W^k O[k], X[NF-k] from the same pair
  //      E = (Z[k] + conj Z[NF-k])/2,  O = -j (Z[k] - conj Z[NF-k])/2,  W = e^{+j 2 pi / M}
  for (int idx = tid; idx < TT * (NF / 2 + 1); idx += NT) {
    const int f = idx / (NF / 2 + 1), k = idx % (NF / 2 + 1);
    float2* row = zbuf + f * STRIDE;
    const float2 zk = row[k];
    const float2 zc = cconjf(k == 0 ? row[0] : row[NF - k]);
    const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
    const float2 dd = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
    const float2 o = make_float2(dd.y, -dd.x);                    // -j * dd
    const float2 wo = cmulf(tw[k], o);
    const float2 xk = caddf(e, wo);
    // X[NF-k] = conj(E[k]) + W^{NF-k} conj(O[k]),  W^{NF-k} = -conj(W^k)  => conj(E - W^k O)
    const float2 xm = cconjf(csubf(e, wo));
    row[k] = make_float2(xk.x * gain, xk.y * gain);
    row[NF - k] = make_float2(xm.x * gain, xm.y * gain);          // k==0 writes bin NF; k==NF/2 rewrites itself
  }
  __syncthreads();

  // ---- store bins 0..NF: for each k a run of TT frames (TT*8 B contiguous)
  const int s = chan / N, nch = chan % N;
  float2* xo = X + ((long)s * K * N + nch) * T_stride + tt0;
  for (int idx = tid; idx < TT * (NF + 1); idx += NT) {
    const int f = idx % TT, k = idx / TT;
    if (tt0 + f < tcount && k >= k0 && k < k1) xo[(long)(k - k0) * N * T_stride + f] = zbuf[f * STRIDE + k];
  }
}

// ------------------------------------------------------------------ synthesis
template <int LOG2M> struct SynCfg {
  static constexpr int M = 1 << LOG2M;
  static constexpr int NF = M / 2;
  static constexpr int LOG2NF = LOG2M - 1;
  static constexpr int FR = (4096 / M) > 32 ? 32 : (4096 / M);                    // frames per FFT round
  static constexpr int STRIDE = NF + 1;
};

// One workgroup: stream s, output blocks [bt0, bt0+TB).  Needs v_f for f in [f_lo, f_hi],
// f_lo = bt0 + pd - (R-1) - R (m-1), f_hi = bt0 + TB - 1 + pd.  v rows live in LDS as
// vbuf[(f - f_lo)][M] floats.
template <int LOG2M, int FR>
__global__ __launch_bounds__(NT)
void synthesis_kernel(const float2* __restrict__ Y, long nframes, long T_stride, int K,
                      const float* __restrict__ proto, const float2* __restrict__ twg,
                      int m, int R, int D, int pd, float gain, int TB,
                      float* __restrict__ out, long out_stride, long b0, long bcount)
{
  using C = SynCfg<LOG2M>;
  constexpr int M = C::M, NF = C::NF, STRIDE = C::STRIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* zbuf = reinterpret_cast<float2*>(smem);                 // [FR][STRIDE]
  float2* tw = zbuf + FR * STRIDE;                                // [M]
  float* vbuf = reinterpret_cast<float*>(tw + M);                 // [NV][M]
  const int NV = TB + R * m - 1;                                  // frames of v needed

  const int tid = threadIdx.x;
  const int s = blockIdx.y;
  const long bt0 = b0 + (long)blockIdx.x * TB;
  const long f_lo = bt0 + pd - (R - 1) - (long)R * (m - 1);
  const float2* Ys = Y + (long)s * K * T_stride;

  for (int j = tid; j < M; j += NT) tw[j] = twg[j];
  __syncthreads();

  for (int r0 = 0; r0 < NV; r0 += FR) {
    // ---- Hermitian pre-pass: Zc[k] = (Y[k] + conj Y[NF-k]) + j W^{-k} (Y[k] - conj Y[NF-k]),
    //      W = e^{+j 2 pi/M}; imaginary parts of Y[0], Y[NF] are ignored like the real part
    //      taken at modulated.cc:562-563.
    for (int idx = tid; idx < FR * NF; idx += NT) {
      const int fr = idx % FR, k = idx / FR;                      // frames fastest: coalesced along t
      const long f = f_lo + r0 + fr;
      float2 z = make_float2(0.f, 0.f);
      if (r0 + fr < NV && f >= 0 && f < nframes) {
        float2 a = Ys[(long)k * T_stride + f];
        float2 bq = Ys[(long)(NF - k) * T_stride + f];
        if (k == 0) { a.y = 0.f; bq.y = 0.f; }
        const float2 bc = cconjf(bq);
        const float2 sm = caddf(a, bc), df = csubf(a, bc);
        const float2 t = cmulf(cconjf(tw[k]), df);                // W^{-k} (Y[k] - conj Y[NF-k])
        z = make_float2(sm.x - t.y, sm.y + t.x);                  // sm + j t
      }
      zbuf[fr * STRIDE + k] = z;
    }
    __syncthreads();
    fft_lds<C::LOG2NF, FR, STRIDE, NT, -1>(zbuf, tw, tid);
    // v[2n] = Re z[n], v[2n+1] = Im z[n]
    for (int idx = tid; idx < FR * NF; idx += NT) {
      const int fr = idx / NF, n = idx % NF;
      if (r0 + fr < NV)
        *reinterpret_cast<float2*>(vbuf + (long)(r0 + fr) * M + 2 * n) = zbuf[fr * STRIDE + n];
    }
    __syncthreads();
  }

  // ---- polyphase + overlap-add.  s_f[i] = sum_k g[M-1-i+M k] v_{f-Rk}[i]; block b (newest
  //      frame f = b+pd) sums s_{f-(R-1-j)}[d + j D] over j in the reference's order with a
  //      float32 running sum (modulated.cc:594-606).
  float* os = out + (long)s * out_stride;
  for (int idx = tid; idx < TB * D; idx += NT) {
    const int bb = idx / D, d = idx % D;
    const long bglob = bt0 + bb;
    if (bglob - b0 >= bcount) continue;
    float acc = 0.f;
    for (int j = 0; j < R; j++) {
      // gsi_ only holds s-frames computed by earlier next() calls: before block 0 the ring is
      // still zero (nothing is pushed to gsi_ while priming, modulated.cc:574-578,600)
      if (bglob - (R - 1 - j) < 0) continue;
      const int i = d + j * D;
      const int vrow = bb + j + R * (m - 1);                      // row of v_{f-(R-1-j)} in vbuf
      float sv = 0.f;
      for (int k = 0; k < m; k++)
        sv = fmaf(proto[(M - 1 - i) + M * k], vbuf[(long)(vrow - R * k) * M + i], sv);
      acc += sv;
    }
    if (gain > 0.f) acc *= gain;
    os[(bglob - b0) * D + (D - 1 - d)] = acc;
  }
}

template <int LOG2M>
int launch_analysis(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N,
                    float2* X, float* P, long T_stride, long t0, long tcount, hipStream_t st)
{
  using C = FbCfg<LOG2M>;
  const int nchan = S * N;
  const int ntiles = (int)((tcount + C::TT - 1) / C::TT);
  const int chan_groups = (nchan + 7) / 8;
  const long nblocks = (long)chan_groups * ntiles * 8;
  const size_t lds = sizeof(float2) * (C::TT * C::STRIDE + C::M) + sizeof(float) * ((C::TT - 1) * fb->D + fb->m * C::M);
  if (lds > 160 * 1024) return btk_set_error(BTK_ERR_PARAMETER, "analysis tile needs %zu B of LDS (m too large)", lds);
  const float gain = fb->gain_factor > 0 ? (float)fb->gain_factor : 1.0f;
#define BTK_LAUNCH_ANA(MT, PO)                                                                          \
  do {                                                                                                  \
    auto kern = analysis_kernel<LOG2M, MT, PO>;                                                         \
    if (lds > 64 * 1024)                                                                                \
      BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                            \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(NT), lds, st, pcm, nsamples, pcm_stride,     \
                       fb->d_proto, fb->d_tw, fb->m, fb->D, fb->laN, gain, N, fb->kx1 - fb->kx0, X, P,  \
                       T_stride, t0, tcount, ntiles, nchan, fb->kx0, fb->kx1);                          \
  } while (0)
  if (P) {
    if (fb->m == 4) BTK_LAUNCH_ANA(4, true); else BTK_LAUNCH_ANA(0, true);
  } else {
    if (fb->m == 4) BTK_LAUNCH_ANA(4, false);
    else if (fb->m == 2) BTK_LAUNCH_ANA(2, false);
    else BTK_LAUNCH_ANA(0, false);
  }
#undef BTK_LAUNCH_ANA
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

template <int LOG2M, int FR>
int launch_synthesis_fr(const btk_fb* fb, const float2* Y, long nframes, long T_stride, int S,
                        float* out, long out_stride, long b0, long bcount, hipStream_t st)
{
  using C = SynCfg<LOG2M>;
  // TB output blocks per workgroup, bounded by LDS: (TB + R m - 1) rows of M floats
  const int extra = fb->R * fb->m - 1;
  const long lds_cap = (C::M <= 512 ? 96 : (FR == C::FR ? 150 : 160)) * 1024;
  long budget = (lds_cap - (long)sizeof(float2) * (FR * C::STRIDE + C::M)) / (long)(sizeof(float) * C::M);
  int TB = (int)(budget - extra);
  if (TB > 32) TB = 32;
  if (TB < 1) return 0;
  const size_t lds = sizeof(float2) * (FR * C::STRIDE + C::M) + sizeof(float) * (size_t)C::M * (TB + extra);
  auto kern = synthesis_kernel<LOG2M, FR>;
  if (lds > 64 * 1024)
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned gx = (unsigned)((bcount + TB - 1) / TB);
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)S), dim3(NT), lds, st, Y, nframes, T_stride, fb->K,
                     fb->d_proto, fb->d_tw, fb->m, fb->R, fb->D, fb->pd, (float)fb->gain_factor, TB,
                     out, out_stride, b0, bcount);
  BTK_HIP_CHECK(hipGetLastError());
  return 1;
}

template <int LOG2M>
int launch_synthesis(const btk_fb* fb, const float2* Y, long nframes, long T_stride, int S,
                     float* out, long out_stride, long b0, long bcount, hipStream_t st)
{
  using C = SynCfg<LOG2M>;
  int rc = launch_synthesis_fr<LOG2M, C::FR>(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
  // the largest geometry (M = 2048, m R = 16 history rows) only fits with one frame per FFT round
  if constexpr (LOG2M == 11) {
    if (rc == 0) rc = launch_synthesis_fr<LOG2M, 1>(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
  }
  if (rc == 0) return btk_set_error(BTK_ERR_PARAMETER, "synthesis tile does not fit LDS (M=%d m=%d r=%d)", fb->M, fb->m, fb->r);
  return rc > 0 ? BTK_OK : rc;
}

}  // namespace

extern "C" {

int btk_fb_analysis(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride,
                    int S, int N, void* X, long T_stride, long t0, long tcount, void* stream)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis: not an analysis plan");
  if (S <= 0 || N <= 0 || tcount < 0 || T_stride < tcount || pcm_stride < nsamples)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_analysis: bad sizes S=%d N=%d tcount=%ld T_stride=%ld", S, N, tcount, T_stride);
  if (tcount == 0) return BTK_OK;
  hipStream_t st = as_stream(stream);
  float2* Xp = static_cast<float2*>(X);
  const bool no512 = btk_switches().disable_analysis512;                         // A/B switches of profiles/ (btk_internal.h)
  if (!no512) {
    const int rc = btk_analysis512_try(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, st);
    if (rc != 0) return rc > 0 ? BTK_OK : rc;
  }
  const bool nofast = btk_switches().disable_fast;
  if (!nofast) {
    const int rc = btk_fast_analysis_try(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, st);
    if (rc != 0) return rc > 0 ? BTK_OK : rc;
  }
  switch (fb->M) {
    case 64:   return launch_analysis<6>(fb, pcm, nsamples, pcm_stride, S, N, Xp, nullptr, T_stride, t0, tcount, st);
    case 128:  return launch_analysis<7>(fb, pcm, nsamples, pcm_stride, S, N, Xp, nullptr, T_stride, t0, tcount, st);
    case 256:  return launch_analysis<8>(fb, pcm, nsamples, pcm_stride, S, N, Xp, nullptr, T_stride, t0, tcount, st);
    case 512:  return launch_analysis<9>(fb, pcm, nsamples, pcm_stride, S, N, Xp, nullptr, T_stride, t0, tcount, st);
    case 1024: return launch_analysis<10>(fb, pcm, nsamples, pcm_stride, S, N, Xp, nullptr, T_stride, t0, tcount, st);
    case 2048: return launch_analysis<11>(fb, pcm, nsamples, pcm_stride, S, N, Xp, nullptr, T_stride, t0, tcount, st);
  }
  return btk_set_error(BTK_ERR_PARAMETER, "unsupported M=%d", fb->M);
}

int btk_fb_analysis_i16_direct(const btk_fb_t* fb)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_i16_direct: not an analysis plan");
  if (fb->m != 4 || !(fb->R == 1 || fb->R == 2 || fb->R == 4)) return 0;
  if (fb->M == 512 && !btk_switches().disable_analysis512) return 1;
  return !btk_switches().disable_fast && (fb->M == 256 || fb->M == 512 || fb->M == 1024 || fb->M == 2048);
}

int btk_fb_analysis_i16(const btk_fb_t* fb, const short* pcm, long nsamples, long pcm_stride,
                        int S, int N, void* X, long T_stride, long t0, long tcount, void* stream)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_i16: not an analysis plan");
  if (S <= 0 || N <= 0 || tcount < 0 || T_stride < tcount || pcm_stride < nsamples)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_analysis_i16: bad sizes S=%d N=%d tcount=%ld T_stride=%ld", S, N, tcount, T_stride);
  if (tcount == 0) return BTK_OK;
  if (!pcm || !X) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_i16: null argument");
  if (btk_fb_analysis_i16_direct(fb) != 1)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_i16: no int16 kernel for M=%d m=%d r=%d (btk_pcm_i16_to_f32 + btk_fb_analysis)", fb->M, fb->m, fb->r);
  int rc = btk_switches().disable_analysis512 ? 0 : btk_analysis512_i16_try(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, as_stream(stream));
  if (rc == 0) rc = btk_fast_analysis_i16_try(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, as_stream(stream));
  if (rc == 0) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_i16: geometry not covered");
  return rc > 0 ? BTK_OK : rc;
}

int btk_fb_analysis_bins(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                         long T_stride, long t0, long tcount, int k0, int k1, void* stream)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bins: not an analysis plan");
  if (k0 < 0 || k1 > fb->K || k0 > k1) return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_analysis_bins: bin range [%d, %d) outside [0, %d]", k0, k1, fb->K);
  if (k0 == k1) return BTK_OK;                           // the empty shard of a trailing rank
  btk_fb shard = *fb;                                    // the plan itself is shared and stays whole
  shard.kx0 = k0; shard.kx1 = k1;
  return btk_fb_analysis(&shard, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, stream);
}

int btk_fb_analysis_polyphase(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride,
                              int S, int N, float* P, long t0, long tcount, void* stream)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_polyphase: not an analysis plan");
  if (S <= 0 || N <= 0 || tcount < 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_analysis_polyphase: bad sizes");
  if (tcount == 0) return BTK_OK;
  hipStream_t st = as_stream(stream);
  switch (fb->M) {
    case 64:   return launch_analysis<6>(fb, pcm, nsamples, pcm_stride, S, N, nullptr, P, tcount, t0, tcount, st);
    case 128:  return launch_analysis<7>(fb, pcm, nsamples, pcm_stride, S, N, nullptr, P, tcount, t0, tcount, st);
    case 256:  return launch_analysis<8>(fb, pcm, nsamples, pcm_stride, S, N, nullptr, P, tcount, t0, tcount, st);
    case 512:  return launch_analysis<9>(fb, pcm, nsamples, pcm_stride, S, N, nullptr, P, tcount, t0, tcount, st);
    case 1024: return launch_analysis<10>(fb, pcm, nsamples, pcm_stride, S, N, nullptr, P, tcount, t0, tcount, st);
    case 2048: return launch_analysis<11>(fb, pcm, nsamples, pcm_stride, S, N, nullptr, P, tcount, t0, tcount, st);
  }
  return btk_set_error(BTK_ERR_PARAMETER, "unsupported M=%d", fb->M);
}

int btk_fb_synthesis(const btk_fb_t* fb, const void* Y, long nframes, long T_stride, int S,
                     float* out, long out_stride, long b0, long bcount, void* stream)
{
  if (!fb || !fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_synthesis: not a synthesis plan");
  if (S <= 0 || bcount < 0 || nframes > T_stride || out_stride < bcount * fb->D)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_synthesis: bad sizes S=%d bcount=%ld nframes=%ld T_stride=%ld", S, bcount, nframes, T_stride);
  if (bcount == 0) return BTK_OK;
  hipStream_t st = as_stream(stream);
  const float2* Yp = static_cast<const float2*>(Y);
  const bool no512 = btk_switches().disable_synthesis512;
  if (!no512) {
    const int rc = btk_synthesis512_try(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    if (rc != 0) return rc > 0 ? BTK_OK : rc;
  }
  const bool nofast = btk_switches().disable_fast;
  if (!nofast) {
    const int rc = btk_fast_synthesis_try(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    if (rc != 0) return rc > 0 ? BTK_OK : rc;
  }
  switch (fb->M) {
    case 64:   return launch_synthesis<6>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 128:  return launch_synthesis<7>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 256:  return launch_synthesis<8>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 512:  return launch_synthesis<9>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 1024: return launch_synthesis<10>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 2048: return launch_synthesis<11>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
  }
  return btk_set_error(BTK_ERR_PARAMETER, "unsupported M=%d", fb->M);
}

int btk_fb_synthesis_aligned_form(const btk_fb_t* fb)
{
  if (!fb || !fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_synthesis_aligned_form: not a synthesis plan");
  if (btk_switches().syn_narrow || fb->m != 4 || fb->R > 2) return 0;
  if (fb->M == 512) return btk_switches().disable_synthesis512 ? 0 : 1;      // synthesis512w_kernel (fb_analysis512.hip)
  return (fb->M == 1024 || fb->M == 2048) && !btk_switches().disable_fast;   // fast_synthesis_w_kernel (fb_fast.hip)
}

// Fused OverSampledDFTAnalysisBank xN -> SubbandDS/GSC/MVDR::next for static weights.  Falls back to the staged
// pair (btk_fb_analysis into `scratch` + btk_bf_apply) for geometries the fused kernel does not cover.
int btk_fb_analysis_bf(const btk_fb_t* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N,
                       const void* W, int per_stream_weights, void* Y, long T_stride, long t0, long tcount,
                       void* scratch, long scratch_bytes, void* stream)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf: not an analysis plan");
  if (!pcm || !W || !Y || !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf: null argument");
  if (S <= 0 || N <= 0 || tcount < 0 || T_stride < tcount || pcm_stride < nsamples)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_analysis_bf: bad sizes S=%d N=%d tcount=%ld T_stride=%ld", S, N, tcount, T_stride);
  if (tcount == 0) return BTK_OK;
  if (reinterpret_cast<uintptr_t>(scratch) & 15) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf: scratch must be 16-byte aligned");
  // fused geometries stage their weights in `scratch`: pairs [Sw][N][320] float4 at M = 512 (fb_analysis512.hip), the
  // transposed matrix [Sw][N][K] complex64 at M = 256 (fb_fast.hip); the staged fall-back checks its own size below
  const bool fuse512 = fb->M == 512 && fb->m == 4;
  const bool fusefast = fb->m == 4 && fb->M == 256 && (fb->R == 1 || fb->R == 2 || fb->R == 4);
  const long Sw = per_stream_weights ? S : 1;
  const long big_bytes = btk_big_analysis_bf_scratch_bytes(fb, S, N, per_stream_weights, tcount);      // M = 1024 / 2048 (fb_fused_big.hip)
  const long wt_bytes = fuse512 ? (long)sizeof(float4) * Sw * 320 * N : fusefast ? (long)sizeof(float2) * Sw * fb->K * N : big_bytes;
  hipStream_t st = as_stream(stream);
  const bool nofuse = btk_switches().disable_fused;
  if (!nofuse) {
    if ((fuse512 || fusefast || big_bytes) && scratch_bytes < wt_bytes)
      return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf: scratch too small (%ld < %ld)", scratch_bytes, wt_bytes);
    int rc = btk_analysis512_bf_try(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream_weights, scratch, Y, T_stride, t0, tcount, st);
    if (rc != 0) return rc > 0 ? BTK_OK : rc;
    if (big_bytes) {
      rc = btk_big_analysis_bf_try(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream_weights, scratch, Y, T_stride, t0, tcount, st);
      if (rc != 0) return rc > 0 ? BTK_OK : rc;
    }
    if (fusefast) {
      rc = btk_fast_analysis_bf_try(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream_weights, scratch, Y, T_stride, t0, tcount, st);
      if (rc != 0) return rc > 0 ? BTK_OK : rc;
    }
  }
  const long x_bytes = (long)sizeof(float2) * S * fb->K * N * tcount;
  if (scratch_bytes < x_bytes)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf: staged fall-back needs %ld bytes of scratch for the snapshots", x_bytes);
  int rc = btk_fb_analysis(fb, pcm, nsamples, pcm_stride, S, N, scratch, tcount, t0, tcount, stream);
  if (rc) return rc;
  // staged layout has T_stride == tcount for X; Y keeps the caller's stride only when they agree
  if (T_stride != tcount) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf: staged fall-back needs T_stride == tcount");
  return btk_bf_apply(W, per_stream_weights, scratch, Y, S, fb->K, N, tcount, tcount, stream);
}

int btk_fb_analysis_bf_fused(const btk_fb_t* fb)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_fused: not an analysis plan");
  if (btk_switches().disable_fused || fb->m != 4 || fb->kx0 != 0 || fb->kx1 != fb->K) return 0;
  const bool rok = fb->R == 1 || fb->R == 2 || fb->R == 4;
  if ((fb->M == 512 || fb->M == 256) && rok) return 1;                       // fb_analysis512.hip, fb_fast.hip
  return (fb->M == 1024 || fb->M == 2048) && fb->R == 2;                      // fb_fused_big.hip
}

int btk_fb_analysis_bf_i16_fused(const btk_fb_t* fb)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16_fused: not an analysis plan");
  if (btk_switches().disable_fused || fb->m != 4 || fb->kx0 != 0 || fb->kx1 != fb->K) return 0;
  if (fb->M == 256 && btk_switches().disable_fast) return 0;
  if (fb->M == 512 || fb->M == 256) return fb->R == 1 || fb->R == 2 || fb->R == 4;
  return (fb->M == 1024 || fb->M == 2048) && fb->R == 2;
}

int btk_fb_analysis_bf_i16(const btk_fb_t* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N,
                           const void* W, int per_stream_weights, void* Y, long T_stride, long t0, long tcount,
                           void* scratch, long scratch_bytes, void* stream)
{
  if (!fb || fb->synthesis) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: not an analysis plan");
  if (!pcm || !W || !Y || !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: null argument");
  if (S <= 0 || N <= 0 || tcount < 0 || T_stride < tcount || pcm_stride < nsamples)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_fb_analysis_bf_i16: bad sizes S=%d N=%d tcount=%ld T_stride=%ld", S, N, tcount, T_stride);
  if (tcount == 0) return BTK_OK;
  if (reinterpret_cast<uintptr_t>(scratch) & 15) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: scratch must be 16-byte aligned");
  if (reinterpret_cast<uintptr_t>(pcm) & 3) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: pcm must be 4-byte aligned");
  if (btk_fb_analysis_bf_i16_fused(fb) != 1)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: no int16 kernel for M=%d m=%d r=%d (btk_pcm_i16_to_f32 + btk_fb_analysis_bf)", fb->M, fb->m, fb->r);
  const long need = btk_fb_analysis_bf_scratch_bytes(fb, S, N, per_stream_weights, tcount);
  if (scratch_bytes < need) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: scratch too small (%ld < %ld)", scratch_bytes, need);
  hipStream_t st = as_stream(stream);
  int rc = btk_analysis512_bf_i16_try(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream_weights, scratch, Y, T_stride, t0, tcount, st);
  if (rc == 0) rc = btk_big_analysis_bf_i16_try(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream_weights, scratch, Y, T_stride, t0, tcount, st);
  if (rc == 0 && !btk_switches().disable_fast) rc = btk_fast_analysis_bf_i16_try(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream_weights, scratch, Y, T_stride, t0, tcount, st);
  if (rc == 0) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_analysis_bf_i16: geometry not covered");
  return rc > 0 ? BTK_OK : rc;
}

long btk_fb_analysis_bf_scratch_bytes(const btk_fb_t* fb, int S, int N, int per_stream_weights, long tcount)
{
  if (!fb) return -1;
  const bool rok = fb->R == 1 || fb->R == 2 || fb->R == 4;
  const bool nofuse = btk_switches().disable_fused;
  if (!nofuse && rok && fb->m == 4 && fb->M == 512)
    return (long)sizeof(float4) * (per_stream_weights ? S : 1) * 320 * N;                // weight pairs [Sw][N][320], see fb_analysis512.hip
  if (!nofuse && rok && fb->m == 4 && fb->M == 256)
    return (long)sizeof(float2) * (per_stream_weights ? S : 1) * fb->K * N;              // transposed weights [Sw][N][K], see fb_fast.hip
  if (!nofuse) {
    const long big = btk_big_analysis_bf_scratch_bytes(fb, S, N, per_stream_weights, tcount);   // weight pairs + channel-group partial blocks
    if (big) return big;
  }
  return (long)sizeof(float2) * S * fb->K * N * tcount;
}

}  // extern "C"
