// wpe_kernels.hip -- multi-channel WPE dereverberation (weighted prediction error) for gfx950.
//
// Replaces MultiChannelWPEDereverberation::{estimate_filter, calc_every_channel_output}
// (reference dereverberation/dereverberation.cc:312-698).  Per stream s, bin k and target channel c:
//   lag vector   ybar(t) = [ y_{c'}(t - lowerN - l) ],  c' < C channel-major, l < L = upperN-lowerN+1   (:540-555)
//   theta_c(t)   = max(|y_c(t) - g_c^H ybar(t)|, 1e-3)^2                                                (:619-646)
//   R_c = sum_{t>=lowerN} ybar ybar^H / theta_c + bias I,   r_c = sum conj(y_c) ybar / theta_c          (:557-615)
//   R_ii <- |R_ii| + max_i |R_ii| 10^{load_db/10}                                                       (:648-663)
//   g_c = R_c^{-1} r_c by complex Cholesky                                                              (:665-690)
//   output   y_c(t) - g_c^H ybar(t)   for t >= lowerN                                                   (:444-501)
// The lag matrix A [C*L][T] is never materialised: every kernel reads the snapshot rows
// X[s][k][c'][.] (frames contiguous) with a per-row shift.
//   wpe_predict_kernel : lanes own frames, (c',l) loop in registers, filter taps wave-uniform -> theta^-1 or output
//   wpe_herk_kernel    : the normal equations are a weighted HERK A diag(1/theta) A^H -> fp32 MFMA
//                        (v_mfma_f32_32x32x2_f32), 64x64 tiles of the lower triangle only
//   wpe_rvec_kernel    : r_c
//   wpe_solve_kernel   : diagonal bias + loading + in-place Cholesky (matrix in global memory / L2) + solves
#include "btk_internal.h"
#include "chol_blocked.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WT_ = 32;                // frames per LDS tile

struct WpeGeom { int K, C, L, lowerN, lower_bw, upper_bw; long T_stride, T; };

__device__ __forceinline__ bool bin_active(const WpeGeom& g, int k) { return !(k > g.lower_bw && k < g.upper_bw); }

// mode 0: Winv[sc][k][t] = 1 / max(|y - pred|, 1e-3)^2 (0 for t < lowerN is handled by the consumers)
// mode 1: OUT[s][k][c][t] = y - pred for t >= lowerN (apply-time ring rule), y otherwise
__global__ __launch_bounds__(256)
void wpe_predict_kernel(const float2* __restrict__ X, const float2* __restrict__ G, WpeGeom g, int mode,
                        float* __restrict__ Winv, float2* __restrict__ OUT)
{
  const int k = blockIdx.y, sc = blockIdx.z;
  const int s = sc / g.C, c = sc % g.C;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= g.T) return;
  const float2* Xk = X + ((long)s * g.K + k) * g.C * g.T_stride;
  const float2 y = Xk[(long)c * g.T_stride + t];
  float pr = 0.f, pi = 0.f;
  const bool active = bin_active(g, k) && t >= g.lowerN;
  if (active) {
    const float2* gk = G + (((long)s * g.C + c) * g.K + k) * (long)(g.C * g.L);
    const long newest = t - g.lowerN;
    // apply-time ring holds the last L frames only (dereverberation.cc:471-480); estimation sees all history
    const long first = (mode == 1) ? ((t + 1 > g.L) ? t + 1 - g.L : 0) : 0;
    for (int cc = 0; cc < g.C; cc++) {
      const float2* xr = Xk + (long)cc * g.T_stride;
      for (int l = 0; l < g.L; l++) {
        const long idx = newest - l;
        if (idx < first) break;
        const float2 gv = gk[cc * g.L + l];
        const float2 v = xr[idx];
        pr = fmaf(gv.x, v.x, fmaf(gv.y, v.y, pr));              // conj(g) * x
        pi = fmaf(gv.x, v.y, fmaf(-gv.y, v.x, pi));
      }
    }
  }
  const float dr = y.x - pr, di = y.y - pi;
  if (mode == 0) {
    float th = sqrtf(dr * dr + di * di);
    if (th < 1.0e-3f) th = 1.0e-3f;                               // subband_floor_
    Winv[((long)sc * g.K + k) * g.T_stride + t] = 1.0f / (th * th);
  } else {
    OUT[(((long)s * g.K + k) * g.C + c) * g.T_stride + t] = make_float2(dr, di);
  }
}


// grid: (lower-triangle tile pairs, K, S*C/CB).  The lag matrix is the same for all target channels -- only the weights
// 1/theta_c differ -- so one workgroup accumulates the tiles of CB channels from ONE staging of the lag rows.
// Staging: row (c', l) of the lag matrix is the snapshot row of channel c' shifted by l, so a 64-row tile x WT_ frames
// only touches a (WT_ + L - 1)-sample span of at most 64/L + 2 channels.  The spans go to LDS (a few hundred
// elements instead of 64 x WT_ per tile) and every lane reads its row through its own shift -- consecutive lags
// are consecutive addresses, i.e. conflict-free.
template <int CB>
__global__ __launch_bounds__(256)
void wpe_herk_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, int ntile,
                     float2* __restrict__ R, int skip_unused, int nspan)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int SPW = WT_ + g.L - 1;                                   // samples per channel span
  float2* spanI = reinterpret_cast<float2*>(smem);                 // [nspan][SPW]
  float2* spanJ = spanI + nspan * SPW;
  float* wrow = reinterpret_cast<float*>(spanJ + nspan * SPW);     // [CB][WT_]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.y;
  if (!bin_active(g, k)) return;
  const int ncg = g.C / CB;
  const int s = blockIdx.z / ncg, c0 = (blockIdx.z % ncg) * CB;
  const int P = g.C * g.L;
  // blockIdx.x -> (ti >= tj)
  int ti = 0, rem = blockIdx.x;
  while (rem > ti) { rem -= ti + 1; ti++; }
  const int tj = rem;
  const float2* Xk = X + ((long)s * g.K + k) * g.C * g.T_stride;
  const int qi = wave >> 1, qj = wave & 1;
  f32x16 rr[CB], ri[CB];
#pragma unroll
  for (int cb = 0; cb < CB; cb++) { rr[cb] = f32x16{0}; ri[cb] = f32x16{0}; }
  const int li = lane & 31, lk = lane >> 5;
  // a wavefront owns one 32x32 quadrant; quadrants strictly above the diagonal or beyond P are never read by the
  // solver -> their MFMAs are skipped (P = 160: 15 of 24 quadrants remain, P = 264: 45 of 60)
  const int bi = ti * 2 + qi, bj = tj * 2 + qj;
  const bool quad_active = !skip_unused || (bi >= bj && bi * 32 < P && bj * 32 < P);
  // this lane's rows of the two tiles -> (channel, lag) -> offset into the spans
  const int chI0 = (ti * 64) / g.L, chJ0 = (tj * 64) / g.L;
  const int pI = ti * 64 + qi * 32 + li, pJ = tj * 64 + qj * 32 + li;
  const bool vI = pI < P, vJ = pJ < P;
  const int offI = vI ? (pI / g.L - chI0) * SPW + (g.L - 1 - pI % g.L) : 0;
  const int offJ = vJ ? (pJ / g.L - chJ0) * SPW + (g.L - 1 - pJ % g.L) : 0;
  // register prefetch of the next tile's spans and weights (the global loads fly under the MFMAs of the current tile);
  // geometries whose spans exceed WPF elements per thread (very short filters) stage directly
  constexpr int WPF = 6;
  const int nelem = 2 * nspan * SPW;
  const bool use_pf = nelem <= WPF * 256;
  float2 pf[WPF];
  float wpf = 0.f;
  auto span_load = [&](int idx, long t0) -> float2 {
    const int which = idx / (nspan * SPW), e = idx % (nspan * SPW);
    const int ch = (which ? chJ0 : chI0) + e / SPW;
    const long i = t0 - g.lowerN - (g.L - 1) + e % SPW;          // span element j holds sample i (zero outside [0, T))
    return (ch < g.C && i >= 0 && i < g.T) ? Xk[(long)ch * g.T_stride + i] : make_float2(0.f, 0.f);
  };
  auto weight_load = [&](long t0) -> float {
    const int cb = tid / WT_, tt = tid % WT_;
    const long t = t0 + tt;
    const float* w = Winv + (((long)s * g.C + c0 + cb) * g.K + k) * g.T_stride;
    return (t < g.T && t >= g.lowerN) ? w[t] : 0.f;
  };
  auto prefetch = [&](long t0) {
#pragma unroll
    for (int q = 0; q < WPF; q++) { const int idx = tid + q * 256; if (idx < nelem) pf[q] = span_load(idx, t0); }
    if (tid < CB * WT_) wpf = weight_load(t0);
  };
  if (use_pf) prefetch(0);
  for (long t0 = 0; t0 < g.T; t0 += WT_) {
    __syncthreads();
    if (use_pf) {
#pragma unroll
      for (int q = 0; q < WPF; q++) { const int idx = tid + q * 256; if (idx < nelem) spanI[idx] = pf[q]; }
      if (tid < CB * WT_) wrow[tid] = wpf;
    } else {
      for (int idx = tid; idx < nelem; idx += 256) spanI[idx] = span_load(idx, t0);
      if (tid < CB * WT_) wrow[tid] = weight_load(t0);
    }
    __syncthreads();
    if (use_pf && t0 + WT_ < g.T) prefetch(t0 + WT_);
    if (!quad_active) continue;
#pragma unroll 2
    for (int kk = 0; kk < WT_; kk += 2) {
      float2 a = spanI[offI + kk + lk];
      float2 b = spanJ[offJ + kk + lk];
      if (!vI) a = make_float2(0.f, 0.f);
      if (!vJ) b = make_float2(0.f, 0.f);
#pragma unroll
      for (int cb = 0; cb < CB; cb++) {
        const float wv = wrow[cb * WT_ + kk + lk];
        const float bx = b.x * wv, by = b.y * wv;
        rr[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bx, rr[cb], 0, 0, 0);
        rr[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, by, rr[cb], 0, 0, 0);
        ri[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bx, ri[cb], 0, 0, 0);
        ri[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.x, by, ri[cb], 0, 0, 0);
      }
    }
  }
  if (!quad_active) return;
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    float2* Rk = R + (((long)s * g.C + c0 + cb) * g.K + k) * (long)P * P;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row = ti * 64 + qi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int col = tj * 64 + qj * 32 + (lane & 31);
      if (row < P && col < P) Rk[(long)row * P + col] = make_float2(rr[cb][reg], ri[cb][reg]);
    }
  }
}

// The same HERK with ONE wavefront per workgroup and one 32 x 32 block of the lower triangle per wavefront.  In the 64 x 64-tile form
// a quarter of the wavefronts own a block the solver never reads (above the diagonal, or beyond P) and idle on their SIMD for the
// whole launch (P = 264: 15 of 60; the MFMA pipes measured 51 % busy); here every launched wavefront multiplies.  The staging is
// per wavefront (its two 32-row spans: <= 31/L + 2 channels each), the tile loop needs no workgroup barrier.
template <int CB>
__global__ __launch_bounds__(64)
void wpe_herk32_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ R, int nspan)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int SPW = WT_ + g.L - 1;                                   // samples per channel span
  float2* spanI = reinterpret_cast<float2*>(smem);                 // [nspan][SPW]
  float2* spanJ = spanI + nspan * SPW;
  float* wrow = reinterpret_cast<float*>(spanJ + nspan * SPW);     // [CB][WT_]
  const int lane = threadIdx.x;
  const int k = blockIdx.y;
  if (!bin_active(g, k)) return;
  const int ncg = g.C / CB;
  const int s = blockIdx.z / ncg, c0 = (blockIdx.z % ncg) * CB;
  const int P = g.C * g.L;
  int bi = 0, rem = blockIdx.x;                                    // blockIdx.x -> (bi >= bj)
  while (rem > bi) { rem -= bi + 1; bi++; }
  const int bj = rem;
  const float2* Xk = X + ((long)s * g.K + k) * g.C * g.T_stride;
  f32x16 rr[CB], ri[CB];
#pragma unroll
  for (int cb = 0; cb < CB; cb++) { rr[cb] = f32x16{0}; ri[cb] = f32x16{0}; }
  const int li = lane & 31, lk = lane >> 5;
  const int chI0 = (bi * 32) / g.L, chJ0 = (bj * 32) / g.L;
  const int pI = bi * 32 + li, pJ = bj * 32 + li;
  const bool vI = pI < P, vJ = pJ < P;
  const int offI = vI ? (pI / g.L - chI0) * SPW + (g.L - 1 - pI % g.L) : 0;
  const int offJ = vJ ? (pJ / g.L - chJ0) * SPW + (g.L - 1 - pJ % g.L) : 0;
  constexpr int WPF = 12;                                           // span elements a lane prefetches (L >= 6 at WT_ = 32)
  const int nelem = 2 * nspan * SPW;
  const bool use_pf = nelem <= WPF * 64;
  float2 pf[WPF];
  float wpf[(CB * WT_ + 63) / 64];
  auto span_load = [&](int idx, long t0) -> float2 {
    const int which = idx / (nspan * SPW), e = idx % (nspan * SPW);
    const int ch = (which ? chJ0 : chI0) + e / SPW;
    const long i = t0 - g.lowerN - (g.L - 1) + e % SPW;          // span element j holds sample i (zero outside [0, T))
    return (ch < g.C && i >= 0 && i < g.T) ? Xk[(long)ch * g.T_stride + i] : make_float2(0.f, 0.f);
  };
  auto weight_load = [&](int idx, long t0) -> float {
    const int cb = idx / WT_, tt = idx % WT_;
    const long t = t0 + tt;
    const float* w = Winv + (((long)s * g.C + c0 + cb) * g.K + k) * g.T_stride;
    return (t < g.T && t >= g.lowerN) ? w[t] : 0.f;
  };
  auto prefetch = [&](long t0) {
#pragma unroll
    for (int q = 0; q < WPF; q++) { const int idx = lane + q * 64; if (idx < nelem) pf[q] = span_load(idx, t0); }
#pragma unroll
    for (int q = 0; q < (CB * WT_ + 63) / 64; q++) { const int idx = lane + q * 64; if (idx < CB * WT_) wpf[q] = weight_load(idx, t0); }
  };
  if (use_pf) prefetch(0);
  for (long t0 = 0; t0 < g.T; t0 += WT_) {
    __syncthreads();                                               // (one wavefront: orders the LDS reads of the last tile before the writes)
    if (use_pf) {
#pragma unroll
      for (int q = 0; q < WPF; q++) { const int idx = lane + q * 64; if (idx < nelem) spanI[idx] = pf[q]; }
#pragma unroll
      for (int q = 0; q < (CB * WT_ + 63) / 64; q++) { const int idx = lane + q * 64; if (idx < CB * WT_) wrow[idx] = wpf[q]; }
    } else {
      for (int idx = lane; idx < nelem; idx += 64) spanI[idx] = span_load(idx, t0);
      for (int idx = lane; idx < CB * WT_; idx += 64) wrow[idx] = weight_load(idx, t0);
    }
    __syncthreads();
    if (use_pf && t0 + WT_ < g.T) prefetch(t0 + WT_);
#pragma unroll 2
    for (int kk = 0; kk < WT_; kk += 2) {
      float2 a = spanI[offI + kk + lk];
      float2 b = spanJ[offJ + kk + lk];
      if (!vI) a = make_float2(0.f, 0.f);
      if (!vJ) b = make_float2(0.f, 0.f);
#pragma unroll
      for (int cb = 0; cb < CB; cb++) {
        const float wv = wrow[cb * WT_ + kk + lk];
        const float bx = b.x * wv, by = b.y * wv;
        rr[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bx, rr[cb], 0, 0, 0);
        rr[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, by, rr[cb], 0, 0, 0);
        ri[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bx, ri[cb], 0, 0, 0);
        ri[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.x, by, ri[cb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    float2* Rk = R + (((long)s * g.C + c0 + cb) * g.K + k) * (long)P * P;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row = bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int col = bj * 32 + (lane & 31);
      if (row < P && col < P) Rk[(long)row * P + col] = make_float2(rr[cb][reg], ri[cb][reg]);
    }
  }
}

// The rows beyond the last full 32-row block (P = 264: rows 256..263) as a 16-row strip on v_mfma_f32_16x16x4_f32: padding them to
// a ninth 32-row block row costs 9 blocks of which a quarter is used (15 % of the launch's MFMA work); a strip of 16 x 16 tiles
// costs 4.25 block equivalents.  One wavefront owns the strip rows against 64 columns (four tiles): A[i][k] comes from lane
// i + 16 k, B[k][j] from lane j + 16 k (k = four consecutive frames), D[i][j] lands in register v of lane l with i = 4 (l / 16) + v,
// j = l % 16.  Same staging and arithmetic as wpe_herk32_kernel.
template <int CB>
__global__ __launch_bounds__(64)
void wpe_herk16_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ R, int nspan, int row0)
{
  constexpr int CT = 4;                                            // 16-column tiles per wavefront
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int SPW = WT_ + g.L - 1;
  float2* spanI = reinterpret_cast<float2*>(smem);                 // [nspan][SPW]
  float2* spanJ = spanI + nspan * SPW;
  float* wrow = reinterpret_cast<float*>(spanJ + nspan * SPW);     // [CB][WT_]
  const int lane = threadIdx.x;
  const int k = blockIdx.y;
  if (!bin_active(g, k)) return;
  const int ncg = g.C / CB;
  const int s = blockIdx.z / ncg, c0 = (blockIdx.z % ncg) * CB;
  const int P = g.C * g.L;
  const int col0 = blockIdx.x * (16 * CT);
  const float2* Xk = X + ((long)s * g.K + k) * g.C * g.T_stride;
  f32x4 rr[CT][CB], ri[CT][CB];
#pragma unroll
  for (int ct = 0; ct < CT; ct++)
#pragma unroll
    for (int cb = 0; cb < CB; cb++) { rr[ct][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; ri[ct][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int li = lane & 15, lk = lane >> 4;
  const int chI0 = row0 / g.L, chJ0 = col0 / g.L;
  const int pI = row0 + li;
  const bool vI = pI < P;
  const int offI = vI ? (pI / g.L - chI0) * SPW + (g.L - 1 - pI % g.L) : 0;
  int offJ[CT]; bool vJ[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    const int pJ = col0 + 16 * ct + li;
    vJ[ct] = pJ < P;
    offJ[ct] = vJ[ct] ? (pJ / g.L - chJ0) * SPW + (g.L - 1 - pJ % g.L) : 0;
  }
  const int nelem = 2 * nspan * SPW;
  auto span_load = [&](int idx, long t0) -> float2 {
    const int which = idx / (nspan * SPW), e = idx % (nspan * SPW);
    const int ch = (which ? chJ0 : chI0) + e / SPW;
    const long i = t0 - g.lowerN - (g.L - 1) + e % SPW;
    return (ch < g.C && i >= 0 && i < g.T) ? Xk[(long)ch * g.T_stride + i] : make_float2(0.f, 0.f);
  };
  auto weight_load = [&](int idx, long t0) -> float {
    const int cb = idx / WT_, tt = idx % WT_;
    const long t = t0 + tt;
    const float* w = Winv + (((long)s * g.C + c0 + cb) * g.K + k) * g.T_stride;
    return (t < g.T && t >= g.lowerN) ? w[t] : 0.f;
  };
  for (long t0 = 0; t0 < g.T; t0 += WT_) {
    __syncthreads();
    for (int idx = lane; idx < nelem; idx += 64) spanI[idx] = span_load(idx, t0);
    for (int idx = lane; idx < CB * WT_; idx += 64) wrow[idx] = weight_load(idx, t0);
    __syncthreads();
#pragma unroll 2
    for (int kk = 0; kk < WT_; kk += 4) {
      float2 a = spanI[offI + kk + lk];
      if (!vI) a = make_float2(0.f, 0.f);
      float wv[CB];
#pragma unroll
      for (int cb = 0; cb < CB; cb++) wv[cb] = wrow[cb * WT_ + kk + lk];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) {
        float2 b = spanJ[offJ[ct] + kk + lk];
        if (!vJ[ct]) b = make_float2(0.f, 0.f);
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
          const float bx = b.x * wv[cb], by = b.y * wv[cb];
          rr[ct][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bx, rr[ct][cb], 0, 0, 0);
          rr[ct][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, by, rr[ct][cb], 0, 0, 0);
          ri[ct][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bx, ri[ct][cb], 0, 0, 0);
          ri[ct][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.x, by, ri[ct][cb], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    float2* Rk = R + (((long)s * g.C + c0 + cb) * g.K + k) * (long)P * P;
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int row = row0 + 4 * lk + v;
        const int col = col0 + 16 * ct + li;
        if (row < P && col < P) Rk[(long)row * P + col] = make_float2(rr[ct][cb][v], ri[ct][cb][v]);
      }
  }
}

// r_c[p] = sum_t conj(y_c(t)) ybar_p(t) / theta_c(t)
__global__ __launch_bounds__(256)
void wpe_rvec_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ rvec)
{
  const int k = blockIdx.x, sc = blockIdx.y;
  if (!bin_active(g, k)) return;
  const int s = sc / g.C, c = sc % g.C;
  const int P = g.C * g.L;
  const float2* Xk = X + ((long)s * g.K + k) * g.C * g.T_stride;
  const float2* yc = Xk + (long)c * g.T_stride;
  const float* w = Winv + ((long)sc * g.K + k) * g.T_stride;
  for (int p = threadIdx.x; p < P; p += 256) {
    const int cc = p / g.L, l = p % g.L;
    const float2* xr = Xk + (long)cc * g.T_stride;
    float ar = 0.f, ai = 0.f;
    for (long t = g.lowerN + l; t < g.T; t++) {
      const float2 y = yc[t];
      const float2 v = xr[t - g.lowerN - l];
      const float wv = w[t];
      ar = fmaf(wv, y.x * v.x + y.y * v.y, ar);                   // conj(y) * v
      ai = fmaf(wv, y.x * v.y - y.y * v.x, ai);
    }
    rvec[((long)sc * g.K + k) * P + p] = make_float2(ar, ai);
  }
}

// One workgroup per (sc,k): diagonal bias + loading, then the blocked Cholesky solve of chol_blocked.h.
__global__ __launch_bounds__(256)
void wpe_solve_kernel(float2* __restrict__ R, const float2* __restrict__ rvec, WpeGeom g, float load_factor,
                      float diagonal_bias, float2* __restrict__ G, int* __restrict__ fail_count,
                      unsigned long long* __restrict__ phase_cycles /* BTK_WPE_TIMING diagnostics, else null */)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, sc = blockIdx.y;
  if (!bin_active(g, k)) return;
  const int P = g.C * g.L;
  const int tid = threadIdx.x;
  const int Ppad = (P + 1) & ~1;
  float2* rhs = reinterpret_cast<float2*>(smem);                   // [Ppad]
  float* red = reinterpret_cast<float*>(rhs + Ppad);               // [512]
  float2* panel = reinterpret_cast<float2*>(red + 512);            // [P][cholb::CH_LD]
  float2* mat = R + ((long)sc * g.K + k) * (long)P * P;
  const int s = sc / g.C, c = sc % g.C;
  float2* gout = G + (((long)s * g.C + c) * g.K + k) * (long)P;
  if (rvec) {
    for (int p = tid; p < P; p += 256) rhs[p] = rvec[((long)sc * g.K + k) * P + p];
  } else {
    // lowerN == 0: y_c(t) is itself a row of the lag matrix (channel c, lag 0), so r_c[p] = sum_t ybar_p conj(y_c) / theta_c is column
    // q = c L of the matrix the HERK just produced (lower triangle stored: R[p][q] for p >= q, conj(R[q][p]) above) -- read
    // before the diagonal is biased and loaded
    const int q = c * g.L;
    for (int p = tid; p < P; p += 256) {
      const float2 v = (p >= q) ? mat[(long)p * P + q] : mat[(long)q * P + p];
      rhs[p] = (p >= q) ? v : make_float2(v.x, -v.y);
    }
    __syncthreads();
  }
  // diagonal bias (dereverberation.cc:574-577) then load_R_ (:648-663)
  float mx = 0.f;
  for (int p = tid; p < P; p += 256) {
    float2 d = mat[(long)p * P + p];
    d.x += diagonal_bias;
    const float a = sqrtf(d.x * d.x + d.y * d.y);
    mat[(long)p * P + p] = make_float2(a, 0.f);
    mx = fmaxf(mx, a);
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
  const float load = red[0] * load_factor;
  __syncthreads();
  for (int p = tid; p < P; p += 256) mat[(long)p * P + p].x += load;
  __syncthreads();

  const int wrot = blockIdx.x + blockIdx.y;
  long long tm[7] = {0, 0, 0, 0, 0, 0, 0};
  long long tlast = phase_cycles ? clock64() : 0;
  auto mark = [&](int i) { if (phase_cycles) { const long long c = clock64(); tm[i] += c - tlast; tlast = c; } };
  if (!cholb::solve(mat, P, rhs, red, panel, 0.f, wrot, mark)) { if (tid == 0) atomicAdd(fail_count, 1); return; }
  for (int p = tid; p < P; p += 256) gout[p] = rhs[p];
  mark(5);                                                           // back substitution
  if (phase_cycles && tid == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(phase_cycles + i, (unsigned long long)tm[i]);
    atomicAdd(phase_cycles + 6, 1ull);
  }
}

WpeGeom make_geom(int K, int C, int lowerN, int upperN, int lower_bw, int upper_bw, long T_stride, long T)
{
  WpeGeom g;
  g.K = K; g.C = C; g.L = upperN - lowerN + 1; g.lowerN = lowerN; g.lower_bw = lower_bw; g.upper_bw = upper_bw;
  g.T_stride = T_stride; g.T = T;
  return g;
}

}  // namespace

extern "C" {

long btk_wpe_workspace_bytes(int S, int K, int C, int lowerN, int upperN, long T_stride)
{
  const long P = (long)C * (upperN - lowerN + 1);
  const long nb = (long)S * C * K;
  return nb * P * P * 8 + nb * P * 8 + nb * T_stride * 4 + 256;
}

int btk_wpe_estimate(const void* X, int S, int K, int C, long T_stride, long T, int lowerN, int upperN, int iterations,
                     double load_db, double diagonal_bias, int lower_bw, int upper_bw, void* G, void* workspace,
                     int* fail_count, void* stream)
{
  if (!X || !G || !workspace || !fail_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_wpe_estimate: null argument");
  if (S <= 0 || K <= 0 || C <= 0 || T < 0 || T_stride < T || upperN < lowerN || lowerN < 0)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_wpe_estimate: bad sizes S=%d K=%d C=%d T=%ld lags %d..%d", S, K, C, T, lowerN, upperN);
  if (T == 0) return BTK_OK;
  const WpeGeom g = make_geom(K, C, lowerN, upperN, lower_bw, upper_bw, T_stride, T);
  const long P = (long)C * g.L;
  const long nb = (long)S * C * K;
  hipStream_t st = as_stream(stream);
  float2* R = static_cast<float2*>(workspace);
  float2* rvec = R + nb * P * P;
  float* Winv = reinterpret_cast<float*>(rvec + nb * P);
  const float2* Xp = static_cast<const float2*>(X);
  float2* Gp = static_cast<float2*>(G);
  const int ntile = (int)((P + 63) / 64);
  const int nspan = 63 / g.L + 2;                                  // channels a 64-row tile can touch
  const size_t lds_herk = sizeof(float2) * 2 * (size_t)nspan * (WT_ + g.L - 1) + sizeof(float) * 4 * WT_;
  const float load_factor = (float)pow(10.0, load_db / 10.0);
  const size_t lds_solve = sizeof(float2) * ((P + 1) & ~1L) + sizeof(float) * 512 + sizeof(float2) * (size_t)P * cholb::CH_LD;
  if (lds_solve > 160 * 1024)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_wpe_estimate: C*(upperN-lowerN+1) = %ld taps exceed the LDS-resident solver (max ~560)", P);
  if (lds_solve > 64 * 1024)
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wpe_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
  unsigned long long* phase = nullptr;                              // BTK_WPE_TIMING=1: shader cycles per phase of the solver, printed per call
  if (btk_switches().wpe_timing) {
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) BTK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbuf), 8 * sizeof(unsigned long long)));
    BTK_HIP_CHECK(hipMemsetAsync(dbuf, 0, 8 * sizeof(unsigned long long), st));
    phase = dbuf;
  }
  for (int it = 0; it < iterations; it++) {
    hipLaunchKernelGGL(wpe_predict_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)K, (unsigned)(S * C)), dim3(256), 0, st,
                       Xp, Gp, g, 0, Winv, static_cast<float2*>(nullptr));
    const int skip = btk_switches().wpe_noskip ? 0 : 1;            // A/B switch of profiles/ (btk_internal.h)
    const dim3 hgrid1((unsigned)(ntile * (ntile + 1) / 2), (unsigned)K, (unsigned)(S * C));
    // up to 16 rows beyond the last full 32-row block go to the 16-row strip kernel instead of a padded block row
    const int rem = (int)(P % 32);
    const bool strip = rem > 0 && rem <= 16 && P >= 32;
    const int nb32 = strip ? (int)(P / 32) : (int)((P + 31) / 32);
    const unsigned nblk = (unsigned)(nb32 * (nb32 + 1) / 2);          // 32 x 32 blocks of the lower triangle
    const int nspan32 = 31 / g.L + 2, nspan64 = 63 / g.L + 2;
    const size_t lds32 = sizeof(float2) * 2 * (size_t)nspan32 * (WT_ + g.L - 1) + sizeof(float) * 4 * WT_;
    const size_t lds16 = sizeof(float2) * 2 * (size_t)nspan64 * (WT_ + g.L - 1) + sizeof(float) * 4 * WT_;
    const unsigned nstrip = (unsigned)((P + 63) / 64);
    if (skip && C % 4 == 0) {
      hipLaunchKernelGGL(wpe_herk32_kernel<4>, dim3(nblk, (unsigned)K, (unsigned)(S * C / 4)), dim3(64), lds32, st, Xp, Winv, g, R, nspan32);
      if (strip) hipLaunchKernelGGL(wpe_herk16_kernel<4>, dim3(nstrip, (unsigned)K, (unsigned)(S * C / 4)), dim3(64), lds16, st, Xp, Winv, g, R, nspan64, nb32 * 32);
    } else if (skip && C % 2 == 0) {
      hipLaunchKernelGGL(wpe_herk32_kernel<2>, dim3(nblk, (unsigned)K, (unsigned)(S * C / 2)), dim3(64), lds32, st, Xp, Winv, g, R, nspan32);
      if (strip) hipLaunchKernelGGL(wpe_herk16_kernel<2>, dim3(nstrip, (unsigned)K, (unsigned)(S * C / 2)), dim3(64), lds16, st, Xp, Winv, g, R, nspan64, nb32 * 32);
    } else if (skip) {
      hipLaunchKernelGGL(wpe_herk32_kernel<1>, dim3(nblk, (unsigned)K, (unsigned)(S * C)), dim3(64), lds32, st, Xp, Winv, g, R, nspan32);
      if (strip) hipLaunchKernelGGL(wpe_herk16_kernel<1>, dim3(nstrip, (unsigned)K, (unsigned)(S * C)), dim3(64), lds16, st, Xp, Winv, g, R, nspan64, nb32 * 32);
    }
    else if (C % 4 == 0)
      hipLaunchKernelGGL(wpe_herk_kernel<4>, dim3(hgrid1.x, hgrid1.y, (unsigned)(S * C / 4)), dim3(256), lds_herk, st, Xp, Winv, g, ntile, R, skip, nspan);
    else if (C % 2 == 0)
      hipLaunchKernelGGL(wpe_herk_kernel<2>, dim3(hgrid1.x, hgrid1.y, (unsigned)(S * C / 2)), dim3(256), lds_herk, st, Xp, Winv, g, ntile, R, skip, nspan);
    else
      hipLaunchKernelGGL(wpe_herk_kernel<1>, hgrid1, dim3(256), lds_herk, st, Xp, Winv, g, ntile, R, skip, nspan);
    const bool rvec_from_R = (lowerN == 0) && skip;            // the lag-0 row of the target channel is y_c itself
    if (!rvec_from_R) hipLaunchKernelGGL(wpe_rvec_kernel, dim3((unsigned)K, (unsigned)(S * C)), dim3(256), 0, st, Xp, Winv, g, rvec);
    hipLaunchKernelGGL(wpe_solve_kernel, dim3((unsigned)K, (unsigned)(S * C)), dim3(256), lds_solve, st,
                       R, rvec_from_R ? static_cast<const float2*>(nullptr) : rvec, g, load_factor, (float)diagonal_bias, Gp, fail_count, phase);
    BTK_HIP_CHECK(hipGetLastError());
  }
  if (phase) {
    unsigned long long h[8];
    BTK_HIP_CHECK(hipMemcpyAsync(h, phase, sizeof(h), hipMemcpyDeviceToHost, st));
    BTK_HIP_CHECK(hipStreamSynchronize(st));
    static const char* names[6] = {"panel_load", "mfma_update", "diagonal_block", "row_solves", "writeback_forward", "back_substitution"};
    double tot = 0; for (int i = 0; i < 6; i++) tot += (double)h[i];
    fprintf(stderr, "wpe_solve phases (%llu systems, P = %ld): %.0f cycles per system:", h[6], P, tot / (h[6] ? h[6] : 1));
    for (int i = 0; i < 6; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * h[i] / (tot > 0 ? tot : 1));
    fprintf(stderr, "\n");
  }
  return BTK_OK;
}

int btk_wpe_apply(const void* X, const void* G, void* OUT, int S, int K, int C, long T_stride, long T,
                  int lowerN, int upperN, int lower_bw, int upper_bw, void* stream)
{
  if (!X || !G || !OUT) return btk_set_error(BTK_ERR_PARAMETER, "btk_wpe_apply: null argument");
  if (S <= 0 || K <= 0 || C <= 0 || T < 0 || T_stride < T || upperN < lowerN)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_wpe_apply: bad sizes");
  if (T == 0) return BTK_OK;
  const WpeGeom g = make_geom(K, C, lowerN, upperN, lower_bw, upper_bw, T_stride, T);
  hipLaunchKernelGGL(wpe_predict_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)K, (unsigned)(S * C)), dim3(256), 0,
                     as_stream(stream), static_cast<const float2*>(X), static_cast<const float2*>(G), g, 1,
                     static_cast<float*>(nullptr), static_cast<float2*>(OUT));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // extern "C"
