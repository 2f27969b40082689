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
#include "chol_reg.h"
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


// ---- round 3: the prediction y_c(t) - g_c^H ybar(t) on the matrix cores (C >= 4 channels).  Per stream and bin it is a complex GEMM
// [C x P] . [P x T]: the C target channels share the lag matrix.  The vector kernel above spends 264 dependent multiply-adds per
// output (15 TFLOP/s, 4.7 ms per call at 16 streams -- twice per estimate and once per apply).  Here a workgroup of four wavefronts owns
// 256 frames of one (stream, bin): conj(G) [C][P] and the C channel spans are staged in LDS once, a wavefront multiplies its 64 frames
// (four 16 x 16 tiles) on v_mfma_f32_16x16x4_f32: rows = target channel (C <= 16 of 16 used), columns = frames, four taps per step.
// A[i][k] comes from lane i + 16 k (G), B[k][j] from lane j + 16 k (the sample y_c'(t_j - lowerN - l) of tap p = 4 q + k = c' L + l),
// D[i][j] lands in register v of lane l with i = 4 (l / 16) + v, j = l % 16.
constexpr int PM_FW = 64, PM_FG = 4 * PM_FW;                       // frames per wavefront / workgroup
__global__ __launch_bounds__(256)
void wpe_predict_mfma_kernel(const float2* __restrict__ X, const float2* __restrict__ G, WpeGeom g, int mode,
                             float* __restrict__ Winv, float2* __restrict__ OUT, int g_ld, int sp_ld)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* gs = reinterpret_cast<float2*>(smem);                    // [C][g_ld]: conj-ready filter taps (zero where a tap does not apply)
  float2* sp = gs + g.C * g_ld;                                    // [C][sp_ld]: samples t0 - lowerN - (L - 1) .. t0 + PM_FG - 1 of every channel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.y, s = blockIdx.z;
  const long t0 = (long)blockIdx.x * PM_FG;
  const int C = g.C, L = g.L, P = C * L;
  const float2* Xk = X + ((long)s * g.K + k) * C * g.T_stride;
  const bool active = bin_active(g, k);
  // apply-time ring (dereverberation.cc:471-480): only the last L frames exist, i.e. taps l > L - 1 - lowerN never apply
  const int lmax = (mode == 1) ? L - 1 - g.lowerN : L - 1;
  for (int idx = tid; idx < C * g_ld; idx += 256) {
    const int c = idx / g_ld, p = idx - c * g_ld;
    float2 v = make_float2(0.f, 0.f);
    if (active && p < P && (p % L) <= lmax) v = G[(((long)s * C + c) * g.K + k) * (long)P + p];
    gs[idx] = v;
  }
  const int SPW = PM_FG + L - 1 + g.lowerN;                        // ... up to sample t0 + PM_FG - 1: y_c(t) itself is read from the span too
  for (int idx = tid; idx < C * SPW; idx += 256) {
    const int c = idx / SPW, e = idx - c * SPW;
    const long i = t0 - g.lowerN - (L - 1) + e;
    sp[c * sp_ld + e] = (i >= 0 && i < g.T) ? Xk[(long)c * g.T_stride + i] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  const int mi = lane & 15, mk = lane >> 4;
  f32x4 dr[4], di[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++) { dr[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; di[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // tap of this lane group in step q: p = 4 q + mk = cp L + l
  int cp = mk / L, l = mk % L;
  const int fbase = wave * PM_FW + mi + (L - 1);                   // span index of frame (wave, tile 0, column mi) at lag 0
  const int nstep = (P + 3) / 4;
  for (int q = 0; q < nstep; q++) {
    const float2 gv = (mi < C && 4 * q + mk < P) ? gs[mi * g_ld + 4 * q + mk] : make_float2(0.f, 0.f);
    const float ngi = -gv.y;
    const bool pv = cp < C;
    const float2* row = sp + (pv ? cp : 0) * sp_ld + fbase - l;
#pragma unroll
    for (int cb = 0; cb < 4; cb++) {
      float2 y = row[16 * cb];
      if (!pv) y = make_float2(0.f, 0.f);
      // conj(g) y = (gr yr + gi yi) + i (gr yi - gi yr)
      dr[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, y.x, dr[cb], 0, 0, 0);
      dr[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, y.y, dr[cb], 0, 0, 0);
      di[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, y.y, di[cb], 0, 0, 0);
      di[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ngi, y.x, di[cb], 0, 0, 0);
    }
    l += 4;
    while (l >= L) { l -= L; cp++; }
  }
  // ---- epilogue: rows c = 4 mk + v < C, frame t = t0 + 64 wave + 16 cb + mi
#pragma unroll
  for (int cb = 0; cb < 4; cb++) {
    const long t = t0 + wave * PM_FW + 16 * cb + mi;
    if (t >= g.T) continue;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int c = 4 * mk + v;
      if (c >= C) continue;
      const float2 y = sp[c * sp_ld + wave * PM_FW + 16 * cb + mi + (L - 1) + g.lowerN];       // y_c(t): the span starts lowerN + L - 1 before t0
      const bool pred = t >= g.lowerN;
      const float drr = y.x - (pred ? dr[cb][v] : 0.f), dii = y.y - (pred ? di[cb][v] : 0.f);
      if (mode == 0) {
        float th = sqrtf(drr * drr + dii * dii);
        if (th < 1.0e-3f) th = 1.0e-3f;                               // subband_floor_
        Winv[(((long)s * C + c) * g.K + k) * g.T_stride + t] = 1.0f / (th * th);
      } else {
        OUT[(((long)s * g.K + k) * C + c) * g.T_stride + t] = make_float2(drr, dii);
      }
    }
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

// ---- round 3: the normal equations as LAG PRODUCTS.  Entry [(c1,l1)][(c2,l2)] of R_c is, with u = t - lowerN - l1 and d = l1 - l2,
//   sum_u  w_c(u + lowerN + l1) * [ y_c1(u) conj(y_c2(u + d)) ]:
// the bracket does not depend on the target channel c nor on l1, only on (c1, c2, d).  The block HERK above recomputes that complex
// product for every one of the L - |d| entries of a diagonal and for each of the C target channels (four real matrix instructions per
// complex block product).  Here the products are formed on the vector ALU -- exactly, in fp32 -- and the matrix cores only apply the
// REAL weights: a real GEMM
//   Out[(l1, c)] [(c1, c2, re|im)] = sum_u  Wh[(l1, c)][u] * Pd[u][(c1, c2, re|im)],     Wh[(l1, c)][u] = w_c(u + lowerN + l1)
// per lag difference d >= 0 (entries with d < 0 are the conjugates of the swapped ordered pair, so all C^2 ordered pairs with d >= 0
// cover the lower triangle once; d = 0 is computed for both orders and stored from one).  Rows: 32 / C consecutive l1 x C target
// channels per 32-row block, blocks counted down from l1 = L - 1 so that only the last one is partial; columns: 2 C^2 = 128 at C = 8 =
// four 32-column blocks.  P = 264: 612 block products of ONE real matrix instruction per two frames against 36 x 8 complex block
// products of FOUR (1 152): 0.53 x the matrix-core work, no strip kernel, no wasted upper halves of diagonal blocks.
// One wavefront per workgroup; a task is (d, up to four row blocks): the wavefront holds up to 4 x 4 accumulator blocks (256 AGPRs, one
// wavefront per SIMD) so that every product feeds four matrix instructions -- on this chip a vector instruction between matrix
// instructions is not free: profiles/ubench/mfma_war.hip measures 145 TFLOP/s for v_mfma_f32_32x32x2_f32 alone with ONE wavefront per
// SIMD and 98-107 with two vector instructions per matrix instruction (the one-row-block form of this kernel, 3.5 other instructions per
// matrix instruction, kept the pipe 52 % busy).  Spans of the C snapshot rows and weight rows of a 64-frame tile are staged per
// wavefront (register prefetch one tile ahead), operands of frame group g + 1 are read from LDS before the matrix instructions of g.
constexpr int LP_WT = 64, LP_RMAX = 4;
constexpr long LP16_SEG = 2048;                                  // frames per accumulator flush of the float16 lag-product kernel (multiple of LP_WT)

template <int C, int NR>
__device__ __forceinline__ void lagprod_task(const float2* __restrict__ Xk, const float* __restrict__ Wk, const WpeGeom& g, float2* __restrict__ R,
                                             int ys_ld, int ws_ld, int d, int la, int s, int k, float2* ys, float* ws)
{
  constexpr int NCB = 2 * C * C / 32;                              // 32-column blocks: col = 2 (c1 C + c2) + (0 re | 1 im)
  constexpr int RL = 32 / C;                                       // l1 values per 32-row block: row m of block j -> (l1 = la + RL j + m / C, c = m % C)
  const int lane = threadIdx.x;
  const int L = g.L, P = C * L;
  const int YN = LP_WT + L - 1;                                    // samples per channel span (<= 128: L <= 65)
  constexpr int WN = LP_WT + RL * NR - 1;                          // weights per target-channel span (<= 128)
  // staging: lane e (and e + 64) of every channel row -- no index arithmetic per element (a flat index over the C x YN tile costs an
  // integer division by YN per element and tile: a quarter of the tile's cycles at one wavefront per SIMD)
  float2 ypf[C][2];
  float wpf[C][2];
  auto prefetch = [&](long u0) {
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int e = lane + 64 * q;
        const long u = u0 + e;
        ypf[c][q] = (e < YN && u < g.T) ? Xk[(long)c * g.T_stride + u] : make_float2(0.f, 0.f);
        const long t = u0 + g.lowerN + la + e;
        wpf[c][q] = (e < WN && t >= g.lowerN && t < g.T) ? Wk[(long)c * g.K * g.T_stride + t] : 0.f;
      }
  };
  f32x16 acc[NR][NCB];
#pragma unroll
  for (int j = 0; j < NR; j++)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[j][cb] = f32x16{0};
  const int m = lane & 31, lk = lane >> 5;
  // frames of a 4-frame group: lane half lk multiplies frames 4 g + 2 lk + h in the two matrix instructions h = 0, 1, so that its
  // two operands of a group are ADJACENT in LDS: the first factors come as one 16-byte read
  const int aoff = (m % C) * ws_ld + m / C + 2 * lk;               // A of block j: Wh[(la + RL j + m / C, m % C)][u0 + 4 g + 2 lk + h] at + RL j
  const bool im = lane & 1;
  // B: x = y_c1(u), y = y_c2(u + d), pair = 16 cb + (lane & 31) / 2 = c1 C + c2; 16 % C == 0: c2 = (m / 2) % C for every cb
  int boffx[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) boffx[cb] = (((16 * cb + (m >> 1)) / C) * ys_ld) / 2 + lk;       // in float4 units (ys_ld is even)
  const int boffy = ((m >> 1) % C) * ys_ld + 2 * lk + d;
  static_assert(16 % C == 0, "the second factor's channel must not depend on the column block");
  prefetch(0);
  for (long u0 = 0; u0 < g.T; u0 += LP_WT) {
    __syncthreads();                                               // (one wavefront: orders the LDS reads of the last tile before the writes)
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int e = lane + 64 * q;
        if (e < YN) ys[c * ys_ld + e] = ypf[c][q];
        if (e < WN) ws[c * ws_ld + e] = wpf[c][q];
      }
    __syncthreads();
    if (u0 + LP_WT < g.T) prefetch(u0 + LP_WT);
    struct Ops { float a[NR][2]; float2 y[2]; float2 x[NCB][2]; };
    auto fetch = [&](Ops& o, int kk) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int j = 0; j < NR; j++) o.a[j][h] = ws[aoff + RL * j + kk + h];
        o.y[h] = ys[boffy + kk + h];                             // (odd d: not 16-byte aligned)
      }
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        const float4 x4 = reinterpret_cast<const float4*>(ys)[boffx[cb] + kk / 2];
        o.x[cb][0] = make_float2(x4.x, x4.y); o.x[cb][1] = make_float2(x4.z, x4.w);
      }
    };
    auto mult = [&](const Ops& o) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        // x conj(y): re = x.x y.x + x.y y.y, im = x.y y.x - x.x y.y
        const float p = im ? -o.y[h].y : o.y[h].x, q = im ? o.y[h].x : o.y[h].y;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          const float b = fmaf(o.x[cb][h].x, p, o.x[cb][h].y * q);
#pragma unroll
          for (int j = 0; j < NR; j++) acc[j][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[j][h], b, acc[j][cb], 0, 0, 0);
        }
      }
    };
    Ops o0, o1;
    fetch(o0, 0);
#pragma unroll 1
    for (int kk = 0; kk < LP_WT; kk += 8) {
      fetch(o1, kk + 4);
      __builtin_amdgcn_sched_barrier(0);                            // (the scheduler otherwise sinks the reads to their first use)
      mult(o0);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 8 < LP_WT) fetch(o0, kk + 8);
      __builtin_amdgcn_sched_barrier(0);
      mult(o1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- store: register r of lane -> row (r & 3) + 8 (r >> 2) + 4 lk, column lane & 31.  With p = c1 L + l1, q = c2 L + l1 - d the
  // entry goes to [p][q] when c1 >= c2 and, conjugated, to [q][p] otherwise; both offsets are  const(c1, c2, d) + l1 (P + 1)
  static_assert(C == 8 || C == 4, "row -> (l1, c) below");
  const int c1c2 = m >> 1;
  float* Rc[4];                                                    // matrices of the target channels of this lane's rows
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = (C == 8) ? i + 4 * lk : i;                       // row % C for row = i + 8 (r >> 2) + 4 lk
    Rc[i] = reinterpret_cast<float*>(R + (((long)s * C + c) * g.K + k) * (long)P * P) + (im ? 1 : 0);
  }
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) {
    const int pair = 16 * cb + c1c2, c1 = pair / C, c2 = pair % C;
    const bool lower = c1 >= c2;
    const bool skip = (d == 0 && c1 < c2);                         // the swapped pair stores this entry
    const long off0 = lower ? (long)c1 * L * P + (long)c2 * L - d : ((long)c2 * L - d) * P + (long)c1 * L;
    const float sgn = (!lower && im) ? -1.f : 1.f;
#pragma unroll
    for (int j = 0; j < NR; j++) {
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
        const int l1 = la + RL * j + row / C;                      // C = 8: la + 4 j + (reg >> 2); C = 4: la + 8 j + 2 (reg >> 2) + lk
        if (l1 < d || skip) continue;                              // (also l1 < 0: la < 0 only in the partial block)
        Rc[reg & 3][2 * (off0 + (long)l1 * (P + 1))] = sgn * acc[j][cb][reg];
      }
      __builtin_amdgcn_sched_barrier(0);                            // (one accumulator block at a time out of the AGPRs)
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same lag-product GEMM on the float16 matrix instruction (v_mfma_f32_32x32x16_f16: 16 frames per instruction at 16 x the
// rate of v_mfma_f32_32x32x2_f32) with BOTH operands split into a high and a low float16 part, a = ah + al, b = bh + bl, and three
// products ah bh + ah bl + al bh accumulated in float32: the dropped al bl term and the parts' own rounding are 2^-22 of |a| |b|, i.e.
// the accuracy of the float32 instruction it replaces (one bin, 8 channels x 33 lags, 1000 frames against float64: max |R - R64| /
// max |R64| 1.1e-7 for the split, 1.3e-7 for float32 products; filter taps 4.7e-6 vs 5.3e-6).  float16's range is met by one power-of-two
// scale per (stream, bin) and operand (wpe_lp_scale_kernel: 2^14 / max, exact to undo).  48 matrix instructions of 32 cycles per 16 frames
// of a task instead of 128 of 64.  The Hankel operand Wh[(l1, c)][u] = w_c(u + l1) needs 8 consecutive float16 values starting at ANY
// index: the weight span of a tile is kept in LDS as its eight one-frame shifts, split, so that every read is one aligned 16-byte word.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_hi(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ f16x8 mk8(unsigned a, unsigned b, unsigned c, unsigned d)
{
  const uint4 v = make_uint4(a, b, c, d);
  return __builtin_bit_cast(f16x8, v);
}

// scales[(s K + k) 2 + {0, 1}] = power of two that brings the largest weight / the bound 2 max|y|^2 of the products to < 2^14
__global__ __launch_bounds__(256)
void wpe_lp_scale_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float* __restrict__ scales,
                         int* __restrict__ tile_exp /* [S][K][nt_stride] */, int nt_stride)
{
  __shared__ float red[2][4];
  __shared__ float sab[2];
  const int k = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  float wm = 0.f, ym = 0.f;
  if (bin_active(g, k)) {
    for (int c = 0; c < g.C; c++) {
      const float* w = Winv + (((long)s * g.C + c) * g.K + k) * g.T_stride;
      const float2* y = X + (((long)s * g.K + k) * g.C + c) * g.T_stride;
      for (long t = tid; t < g.T; t += 256) {
        if (t >= g.lowerN) wm = fmaxf(wm, w[t]);
        const float2 v = y[t];
        ym = fmaxf(ym, fmaf(v.x, v.x, v.y * v.y));
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { wm = fmaxf(wm, __shfl_xor(wm, o)); ym = fmaxf(ym, __shfl_xor(ym, o)); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = wm; red[1][tid >> 6] = ym; }
  __syncthreads();
  if (tid == 0) {
    wm = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    ym = 2.f * fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    int ea = 0, eb = 0;
    if (wm > 0.f && wm < 3.0e38f) (void)frexpf(wm, &ea);                     // wm < 2^ea
    if (ym > 0.f && ym < 3.0e38f) (void)frexpf(ym, &eb);
    // (clamped: a bin of near-silence must not drive 2^(14 - e) -- or 1 / (sa sb) at the store -- out of float32's range)
    const int xa = 14 - ea < -60 ? -60 : (14 - ea > 60 ? 60 : 14 - ea), xb = 14 - eb < -60 ? -60 : (14 - eb > 60 ? 60 : 14 - eb);
    scales[((long)s * g.K + k) * 2 + 0] = ldexpf(1.f, xa);
    scales[((long)s * g.K + k) * 2 + 1] = ldexpf(1.f, xb);
    sab[0] = ldexpf(1.f, xa); sab[1] = ldexpf(1.f, xb);
  }
  __syncthreads();
  // Exponent balance per 64-frame tile (round 6; see lagprod16_task): half the distance between the exponents of the largest scaled
  // weight and the largest scaled product bound among the frames a task stages for the tile.  The samples' span does not depend on the
  // task; the weights' span starts la frames in and is LP_WT + RL nb - 1 long (nb = the task's row blocks, 1 .. LP_RMAX): one value per
  // (tile, q = (L - la) / RL, nb), the SAME numbers the tasks used to find themselves -- every tile, in the wavefronts that prepare the
  // operands, 52 tasks per bin at L = 33: 7 % of the kernel.  te[(j NLA + q) LP_RMAX + nb - 1].
  if (!bin_active(g, k)) return;
  const float sa = sab[0], sb = sab[1];
  const int RLs = 32 / g.C, NLA = (g.L + RLs - 1) / RLs + 1;
  const long ntile = (g.T + LP_WT - 1) / LP_WT;
  int* te = tile_exp + ((long)s * g.K + k) * nt_stride;
  // (16 tiles at a time: the largest weight over the channels of every frame the 16 tiles' spans can touch goes to LDS first, the spans
  //  of the (tile, q, nb) entries are then maxima over LDS words -- the table costs 30 us per launch instead of 70)
  constexpr int WMX = 16 * LP_WT + 256;
  __shared__ float wmx[WMX];
  for (long j0 = 0; j0 < ntile; j0 += 16) {
    long tA = j0 * LP_WT + g.lowerN - RLs, tB = (j0 + 16) * LP_WT + g.lowerN + g.L + RLs * LP_RMAX;
    if (tA < g.lowerN) tA = g.lowerN;
    if (tB > g.T) tB = g.T;
    const bool in_lds = tB - tA <= WMX;                              // (L <= ~170 at 8 channels; beyond that the spans read global memory)
    __syncthreads();
    if (in_lds)
      for (long i = tid; i < tB - tA; i += 256) {
        float m = 0.f;
        for (int c = 0; c < g.C; c++) m = fmaxf(m, Winv[(((long)s * g.C + c) * g.K + k) * g.T_stride + tA + i]);
        wmx[i] = m;
      }
    __syncthreads();
    const long j = j0 + (tid >> 4);
    const int sub = tid & 15;
    const long u0 = j * LP_WT;
    float ymj = 0.f;
    if (j < ntile) {
      long ty1 = u0 + LP_WT + g.L - 1;
      if (ty1 > g.T) ty1 = g.T;
      for (int c = 0; c < g.C; c++) {
        const float2* y = X + (((long)s * g.K + k) * g.C + c) * g.T_stride;
        for (long u = u0 + sub; u < ty1; u += 16) { const float2 v = y[u]; ymj = fmaxf(ymj, fmaf(v.x, v.x, v.y * v.y)); }
      }
    }
    for (int o = 8; o > 0; o >>= 1) ymj = fmaxf(ymj, __shfl_xor(ymj, o));
    if (j < ntile) {
      const float bm2 = ymj * sb;
      for (int q = sub; q < NLA; q += 16) {
        const long la = (long)g.L - (long)RLs * q;
        long t = u0 + g.lowerN + la;                                 // first weight of the span (entries before lowerN / behind T are zeros)
        float wmq = 0.f;
        long t1 = t + LP_WT - 1;                                     // the span of nb row blocks ends at t + LP_WT + RL nb - 1
        for (int nbk = 1; nbk <= LP_RMAX; nbk++) {
          t1 += RLs;
          const long lo = t < g.lowerN ? g.lowerN : t, hi = t1 < g.T ? t1 : g.T;
          if (in_lds) {
            for (long tt = lo; tt < hi; tt++) wmq = fmaxf(wmq, wmx[tt - tA]);
          } else {
            for (int c = 0; c < g.C; c++) {
              const float* w = Winv + (((long)s * g.C + c) * g.K + k) * g.T_stride;
              for (long tt = lo; tt < hi; tt++) wmq = fmaxf(wmq, w[tt]);
            }
          }
          t = hi > t ? hi : t;                                       // the next, longer span adds only its tail
          const float wm2 = wmq * sa;
          int ea = 0, eb = 0;
          if (wm2 > 0.f && bm2 > 0.f) { (void)frexpf(wm2, &ea); (void)frexpf(bm2, &eb); }
          te[(j * NLA + q) * LP_RMAX + nbk - 1] = (ea - eb) >> 1;    // |e| <= 64: exact powers of two
        }
      }
    }
  }
}

// 8-frame blocks of a shifted weight span: a lane's stream runs LP_WT + RL LP_RMAX frames from its copy's start (a row's shift within
// its task is at most RL LP_RMAX - 1 frames: 15 at 8 channels)
__host__ __device__ constexpr int lp16_nb(int C) { return (64 + 32 / C * 4 + 7) / 8; }

// b - (float)h as ONE instruction: v_fma_mix_f32 reads the float16 operand in place (no separate conversion)
__device__ __forceinline__ float sub_h_lo(float b, unsigned hpair)
{
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(b));
  return r;
}
__device__ __forceinline__ float sub_h_hi(float b, unsigned hpair)
{
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(b));
  return r;
}
// (a, b) -> packed float16 high parts (round toward zero) and packed float16 remainders: 4 instructions per pair
__device__ __forceinline__ void split2m(float a, float b, unsigned& hi, unsigned& lo)
{
  hi = pk_hi(a, b);
  lo = pk_hi(sub_h_lo(a, hi), sub_h_hi(b, hi));
}

// BTK_LP_TIMING build (profiles/): shader cycles of wavefront 0 of every task by phase -- 0 waiting for the partner wavefront at the top of a
// tile, 1 staging, 2 shifted copies, 3 the products, 4 the epilogue -- summed into lp_phase_cycles (read back by btk_wpe_estimate, BTK_WPE_TIMING=1)
#ifdef BTK_LP_TIMING
__device__ unsigned long long lp_phase_cycles[8];
#define LP_MARK(i) do { const long long cy_ = __builtin_readcyclecounter(); lp_tm[i] += cy_ - lp_last; lp_last = cy_; } while (0)
#else
#define LP_MARK(i) do { } while (0)
#endif
constexpr int LP16_ROWH = 8 * (lp16_nb(8) + 1);                        // halfs per row of a shifted copy: [8 pad][8 LP16_NB values] (8 channels)
constexpr int LP16_NCP = 4;                                        // shifted copies of the weight span kept in LDS
typedef unsigned u32x12 __attribute__((ext_vector_type(12)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// registers 0 .. 2 NR + 1 of a lane's operand stream from p[0], p[1], p[2]
template <int NR>
__device__ __forceinline__ void lp16_window(const uint4* p, u32x12& w)
{
  const u32x4 r0 = *reinterpret_cast<const u32x4*>(p);
  if constexpr (NR == 1) w = __builtin_shufflevector(r0, r0, 0, 1, 2, 3, -1, -1, -1, -1, -1, -1, -1, -1);
  if constexpr (NR == 2) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p + 1);
    const u32x4 r1 = __builtin_shufflevector(t, t, 0, 1, -1, -1);
    w = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, -1, -1, -1, -1, -1, -1);
  }
  if constexpr (NR >= 3) {
    const u32x4 r1 = *reinterpret_cast<const u32x4*>(p + 1);
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    const u32x8 r01 = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (NR == 3) w = __builtin_shufflevector(r01, r01, 0, 1, 2, 3, 4, 5, 6, 7, -1, -1, -1, -1);
    else {
      const u32x2 t = *reinterpret_cast<const u32x2*>(p + 2);
      const u32x8 r2 = __builtin_shufflevector(t, t, 0, 1, -1, -1, -1, -1, -1, -1);
      w = __builtin_shufflevector(r01, r2, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, -1, -1);
    }
  }
}

template <int J>
__device__ __forceinline__ f16x8 lp16_block(const u32x12& w)
{
  return __builtin_bit_cast(f16x8, __builtin_shufflevector(w, w, 2 * J, 2 * J + 1, 2 * J + 2, 2 * J + 3));
}

// One task = (lag difference d, up to four 32-row blocks) as before, but run by TWO wavefronts that share the staged tile: wavefront wv
// owns the column blocks 2 wv, 2 wv + 1 (8 accumulator blocks = 128 registers), so that two wavefronts fit a SIMD and one's operand
// preparation (LDS reads, products, float16 splits: ~3 vector instructions per matrix instruction) runs under the other's matrix
// instructions.  (One wavefront per SIMD with all 16 accumulator blocks measured no faster than the float32 kernel: with nothing else
// resident every LDS wait and every dependent vector instruction is exposed, ~22 cycles per instruction.)
template <int C, int NR, int NCW>
__device__ __forceinline__ void lagprod16_task(const float2* __restrict__ Xk, const float* __restrict__ Wk, const WpeGeom& g, float2* __restrict__ R,
                                               int ys_ld, int d, int la, int s, int k, float2* ys, uint4* wcp,
                                               float sa, float sb, const int* __restrict__ te)
{
  // Columns (round 6): the 2 C^2 = 128 columns are (re | im) x 64 channel pairs; a wavefront's column blocks are the REAL and the
  // IMAGINARY parts of the same 32 pairs (NCW = 2: pairs 32 wv + m; NCW = 1: pairs 32 (wv >> 1) + m, part wv & 1), so both blocks form
  // their products from the same sixteen LDS words -- x = y_c1(u), q = fb y_c2(u + d): re = x.x q.x + x.y q.y, im = x.y q.x - x.x q.y
  // -- where the interleaved columns of the first form (col = 2 pair + part) read x per block and kept q in two pre-multiplied copies.
  static_assert(C == 8 && (NCW == 1 || NCW == 2), "4 / NCW wavefronts x NCW column blocks");
  constexpr int RL = 32 / C;                                       // l1 values per 32-row block: row m of block j -> (l1 = la + RL j + m / C, c = m % C)
  constexpr int LP16_NB = lp16_nb(C);
  static_assert(LP_RMAX == 4, "lp16_nb");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int L = g.L, P = C * L;
  const int YN = LP_WT + L - 1;                                    // samples per channel span (<= 128: L <= 65)
  constexpr int WN = LP_WT + RL * NR - 1;                          // weights per target-channel span
  const int yq_ld = ys_ld - 1;                                     // (odd pitch: the eight rows x two lane halves of a 64-bit read fall on disjoint banks)
  float2* ysq = ys + C * ys_ld;                                    // the second factors, scaled: [C][yq_ld]
  float2 ypf[C];
  float wpf[C];
  int ehpf = 0;
  bool yok = false, wok = false;                                   // the prefetched sample / weight of this thread lies inside the recording
  const int teq = ((L + RL - 1) / RL + 1) * LP_RMAX, teo = ((L - la) / RL) * LP_RMAX + NR - 1;
  auto prefetch = [&](long u0) {                                   // threads 0 .. 127: one sample and one weight per channel and thread
    const int e = tid;
    if (e >= 128) return;
    ehpf = te[(u0 / LP_WT) * teq + teo];                           // the tile's exponent trade, |eh| <= 64 (wpe_lp_scale_kernel)
    // Every lane loads from a valid (clamped) address, raw, into its own register, and the staging step selects and scales: seventeen
    // loads in flight.  (Guarded loads with the scale applied on arrival compiled to a branch around every load and a weight register
    // shared by the eight channels -- eight global round trips in a row per tile, 40 % of the kernel's cycles: profiles/r06_wpe_lagprod_phases.txt)
    const long u = u0 + e;
    const long t = u0 + g.lowerN + la + e;
    yok = e < YN && u < g.T;
    wok = e < WN && t >= g.lowerN && t < g.T;
    const long uc = u < g.T ? u : g.T - 1;
    const long tc = t < 0 ? 0 : (t < g.T ? t : g.T - 1);
#pragma unroll
    for (int c = 0; c < C; c++) {
      ypf[c] = Xk[(long)c * g.T_stride + uc];
      wpf[c] = Wk[(long)c * g.K * g.T_stride + tc];
    }
  };
  f32x16 acc[NR][NCW];
#pragma unroll
  for (int j = 0; j < NR; j++)
#pragma unroll
    for (int cb = 0; cb < NCW; cb++) acc[j][cb] = f32x16{0};
  const int m = lane & 31, lk = lane >> 5;
  // A of block j: the 8 weights w_c(u0 + lowerN + la + e), e = kk + 8 lk + sh .. + 7, sh = RL j + m / C.  With RL = 4 the row blocks of a
  // lane are the windows 4 j .. 4 j + 7 of ONE stream -- copy m / C (a shift of 0 .. 3 frames) from the aligned offset kk + 8 lk on --,
  // so the lane reads 4 + 2 (NR - 1) consecutive registers' worth once (two 16-byte words and an 8-byte one at NR = 4) and block j's
  // operand is registers 2 j .. 2 j + 3 of that tuple: 5 LDS reads per 16 frames instead of 8, and only FOUR shifted copies to build.
  // wcp: row (hl LP16_NCP + copy) C + c of [8 pad][8 LP16_NB] float16 (LP16_ROWH / 8 uint4; the lane's stream starts one uint4 in)
  static_assert(RL == 4, "row blocks as windows of one shifted copy");
#if defined(BTK_LP_ABLATE) && (BTK_LP_ABLATE & 4)              // ablation build: every lane of a half reads the same words
  const int abase = 1 + lk;
#else
  const int abase = m * (LP16_ROWH / 8) + 1 + lk;                  // (copy C + c = m; uint4 units, behind the row's pad)
#endif
  // B: x = y_c1(u), q = fb y_c2(u + d) of the lane's pair c1 C + c2
  const int pairl = 32 * (NCW == 2 ? wv : (wv >> 1)) + m;
  const bool imw = (NCW == 1) && (wv & 1);                         // NCW = 1: the wavefront's part
#if defined(BTK_LP_ABLATE) && (BTK_LP_ABLATE & 8)              // ablation build: every lane of a half reads the same words (no bank conflicts possible)
  const int boffx = 4 * lk, boffq = 8 * lk + d;
#else
  const int boffx = ((pairl / C) * ys_ld) / 2 + 4 * lk;            // in float4 units (ys_ld is even)
  const int boffq = (pairl % C) * yq_ld + 8 * lk + d;
#endif
#ifdef BTK_LP_TIMING
  long long lp_tm[5] = {0, 0, 0, 0, 0};
  long long lp_last = __builtin_readcyclecounter();
#endif
  prefetch(0);
  // Segments of LP16_SEG frames: the low-part products are 2^-11 of the high-part ones and, on the diagonal of R, of one sign; added to
  // an accumulator that has grown over very many frames they would fall under half an ulp and vanish (a bias, not noise).  So the
  // accumulators are flushed to R every LP16_SEG frames (first segment: store, later ones: add -- same task, fixed order).
  for (long seg0 = 0; seg0 < g.T; seg0 += LP16_SEG) {
  const long seg1 = seg0 + LP16_SEG < g.T ? seg0 + LP16_SEG : g.T;
  int tpar = 0;
  for (long u0 = seg0; u0 < seg1; u0 += LP_WT, tpar ^= 1) {
    // Exponent balance of the tile (round 6).  The weights are 1 / |y|^2: in a quiet stretch of a non-stationary signal a large weight
    // meets tiny products, in a loud one the reverse, and with the one scale per (stream, bin) and operand above the small operand of
    // either kind sits so low in float16's range that its low part is subnormal or gone -- 11 bits instead of 22 on terms that are NOT
    // small (measured, segments 40 / 60 dB apart: taps 1.4e-3 / 1.9e-3 of the largest against the float64 oracle, the float32 kernel
    // 3e-5 / 1.5e-4; profiles/r06_wpe_envelope.txt).  Per tile the two operands trade a power of two, a 2^-e and b 2^+e with
    // e = half the distance between the exponents of the tile's largest weight and largest product bound: both then peak at the same
    // height (<= 2^14), the product -- hence the accumulators' scale -- is unchanged, and nothing is rounded by it.
    // (the exponent of the tile comes from wpe_lp_scale_kernel: one value per (stream, bin, tile), the same for every task)
    __syncthreads();                                               // the reads of the last tile are done
    LP_MARK(0);
#if defined(BTK_LP_ABLATE) && (BTK_LP_ABLATE & 16)             // ablation build: tiles after the first are neither staged nor copied
    if (tid < 128 && u0 == 0) {
#else
    if (tid < 128) {
#endif
      const int eh = ehpf;                                         // (loaded with the tile's samples, a tile ahead)
      const float fa = ldexpf(1.f, -eh), fb = sb * ldexpf(1.f, eh);
      const int e = tid;
      if (e < YN) {                                                // (one branch per kind of row, not one per channel)
#pragma unroll
        for (int c = 0; c < C; c++) {
          const float2 v = yok ? ypf[c] : make_float2(0.f, 0.f);
          ys[c * ys_ld + e] = v;
          ysq[c * yq_ld + e] = make_float2(v.x * fb, v.y * fb);
        }
      }
      // The weight span as its four one-frame shifts, split into float16 high / low parts, written by the thread that holds the value:
      // copy_s[p] = w[p + s], so thread e writes position e - s of copy s (16-bit stores; two channels share the conversions).  A row
      // is [8 pad][8 LP16_NB values]: positions -3 .. -1 fall into the row's own pad, positions past the end into the next row's.
      // (Until here the span went through a float row in LDS and a second step -- 80 threads, twelve dependent LDS reads each, a
      //  barrier of its own -- rebuilt it as shifted copies.)
      if (e < 8 * LP16_NB + 8) {
        const bool wl = e < WN && wok;
        unsigned short* wch = reinterpret_cast<unsigned short*>(wcp);
#pragma unroll
        for (int c = 0; c < C; c += 2) {
          const float w0 = wl ? (wpf[c] * sa) * fa : 0.f, w1 = wl ? (wpf[c + 1] * sa) * fa : 0.f;
          const unsigned hi = pk_hi(w0, w1);
          const unsigned lo = pk_hi(sub_h_lo(w0, hi), sub_h_hi(w1, hi));
#pragma unroll
          for (int sft = 0; sft < LP16_NCP; sft++) {
            const int pos = 8 + e - sft;                           // halfs from the row's start
            wch[((0 * LP16_NCP + sft) * C + c) * LP16_ROWH + pos] = (unsigned short)(hi & 0xffffu);
            wch[((0 * LP16_NCP + sft) * C + c + 1) * LP16_ROWH + pos] = (unsigned short)(hi >> 16);
            wch[((1 * LP16_NCP + sft) * C + c) * LP16_ROWH + pos] = (unsigned short)(lo & 0xffffu);
            wch[((1 * LP16_NCP + sft) * C + c + 1) * LP16_ROWH + pos] = (unsigned short)(lo >> 16);
          }
        }
      }
    }
    LP_MARK(1);
    __syncthreads();
    if (u0 + LP_WT < g.T) prefetch(u0 + LP_WT);
    LP_MARK(2);
#pragma unroll 1
    for (int kk = 0; kk < LP_WT; kk += 16) {
      f16x8 ah[NR], al[NR];
      {
        u32x12 wh, wl;                                             // the lane's stream, 2 NR + 2 registers of it
        lp16_window<NR>(wcp + abase + kk / 8, wh);
        lp16_window<NR>(wcp + LP16_NCP * C * (LP16_ROWH / 8) + abase + kk / 8, wl);
        if constexpr (NR > 1) asm volatile("" : "+v"(wh), "+v"(wl));   // (one register tuple each: the blocks' operands are sub-ranges of it, not copies)
        ah[0] = lp16_block<0>(wh); al[0] = lp16_block<0>(wl);
        if constexpr (NR > 1) { ah[1] = lp16_block<1>(wh); al[1] = lp16_block<1>(wl); }
        if constexpr (NR > 2) { ah[2] = lp16_block<2>(wh); al[2] = lp16_block<2>(wl); }
        if constexpr (NR > 3) { ah[3] = lp16_block<3>(wh); al[3] = lp16_block<3>(wl); }
      }
      float2 yq[8];
#pragma unroll
      for (int i = 0; i < 8; i++) yq[i] = ysq[boffq + kk + i];     // (odd d: not 16-byte aligned)
      float4 x4[4];
#pragma unroll
      for (int i2 = 0; i2 < 4; i2++) x4[i2] = reinterpret_cast<const float4*>(ys)[boffx + kk / 2 + i2];
#pragma unroll
      for (int cb = 0; cb < NCW; cb++) {
        unsigned bh[4], bl[4];
        const bool imc = (NCW == 2) ? (cb == 1) : imw;
#pragma unroll
        for (int i2 = 0; i2 < 4; i2++) {
          const float4 x = x4[i2];
          const float2 q0 = yq[2 * i2], q1 = yq[2 * i2 + 1];
          // x conj(y): re = x.x q.x + x.y q.y, im = x.x (-q.y) + x.y q.x (the same two roundings as the pre-multiplied form)
          const float b0 = imc ? fmaf(x.x, -q0.y, x.y * q0.x) : fmaf(x.x, q0.x, x.y * q0.y);
          const float b1 = imc ? fmaf(x.z, -q1.y, x.w * q1.x) : fmaf(x.z, q1.x, x.w * q1.y);
          split2m(b0, b1, bh[i2], bl[i2]);
        }
        const f16x8 Bh = mk8(bh[0], bh[1], bh[2], bh[3]), Bl = mk8(bl[0], bl[1], bl[2], bl[3]);
#if defined(BTK_LP_ABLATE) && (BTK_LP_ABLATE & 1)              // ablation build (profiles/): the operands are prepared, no matrix instruction consumes them
#pragma unroll
        for (int j = 0; j < NR; j++) asm volatile("" :: "v"(ah[j]), "v"(al[j]), "v"(Bh), "v"(Bl));
#else
#pragma unroll
        for (int j = 0; j < NR; j++) acc[j][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], Bh, acc[j][cb], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NR; j++) acc[j][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], Bl, acc[j][cb], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NR; j++) acc[j][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], Bh, acc[j][cb], 0, 0, 0);
#endif
      }
    }
  }
  LP_MARK(3);
  // ---- store the segment (as lagprod_task), with the two scales undone
  // (the lane indices pass through an opaque move: the addresses below are then computed here, per flush, instead of being hoisted
  // out of the segment loop and held in ~80 registers across the tile loop, which spilled the accumulators)
  int ms = pairl, lks = lk;
  asm volatile("" : "+v"(ms), "+v"(lks));
  const float unscale = 1.0f / (sa * sb);
  float2* Rc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = i + 4 * lks;                                     // row % C for row = i + 8 (r >> 2) + 4 lk
    Rc[i] = R + (((long)s * C + c) * g.K + k) * (long)P * P;
  }
  {
    const int c1 = ms / C, c2 = ms % C;
    const bool lower = c1 >= c2;
    const bool skip = (d == 0 && c1 < c2);                         // the swapped pair stores this entry
    const long off0 = lower ? (long)c1 * L * P + (long)c2 * L - d : ((long)c2 * L - d) * P + (long)c1 * L;
    const float sre = unscale, sim = lower ? unscale : -unscale;   // (the upper triangle holds the conjugate)
#pragma unroll
    for (int j = 0; j < NR; j++) {
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * lks;
        const int l1 = la + RL * j + row / C;
#if defined(BTK_LP_ABLATE) && (BTK_LP_ABLATE & 32)             // ablation build: nothing is stored (the condition is never true)
        if (l1 < d || skip || g.T > 0) continue;
#else
        if (l1 < d || skip) continue;
#endif
        float2* dst = &Rc[reg & 3][off0 + (long)l1 * (P + 1)];
        if constexpr (NCW == 2) {                                  // both parts of the entry are this lane's: one 8-byte store
          const float2 v = make_float2(sre * acc[j][0][reg], sim * acc[j][1][reg]);
          if (seg0 > 0) { const float2 o = *dst; *dst = make_float2(o.x + v.x, o.y + v.y); }
          else *dst = v;
          acc[j][0][reg] = 0.f; acc[j][1][reg] = 0.f;
        } else {
          float* dp = reinterpret_cast<float*>(dst) + (imw ? 1 : 0);
          const float v = (imw ? sim : sre) * acc[j][0][reg];
          *dp = seg0 > 0 ? *dp + v : v;
          acc[j][0][reg] = 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  LP_MARK(4);
  }
#ifdef BTK_LP_TIMING
  if (tid == 0) {
    for (int i = 0; i < 5; i++) atomicAdd(lp_phase_cycles + i, (unsigned long long)lp_tm[i]);
    atomicAdd(lp_phase_cycles + 5, 1ull);
  }
#endif
}

template <int C, int NCW>
__device__ __forceinline__ void wpe_lagprod16_body(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ R, int ys_ld,
                          const float* __restrict__ scales, const int* __restrict__ tile_exp, int nt_stride, int nstreams)
{
  constexpr int RL = 32 / C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(LP16_ROWH == 8 * (lp16_nb(C) + 1), "rows of the shifted copies");
  uint4* wcp = reinterpret_cast<uint4*>(smem);                     // [2 (high | low)][LP16_NCP shifts][C] rows of [8 pad][8 LP16_NB] float16, + one pad behind the last row
  float2* ys = reinterpret_cast<float2*>(wcp + 2 * LP16_NCP * C * (LP16_ROWH / 8) + 1);   // [C][ys_ld]: sample u0 + e of channel c'
  // (ys is followed by the scaled second factors, pitch ys_ld - 1)
  // Launch order (round 6): the tasks of a (stream, bin) write the diagonals of the same eight matrices -- every 128-byte line of R takes
  // its sixteen entries from sixteen different tasks.  Workgroups go to the eight XCDs in turn, so with one grid row per bin a line was
  // assembled in eight L2s and left each of them partly written; here blockIdx.x = XCD + 8 task and blockIdx.y = a group of eight bins,
  // one per XCD: a bin's tasks run on ONE XCD, back to back, and its lines fill up in that L2.
  const int bin = 8 * (int)blockIdx.y + (int)(blockIdx.x & 7);
  if (bin >= g.K * nstreams) return;
  const int k = bin % g.K, s = bin / g.K;
  if (!bin_active(g, k)) return;
  const int L = g.L;
  int d = 0, grp = blockIdx.x >> 3;
  auto nblk = [&](int dd) { return (L - dd + RL - 1) / RL; };
  auto ngrp = [&](int dd) { return (nblk(dd) + LP_RMAX - 1) / LP_RMAX; };
  while (grp >= ngrp(d)) { grp -= ngrp(d); d++; }
  const int nbd = nblk(d), ng = ngrp(d), base = nbd / ng, extra = nbd % ng;
  const int nb = base + (grp < extra ? 1 : 0);
  const int first = grp * base + (grp < extra ? grp : extra);
  const int la = L - RL * (first + nb);
  const float2* Xk = X + ((long)s * g.K + k) * C * g.T_stride;
  const float* Wk = Winv + ((long)s * C * g.K + k) * g.T_stride;
  const float sa = scales[((long)s * g.K + k) * 2], sb = scales[((long)s * g.K + k) * 2 + 1];
  const int* te = tile_exp + ((long)s * g.K + k) * nt_stride;
  switch (nb) {
    case 4: lagprod16_task<C, 4, NCW>(Xk, Wk, g, R, ys_ld, d, la, s, k, ys, wcp, sa, sb, te); break;
    case 3: lagprod16_task<C, 3, NCW>(Xk, Wk, g, R, ys_ld, d, la, s, k, ys, wcp, sa, sb, te); break;
    case 2: lagprod16_task<C, 2, NCW>(Xk, Wk, g, R, ys_ld, d, la, s, k, ys, wcp, sa, sb, te); break;
    default: lagprod16_task<C, 1, NCW>(Xk, Wk, g, R, ys_ld, d, la, s, k, ys, wcp, sa, sb, te); break;
  }
}

// two wavefronts x two column blocks (256 registers per lane: two wavefronts per SIMD) / four x one (168: three per SIMD)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wpe_lagprod16_w2_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ R, int ys_ld,
                             const float* __restrict__ scales, const int* __restrict__ tile_exp, int nt_stride, int nstreams)
{
  wpe_lagprod16_body<8, 2>(X, Winv, g, R, ys_ld, scales, tile_exp, nt_stride, nstreams);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void wpe_lagprod16_w4_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ R, int ys_ld,
                             const float* __restrict__ scales, const int* __restrict__ tile_exp, int nt_stride, int nstreams)
{
  wpe_lagprod16_body<8, 1>(X, Winv, g, R, ys_ld, scales, tile_exp, nt_stride, nstreams);
}

template <int C>
__global__ __launch_bounds__(64)
void wpe_lagprod_kernel(const float2* __restrict__ X, const float* __restrict__ Winv, WpeGeom g, float2* __restrict__ R, int ys_ld, int ws_ld)
{
  constexpr int RL = 32 / C;
  static_assert(2 * C * C / 32 >= 1 && 2 * C * C / 32 <= 4 && 32 % C == 0, "C in {4, 8}");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* ys = reinterpret_cast<float2*>(smem);                    // [C][ys_ld]: sample u0 + e of channel c'
  float* ws = reinterpret_cast<float*>(ys + C * ys_ld);            // [C][ws_ld]: w_c(u0 + lowerN + la + e)
  const int k = blockIdx.y, s = blockIdx.z;
  if (!bin_active(g, k)) return;
  const int L = g.L;
  // task -> (d, group of up to LP_RMAX row blocks counted down from l1 = L - 1)
  int d = 0, grp = blockIdx.x;
  auto nblk = [&](int dd) { return (L - dd + RL - 1) / RL; };
  auto ngrp = [&](int dd) { return (nblk(dd) + LP_RMAX - 1) / LP_RMAX; };
  while (grp >= ngrp(d)) { grp -= ngrp(d); d++; }
  // the row blocks of d are spread evenly over its groups (9 -> 3 + 3 + 3, 5 -> 3 + 2): a one-block task runs at half the efficiency
  const int nbd = nblk(d), ng = ngrp(d), base = nbd / ng, extra = nbd % ng;
  const int nb = base + (grp < extra ? 1 : 0);
  const int first = grp * base + (grp < extra ? grp : extra);      // row blocks above this group (counted down from l1 = L - 1)
  const int la = L - RL * (first + nb);                            // lowest l1 of the task (rows with l1 < d are not stored)
  const float2* Xk = X + ((long)s * g.K + k) * C * g.T_stride;
  const float* Wk = Winv + ((long)s * C * g.K + k) * g.T_stride;   // + c K T_stride
  switch (nb) {
    case 4: lagprod_task<C, 4>(Xk, Wk, g, R, ys_ld, ws_ld, d, la, s, k, ys, ws); break;
    case 3: lagprod_task<C, 3>(Xk, Wk, g, R, ys_ld, ws_ld, d, la, s, k, ys, ws); break;
    case 2: lagprod_task<C, 2>(Xk, Wk, g, R, ys_ld, ws_ld, d, la, s, k, ys, ws); break;
    default: lagprod_task<C, 1>(Xk, Wk, g, R, ys_ld, ws_ld, d, la, s, k, ys, ws); break;
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

// Round 4: the same solve with the matrix resident in the accumulator registers of one 512-thread workgroup (chol_reg.h), P <= 271.
// R is only read (the lower triangle, once); diagonal bias and loading as above, applied on the way into the registers.
__global__ __launch_bounds__(cholr::NTH)
void wpe_solve_reg_kernel(const float2* __restrict__ R, const float2* __restrict__ rvec, WpeGeom g, float load_factor,
                          float diagonal_bias, float2* __restrict__ G, int* __restrict__ fail_count,
                          unsigned long long* __restrict__ phase_cycles)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, sc = blockIdx.y;
  if (!bin_active(g, k)) return;
  const int P = g.C * g.L;
  const int tid = threadIdx.x;
  const float2* mat = R + ((long)sc * g.K + k) * (long)P * P;
  const int s = sc / g.C, c = sc % g.C;
  float2* gout = G + (((long)s * g.C + c) * g.K + k) * (long)P;
  float* diagv = reinterpret_cast<float*>(reinterpret_cast<float2*>(smem) + cholr::OFF_DIAG);
  float* red = diagv + cholr::ROWS;
  // diagonal bias (dereverberation.cc:574-577) then load_R_ (:648-663): |d + bias| + max |.| * load_factor
  float a = 0.f;
  if (tid < P) {
    float2 d = mat[(long)tid * P + tid];
    d.x += diagonal_bias;
    a = sqrtf(d.x * d.x + d.y * d.y);
  }
  float mx = a;
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < cholr::NWAVE; w++) mx = fmaxf(mx, red[w]);
  const float load = mx * load_factor;
  if (tid < P) diagv[tid] = a + load;
  __syncthreads();
  long long tm[5] = {0, 0, 0, 0, 0};
  long long tlast = phase_cycles ? clock64() : 0;
  auto mark = [&](int i) { if (phase_cycles) { const long long cy = clock64(); tm[i] += cy - tlast; tlast = cy; } };
  const int q = c * g.L;                                            // lowerN == 0: r_c is column c L of R (see wpe_solve_kernel)
  auto rhs_conj = [&](int p) {
    if (rvec) { const float2 v = rvec[((long)sc * g.K + k) * P + p]; return make_float2(v.x, -v.y); }
    const float2 v = (p >= q) ? mat[(long)p * P + q] : mat[(long)q * P + p];
    return (p >= q) ? make_float2(v.x, -v.y) : v;
  };
  const float2* x = cholr::solve(mat, P, rhs_conj, [&](int p) { return diagv[p]; }, 0.f, smem, mark);
  if (!x) { if (tid == 0) atomicAdd(fail_count, 1); return; }
  if (tid < P) gout[tid] = x[tid];
  if (phase_cycles && tid == 0) {
    for (int i = 0; i < 5; i++) atomicAdd(phase_cycles + i, (unsigned long long)tm[i]);
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


// prediction pass: matrix-core kernel for 4..16 channels, the vector kernel otherwise (and with BTK_WPE_PREDICT_VALU set)
int launch_wpe_predict(const float2* X, const float2* G, const WpeGeom& g, int S, int mode, float* Winv, float2* OUT, hipStream_t st)
{
  const int C = g.C, K = g.K;
  const long T = g.T;
  const int P = C * g.L;
  if (C >= 4 && C <= 16 && g.L >= 1 && !btk_switches().wpe_predict_valu) {
    int g_ld = P; while (g_ld % 32 != 4) g_ld++;                   // filter rows 8 banks apart
    const int sp_ld = PM_FG + g.L - 1 + g.lowerN;
    const size_t lds = sizeof(float2) * ((size_t)C * g_ld + (size_t)C * sp_ld);
    if (lds <= 64 * 1024) {
      hipLaunchKernelGGL(wpe_predict_mfma_kernel, dim3((unsigned)((T + PM_FG - 1) / PM_FG), (unsigned)K, (unsigned)S), dim3(256), lds, st,
                         X, G, g, mode, Winv, OUT, g_ld, sp_ld);
      BTK_HIP_CHECK(hipGetLastError());
      return BTK_OK;
    }
  }
  hipLaunchKernelGGL(wpe_predict_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)K, (unsigned)(S * C)), dim3(256), 0, st, X, G, g, mode, Winv, OUT);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // namespace

extern "C" {

long btk_wpe_workspace_bytes(int S, int K, int C, int lowerN, int upperN, long T_stride)
{
  const long P = (long)C * (upperN - lowerN + 1);
  const long nb = (long)S * C * K;
  return nb * P * P * 8 + nb * P * 8 + nb * T_stride * 4 + (long)S * K * 8 + (long)S * K * (T_stride / 64 + 2) * (((upperN - lowerN + 1) + 32 / (C > 0 && C <= 32 ? C : 32) - 1) / (32 / (C > 0 && C <= 32 ? C : 32)) + 1) * 4 * 4 + 256;
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
  float* lp_scales = Winv + nb * T_stride;                          // [S][K][2]: operand scales of the float16 lag-product kernel
  const int nla = C <= 32 ? ((g.L + 32 / C - 1) / (32 / C) + 1) : 1;
  const int nt_stride = (int)(T_stride / 64 + 2) * nla * 4;          // per (stream, bin): [tiles][q][row blocks] (wpe_lp_scale_kernel)
  int* lp_tile_exp = reinterpret_cast<int*>(lp_scales + (long)S * K * 2);   // [S][K][nt_stride]: its per-tile exponent trades
  const float2* Xp = static_cast<const float2*>(X);
  float2* Gp = static_cast<float2*>(G);
  const int ntile = (int)((P + 63) / 64);
  const int nspan = 63 / g.L + 2;                                  // channels a 64-row tile can touch
  const size_t lds_herk = sizeof(float2) * 2 * (size_t)nspan * (WT_ + g.L - 1) + sizeof(float) * 4 * WT_;
  const float load_factor = (float)pow(10.0, load_db / 10.0);
  const size_t lds_solve = sizeof(float2) * ((P + 1) & ~1L) + sizeof(float) * 512 + sizeof(float2) * (size_t)P * cholb::CH_LD;
  if (lds_solve > 160 * 1024)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_wpe_estimate: C*(upperN-lowerN+1) = %ld taps exceed the LDS-resident solver (max ~560)", P);
  // register-resident solver (chol_reg.h) for 112 <= P <= 271 (profiles/r04_wpe_solver_sweep.txt: the panel solver's four systems per CU win below;
  // BTK_WPE_SOLVE_REG=1: every P <= 271; BTK_WPE_SOLVE_PANEL=1: never)
  const bool solve_reg = P <= cholr::P_MAX && !btk_switches().wpe_solve_panel && (P >= 112 || btk_switches().wpe_solve_reg);
  if (solve_reg)
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wpe_solve_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cholr::lds_bytes()));
  else if (lds_solve > 64 * 1024)
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wpe_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
  unsigned long long* phase = nullptr;                              // BTK_WPE_TIMING=1: shader cycles per phase of the solver, printed per call
  if (btk_switches().wpe_timing) {
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) BTK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbuf), 8 * sizeof(unsigned long long)));
    BTK_HIP_CHECK(hipMemsetAsync(dbuf, 0, 8 * sizeof(unsigned long long), st));
    phase = dbuf;
  }
  for (int it = 0; it < iterations; it++) {
    { const int rc = launch_wpe_predict(Xp, Gp, g, S, 0, Winv, static_cast<float2*>(nullptr), st); if (rc != BTK_OK) return rc; }
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
    const bool lagprod = skip && !btk_switches().wpe_herk_blocks && (C == 8 || C == 4) && g.L <= 65 && g.L >= 32 / C;
    if (lagprod) {
      // tasks: for every lag difference d the row blocks of l1 = d .. L-1 in groups of LP_RMAX
      const int RL = 32 / C;
      unsigned ntask = 0;
      for (int d = 0; d < g.L; d++) ntask += (unsigned)(((g.L - d + RL - 1) / RL + LP_RMAX - 1) / LP_RMAX);
      int ys_ld = LP_WT + g.L - 1; while (ys_ld % 32 != 4) ys_ld++;          // float2 row pitch: channel rows 8 banks apart (16-byte reads of the two lane halves 4 apart)
      int ws_ld = LP_WT + RL * LP_RMAX - 1; while (ws_ld % 64 != (C == 8 ? 8 : 16)) ws_ld++;    // weight rows 8 / 16 banks apart: a wavefront's read2 touches 7 / 13 consecutive words per row
      const size_t lds_lp = sizeof(float2) * (size_t)C * ys_ld + sizeof(float) * (size_t)C * ws_ld;
      if (!btk_switches().wpe_lagprod_f32 && C == 8) {
        // round 5: float16-split operands on the 16 x faster matrix instruction (see lagprod16_task)
        int ys16 = LP_WT + g.L; while (ys16 % 32 != 2) ys16++;                 // float2 row pitch of the samples: the four rows x two lane halves of a 16-byte read on disjoint banks; the scaled second factors follow at pitch ys16 - 1
        const size_t lds16p = sizeof(uint4) * (2 * LP16_NCP * (size_t)C * (LP16_ROWH / 8) + 1) + sizeof(float2) * (size_t)C * (2 * ys16 - 1);   // 23.8 KB at 33 lags
        hipLaunchKernelGGL(wpe_lp_scale_kernel, dim3((unsigned)K, (unsigned)S), dim3(256), 0, st, Xp, Winv, g, lp_scales, lp_tile_exp, nt_stride);
        if (btk_switches().wpe_lagprod_waves == 2)
          hipLaunchKernelGGL(wpe_lagprod16_w2_kernel, dim3(8 * ntask, (unsigned)((K * S + 7) / 8)), dim3(128), lds16p, st, Xp, Winv, g, R, ys16, lp_scales, lp_tile_exp, nt_stride, S);
        else
          hipLaunchKernelGGL(wpe_lagprod16_w4_kernel, dim3(8 * ntask, (unsigned)((K * S + 7) / 8)), dim3(256), lds16p, st, Xp, Winv, g, R, ys16, lp_scales, lp_tile_exp, nt_stride, S);
      }
      else if (C == 8) hipLaunchKernelGGL(wpe_lagprod_kernel<8>, dim3(ntask, (unsigned)K, (unsigned)S), dim3(64), lds_lp, st, Xp, Winv, g, R, ys_ld, ws_ld);
      else        hipLaunchKernelGGL(wpe_lagprod_kernel<4>, dim3(ntask, (unsigned)K, (unsigned)S), dim3(64), lds_lp, st, Xp, Winv, g, R, ys_ld, ws_ld);
    } else if (skip && C % 4 == 0) {
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
    if (solve_reg)
      hipLaunchKernelGGL(wpe_solve_reg_kernel, dim3((unsigned)K, (unsigned)(S * C)), dim3(cholr::NTH), cholr::lds_bytes(), st,
                         R, rvec_from_R ? static_cast<const float2*>(nullptr) : rvec, g, load_factor, (float)diagonal_bias, Gp, fail_count, phase);
    else
      hipLaunchKernelGGL(wpe_solve_kernel, dim3((unsigned)K, (unsigned)(S * C)), dim3(256), lds_solve, st,
                         R, rvec_from_R ? static_cast<const float2*>(nullptr) : rvec, g, load_factor, (float)diagonal_bias, Gp, fail_count, phase);
    BTK_HIP_CHECK(hipGetLastError());
  }
#ifdef BTK_LP_TIMING
  {
    unsigned long long h[8];
    BTK_HIP_CHECK(hipStreamSynchronize(st));
    BTK_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(lp_phase_cycles), sizeof(h)));
    static const char* nm[5] = {"partner_wait", "staging", "copies", "products", "epilogue"};
    unsigned long long tot = 0;
    for (int i = 0; i < 5; i++) tot += h[i];
    fprintf(stderr, "[BTK_LP_TIMING] %llu tasks (cumulative), %.0f cycles per task:", h[5], h[5] ? (double)tot / h[5] : 0.0);
    for (int i = 0; i < 5; i++) fprintf(stderr, " %s %.1f%%", nm[i], tot ? 100.0 * h[i] / tot : 0.0);
    fprintf(stderr, "\n");
  }
#endif
  if (phase) {
    unsigned long long h[8];
    BTK_HIP_CHECK(hipMemcpyAsync(h, phase, sizeof(h), hipMemcpyDeviceToHost, st));
    BTK_HIP_CHECK(hipStreamSynchronize(st));
    static const char* names_panel[6] = {"panel_load", "mfma_update", "diagonal_block", "row_solves", "writeback_forward", "back_substitution"};
    static const char* names_reg[6] = {"load", "diagonal_block", "row_solves", "trailing_update", "back_substitution", "-"};
    const char** names = solve_reg ? names_reg : names_panel;
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
  return launch_wpe_predict(static_cast<const float2*>(X), static_cast<const float2*>(G), g, S, 1, static_cast<float*>(nullptr),
                            static_cast<float2*>(OUT), as_stream(stream));
}

}  // extern "C"
