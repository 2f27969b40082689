// cov_kernels.hip -- per-bin spatial covariance accumulation (weighted HERK) for gfx950.
//
// Replaces the numpy.outer accumulation loops of SubbandSMIMVDRBeamformer.accu_stats_from_label
// and SubbandSOSBatchBeamformer.accu_stats_from_{label,tfmask}
// (reference lib/pybeamformer.py:967-985, 1075-1093, 1129-1147):
//     R_k += sum_t w[k][t] x_k[t] x_k[t]^H        (w = 0/1 frame gate or TF-mask value)
// For one (stream, bin) this is a weighted rank-T update A diag(w) A^H with A = X[s][k] (N x T,
// frames contiguous): a dense complex GEMM, 8 N^2 flop per frame and bin.
//
//   cov_mfma_kernel : v_mfma_f32_32x32x2_f32 (exact fp32 matrix cores).  A workgroup owns one
//                     64x64 tile of R_k; its four wavefronts own the 32x32 quadrants.  The real
//                     and imaginary parts use  Rr = Ar Ar^T + Ai Ai^T,  Ri = Ai Ar^T - Ar Ai^T
//                     (4 real MFMA chains, 64 accumulator registers per lane).  Frames are staged
//                     through LDS in 32-frame tiles with 128-byte coalesced row reads.
//   cov_valu_kernel : same tiling on the vector ALU, used for parity cross-checks and N < 16.
//
// Weights: wt [S][K][T] float32 (TF mask, frames contiguous) or wf [S][T] (frame gate); both
// optional.  R [S][K][N][N] complex64 row-major, accumulated in place (+=).
#include "btk_internal.h"

namespace {

constexpr int CT = 32;                 // frames per LDS tile
constexpr int CLD = CT + 1;            // padded row (float2 units)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Stage rows [row0, row0+64) of A_k (zero beyond N) for frames [t0, t0+CT) scaled by nothing;
// weight vector for the tile goes to wrow[CT].
__device__ __forceinline__ void stage_tile(const float2* __restrict__ Xk, int N, long T_stride, long T,
                                           int row0, long t0, float2* __restrict__ dst, int tid, int nthreads)
{
  // 16 lanes x float4 (2 frames) = 32 frames = 256 B per row
  for (int idx = tid; idx < 64 * (CT / 2); idx += nthreads) {
    const int r = idx / (CT / 2), c2 = idx % (CT / 2);
    const int n = row0 + r;
    const long t = t0 + 2 * c2;
    float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
    if (n < N) {
      const float2* p = Xk + (long)n * T_stride + t;
      if (t < T) a = p[0];
      if (t + 1 < T) b = p[1];
    }
    dst[r * CLD + 2 * c2] = a;
    dst[r * CLD + 2 * c2 + 1] = b;
  }
}

__device__ __forceinline__ float tile_weight(const float* __restrict__ wt, const float* __restrict__ wf, long t, long T)
{
  if (t >= T) return 0.f;
  float w = 1.f;
  if (wt) w *= wt[t];
  if (wf) w *= wf[t];
  return w;
}

// grid: (tiles_i * tiles_j, K, S); block 256 threads = 4 waves (2x2 quadrants of 32x32)
__global__ __launch_bounds__(256)
void cov_mfma_kernel(const float2* __restrict__ X, const float* __restrict__ WT, const float* __restrict__ WF,
                     float2* __restrict__ R, int K, int N, long T_stride, long T, int ntile)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* Ai = reinterpret_cast<float2*>(smem);          // [64][CLD] rows of the i-tile
  float2* Aj = Ai + 64 * CLD;                            // [64][CLD] rows of the j-tile (weighted)
  float* wrow = reinterpret_cast<float*>(Aj + 64 * CLD); // [CT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.y, s = blockIdx.z;
  const int ti = blockIdx.x / ntile, tj = blockIdx.x % ntile;
  const float2* Xk = X + ((long)s * K + k) * N * T_stride;
  const float* wt = WT ? WT + ((long)s * K + k) * T_stride : nullptr;
  const float* wf = WF ? WF + (long)s * T_stride : nullptr;
  const int qi = wave >> 1, qj = wave & 1;               // quadrant of the 64x64 tile
  f32x16 rr = {0}, ri = {0};                             // Re, Im accumulators (32x32 per wave)
  const int li = lane & 31, lk = lane >> 5;              // MFMA 32x32x2: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31]
  const bool same = ti == tj;                            // diagonal tile: the i- and j-rows are the same rows, staged once
  const float2* Bj = same ? Ai : Aj;

  // register prefetch of the next frame tile: the global loads fly while the MFMAs of the current tile run
  constexpr int PER = 64 * (CT / 2) / 256;               // float4 pairs per thread and row tile (= 4)
  float4 pi_[PER], pj_[PER];
  float wnext = 0.f;
  auto prefetch = [&](long t0) {
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int idx = tid + q * 256;
      const int r = idx / (CT / 2), c2 = idx % (CT / 2);
      const long t = t0 + 2 * c2;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      const int ni = ti * 64 + r, nj = tj * 64 + r;
      if (ni < N && t < T) {
        const float2* p = Xk + (long)ni * T_stride + t;
        const float2 u = p[0];
        const float2 v = (t + 1 < T) ? p[1] : make_float2(0.f, 0.f);
        a = make_float4(u.x, u.y, v.x, v.y);
      }
      if (!same && nj < N && t < T) {
        const float2* p = Xk + (long)nj * T_stride + t;
        const float2 u = p[0];
        const float2 v = (t + 1 < T) ? p[1] : make_float2(0.f, 0.f);
        b = make_float4(u.x, u.y, v.x, v.y);
      }
      pi_[q] = a; pj_[q] = b;
    }
    if (tid < CT) wnext = tile_weight(wt, wf, t0 + tid, T);
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int idx = tid + q * 256;
      const int r = idx / (CT / 2), c2 = idx % (CT / 2);
      Ai[r * CLD + 2 * c2] = make_float2(pi_[q].x, pi_[q].y);
      Ai[r * CLD + 2 * c2 + 1] = make_float2(pi_[q].z, pi_[q].w);
      if (!same) {
        Aj[r * CLD + 2 * c2] = make_float2(pj_[q].x, pj_[q].y);
        Aj[r * CLD + 2 * c2 + 1] = make_float2(pj_[q].z, pj_[q].w);
      }
    }
    if (tid < CT) wrow[tid] = wnext;
  };

  prefetch(0);
  for (long t0 = 0; t0 < T; t0 += CT) {
    __syncthreads();
    commit();
    __syncthreads();
    if (t0 + CT < T) prefetch(t0 + CT);
#pragma unroll 4
    for (int kk = 0; kk < CT; kk += 2) {
      const float2 a = Ai[(qi * 32 + li) * CLD + kk + lk];
      float2 b = Bj[(qj * 32 + li) * CLD + kk + lk];
      const float w = wrow[kk + lk];
      b.x *= w; b.y *= w;
      // Rr += ar br + ai bi ; Ri += ai br - ar bi      (R_ij = a_i conj(b_j))
      rr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, rr, 0, 0, 0);
      rr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, rr, 0, 0, 0);
      ri = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, ri, 0, 0, 0);
      ri = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.x, b.y, ri, 0, 0, 0);
    }
  }
  // C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float2* Rk = R + ((long)s * K + k) * N * N;
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    const int row = ti * 64 + qi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int col = tj * 64 + qj * 32 + (lane & 31);
    if (row < N && col < N) {
      float2* p = Rk + (long)row * N + col;
      const float2 old = *p;
      *p = make_float2(old.x + rr[reg], old.y + ri[reg]);
    }
  }
}

// Vector-ALU version: thread (a,b) of a 16x16 grid owns the 4x4 sub-block rows a+16p, cols b+16q.
__global__ __launch_bounds__(256)
void cov_valu_kernel(const float2* __restrict__ X, const float* __restrict__ WT, const float* __restrict__ WF,
                     float2* __restrict__ R, int K, int N, long T_stride, long T, int ntile)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* Ai = reinterpret_cast<float2*>(smem);
  float2* Aj = Ai + 64 * CLD;
  float* wrow = reinterpret_cast<float*>(Aj + 64 * CLD);
  const int tid = threadIdx.x;
  const int k = blockIdx.y, s = blockIdx.z;
  const int ti = blockIdx.x / ntile, tj = blockIdx.x % ntile;
  const float2* Xk = X + ((long)s * K + k) * N * T_stride;
  const float* wt = WT ? WT + ((long)s * K + k) * T_stride : nullptr;
  const float* wf = WF ? WF + (long)s * T_stride : nullptr;
  const int a = tid >> 4, b = tid & 15;
  float2 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[p][q] = make_float2(0.f, 0.f);
  for (long t0 = 0; t0 < T; t0 += CT) {
    __syncthreads();
    stage_tile(Xk, N, T_stride, T, ti * 64, t0, Ai, tid, 256);
    stage_tile(Xk, N, T_stride, T, tj * 64, t0, Aj, tid, 256);
    if (tid < CT) wrow[tid] = tile_weight(wt, wf, t0 + tid, T);
    __syncthreads();
    for (int tt = 0; tt < CT; tt++) {
      const float w = wrow[tt];
      float2 xi[4], xj[4];
#pragma unroll
      for (int p = 0; p < 4; p++) xi[p] = Ai[(a + 16 * p) * CLD + tt];
#pragma unroll
      for (int q = 0; q < 4; q++) { const float2 v = Aj[(b + 16 * q) * CLD + tt]; xj[q] = make_float2(w * v.x, w * v.y); }
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          acc[p][q].x = fmaf(xi[p].x, xj[q].x, fmaf(xi[p].y, xj[q].y, acc[p][q].x));
          acc[p][q].y = fmaf(xi[p].y, xj[q].x, fmaf(-xi[p].x, xj[q].y, acc[p][q].y));
        }
    }
  }
  float2* Rk = R + ((long)s * K + k) * N * N;
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = ti * 64 + a + 16 * p, col = tj * 64 + b + 16 * q;
      if (row < N && col < N) {
        float2* ptr = Rk + (long)row * N + col;
        const float2 old = *ptr;
        *ptr = make_float2(old.x + acc[p][q].x, old.y + acc[p][q].y);
      }
    }
}

// Small arrays (N <= 16): the HERK is HBM-bound (arithmetic intensity N flop/B), a 64x64 MFMA tile would be >= 94 % padding.
// Lanes own frames: a lane loads the N channel samples of its frame (coalesced along t), multiplies out the upper
// triangle into registers and the partial sums are reduced once at the end (wave shuffles, then LDS across waves).
// The N(N+1)/2 pairs are split over PS wave groups so that a lane never holds more than ~40 complex accumulators;
// the 4/PS waves of a group split the frames.  One workgroup per (stream, bin).
template <int N>
__global__ __launch_bounds__(256)
void cov_small_kernel(const float2* __restrict__ X, const float* __restrict__ WT, const float* __restrict__ WF,
                      float2* __restrict__ R, int K, long T_stride, long T)
{
  constexpr int NP = N * (N + 1) / 2;
  constexpr int PS = NP <= 40 ? 1 : (NP <= 80 ? 2 : 4);    // pair groups
  constexpr int PPG = (NP + PS - 1) / PS;                 // pairs per group
  constexpr int TS = 4 / PS;                              // waves splitting the frames inside a group
  __shared__ float2 red[4][PPG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pg = wave % PS, ts = wave / PS;
  const int k = blockIdx.x, s = blockIdx.y;
  const float2* Xk = X + ((long)s * K + k) * N * T_stride;
  const float* wt = WT ? WT + ((long)s * K + k) * T_stride : nullptr;
  const float* wf = WF ? WF + (long)s * T_stride : nullptr;
  float2 acc[PPG];
#pragma unroll
  for (int q = 0; q < PPG; q++) acc[q] = make_float2(0.f, 0.f);
  for (long t = (long)ts * 64 + lane; t < T; t += 64 * TS) {
    float w = 1.f;
    if (wt) w *= wt[t];
    if (wf) w *= wf[t];
    if (w == 0.f) continue;
    float2 x[N];
#pragma unroll
    for (int n = 0; n < N; n++) x[n] = Xk[(long)n * T_stride + t];
    // pairs (i <= j) in row-major order of the upper triangle; this wave owns [pg*PPG, (pg+1)*PPG).  The group index is
    // wave-uniform: one unrolled copy per group keeps every accumulator index a compile-time constant.
#pragma unroll
    for (int g = 0; g < PS; g++) {
      if (g != pg) continue;
      int p = 0;
#pragma unroll
      for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i; j < N; j++, p++) {
          if (p >= g * PPG && p < (g + 1) * PPG) {
            const float xr = w * x[j].x, xi = w * x[j].y;
            acc[p - g * PPG].x += x[i].x * xr + x[i].y * xi;
            acc[p - g * PPG].y += x[i].y * xr - x[i].x * xi;
          }
        }
    }
  }
  // reduce over the 64 lanes, then over the TS waves of the group
#pragma unroll
  for (int q = 0; q < PPG; q++) {
    float ar = acc[q].x, ai = acc[q].y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ar += __shfl_xor(ar, o, 64); ai += __shfl_xor(ai, o, 64); }
    if (lane == 0) red[wave][q] = make_float2(ar, ai);
  }
  __syncthreads();
  float2* Rk = R + ((long)s * K + k) * N * N;
  for (int p = tid; p < NP; p += 256) {
    const int g = p / PPG, q = p % PPG;
    float ar = 0.f, ai = 0.f;
#pragma unroll
    for (int tsx = 0; tsx < TS; tsx++) { const float2 v = red[tsx * PS + g][q]; ar += v.x; ai += v.y; }
    // p -> (i, j) of the upper triangle
    int i = 0, rem = p;
    while (rem >= N - i) { rem -= N - i; i++; }
    const int j = i + rem;
    float2 o1 = Rk[(long)i * N + j];
    Rk[(long)i * N + j] = make_float2(o1.x + ar, o1.y + ai);
    if (i != j) {
      float2 o2 = Rk[(long)j * N + i];
      Rk[(long)j * N + i] = make_float2(o2.x + ar, o2.y - ai);
    }
  }
}

// frame gate of accu_stats_from_label: w[s][t] = (energy > threshold) && label[s][t]
__global__ void cov_gate_kernel(const float* __restrict__ energy, const float* __restrict__ label, long T, long T_stride,
                                float threshold, float* __restrict__ wf, float* __restrict__ count /* [S] */)
{
  const int s = blockIdx.x;
  float cnt = 0.f;
  for (long t = threadIdx.x; t < T; t += blockDim.x) {
    const float w = (energy[(long)s * T_stride + t] > threshold) ? (label ? label[(long)s * T_stride + t] : 1.f) : 0.f;
    wf[(long)s * T_stride + t] = w;
    cnt += w;
  }
  __shared__ float red[256];
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[s] += red[0];
}


// finalize_stats (pybeamformer.py:994-1000, 1249-1263) + improve_matrix_condition (:1200-1207):
// R <- R / count ; if gamma > 0: R <- (R + gamma tr(R)/N I) / (1 + gamma).  One workgroup per (s,k).
__global__ __launch_bounds__(256)
void cov_finalize_kernel(float2* __restrict__ R, const float* __restrict__ count, int count_per_bin,
                         int K, int N, float gamma)
{
  const int k = blockIdx.x, s = blockIdx.y;
  float2* Rk = R + ((long)s * K + k) * N * N;
  const float c = count_per_bin ? count[(long)s * K + k] : count[s];
  const float inv = 1.0f / c;
  __shared__ float red[256];
  float tr = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) tr += Rk[(long)i * N + i].x * inv;
  red[threadIdx.x] = tr;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float load = gamma > 0.f ? gamma * red[0] / (float)N : 0.f;
  const float post = gamma > 0.f ? 1.0f / (1.0f + gamma) : 1.0f;
  for (int idx = threadIdx.x; idx < N * N; idx += 256) {
    float2 v = Rk[idx];
    v.x *= inv; v.y *= inv;
    if (idx / N == idx % N) v.x += load;
    Rk[idx] = make_float2(v.x * post, v.y * post);
  }
}

}  // namespace

extern "C" {

int btk_cov_accumulate(const void* X, const float* tf_weights, const float* frame_weights, void* R,
                       int S, int K, int N, long T_stride, long T, int use_mfma, void* stream)
{
  if (!X || !R) return btk_set_error(BTK_ERR_PARAMETER, "btk_cov_accumulate: null argument");
  if (S <= 0 || K <= 0 || N <= 0 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_cov_accumulate: bad sizes S=%d K=%d N=%d T=%ld", S, K, N, T);
  if (T == 0) return BTK_OK;
  if (N <= 16 && use_mfma != 2) {                        // small arrays: HBM-bound lanes-own-frames kernel (use_mfma == 2 forces the tiled ones)
    const dim3 g2((unsigned)K, (unsigned)S);
    const float2* Xp = static_cast<const float2*>(X);
    float2* Rp = static_cast<float2*>(R);
    hipStream_t st = as_stream(stream);
    switch (N) {
#define BTK_COV_SMALL(NN) case NN: hipLaunchKernelGGL(cov_small_kernel<NN>, g2, dim3(256), 0, st, Xp, tf_weights, frame_weights, Rp, K, T_stride, T); break;
      BTK_COV_SMALL(1) BTK_COV_SMALL(2) BTK_COV_SMALL(3) BTK_COV_SMALL(4) BTK_COV_SMALL(5) BTK_COV_SMALL(6) BTK_COV_SMALL(7) BTK_COV_SMALL(8)
      BTK_COV_SMALL(9) BTK_COV_SMALL(10) BTK_COV_SMALL(11) BTK_COV_SMALL(12) BTK_COV_SMALL(13) BTK_COV_SMALL(14) BTK_COV_SMALL(15) BTK_COV_SMALL(16)
#undef BTK_COV_SMALL
    }
    BTK_HIP_CHECK(hipGetLastError());
    return BTK_OK;
  }
  const int ntile = (N + 63) / 64;
  const size_t lds = sizeof(float2) * 2 * 64 * CLD + sizeof(float) * CT;
  dim3 grid((unsigned)(ntile * ntile), (unsigned)K, (unsigned)S);
  if (use_mfma)
    hipLaunchKernelGGL(cov_mfma_kernel, grid, dim3(256), lds, as_stream(stream), static_cast<const float2*>(X),
                       tf_weights, frame_weights, static_cast<float2*>(R), K, N, T_stride, T, ntile);
  else
    hipLaunchKernelGGL(cov_valu_kernel, grid, dim3(256), lds, as_stream(stream), static_cast<const float2*>(X),
                       tf_weights, frame_weights, static_cast<float2*>(R), K, N, T_stride, T, ntile);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_cov_frame_gate(const float* energy, const float* label, int S, long T_stride, long T,
                       float energy_threshold, float* frame_weights, float* frame_count, void* stream)
{
  if (!energy || !frame_weights || !frame_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_cov_frame_gate: null argument");
  if (S <= 0 || T < 0 || T_stride < T) return btk_set_error(BTK_ERR_DIMENSION, "btk_cov_frame_gate: bad sizes");
  if (T == 0) return BTK_OK;
  hipLaunchKernelGGL(cov_gate_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), energy, label, T, T_stride,
                     energy_threshold, frame_weights, frame_count);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_cov_finalize(void* R, const float* count, int count_per_bin, int S, int K, int N, float gamma, void* stream)
{
  if (!R || !count) return btk_set_error(BTK_ERR_PARAMETER, "btk_cov_finalize: null argument");
  if (S <= 0 || K <= 0 || N <= 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_cov_finalize: bad sizes");
  hipLaunchKernelGGL(cov_finalize_kernel, dim3((unsigned)K, (unsigned)S), dim3(256), 0, as_stream(stream),
                     static_cast<float2*>(R), count, count_per_bin, K, N, gamma);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // extern "C"
