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

  for (long t0 = 0; t0 < T; t0 += CT) {
    __syncthreads();
    stage_tile(Xk, N, T_stride, T, ti * 64, t0, Ai, tid, 256);
    stage_tile(Xk, N, T_stride, T, tj * 64, t0, Aj, tid, 256);
    if (tid < CT) wrow[tid] = tile_weight(wt, wf, t0 + tid, T);
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < CT; kk += 2) {
      const float2 a = Ai[(qi * 32 + li) * CLD + kk + lk];
      float2 b = Aj[(qj * 32 + li) * CLD + kk + lk];
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
