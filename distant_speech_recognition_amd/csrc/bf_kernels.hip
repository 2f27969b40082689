// bf_kernels.hip -- fixed-weight subband beamformer apply for gfx950.
//
// y_k[t] = w_k^H x_k[t] for every stream s, bin k <= M/2 and frame t: the per-bin zdotc loop of
// SubbandDS::next / SubbandGSC::next / SubbandMVDR::next (reference beamformer/beamformer.cc
// :1132-1151, :1298-1310, :2566-2581) batched over frames.
//
// HBM-bound: 8 K (N+1) algorithmic bytes per frame.  X[S][K][N][T] keeps frames contiguous, so
// for a fixed (s,k,n) a wavefront streams 64 lanes x 16 B = 1 KiB of consecutive frames per load
// instruction; lanes own frames, the channel loop runs in registers, and the weights of the bin
// are wave-uniform (scalar loads).  No cross-lane reduction, no LDS.
#include "btk_internal.h"

namespace {

constexpr int BF_NT = 256;
constexpr int UNROLL = 8;

// VEC = 2: each lane owns two consecutive frames (float4 loads), requires even T_stride.
template <int VEC>
__global__ __launch_bounds__(BF_NT)
void bf_apply_kernel(const float2* __restrict__ W, long w_stream_stride, const float2* __restrict__ X,
                     float2* __restrict__ Y, int K, int N, long T_stride, long T)
{
  const int k = blockIdx.y, s = blockIdx.z;
  const long t = ((long)blockIdx.x * BF_NT + threadIdx.x) * VEC;
  if (t >= T) return;
  const float2* w = W + s * w_stream_stride + (long)k * N;
  const float2* x = X + ((long)s * K + k) * N * T_stride + t;
  float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
  int n = 0;
  for (; n + UNROLL <= N; n += UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      // (non-temporal loads and stores: every snapshot is read once, Y is read by another kernel; C0 launch 2.91 -> 2.77 ms by the
      //  loads, a further 3 % by the stores: profiles/r06_nt_hints.txt)
      if (VEC == 2) v[u] = btk_ld<true>(reinterpret_cast<const float4*>(x + (long)(n + u) * T_stride));
      else { const float2 q = x[(long)(n + u) * T_stride]; v[u] = make_float4(q.x, q.y, 0.f, 0.f); }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const float2 wn = w[n + u];
      // conj(w) * x
      acc0.x = fmaf(wn.x, v[u].x, fmaf(wn.y, v[u].y, acc0.x));
      acc0.y = fmaf(wn.x, v[u].y, fmaf(-wn.y, v[u].x, acc0.y));
      if (VEC == 2) {
        acc1.x = fmaf(wn.x, v[u].z, fmaf(wn.y, v[u].w, acc1.x));
        acc1.y = fmaf(wn.x, v[u].w, fmaf(-wn.y, v[u].z, acc1.y));
      }
    }
  }
  for (; n < N; n++) {
    const float2 wn = w[n];
    if (VEC == 2) {
      const float4 q = *reinterpret_cast<const float4*>(x + (long)n * T_stride);
      acc0.x = fmaf(wn.x, q.x, fmaf(wn.y, q.y, acc0.x));
      acc0.y = fmaf(wn.x, q.y, fmaf(-wn.y, q.x, acc0.y));
      acc1.x = fmaf(wn.x, q.z, fmaf(wn.y, q.w, acc1.x));
      acc1.y = fmaf(wn.x, q.w, fmaf(-wn.y, q.z, acc1.y));
    } else {
      const float2 q = x[(long)n * T_stride];
      acc0.x = fmaf(wn.x, q.x, fmaf(wn.y, q.y, acc0.x));
      acc0.y = fmaf(wn.x, q.y, fmaf(-wn.y, q.x, acc0.y));
    }
  }
  float2* y = Y + ((long)s * K + k) * T_stride + t;
  if (VEC == 2) {
    if (t + 1 < T) btk_st<true>(reinterpret_cast<float4*>(y), make_float4(acc0.x, acc0.y, acc1.x, acc1.y));
    else y[0] = acc0;
  } else {
    y[0] = acc0;
  }
}

}  // namespace

extern "C" int btk_bf_apply(const void* W, int per_stream_weights, const void* X, void* Y,
                            int S, int K, int N, long T_stride, long T, void* stream)
{
  if (S <= 0 || K < 0 || N <= 0 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_bf_apply: bad sizes S=%d K=%d N=%d T=%ld T_stride=%ld", S, K, N, T, T_stride);
  if (T == 0 || K == 0) return BTK_OK;                  // K == 0: the empty bin shard of a trailing rank (sharding.py)
  if (!W || !X || !Y) return btk_set_error(BTK_ERR_PARAMETER, "btk_bf_apply: null argument");
  const long wss = per_stream_weights ? (long)K * N : 0;
  const bool vec = (T_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
  if (vec) {
    dim3 grid((unsigned)((T + 2 * BF_NT - 1) / (2 * BF_NT)), (unsigned)K, (unsigned)S);
    hipLaunchKernelGGL(bf_apply_kernel<2>, grid, dim3(BF_NT), 0, as_stream(stream),
                       static_cast<const float2*>(W), wss, static_cast<const float2*>(X),
                       static_cast<float2*>(Y), K, N, T_stride, T);
  } else {
    dim3 grid((unsigned)((T + BF_NT - 1) / BF_NT), (unsigned)K, (unsigned)S);
    hipLaunchKernelGGL(bf_apply_kernel<1>, grid, dim3(BF_NT), 0, as_stream(stream),
                       static_cast<const float2*>(W), wss, static_cast<const float2*>(X),
                       static_cast<float2*>(Y), K, N, T_stride, T);
  }
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}
