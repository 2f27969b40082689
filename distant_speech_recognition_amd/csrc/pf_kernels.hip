// pf_kernels.hip -- Zelinski post-filter for gfx950.
//
// Replaces ZelinskiFilter_f / ZelinskiFilter / ZelinskiPostFilter::next
// (reference postfilter/postfilter.cc:57-219, 424-491).
//
// The reference keeps N(N-1)/2 cross spectral densities phi_ij per bin and updates each of them
// per frame (O(N^2)); only their SUM and the sum of the auto spectral densities enter the gain:
//     W = clamp( f(sum_{i<j} phi_ij) / sum_i phi_ii * 2/(N-1), 1e-4, 1 ),  f = |.| or max(Re .,0)
// Every phi obeys the same first-order recursion with the same alpha, so the sums obey it too:
//     Phi_t = a_t Phi_{t-1} + b_t c_t,   c_t = sum_{i<j} x'_i conj(x'_j),   x'_i = conj(d_i) x_i
//     Psi_t = a_t Psi_{t-1} + b_t e_t,   e_t = sum_i |x'_i|^2
// (a_t, b_t) = (0, 1) while frame_no_ <= 0, i.e. for the first two frames (postfilter.cc:460-463),
// (alpha, 1-alpha) afterwards.  c_t is an O(N) prefix-sum form: sum_j conj(x'_j) sum_{i<j} x'_i.
//
//   1. bf_apply_stats_kernel : one pass over X (lanes own frames, channels in registers):
//                              y_t = w^H x_t, c_t, e_t  -- the snapshot is read ONCE for beamformer
//                              and post-filter together.
//   2. zelinski_iir_kernel   : one wavefront per (stream, bin) row; 64-frame chunks scanned with a
//                              Hillis-Steele linear-recurrence scan; gain applied to Y in place.
#include "btk_internal.h"

namespace {

constexpr int PF_NT = 256;

__global__ __launch_bounds__(PF_NT)
void bf_apply_stats_kernel(const float2* __restrict__ W, long w_stream_stride,
                           const float2* __restrict__ Dv /* alignment vector d, same layout as W */,
                           const float2* __restrict__ X, float2* __restrict__ Y,
                           float2* __restrict__ Cc, float* __restrict__ Ee,
                           int K, int N, long T_stride, long T)
{
  const int k = blockIdx.y, s = blockIdx.z;
  const long t = (long)blockIdx.x * PF_NT + threadIdx.x;
  if (t >= T) return;
  const float2* w = W + s * w_stream_stride + (long)k * N;
  const float2* d = Dv + s * w_stream_stride + (long)k * N;
  const float2* x = X + ((long)s * K + k) * N * T_stride + t;
  float yr = 0.f, yi = 0.f, pr = 0.f, pi = 0.f, cr = 0.f, ci = 0.f, e = 0.f;
#pragma unroll 8
  for (int n = 0; n < N; n++) {
    const float2 v = x[(long)n * T_stride];
    const float2 wn = w[n], dn = d[n];
    yr = fmaf(wn.x, v.x, fmaf(wn.y, v.y, yr));                  // conj(w) x
    yi = fmaf(wn.x, v.y, fmaf(-wn.y, v.x, yi));
    const float ar = fmaf(dn.x, v.x, dn.y * v.y);               // x' = conj(d) x
    const float ai = fmaf(dn.x, v.y, -dn.y * v.x);
    cr = fmaf(pr, ar, fmaf(pi, ai, cr));                        // prefix * conj(x')
    ci = fmaf(pi, ar, fmaf(-pr, ai, ci));
    pr += ar; pi += ai;
    e = fmaf(ar, ar, fmaf(ai, ai, e));
  }
  const long o = ((long)s * K + k) * T_stride + t;
  Y[o] = make_float2(yr, yi);
  Cc[o] = make_float2(cr, ci);
  Ee[o] = e;
}

// One wavefront per (s,k).  frame_base = number of frames this post-filter has already produced
// (ZelinskiPostFilter::frame_no_ + 1).
__global__ __launch_bounds__(64)
void zelinski_iir_kernel(float2* __restrict__ Y, const float2* __restrict__ Cc, const float* __restrict__ Ee,
                         int K, int N, long T_stride, long T, float alpha, int type, int min_frames,
                         long frame_base, float2* __restrict__ Phi /* [S][K] */, float* __restrict__ Psi /* [S][K] */,
                         float* __restrict__ Wlast /* [S][K] */)
{
  const int k = blockIdx.x, s = blockIdx.y;
  const int lane = threadIdx.x;
  const long row = ((long)s * K + k);
  float2 phi_c = Phi[row];
  float psi_c = Psi[row];
  float wlast = Wlast[row];
  const float scale = 2.0f / ((float)N - 1.0f);
  for (long t0 = 0; t0 < T; t0 += 64) {
    const long t = t0 + lane;
    const bool ok = t < T;
    const long g = frame_base + t;                              // global frame index; frame_no_ before increment = g-1
    float a = ok ? ((g >= 2) ? alpha : 0.f) : 1.f;              // identity element beyond the end
    const float bsc = (g >= 2 && alpha > 0.f) ? 1.f - alpha : 1.f;
    if (alpha <= 0.f) a = ok ? 0.f : 1.f;                       // calc_CSD_: alpha <= 0 -> no memory
    float2 c = ok ? Cc[row * T_stride + t] : make_float2(0.f, 0.f);
    float e = ok ? Ee[row * T_stride + t] : 0.f;
    float br = ok ? bsc * c.x : 0.f, bi = ok ? bsc * c.y : 0.f, be = ok ? bsc * e : 0.f;
    // inclusive scan of the affine maps v -> a v + b
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const float a2 = __shfl_up(a, dlt, 64);
      const float r2 = __shfl_up(br, dlt, 64), i2 = __shfl_up(bi, dlt, 64), e2 = __shfl_up(be, dlt, 64);
      if (lane >= dlt) {
        br = fmaf(a, r2, br); bi = fmaf(a, i2, bi); be = fmaf(a, e2, be);
        a *= a2;
      }
    }
    const float phr = fmaf(a, phi_c.x, br), phim = fmaf(a, phi_c.y, bi), ps = fmaf(a, psi_c, be);
    if (ok) {
      const bool apply = (g - 1) >= (long)min_frames;           // frame_no_ (= g-1, pre-increment) < min_frames -> NO_USE_POST_FILTER
      const int pft = apply ? type : 0;
      float num = (pft & 1) ? fmaxf(phr, 0.f) : sqrtf(phr * phr + phim * phim);
      float Wf = (num / ps) * scale;
      if (Wf >= 1.0f) Wf = 1.0f;
      if (Wf < 1.0e-4f) Wf = 1.0e-4f;
      if (apply) {
        float2 y = Y[row * T_stride + t];
        Y[row * T_stride + t] = make_float2(Wf * y.x, Wf * y.y);
      }
      wlast = Wf;
    }
    // carry = state after the last valid frame of the chunk
    const int last = (T - t0) >= 64 ? 63 : (int)(T - t0) - 1;
    phi_c = make_float2(__shfl(phr, last, 64), __shfl(phim, last, 64));
    psi_c = __shfl(ps, last, 64);
    wlast = __shfl(wlast, last, 64);
  }
  if (lane == 0) { Phi[row] = phi_c; Psi[row] = psi_c; Wlast[row] = wlast; }
}

}  // namespace

extern "C" {

int btk_bf_apply_stats(const void* W, const void* D, int per_stream_weights, const void* X, void* Y,
                       void* C, float* E, int S, int K, int N, long T_stride, long T, void* stream)
{
  if (!W || !D || !X || !Y || !C || !E) return btk_set_error(BTK_ERR_PARAMETER, "btk_bf_apply_stats: null argument");
  if (S <= 0 || K <= 0 || N <= 0 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_bf_apply_stats: bad sizes S=%d K=%d N=%d T=%ld", S, K, N, T);
  if (T == 0) return BTK_OK;
  const long wss = per_stream_weights ? (long)K * N : 0;
  dim3 grid((unsigned)((T + PF_NT - 1) / PF_NT), (unsigned)K, (unsigned)S);
  hipLaunchKernelGGL(bf_apply_stats_kernel, grid, dim3(PF_NT), 0, as_stream(stream),
                     static_cast<const float2*>(W), wss, static_cast<const float2*>(D),
                     static_cast<const float2*>(X), static_cast<float2*>(Y), static_cast<float2*>(C), E,
                     K, N, T_stride, T);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_zelinski_process(void* Y, const void* C, const float* E, int S, int K, int N, long T_stride, long T,
                         double alpha, int type, int min_frames, long frames_done,
                         void* phi_state, float* psi_state, float* w_last, void* stream)
{
  if (!Y || !C || !E || !phi_state || !psi_state || !w_last)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_zelinski_process: null argument");
  if (N <= 1) return btk_set_error(BTK_ERR_DIMENSION, "The number of channels %d is <= 1 ", N);
  if (S <= 0 || K <= 0 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_zelinski_process: bad sizes S=%d K=%d T=%ld", S, K, T);
  if (T == 0) return BTK_OK;
  hipLaunchKernelGGL(zelinski_iir_kernel, dim3((unsigned)K, (unsigned)S), dim3(64), 0, as_stream(stream),
                     static_cast<float2*>(Y), static_cast<const float2*>(C), E, K, N, T_stride, T,
                     (float)alpha, type, min_frames, frames_done, static_cast<float2*>(phi_state), psi_state, w_last);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // extern "C"
