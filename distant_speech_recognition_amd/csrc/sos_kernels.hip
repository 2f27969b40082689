// Batch second-order-statistics beamformers (gfx950): the weight design that follows covariance accumulation.
//
// Replaces (lib/pybeamformer.py)
//   SubbandSOSBatchBeamformer.accu_stats_from_tfmask  per-bin frame counters            (:1127-1147)
//   SubbandBlindMVDRBeamformer.calc_beamformer_weights  w^H = conj(inv(Rn) Rt u / (offset + tr(inv(Rn) Rt)))  (:1225-1247)
//   SubbandGEVBeamformer.finalize_stats                 Rn <- Rn / (tr(Rn)/N)                                   (:1326)
//   SubbandGEVBeamformer.calc_beamformer_weights        principal generalised eigenvector of (Rt, Rn), phase aligned
//                                                        bin to bin, conjugated                                  (:1280-1303)
// One workgroup per frequency bin, float64 arithmetic (the reference uses numpy.linalg.inv / scipy.linalg.eigh in
// float64); matrices live in LDS when they fit and in an L2-resident scratch otherwise.  These are one-off O(K N^3)
// design steps -- the per-frame work stays in btk_bf_apply.
//
// GEV: Rn = L L^H, C = L^-1 Rt L^-H (Hermitian PSD), principal eigenvector of C by repeated squaring
// C <- C^2 / tr(C^2) (32 squarings = power 2^32: separates eigenvalue ratios up to 1 - 1e-9), v = the column of
// the resulting rank-one matrix with the largest diagonal, w = L^-H v / |v|  (w^H Rn w = 1 like scipy's eigh).
#include "btk_internal.h"

namespace {

struct zd { double x, y; };
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ zd zmk(double x, double y) { zd r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ zd zconj(zd a) { return zmk(a.x, -a.y); }
__device__ __forceinline__ zd zscale(zd a, double s) { return zmk(a.x * s, a.y * s); }
// c + a b
__device__ __forceinline__ zd zfma(zd a, zd b, zd c)
{
  c.x = fma(a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
  c.y = fma(a.x, b.y, c.y); c.y = fma(a.y, b.x, c.y);
  return c;
}
// c - a b
__device__ __forceinline__ zd zfms(zd a, zd b, zd c) { return zfma(zmk(-a.x, -a.y), b, c); }

constexpr int SOS_NT = 256;
constexpr int SOS_TAIL = 2 * SOS_NT * 8 + 64;      // reduction scratch ahead of the matrices in dynamic LDS

__global__ __launch_bounds__(256)
void cov_mask_count_kernel(const float* __restrict__ tf, const float* __restrict__ fw, int K, long T_stride, long T,
                           float* __restrict__ count)
{
  __shared__ float red[256];
  const int k = blockIdx.x, s = blockIdx.y;
  const float* m = tf + ((long)s * K + k) * T_stride;
  float acc = 0.f;
  for (long t = threadIdx.x; t < T; t += 256) {
    const float v = m[t];
    // the reference's counters are integer arrays: count[m] += mask truncates (pybeamformer.py:1127-1128, :1140, :1144)
    if (v > 0.f) acc += floorf(v) * (fw ? fw[(long)s * T_stride + t] : 1.f);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[(long)s * K + k] += red[0];
}

__global__ __launch_bounds__(64)
void cov_trace_normalize_kernel(float2* __restrict__ R, int N)
{
  float2* Rk = R + (long)blockIdx.x * N * N;
  double tr = 0.0, ti = 0.0;
  for (int i = 0; i < N; i++) { tr += Rk[(long)i * N + i].x; ti += Rk[(long)i * N + i].y; }
  // x / (trace / N) with a complex trace, as numpy does it
  const double dr = tr / N, di = ti / N, dn = dr * dr + di * di;
  __syncthreads();
  for (int e = threadIdx.x; e < N * N; e += 64) {
    const float2 v = Rk[e];
    Rk[e] = make_float2((float)((v.x * dr + v.y * di) / dn), (float)((v.y * dr - v.x * di) / dn));
  }
}

// in-place lower Cholesky of the Hermitian matrix A [N][N]; returns false (uniformly) if a pivot is not positive
__device__ bool cholesky_lower(zd* A, int N, int tid, volatile int* bad)
{
  if (tid == 0) *bad = 0;
  __syncthreads();
  for (int j = 0; j < N; j++) {
    if (tid == 0) {
      const double piv = A[(long)j * N + j].x;
      if (!(piv > 0.0)) *bad = 1;
      else A[(long)j * N + j] = zmk(sqrt(piv), 0.0);
    }
    __syncthreads();
    if (*bad) return false;
    const double inv = 1.0 / A[(long)j * N + j].x;
    for (int i = j + 1 + tid; i < N; i += SOS_NT) A[(long)i * N + j] = zscale(A[(long)i * N + j], inv);
    __syncthreads();
    const int rem = N - j - 1;
    for (int idx = tid; idx < rem * rem; idx += SOS_NT) {
      const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
      if (c <= i) A[(long)i * N + c] = zfms(A[(long)i * N + j], zconj(A[(long)c * N + j]), A[(long)i * N + c]);
    }
    __syncthreads();
  }
  return true;
}

// B <- L^-1 B (column c per thread), L lower triangular with real positive diagonal
__device__ void forward_solve_cols(const zd* L, zd* B, int N, int tid)
{
  for (int c = tid; c < N; c += SOS_NT) {
    for (int i = 0; i < N; i++) {
      zd a = B[(long)i * N + c];
      for (int k = 0; k < i; k++) a = zfms(L[(long)i * N + k], B[(long)k * N + c], a);
      B[(long)i * N + c] = zscale(a, 1.0 / L[(long)i * N + i].x);
    }
  }
  __syncthreads();
}
// B <- L^-H B
__device__ void backward_solve_cols(const zd* L, zd* B, int N, int tid)
{
  for (int c = tid; c < N; c += SOS_NT) {
    for (int i = N - 1; i >= 0; i--) {
      zd a = B[(long)i * N + c];
      for (int k = i + 1; k < N; k++) a = zfms(zconj(L[(long)k * N + i]), B[(long)k * N + c], a);
      B[(long)i * N + c] = zscale(a, 1.0 / L[(long)i * N + i].x);
    }
  }
  __syncthreads();
}

// wqH[k] = conj( Z[:, ref] / (offset + tr Z) ),  Z = inv(Rn) Rt
template <bool IN_LDS>
__global__ __launch_bounds__(SOS_NT)
void bmvdr_weights_kernel(const float2* __restrict__ Rt, const float2* __restrict__ Rn, int N, int ref_mic, double offset,
                          float2* __restrict__ WqH, zd* __restrict__ scratch, int* __restrict__ fail_count)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];       // all LDS carved from the dynamic region (guide G17)
  const int k = blockIdx.x, tid = threadIdx.x;
  double* red_r = reinterpret_cast<double*>(smem);                  // [SOS_NT]
  double* red_i = red_r + SOS_NT;                                   // [SOS_NT]
  volatile int& bad = *reinterpret_cast<volatile int*>(red_i + SOS_NT);
  zd* L = IN_LDS ? reinterpret_cast<zd*>(smem + SOS_TAIL) : scratch + (long)k * 2 * N * N;
  zd* Z = L + (long)N * N;
  for (int e = tid; e < N * N; e += SOS_NT) {
    const float2 a = Rn[(long)k * N * N + e], b = Rt[(long)k * N * N + e];
    L[e] = zmk(a.x, a.y);
    Z[e] = zmk(b.x, b.y);
  }
  __syncthreads();
  if (!cholesky_lower(L, N, tid, &bad)) {
    if (tid == 0) atomicAdd(fail_count, 1);                       // "Matrix inversion failed" (pybeamformer.py:1246-1247)
    for (int c = tid; c < N; c += SOS_NT) WqH[(long)k * N + c] = make_float2(0.f, 0.f);
    return;
  }
  forward_solve_cols(L, Z, N, tid);
  backward_solve_cols(L, Z, N, tid);
  double pr = 0.0, pi = 0.0;
  for (int i = tid; i < N; i += SOS_NT) { pr += Z[(long)i * N + i].x; pi += Z[(long)i * N + i].y; }
  red_r[tid] = pr; red_i[tid] = pi;
  __syncthreads();
  for (int o = SOS_NT / 2; o > 0; o >>= 1) {
    if (tid < o) { red_r[tid] += red_r[tid + o]; red_i[tid] += red_i[tid + o]; }
    __syncthreads();
  }
  const double dr = offset + red_r[0], di = red_i[0], dn = dr * dr + di * di;
  for (int c = tid; c < N; c += SOS_NT) {
    const zd z = Z[(long)c * N + ref_mic];
    const double qr = (z.x * dr + z.y * di) / dn, qi = (z.y * dr - z.x * di) / dn;
    WqH[(long)k * N + c] = make_float2((float)qr, (float)-qi);
  }
}

constexpr int GEV_SQUARINGS = 32;

// V[k] = principal generalised eigenvector of (Rt_k, Rn_k), normalised v^H Rn v = 1, complex128 (phase not yet aligned)
template <bool IN_LDS>
__global__ __launch_bounds__(SOS_NT)
void gev_vectors_kernel(const float2* __restrict__ Rt, const float2* __restrict__ Rn, int N,
                        zd* __restrict__ V, zd* __restrict__ scratch, int* __restrict__ fail_count)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, tid = threadIdx.x;
  double* red = reinterpret_cast<double*>(smem);                    // [SOS_NT]
  int* red_j = reinterpret_cast<int*>(red + SOS_NT);                // [SOS_NT]
  volatile int& bad = *reinterpret_cast<volatile int*>(red_j + SOS_NT);
  zd* L = scratch + (long)k * 3 * N * N;                          // L always in the (L2-resident) scratch
  zd* C = IN_LDS ? reinterpret_cast<zd*>(smem + SOS_TAIL) : L + (long)N * N;
  zd* C2 = C + (long)N * N;
  for (int e = tid; e < N * N; e += SOS_NT) {
    const float2 a = Rn[(long)k * N * N + e], b = Rt[(long)k * N * N + e];
    L[e] = zmk(a.x, a.y);
    C[e] = zmk(b.x, b.y);
  }
  __syncthreads();
  if (!cholesky_lower(L, N, tid, &bad)) {
    if (tid == 0) atomicAdd(fail_count, 1);                       // "GEV failed" (pybeamformer.py:1296-1297)
    for (int c = tid; c < N; c += SOS_NT) V[(long)k * N + c] = zmk(0.0, 0.0);
    return;
  }
  // C <- L^-1 Rt L^-H :  Y = L^-1 Rt, then C = (L^-1 Y^H)^H
  forward_solve_cols(L, C, N, tid);
  for (int e = tid; e < N * N; e += SOS_NT) { const int i = e / N, j = e % N; C2[(long)j * N + i] = zconj(C[e]); }
  __syncthreads();
  forward_solve_cols(L, C2, N, tid);
  for (int e = tid; e < N * N; e += SOS_NT) { const int i = e / N, j = e % N; C[(long)j * N + i] = zconj(C2[e]); }
  __syncthreads();
  // symmetrise (rounding) and scale to unit trace
  auto normalise = [&](zd* A) {
    double p = 0.0;
    for (int i = tid; i < N; i += SOS_NT) p += A[(long)i * N + i].x;
    red[tid] = p;
    __syncthreads();
    for (int o = SOS_NT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double inv = red[0] > 0.0 ? 1.0 / red[0] : 0.0;
    __syncthreads();
    for (int e = tid; e < N * N; e += SOS_NT) A[e] = zscale(A[e], inv);
    __syncthreads();
  };
  for (int e = tid; e < N * N; e += SOS_NT) {
    const int i = e / N, j = e % N;
    if (i < j) {
      const zd a = C[(long)i * N + j], b = C[(long)j * N + i];
      const zd m = zmk(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
      C[(long)i * N + j] = m; C[(long)j * N + i] = zconj(m);
    } else if (i == j) C[e].y = 0.0;
  }
  __syncthreads();
  normalise(C);
  zd* src = C;
  zd* dst = C2;
  for (int it = 0; it < GEV_SQUARINGS; it++) {
    // dst = src src^H = src^2 (Hermitian): only i <= j computed, mirrored
    if (N >= 16) {
      // round 4: on the float64 matrix cores (v_mfma_f64_16x16x4_f64: A[i][kk] from lane i + 16 kk, B[kk][j] from lane j + 16 kk, register v of
      // lane l = D[4 v + l / 16][l % 16] -- profiles/ubench/mfma_f64_layout.hip; the float32 instruction of the same shape has
      // D[4 (l / 16) + v][l % 16]): 16 x 16 tiles of the upper triangle dealt to the four wavefronts, re = ar br + ai bi, im = ai br - ar bi.
      // 64 microphones: 640 matrix instructions per squaring instead of 133 000 dependent float64 multiply-adds out of LDS (10.3 -> 0.9 ms
      // for 257 bins).
      const int nt = (N + 15) >> 4, lane = tid & 63, wave = tid >> 6, mi = lane & 15, mg = lane >> 4;
      for (int t = wave; t < nt * (nt + 1) / 2; t += SOS_NT / 64) {
        int ti = 0, rem = t;
        while (rem >= nt - ti) { rem -= nt - ti; ti++; }
        const int tj = ti + rem;
        const int ra = 16 * ti + mi, rb = 16 * tj + mi;
        d4 cr = {0.0, 0.0, 0.0, 0.0}, ci = {0.0, 0.0, 0.0, 0.0};
        for (int q0 = 0; q0 < N; q0 += 4) {
          const int q = q0 + mg;
          const zd a = (ra < N && q < N) ? src[(long)ra * N + q] : zmk(0.0, 0.0);
          const zd b = (rb < N && q < N) ? src[(long)rb * N + q] : zmk(0.0, 0.0);
          cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, cr, 0, 0, 0);
          cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, cr, 0, 0, 0);
          ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, ci, 0, 0, 0);
          ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.x, b.y, ci, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int i = 16 * ti + 4 * v + mg, j = 16 * tj + mi;
          if (i < N && j < N && i <= j) {
            const zd a = zmk(cr[v], (i == j) ? 0.0 : ci[v]);
            dst[(long)i * N + j] = a;
            if (i != j) dst[(long)j * N + i] = zconj(a);
          }
        }
      }
    } else {
      for (int e = tid; e < N * N; e += SOS_NT) {
        const int i = e / N, j = e % N;
        if (i <= j) {
          zd a = zmk(0.0, 0.0);
          for (int q = 0; q < N; q++) a = zfma(src[(long)i * N + q], zconj(src[(long)j * N + q]), a);
          if (i == j) a.y = 0.0;
          dst[(long)i * N + j] = a;
          if (i != j) dst[(long)j * N + i] = zconj(a);
        }
      }
    }
    __syncthreads();
    normalise(dst);
    zd* t = src; src = dst; dst = t;
  }
  // column with the largest diagonal entry
  double best = -1.0; int bj = 0;
  for (int i = tid; i < N; i += SOS_NT) { const double dgn = src[(long)i * N + i].x; if (dgn > best) { best = dgn; bj = i; } }
  red[tid] = best; red_j[tid] = bj;
  __syncthreads();
  for (int o = SOS_NT / 2; o > 0; o >>= 1) {
    if (tid < o && (red[tid + o] > red[tid] || (red[tid + o] == red[tid] && red_j[tid + o] < red_j[tid]))) {
      red[tid] = red[tid + o]; red_j[tid] = red_j[tid + o];
    }
    __syncthreads();
  }
  const int jstar = red_j[0];
  __syncthreads();
  // y = column jstar (unit norm), then w = L^-H y: reuse dst's first column as a 1-column right-hand side
  double p = 0.0;
  for (int i = tid; i < N; i += SOS_NT) { const zd a = src[(long)i * N + jstar]; p += a.x * a.x + a.y * a.y; }
  red[tid] = p;
  __syncthreads();
  for (int o = SOS_NT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const double invn = 1.0 / sqrt(red[0]);
  __syncthreads();
  for (int i = tid; i < N; i += SOS_NT) dst[i] = zscale(src[(long)i * N + jstar], invn);
  __syncthreads();
  if (tid == 0) {
    for (int i = N - 1; i >= 0; i--) {
      zd a = dst[i];
      for (int q = i + 1; q < N; q++) a = zfms(zconj(L[(long)q * N + i]), dst[q], a);
      dst[i] = zscale(a, 1.0 / L[(long)i * N + i].x);
    }
  }
  __syncthreads();
  for (int i = tid; i < N; i += SOS_NT) V[(long)k * N + i] = dst[i];
}

// sequential over bins: bin 0 rotated to make its largest component real positive; bin m aligned to bin m-1
// (pybeamformer.py:1299-1301), then everything conjugated (:1304-1306)
__global__ __launch_bounds__(64)
void gev_phase_align_kernel(zd* __restrict__ V, int K, int N, float2* __restrict__ WqH)
{
  const int lane = threadIdx.x;
  for (int m = 0; m < K; m++) {
    zd* v = V + (long)m * N;
    double rx, ry;
    if (m == 0) {
      double best = -1.0; int bi = 0;
      for (int i = 0; i < N; i++) { const double a = v[i].x * v[i].x + v[i].y * v[i].y; if (a > best) { best = a; bi = i; } }
      const double a = sqrt(best);
      rx = a > 0.0 ? v[bi].x / a : 1.0; ry = a > 0.0 ? -v[bi].y / a : 0.0;          // multiply by conj(phase)
    } else {
      const zd* u = V + (long)(m - 1) * N;
      double pr = 0.0, pi = 0.0;
      for (int i = lane; i < N; i += 64) {                        // inner(w[m], conj(w[m-1]))
        pr += v[i].x * u[i].x + v[i].y * u[i].y;
        pi += v[i].y * u[i].x - v[i].x * u[i].y;
      }
      for (int o = 32; o > 0; o >>= 1) { pr += __shfl_xor(pr, o, 64); pi += __shfl_xor(pi, o, 64); }
      const double a = sqrt(pr * pr + pi * pi);
      rx = a > 0.0 ? pr / a : 1.0; ry = a > 0.0 ? -pi / a : 0.0;  // exp(-j angle)
    }
    __syncthreads();
    for (int i = lane; i < N; i += 64) {
      const zd a = v[i];
      v[i] = zmk(a.x * rx - a.y * ry, a.x * ry + a.y * rx);
    }
    __syncthreads();
  }
  for (long e = lane; e < (long)K * N; e += 64) WqH[e] = make_float2((float)V[e].x, (float)-V[e].y);
}

}  // namespace

extern "C" {

int btk_cov_mask_count(const float* tf_weights, const float* frame_weights, int S, int K, long T_stride, long T,
                       float* count, void* stream)
{
  if (!tf_weights || !count) return btk_set_error(BTK_ERR_PARAMETER, "btk_cov_mask_count: null argument");
  if (S <= 0 || K <= 0 || T < 0 || T_stride < T) return btk_set_error(BTK_ERR_DIMENSION, "btk_cov_mask_count: bad sizes");
  if (T == 0) return BTK_OK;
  hipLaunchKernelGGL(cov_mask_count_kernel, dim3((unsigned)K, (unsigned)S), dim3(256), 0, as_stream(stream),
                     tf_weights, frame_weights, K, T_stride, T, count);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_cov_trace_normalize(void* R, int nbins, int N, void* stream)
{
  if (!R) return btk_set_error(BTK_ERR_PARAMETER, "btk_cov_trace_normalize: null argument");
  if (nbins <= 0 || N <= 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_cov_trace_normalize: bad sizes");
  hipLaunchKernelGGL(cov_trace_normalize_kernel, dim3((unsigned)nbins), dim3(64), 0, as_stream(stream), static_cast<float2*>(R), N);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

long btk_sos_scratch_bytes(int nbins, int N)
{
  return (long)sizeof(double) * 2 * ((long)nbins * 3 * N * N + (long)nbins * N) + 64;
}

int btk_bmvdr_weights(const void* Rt, const void* Rn, int nbins, int N, int ref_mic, double offset, void* WqH,
                      void* scratch, int* fail_count, void* stream)
{
  if (!Rt || !Rn || !WqH || !scratch || !fail_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_bmvdr_weights: null argument");
  if (nbins <= 0 || N <= 0 || ref_mic < 0 || ref_mic >= N)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_bmvdr_weights: bad sizes nbins=%d N=%d ref_mic=%d", nbins, N, ref_mic);
  if (!(offset >= 0.0 && offset <= 1.0)) return btk_set_error(BTK_ERR_PARAMETER, "The offset value %f is out of [0, 1]", offset);
  const size_t lds = SOS_TAIL + sizeof(zd) * 2 * (size_t)N * N;
  zd* sc = static_cast<zd*>(scratch);
  if (lds <= 140 * 1024) {
    auto kern = bmvdr_weights_kernel<true>;
    if (lds > 48 * 1024)
      BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nbins), dim3(SOS_NT), lds, as_stream(stream), static_cast<const float2*>(Rt),
                       static_cast<const float2*>(Rn), N, ref_mic, offset, static_cast<float2*>(WqH), sc, fail_count);
  } else {
    hipLaunchKernelGGL(bmvdr_weights_kernel<false>, dim3((unsigned)nbins), dim3(SOS_NT), SOS_TAIL, as_stream(stream),
                       static_cast<const float2*>(Rt), static_cast<const float2*>(Rn), N, ref_mic, offset,
                       static_cast<float2*>(WqH), sc, fail_count);
  }
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_gev_weights(const void* Rt, const void* Rn, int K, int N, void* WqH, void* scratch, int* fail_count, void* stream)
{
  if (!Rt || !Rn || !WqH || !scratch || !fail_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_gev_weights: null argument");
  if (K <= 0 || N <= 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_gev_weights: bad sizes K=%d N=%d", K, N);
  zd* sc = static_cast<zd*>(scratch);
  zd* V = sc + (long)K * 3 * N * N;
  const size_t lds = SOS_TAIL + sizeof(zd) * 2 * (size_t)N * N;
  if (lds <= 140 * 1024) {
    auto kern = gev_vectors_kernel<true>;
    if (lds > 48 * 1024)
      BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)K), dim3(SOS_NT), lds, as_stream(stream), static_cast<const float2*>(Rt),
                       static_cast<const float2*>(Rn), N, V, sc, fail_count);
  } else {
    hipLaunchKernelGGL(gev_vectors_kernel<false>, dim3((unsigned)K), dim3(SOS_NT), SOS_TAIL, as_stream(stream),
                       static_cast<const float2*>(Rt), static_cast<const float2*>(Rn), N, V, sc, fail_count);
  }
  BTK_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(gev_phase_align_kernel, dim3(1), dim3(64), 0, as_stream(stream), V, K, N, static_cast<float2*>(WqH));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // extern "C"
