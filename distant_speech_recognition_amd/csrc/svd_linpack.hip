// svd_linpack.hip -- the reference's `pseudoinverse() returned false -> identity` decision, taken the reference's way.
//
// pseudoinverse() (beamformer/beamformer.cc:232-289) runs LINPACK's float32 csvdc (matrix/linpack_c.cc:9516) and returns
// false when csvdc's INFO != 0 or a singular value is below the threshold; calc_mvdr_weights (:2379-2384) and
// LefkimmiatisPostFilter::calc_inverse_noise_spatial_spectral_matrix (postfilter/postfilter.cc:967-980) then use the
// identity.  csvdc_values_kernel is csvdc with job = 0 -- singular values, super-diagonal and INFO, no vectors --, one
// workgroup per matrix, float32 rounding for rounding (csrc/linpack_f32.h holds the body and explains why).  A bin the rule
// sends to the identity gets w = d / (N d^H d) (and Lambda = d^H d); every other bin keeps the answer of the solver that ran
// before.  Compiled with -ffp-contract=off (Makefile): a fused multiply-add anywhere would change INFO on borderline bins.
#include <hip/hip_runtime.h>
#include "btk_internal.h"
#include "linpack_f32.h"
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace {

struct WgCtx {
  __device__ int tid() const { return (int)threadIdx.x; }
  __device__ int nthreads() const { return (int)blockDim.x; }
  __device__ void barrier() const { __syncthreads(); }
};

constexpr int LP_THREADS = 256;

__host__ __device__ inline int lp_m(int n, int p) { return p < n + 1 ? p : n + 1; }
__host__ __device__ inline int lp_ld(int p) { return p | 1; }                                   // odd row pitch (in complex numbers) in LDS
// LDS carve-up in bytes: col[n + 1], ev[p + 1], work[n + 1], sc[n + p + 2], ec[n + p + 2], t[2] complex; s[m], e[m] float; flag[4] int
__host__ __device__ inline size_t lp_small_lds(int n, int p)
{
  return sizeof(float2) * ((size_t)4 * n + (size_t)3 * p + 9) + sizeof(float) * 2 * (size_t)lp_m(n, p) + 16;
}
inline size_t lp_mat_lds(int n, int p) { return sizeof(float2) * (size_t)n * lp_ld(p); }
// Where the working copy of a matrix lives.  A bin is mostly ONE thread's serial work (scnrm2's scaled sums, the QR sweeps: serial
// in the reference's rounding order), so the batch runs at the latency of a bin as long as every bin is resident at once.  LDS holds
// 160 KB / (matrix + work arrays) bins per CU -- four at 64 channels --; a batch beyond that (the designs of several streams in one
// call) keeps its matrices in a global scratch copy (L2) instead, where a CU takes as many bins as it has wavefront slots.
inline bool lp_fits_lds(int n, int p) { return lp_small_lds(n, p) + lp_mat_lds(n, p) <= (size_t)150 * 1024; }
inline bool lp_in_lds(int K, int n, int p)
{
  if (!lp_fits_lds(n, p)) return false;
  const size_t per_cu = ((size_t)160 * 1024) / (lp_small_lds(n, p) + lp_mat_lds(n, p) + 512);
  return (size_t)K <= 256 * (per_cu ? per_cu : 1);
}
// one wavefront per bin up to 64 x 64 (every column / row has its thread, barriers cost nothing), four above
inline int lp_threads(int n, int p) { return (n <= 64 && p <= 64) ? 64 : LP_THREADS; }

// A [K][n][p] complex64 row-major.  s_out / e_out [K][m] (may be null), info_out [K] (may be null).
// rule_flags (may be null) [K]: 1 where pseudoinverse() returns false.  A DC bin (skip_dc: bin 0 of the whole spectrum /
// of every stacked stream) is not decomposed: calc_mvdr_weights starts at bin 1.
template <bool IN_LDS>
__global__ __launch_bounds__(LP_THREADS)
void csvdc_values_kernel(const float2* __restrict__ A, int n, int p, float* __restrict__ s_out, float* __restrict__ e_out,
                         int* __restrict__ info_out, float2* __restrict__ scratch, float threshold, int* __restrict__ rule_flags,
                         int skip_dc, int k_offset, int kper)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using lpk::cf;
  const int k = blockIdx.x, tid = threadIdx.x, m = lp_m(n, p);
  if (skip_dc && (kper > 0 ? (k % kper) == 0 : (k + k_offset) == 0)) {
    if (tid == 0) { if (info_out) info_out[k] = 0; if (rule_flags) rule_flags[k] = 0; }
    for (int i = tid; i < m; i += (int)blockDim.x) { if (s_out) s_out[(long)k * m + i] = 0.f; if (e_out) e_out[(long)k * m + i] = 0.f; }
    return;
  }
  cf* col = reinterpret_cast<cf*>(smem);
  cf* ev = col + (n + 1);
  cf* work = ev + (p + 1);
  cf* sc = work + (n + 1);
  cf* ec = sc + (n + p + 2);
  cf* tt = ec + (n + p + 2);
  float* s = reinterpret_cast<float*>(tt + 2);
  float* e = s + m;
  int* flag = reinterpret_cast<int*>(e + m);
  const size_t small = (lp_small_lds(n, p) + 15) & ~(size_t)15;
  const int ld = IN_LDS ? lp_ld(p) : p;
  cf* x = IN_LDS ? reinterpret_cast<cf*>(smem + small) : reinterpret_cast<cf*>(scratch + (long)k * n * p);
  const float2* Ak = A + (long)k * n * p;
  for (int idx = tid; idx < n * p; idx += (int)blockDim.x) {
    const float2 v = Ak[idx];
    x[(long)(idx / p) * ld + (idx % p)] = lpk::mk(v.x, v.y);
  }
  __syncthreads();
  lpk::Work w{col, ev, work, sc, ec, tt, flag};
  WgCtx cx;
  const int info = lpk::csvdc_values(cx, x, ld, n, p, w, s, e);
  if (tid == 0) {
    if (info_out) info_out[k] = info;
    if (rule_flags) {
      int bad = info != 0;
      for (int i = 0; i < p && i < m && !bad; ++i) bad = fabsf(s[i]) < threshold;           // `abs(s[k]) < dThreshold`, beamformer.cc:262-263
      rule_flags[k] = bad;
    }
  }
  for (int i = tid; i < m; i += (int)blockDim.x) { if (s_out) s_out[(long)k * m + i] = s[i]; if (e_out) e_out[(long)k * m + i] = e[i]; }
}

// invR = identity (beamformer.cc:2381-2396): tmpH = d, Lambda = d^H d, w = d / (N Lambda).  One wavefront per flagged bin;
// the bin leaves the list of bins the pseudo-inverse fall-back still has to visit.
__global__ __launch_bounds__(64)
void identity_rule_kernel(const int* __restrict__ rule_flags, const int* __restrict__ info, const float2* __restrict__ Dq,
                          float2* __restrict__ W, float2* __restrict__ lambda_out, int N, int* __restrict__ fail_flags,
                          int* __restrict__ counts)
{
  const int k = blockIdx.x, lane = threadIdx.x;
  if (!rule_flags[k]) return;
  const float2* d = Dq + (long)k * N;
  float acc = 0.f;
  for (int c = lane; c < N; c += 64) acc += d[c].x * d[c].x + d[c].y * d[c].y;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (W) {
    const float sc = 1.0f / ((float)N * acc);
    for (int c = lane; c < N; c += 64) W[(long)k * N + c] = make_float2(d[c].x * sc, d[c].y * sc);
  }
  if (lane == 0) {
    if (lambda_out) lambda_out[k] = make_float2(acc, 0.f);
    if (fail_flags) fail_flags[k] = 0;
    if (counts) atomicAdd(&counts[info[k] != 0 ? 0 : 1], 1);
  }
}

int launch_values(const void* A, int K, int n, int p, float* s, float* e, int* info, void* scratch, float threshold,
                  int* rule_flags, int skip_dc, int k_offset, int kper, hipStream_t st)
{
  if (!A) return btk_set_error(BTK_ERR_PARAMETER, "btk_csvdc_values: null argument");
  if (K < 1 || n < 1 || p < 1 || n > 2048 || p > 2048) return btk_set_error(BTK_ERR_DIMENSION, "btk_csvdc_values: bad sizes K=%d n=%d p=%d", K, n, p);
  const bool in_lds = lp_in_lds(K, n, p);
  if (!in_lds && !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_csvdc_values: %d matrices of %d x %d need a scratch buffer (btk_csvdc_scratch_bytes)", K, n, p);
  const size_t small = (lp_small_lds(n, p) + 15) & ~(size_t)15;
  const size_t lds = small + (in_lds ? lp_mat_lds(n, p) : 0);
  auto kern = in_lds ? csvdc_values_kernel<true> : csvdc_values_kernel<false>;
  if (lds > 64 * 1024)
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)K), dim3((unsigned)lp_threads(n, p)), lds, st, static_cast<const float2*>(A), n, p, s, e, info,
                     static_cast<float2*>(scratch), threshold, rule_flags, skip_dc, k_offset, kper);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // namespace

extern "C" {

long btk_csvdc_scratch_bytes(int K, int n, int p)
{
  if (K < 1 || n < 1 || p < 1) return 0;
  return lp_in_lds(K, n, p) ? 0 : (long)sizeof(float2) * K * n * p;
}

int btk_csvdc_values(const void* A, int K, int n, int p, float* s, float* e, int* info, void* scratch, void* stream)
{
  return launch_values(A, K, n, p, s, e, info, scratch, 0.f, nullptr, 0, 0, 0, as_stream(stream));
}

long btk_mvdr_linpack_rule_scratch_bytes(int K, int N)
{
  if (K < 1 || N < 1) return 0;
  return btk_csvdc_scratch_bytes(K, N, N) + (long)sizeof(int) * 2 * K + 64;
}

int btk_mvdr_linpack_rule(const void* R, const void* wq, void* W, void* lambda, int K, int N, int first_bin, int kper,
                          int skip_dc, float threshold, int* fail_flags, int* counts, void* scratch, void* stream)
{
  if (!R || !wq || !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_linpack_rule: null argument");
  if (K < 1 || N < 1 || first_bin < 0 || kper < 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_linpack_rule: bad sizes");
  hipStream_t st = as_stream(stream);
  const long mat = btk_csvdc_scratch_bytes(K, N, N);
  char* base = static_cast<char*>(scratch);
  int* rule = reinterpret_cast<int*>(base + ((mat + 63) & ~63L));
  int* info = rule + K;
  const int rc = launch_values(R, K, N, N, nullptr, nullptr, info, mat ? scratch : nullptr, threshold, rule, skip_dc, first_bin, kper, st);
  if (rc != BTK_OK) return rc;
  hipLaunchKernelGGL(identity_rule_kernel, dim3((unsigned)K), dim3(64), 0, st, rule, info, static_cast<const float2*>(wq),
                     static_cast<float2*>(W), static_cast<float2*>(lambda), N, fail_flags, counts);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // extern "C"
