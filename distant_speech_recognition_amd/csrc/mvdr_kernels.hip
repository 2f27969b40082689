// mvdr_kernels.hip -- SubbandMVDR weight design on gfx950: diffuse-noise coherence model, diagonal
// loading and a batched Hermitian solve for the MVDR weights.
//
// Replaces SubbandMVDR::set_diffuse_noise_model / set_all_diagonal_loading / calc_mvdr_weights
// (reference beamformer/beamformer.cc:2442-2523, 2350-2402).  The reference inverts R_k with a
// float32 LINPACK SVD pseudo-inverse (pseudoinverse(), :232-289) and substitutes the identity when a
// singular value falls below dThreshold (:262-270, :2381-2383).  R_k is Hermitian; once it is
// positive definite (diagonal loading) the pseudo-inverse IS the inverse, so the engine solves
// R z = d by a complex Cholesky factorisation (one workgroup per bin, matrix resident in LDS) and
// falls back to z = d (identity) when a pivot drops below the threshold -- the same observable rule.
//     w_k = z / (N d^H z),  w_0 = all ones (:2369-2371),  d = wq_k (carries the 1/N factor, :542).
#include "btk_internal.h"
#include <cstdlib>
#include "chol_blocked.h"
#include "chol_reg.h"

namespace {

// Gamma_mn = sinc(2 fs k d_mn / (M c)), GSL sinc(x) = sin(pi x)/(pi x)  (beamformer.cc:2483-2502)
__global__ __launch_bounds__(256)
void diffuse_model_kernel(const float* __restrict__ mpos /* [N][3] */, int N, int M, float samplerate, float sspeed,
                          float2* __restrict__ R /* [K][N][N] */)
{
  const int k = blockIdx.x;
  const double omega_d_c = 2.0 * (double)samplerate * k / ((double)M * (double)sspeed);
  float2* Rk = R + (long)k * N * N;
  for (int idx = threadIdx.x; idx < N * N; idx += 256) {
    const int a = idx / N, b = idx % N;
    float v = 1.0f;
    if (a != b) {
      const double dx = (double)mpos[a * 3] - mpos[b * 3], dy = (double)mpos[a * 3 + 1] - mpos[b * 3 + 1],
                   dz = (double)mpos[a * 3 + 2] - mpos[b * 3 + 2];
      const double x = omega_d_c * sqrt(dx * dx + dy * dy + dz * dz);
      v = (x == 0.0) ? 1.0f : (float)(sinpi(x) / (M_PI * x));
    }
    Rk[idx] = make_float2(v, 0.f);
  }
}

__global__ void diag_load_kernel(float2* __restrict__ R, int N, float w)
{
  float2* Rk = R + (long)blockIdx.x * N * N;
  for (int c = threadIdx.x; c < N; c += blockDim.x) Rk[(long)c * N + c].x += w;
}

// R_xy /= (1 + mu) for x != y  (SubbandMVDR::divide_nondiagonal_elements, beamformer.cc:2589-2599)
__global__ void divide_nondiag_kernel(float2* __restrict__ R, int N, float inv1pmu)
{
  float2* Rk = R + (long)blockIdx.x * N * N;
  for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x)
    if (idx / N != idx % N) { Rk[idx].x *= inv1pmu; Rk[idx].y *= inv1pmu; }
}

// One workgroup per bin.  A (N x N, row-major, lower triangle used) lives in `mat` (LDS or global).
// Right-looking Cholesky A = L L^H, then forward/back substitution for L y = d, L^H z = y.
template <bool IN_LDS>
__global__ __launch_bounds__(256)
void mvdr_solve_kernel(const float2* __restrict__ R, const float2* __restrict__ Dq /* [K][N] d=wq */,
                       float2* __restrict__ Wout /* [K][N] or null */, float2* __restrict__ scratch,
                       int N, float threshold, int* __restrict__ fallback_count,
                       float2* __restrict__ lambda_out /* [K] or null: d^H invR d */, int k_offset /* global index of bin 0 */, int kper /* > 0: S stacked streams of kper bins each, every stream's bin 0 is a DC bin */,
                       int* __restrict__ fail_flags /* [K] or null: 1 where the Cholesky factorisation stopped */)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x;
  const int tid = threadIdx.x;
  // all LDS carved from the dynamic region (16-B aligned base, guide G17)
  float* red_r = reinterpret_cast<float*>(smem);                   // [256]
  float* red_i = red_r + 256;                                      // [256]
  volatile int& bad = *reinterpret_cast<volatile int*>(red_i + 256);
  float2* rhs = reinterpret_cast<float2*>(smem + 2064);            // [N]
  float2* mat = IN_LDS ? rhs + N : scratch + (long)k * N * N;      // [N][N]
  const float2* Rk = R + (long)k * N * N;
  const bool dc_bin = kper > 0 ? (k % kper) == 0 : (k + k_offset) == 0;
  if (dc_bin && Wout) {                                            // wmvdr_[0] = ones (calc_mvdr_weights starts at bin 1)
    for (int c = tid; c < N; c += 256) Wout[(long)k * N + c] = make_float2(1.f, 0.f);
    if (!lambda_out) { if (fail_flags && tid == 0) fail_flags[k] = 0; return; }
  }
  for (int idx = tid; idx < N * N; idx += 256) mat[idx] = Rk[idx];
  for (int c = tid; c < N; c += 256) rhs[c] = Dq[(long)k * N + c];
  if (tid == 0) bad = 0;
  __syncthreads();

  for (int j = 0; j < N && !bad; j++) {
    // pivot
    if (tid == 0) {
      const float piv = mat[(long)j * N + j].x;
      if (!(piv > threshold)) bad = 1;
      else mat[(long)j * N + j] = make_float2(sqrtf(piv), 0.f);
    }
    __syncthreads();
    if (bad) break;
    const float inv = 1.0f / mat[(long)j * N + j].x;
    for (int i = j + 1 + tid; i < N; i += 256) {
      float2 v = mat[(long)i * N + j];
      mat[(long)i * N + j] = make_float2(v.x * inv, v.y * inv);
    }
    __syncthreads();
    // trailing update of the lower triangle: A[i][c] -= L[i][j] conj(L[c][j]),  j < c <= i
    const int rem = N - j - 1;
    for (int idx = tid; idx < rem * rem; idx += 256) {
      const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
      if (c <= i) {
        const float2 li = mat[(long)i * N + j], lc = mat[(long)c * N + j];
        float2 v = mat[(long)i * N + c];
        v.x -= li.x * lc.x + li.y * lc.y;
        v.y -= li.y * lc.x - li.x * lc.y;
        mat[(long)i * N + c] = v;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (fail_flags && tid == 0) fail_flags[k] = bad ? 1 : 0;
  if (bad) {
    // pseudoinverse() reported failure -> invR = identity -> tmpH = d  (beamformer.cc:2381-2383); callers that need the
    // reference's result for matrices that are not positive definite re-solve the flagged bins (btk_mvdr_pinv_fallback)
    if (tid == 0) atomicAdd(fallback_count, 1);
  } else {
    // forward substitution L y = d (column sweep)
    for (int j = 0; j < N; j++) {
      if (tid == 0) { const float inv = 1.0f / mat[(long)j * N + j].x; rhs[j].x *= inv; rhs[j].y *= inv; }
      __syncthreads();
      const float2 yj = rhs[j];
      for (int i = j + 1 + tid; i < N; i += 256) {
        const float2 l = mat[(long)i * N + j];
        rhs[i].x -= l.x * yj.x - l.y * yj.y;
        rhs[i].y -= l.x * yj.y + l.y * yj.x;
      }
      __syncthreads();
    }
    // back substitution L^H z = y
    for (int j = N - 1; j >= 0; j--) {
      if (tid == 0) { const float inv = 1.0f / mat[(long)j * N + j].x; rhs[j].x *= inv; rhs[j].y *= inv; }
      __syncthreads();
      const float2 zj = rhs[j];
      for (int i = tid; i < j; i += 256) {
        const float2 l = mat[(long)j * N + i];                      // conj(L[j][i]) multiplies z_j
        rhs[i].x -= l.x * zj.x + l.y * zj.y;
        rhs[i].y -= l.x * zj.y - l.y * zj.x;
      }
      __syncthreads();
    }
  }
  // Lambda = d^H z ; w = z / (N Lambda)
  float pr = 0.f, pi = 0.f;
  for (int c = tid; c < N; c += 256) {
    const float2 d = Dq[(long)k * N + c], z = rhs[c];
    pr += d.x * z.x + d.y * z.y;          // conj(z) d summed == zdotc(tmpH, d): real part
    pi += z.x * d.y - z.y * d.x;
  }
  red_r[tid] = pr; red_i[tid] = pi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red_r[tid] += red_r[tid + o]; red_i[tid] += red_i[tid + o]; }
    __syncthreads();
  }
  if (lambda_out && tid == 0) lambda_out[k] = make_float2(red_r[0], red_i[0]);
  if (!Wout || dc_bin) return;
  const float nr = red_r[0] * (float)N, ni = red_i[0] * (float)N;   // norm = Lambda * N (complex)
  const float den = nr * nr + ni * ni;
  for (int c = tid; c < N; c += 256) {
    const float2 z = rhs[c];
    Wout[(long)k * N + c] = make_float2((z.x * nr + z.y * ni) / den, (z.y * nr - z.x * ni) / den);
  }
}

// The same solve for arrays whose matrix does not fit the LDS (N > 136; C4: 256 microphones): the blocked left-looking Cholesky of
// chol_blocked.h (16-column panels through LDS, trailing updates on the fp32 matrix cores) on a copy of R_k in `scratch` -- R itself
// stays intact for the pseudo-inverse fall-back.  Round 2 ran the unblocked kernel above on the global copy: one sweep of the trailing
// matrix per column, 19.8 ms for 1025 matrices of 256 x 256 (2.3 TFLOP/s).
__global__ __launch_bounds__(256)
void mvdr_solve_blocked_kernel(const float2* __restrict__ R, const float2* __restrict__ Dq, float2* __restrict__ Wout,
                               float2* __restrict__ scratch, int N, float threshold, int* __restrict__ fallback_count,
                               float2* __restrict__ lambda_out, int k_offset, int kper, int* __restrict__ fail_flags)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, tid = threadIdx.x;
  float2* rhs = reinterpret_cast<float2*>(smem);                   // [N rounded to even]
  float* red = reinterpret_cast<float*>(rhs + ((N + 1) & ~1));     // [512]
  float2* panel = reinterpret_cast<float2*>(red + 512);            // [N][CH_LD]
  float2* mat = scratch + (long)k * N * N;
  const float2* Rk = R + (long)k * N * N;
  const bool dc_bin = kper > 0 ? (k % kper) == 0 : (k + k_offset) == 0;
  if (dc_bin && Wout) {                                            // wmvdr_[0] = ones (calc_mvdr_weights starts at bin 1)
    for (int c = tid; c < N; c += 256) Wout[(long)k * N + c] = make_float2(1.f, 0.f);
    if (!lambda_out) { if (fail_flags && tid == 0) fail_flags[k] = 0; return; }
  }
  for (int idx = tid; idx < N * N; idx += 256) mat[idx] = Rk[idx];
  for (int c = tid; c < N; c += 256) rhs[c] = Dq[(long)k * N + c];
  __syncthreads();
  const bool ok = cholb::solve(mat, N, rhs, red, panel, threshold, k, [](int) {});
  __syncthreads();
  if (fail_flags && tid == 0) fail_flags[k] = ok ? 0 : 1;
  if (!ok) {
    // pseudoinverse() reported failure -> invR = identity -> tmpH = d  (beamformer.cc:2381-2383); the flagged bins are re-solved
    // by the pseudo-inverse kernel when the caller asks for the reference's full rule
    if (tid == 0) atomicAdd(fallback_count, 1);
    for (int c = tid; c < N; c += 256) rhs[c] = Dq[(long)k * N + c];
    __syncthreads();
  }
  float pr = 0.f, pi = 0.f;
  for (int c = tid; c < N; c += 256) {
    const float2 d = Dq[(long)k * N + c], z = rhs[c];
    pr += d.x * z.x + d.y * z.y;
    pi += z.x * d.y - z.y * d.x;
  }
  float* red_r = red; float* red_i = red + 256;
  red_r[tid] = pr; red_i[tid] = pi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red_r[tid] += red_r[tid + o]; red_i[tid] += red_i[tid + o]; }
    __syncthreads();
  }
  if (lambda_out && tid == 0) lambda_out[k] = make_float2(red_r[0], red_i[0]);
  if (!Wout || dc_bin) return;
  const float nr = red_r[0] * (float)N, ni = red_i[0] * (float)N;
  const float den = nr * nr + ni * ni;
  for (int c = tid; c < N; c += 256) {
    const float2 z = rhs[c];
    Wout[(long)k * N + c] = make_float2((z.x * nr + z.y * ni) / den, (z.y * nr - z.x * ni) / den);
  }
}

// Round 4: the same solve with the matrix resident in the accumulator registers of a 512-thread workgroup (chol_reg.h), 136 < N <= 271
// (C4 / C5: 256 microphones).  R is read once (its lower triangle) and never copied: no scratch buffer.
__global__ __launch_bounds__(cholr::NTH)
void mvdr_solve_reg_kernel(const float2* __restrict__ R, const float2* __restrict__ Dq, float2* __restrict__ Wout, int N, float threshold,
                           int* __restrict__ fallback_count, float2* __restrict__ lambda_out, int k_offset, int kper, int* __restrict__ fail_flags)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, tid = threadIdx.x;
  const float2* Rk = R + (long)k * N * N;
  const float2* dq = Dq + (long)k * N;
  const bool dc_bin = kper > 0 ? (k % kper) == 0 : (k + k_offset) == 0;
  if (dc_bin && Wout) {                                            // wmvdr_[0] = ones (calc_mvdr_weights starts at bin 1)
    if (tid < N) Wout[(long)k * N + tid] = make_float2(1.f, 0.f);
    if (!lambda_out) { if (fail_flags && tid == 0) fail_flags[k] = 0; return; }
  }
  const float2* x = cholr::solve(Rk, N, [&](int p) { const float2 d = dq[p]; return make_float2(d.x, -d.y); },
                                 [&](int p) { return Rk[(long)p * N + p].x; }, threshold, smem, [](int) {});
  if (fail_flags && tid == 0) fail_flags[k] = x ? 0 : 1;
  // pseudoinverse() reported failure -> invR = identity -> tmpH = d  (beamformer.cc:2381-2383); see mvdr_solve_blocked_kernel
  if (!x && tid == 0) atomicAdd(fallback_count, 1);
  const float2 d = (tid < N) ? dq[tid] : make_float2(0.f, 0.f);
  const float2 z = (tid < N) ? (x ? x[tid] : d) : make_float2(0.f, 0.f);
  float pr = d.x * z.x + d.y * z.y, pi = z.x * d.y - z.y * d.x;
  for (int o = 32; o > 0; o >>= 1) { pr += __shfl_xor(pr, o, 64); pi += __shfl_xor(pi, o, 64); }
  __syncthreads();                                                 // (the solver's LDS is free from here on)
  float* red = reinterpret_cast<float*>(smem);
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = pr; red[2 * (tid >> 6) + 1] = pi; }
  __syncthreads();
  float sr = 0.f, si = 0.f;
  for (int w = 0; w < cholr::NWAVE; w++) { sr += red[2 * w]; si += red[2 * w + 1]; }
  if (lambda_out && tid == 0) lambda_out[k] = make_float2(sr, si);
  if (!Wout || dc_bin) return;
  const float nr = sr * (float)N, ni = si * (float)N;
  const float den = nr * nr + ni * ni;
  if (tid < N) Wout[(long)k * N + tid] = make_float2((z.x * nr + z.y * ni) / den, (z.y * nr - z.x * ni) / den);
}

}  // namespace

extern "C" {

int btk_mvdr_diffuse_model(const float* mpos, int N, int M, float samplerate, float sspeed, void* R, void* stream)
{
  if (!mpos || !R) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_diffuse_model: null argument");
  if (N < 1 || M < 2) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_diffuse_model: bad sizes");
  hipLaunchKernelGGL(diffuse_model_kernel, dim3((unsigned)(M / 2 + 1)), dim3(256), 0, as_stream(stream),
                     mpos, N, M, samplerate, sspeed, static_cast<float2*>(R));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_mvdr_diagonal_loading(void* R, int nbins, int N, float weight, void* stream)
{
  if (!R) return btk_set_error(BTK_ERR_PARAMETER, "Construct first a noise covariance matrix");
  hipLaunchKernelGGL(diag_load_kernel, dim3((unsigned)nbins), dim3(64), 0, as_stream(stream), static_cast<float2*>(R), N, weight);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

static int mvdr_solve(const void* R, const void* wq, void* W, void* lambda_out, int K, int N, float threshold,
                      void* scratch, int* fallback_count, void* stream, int k_offset = 0, int* fail_flags = nullptr, int kper = 0)
{
  if (!R || !wq || !fallback_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_weights: null argument");
  if (K < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_weights: bad sizes");
  const size_t lds_mat = 2064 + sizeof(float2) * ((size_t)N * N + N);
  // Which solver (profiles/r04_mvdr_sweep.txt, 513 / 2052 systems): the LDS kernel (unblocked, a barrier per column) wins below 64 channels --
  // several small systems per CU --, the register-resident one from 64 on (N = 64: 0.18 -> 0.11 ms, N = 100: 0.65 -> 0.14, N = 136: 1.33 -> 0.17 ms
  // for 513 bins).  BTK_MVDR_REG_MIN moves the switch-over (A/B).
  const int reg_min = btk_switches().mvdr_reg_min;
  if (lds_mat <= 150 * 1024 && N < reg_min) {
    auto kern = mvdr_solve_kernel<true>;
    if (lds_mat > 64 * 1024)
      BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mat));
    hipLaunchKernelGGL(kern, dim3((unsigned)K), dim3(256), lds_mat, as_stream(stream), static_cast<const float2*>(R),
                       static_cast<const float2*>(wq), static_cast<float2*>(W), nullptr, N, threshold, fallback_count,
                       static_cast<float2*>(lambda_out), k_offset, kper, fail_flags);
  } else if (N <= cholr::P_MAX && !btk_switches().wpe_solve_panel) {
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mvdr_solve_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cholr::lds_bytes()));
    hipLaunchKernelGGL(mvdr_solve_reg_kernel, dim3((unsigned)K), dim3(cholr::NTH), cholr::lds_bytes(), as_stream(stream),
                       static_cast<const float2*>(R), static_cast<const float2*>(wq), static_cast<float2*>(W), N, threshold, fallback_count,
                       static_cast<float2*>(lambda_out), k_offset, kper, fail_flags);
  } else {
    if (!scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_weights: N=%d needs a [K][N][N] complex64 scratch buffer", N);
    const size_t lds = cholb::lds_bytes(N);
    if (lds > 150 * 1024) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_weights: N = %d exceeds the blocked solver's LDS panel", N);
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mvdr_solve_blocked_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mvdr_solve_blocked_kernel, dim3((unsigned)K), dim3(256), lds, as_stream(stream),
                       static_cast<const float2*>(R), static_cast<const float2*>(wq), static_cast<float2*>(W),
                       static_cast<float2*>(scratch), N, threshold, fallback_count, static_cast<float2*>(lambda_out), k_offset, kper, fail_flags);
  }
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

// Bytes of the [K][N][N] scratch copy the solver mvdr_solve() would pick for N channels needs (0: none) -- the one place
// that knows the dispatch rule, so that callers never re-derive it.
long btk_mvdr_scratch_bytes(int K, int N)
{
  if (K < 1 || N < 1) return 0;
  const size_t lds_mat = 2064 + sizeof(float2) * ((size_t)N * N + N);
  if (lds_mat <= 150 * 1024 && N < btk_switches().mvdr_reg_min) return 0;
  if (N <= cholr::P_MAX && !btk_switches().wpe_solve_panel) return 0;
  return (long)sizeof(float2) * K * N * N;
}

int btk_mvdr_weights(const void* R, const void* wq, void* W, int K, int N, float threshold,
                     void* scratch, int* fallback_count, void* stream)
{
  if (!W) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_weights: null argument");
  return mvdr_solve(R, wq, W, nullptr, K, N, threshold, scratch, fallback_count, stream);
}

int btk_mvdr_weights_shard(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                           void* scratch, int* fallback_count, void* stream)
{
  if (!W) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_weights_shard: null argument");
  if (first_bin < 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_weights_shard: first_bin = %d", first_bin);
  return mvdr_solve(R, wq, W, nullptr, K, N, threshold, scratch, fallback_count, stream, first_bin);
}

int btk_mvdr_weights_flags(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                           void* scratch, int* fallback_count, int* fail_flags, void* stream)
{
  if (!W || !fail_flags) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_weights_flags: null argument");
  if (first_bin < 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_weights_flags: first_bin = %d", first_bin);
  return mvdr_solve(R, wq, W, nullptr, K, N, threshold, scratch, fallback_count, stream, first_bin, fail_flags);
}

int btk_mvdr_weights_streams(const void* R, const void* wq, void* W, int S, int K, int N, float threshold,
                             void* scratch, int* fallback_count, int* fail_flags, void* stream)
{
  if (!W || !fail_flags) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_weights_streams: null argument");
  if (S < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_weights_streams: S = %d", S);
  if ((long)S * K > 0x7fffffffL) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_weights_streams: S * K too large");
  return mvdr_solve(R, wq, W, nullptr, S * K, N, threshold, scratch, fallback_count, stream, 0, fail_flags, K);
}

int btk_mvdr_divide_nondiagonal(void* R, int nbins, int N, float mu, void* stream)
{
  if (!R) return btk_set_error(BTK_ERR_PARAMETER, "Construct first a noise covariance matrix");
  if (nbins < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_divide_nondiagonal: bad sizes");
  hipLaunchKernelGGL(divide_nondiag_kernel, dim3((unsigned)nbins), dim3(256), 0, as_stream(stream), static_cast<float2*>(R), N,
                     (float)(1.0 / (1.0 + (double)mu)));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_mvdr_lambda(const void* R, const void* d, void* lambda, int K, int N, float threshold,
                    void* scratch, int* fallback_count, void* stream)
{
  if (!lambda) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_lambda: null argument");
  return mvdr_solve(R, d, nullptr, lambda, K, N, threshold, scratch, fallback_count, stream);
}

}  // extern "C"
