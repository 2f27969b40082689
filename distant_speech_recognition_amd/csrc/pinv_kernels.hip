// pinv_kernels.hip -- batched pseudo-inverse MVDR solve on the GPU for the bins the Cholesky kernel flags (gfx950).
//
// Reference: calc_mvdr_weights (beamformer/beamformer.cc:2372-2397) inverts every R_k with pseudoinverse() (:232-289): the matrix is
// rounded to complex<float>, LINPACK csvdc (matrix/linpack_c.cc:9516) factors it, singular values below dThreshold are dropped AND make
// the call report failure, upon which the caller substitutes the identity (:2381-2383).  The pseudo-inverse is unique, so any accurate
// SVD reproduces it: one workgroup per flagged bin runs a one-sided Jacobi (Hestenes) SVD of the float32-rounded matrix in float64 --
// the same method as the host reference btk_pinv (pinv_host.hip), which stays as the checker of this kernel -- and goes straight to
// the weights, w = t / (N d^H t), t = pinv(R)^H d = G S^-2 V^H d, without forming the inverse (A V = G, columns of G orthogonal).
//
// Parallel ordering: the N columns are paired by the round-robin tournament (N - 1 steps of N/2 disjoint pairs per sweep), every pair
// is owned by a group of TPP lanes that split the rows; the three inner products of a pair are reduced inside the group with
// wave shuffles, the rotation is applied to G and V by the same lanes; one workgroup barrier per step.  G and V live column-major
// (odd pitch) in LDS for N <= 64 (133 KB) and in a per-workgroup slice of a global scratch above that (L2-resident: 2 MB at N = 256).
#include "btk_internal.h"

namespace {

struct cd2 { double x, y; };
__device__ __forceinline__ cd2 cmul(cd2 a, cd2 b) { return cd2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd2 cmulc(cd2 a, cd2 b) { return cd2{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; }     // conj(a) b


__device__ __forceinline__ double group_sum(double v, int tpp)
{
  for (int m = tpp >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// IN_LDS: G and V in LDS, 256 threads (N <= 64: 32 pairs x 8 lanes).  Otherwise they live in the workgroup's scratch slice and the
// workgroup has 1024 threads (N = 256: 8 lanes per column pair instead of 2; 1025 bins 9.9 s -> 4.2 s).
template <bool IN_LDS, int PV_NT>
__global__ __launch_bounds__(PV_NT)
void mvdr_pinv_kernel(const float2* __restrict__ R, const float2* __restrict__ Dq, float2* __restrict__ W, int K, int N,
                      int first_bin, float threshold, const int* __restrict__ fail_flags, int* __restrict__ identity_count,
                      cd2* __restrict__ scratch, int tpp)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int LD = N | 1;
  // LDS: sig2 [N] double | u [N] cd2 | dv [N] cd2 | tv [N] cd2 | red [32] double | flags [4] int | (G, V when IN_LDS)
  double* sig2 = reinterpret_cast<double*>(smem);
  cd2* u = reinterpret_cast<cd2*>(sig2 + ((N + 1) & ~1));
  cd2* dv = u + N;
  cd2* tv = dv + N;
  double* red = reinterpret_cast<double*>(tv + N);
  volatile int* flg = reinterpret_cast<volatile int*>(red + 32);          // [0] rotated in this sweep, [1] singular values below threshold
  cd2* G = IN_LDS ? reinterpret_cast<cd2*>(red + 34) : scratch + (size_t)blockIdx.x * 2 * N * LD;
  cd2* V = G + (size_t)N * LD;

  const int np = (N + 1) / 2, NP = 2 * np;                                 // pairs per step, players (a dummy one when N is odd)
  const int ngroups = PV_NT / tpp, grp = tid / tpp, sub = tid % tpp;

  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    if (!fail_flags[k] || k + first_bin == 0) continue;                    // uniform; global bin 0 keeps its all-ones weight (:2369-2371)
    const float2* Rk = R + (size_t)k * N * N;
    __syncthreads();
    for (int idx = tid; idx < N * N; idx += PV_NT) {
      const int i = idx / N, j = idx % N;
      const float2 r = Rk[idx];                                            // complex64 already == the reference's float32 rounding
      G[(size_t)j * LD + i] = cd2{(double)r.x, (double)r.y};
      V[(size_t)j * LD + i] = cd2{i == j ? 1.0 : 0.0, 0.0};
    }
    for (int i = tid; i < N; i += PV_NT) { const float2 d = Dq[(size_t)k * N + i]; dv[i] = cd2{(double)d.x, (double)d.y}; }
    if (tid == 0) { flg[0] = 0; flg[1] = 0; }
    __syncthreads();

    // a pair counts as orthogonal below sqrt(N) machine epsilons (LAPACK zgesvj's rule): the rounding of an N-term inner product is
    // of that order, and a tighter bound (the host checker's 1e-15) keeps rotating noise for all 60 sweeps
    const double eps = 2.3e-16 * sqrt((double)N) * 2.0;
    for (int sweep = 0; sweep < 60; sweep++) {
      for (int step = 0; step < NP - 1; step++) {
        for (int pi = grp; pi < np; pi += ngroups) {
          int p, q;
          if (pi == 0) { p = NP - 1; q = step; }
          else { p = (step + pi) % (NP - 1); q = (step - pi + NP - 1) % (NP - 1); }
          if (p > q) { const int t = p; p = q; q = t; }
          if (q < N) {                                                     // (uniform in the group) the dummy player of an odd N sits out
            cd2* gp = G + (size_t)p * LD; cd2* gq = G + (size_t)q * LD;
            double alpha = 0, beta = 0, gre = 0, gim = 0;
            for (int i = sub; i < N; i += tpp) {
              const cd2 a = gp[i], b = gq[i];
              alpha += a.x * a.x + a.y * a.y; beta += b.x * b.x + b.y * b.y;
              gre += a.x * b.x + a.y * b.y; gim += a.x * b.y - a.y * b.x;  // conj(a) b
            }
            alpha = group_sum(alpha, tpp); beta = group_sum(beta, tpp); gre = group_sum(gre, tpp); gim = group_sum(gim, tpp);
            const double ag = sqrt(gre * gre + gim * gim);
            if (!(ag <= eps * sqrt(alpha * beta) || ag == 0.0)) {
              if (sub == 0) flg[0] = 1;
              const cd2 phc = cd2{gre / ag, -gim / ag};                    // conj(ph): g_q <- conj(ph) g_q makes the inner product real
              const double zeta = (beta - alpha) / (2.0 * ag);
              const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
              const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
              cd2* vp = V + (size_t)p * LD; cd2* vq = V + (size_t)q * LD;
              for (int i = sub; i < N; i += tpp) {
                const cd2 a = gp[i], b = cmul(phc, gq[i]);
                gp[i] = cd2{c * a.x - s * b.x, c * a.y - s * b.y};
                gq[i] = cd2{s * a.x + c * b.x, s * a.y + c * b.y};
                const cd2 va = vp[i], vb = cmul(phc, vq[i]);
                vp[i] = cd2{c * va.x - s * vb.x, c * va.y - s * vb.y};
                vq[i] = cd2{s * va.x + c * vb.x, s * va.y + c * vb.y};
              }
            }
          }
        }
        __syncthreads();
      }
      const int rot = flg[0];
      __syncthreads();
      if (tid == 0) flg[0] = 0;
      __syncthreads();                                                     // the reset must not overtake the next sweep's first mark
      if (!rot) break;
    }
    __syncthreads();
    // singular values, the reference's threshold rule (beamformer.cc:262-266)
    for (int col = grp; col < N; col += ngroups) {
      double s2 = 0;
      for (int i = sub; i < N; i += tpp) { const cd2 a = G[(size_t)col * LD + i]; s2 += a.x * a.x + a.y * a.y; }
      s2 = group_sum(s2, tpp);
      if (sub == 0) {
        const bool below = (float)sqrt(s2) < threshold;
        if (below) flg[1] = 1;
        sig2[col] = below ? 0.0 : 1.0 / s2;
      }
    }
    __syncthreads();
    const bool ident = flg[1] != 0;                                        // "ret = false" -> gsl_matrix_complex_set_identity(invR)
    if (ident) {
      for (int i = tid; i < N; i += PV_NT) tv[i] = dv[i];
      if (tid == 0) atomicAdd(identity_count, 1);
    } else {
      for (int col = grp; col < N; col += ngroups) {                       // u = S^-2 V^H d
        double ur = 0, ui = 0;
        for (int j = sub; j < N; j += tpp) { const cd2 x = cmulc(V[(size_t)col * LD + j], dv[j]); ur += x.x; ui += x.y; }
        ur = group_sum(ur, tpp); ui = group_sum(ui, tpp);
        if (sub == 0) u[col] = cd2{ur * sig2[col], ui * sig2[col]};
      }
      __syncthreads();
      for (int i = tid; i < N; i += PV_NT) {                               // t = G u  (= pinv(R)^H d)
        double tr = 0, ti = 0;
        for (int col = 0; col < N; col++) { const cd2 x = cmul(G[(size_t)col * LD + i], u[col]); tr += x.x; ti += x.y; }
        tv[i] = cd2{tr, ti};
      }
    }
    __syncthreads();
    // lam = zdotc(t, d); w = t / (N lam)
    double lr = 0, li = 0;
    for (int i = tid; i < N; i += PV_NT) { const cd2 x = cmulc(tv[i], dv[i]); lr += x.x; li += x.y; }
    lr = group_sum(lr, 64); li = group_sum(li, 64);
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = lr; red[2 * (tid >> 6) + 1] = li; }
    __syncthreads();
    lr = 0; li = 0;
#pragma unroll
    for (int w = 0; w < PV_NT / 64; w++) { lr += red[2 * w]; li += red[2 * w + 1]; }
    const double nr = lr * N, ni = li * N, den = nr * nr + ni * ni;
    for (int i = tid; i < N; i += PV_NT) {
      const cd2 t = tv[i];
      W[(size_t)k * N + i] = make_float2((float)((t.x * nr + t.y * ni) / den), (float)((t.y * nr - t.x * ni) / den));
    }
  }
}

inline size_t pv_small_lds(int N) { return sizeof(double) * (((N + 1) & ~1) + 34) + sizeof(cd2) * 3 * N + 16; }
inline bool pv_in_lds(int N) { return pv_small_lds(N) + (size_t)2 * N * (N | 1) * sizeof(cd2) <= 160 * 1024 - 512; }
// persistent workgroups: at most 512, and in the scratch form no more than keep their G / V slices (2 N^2 complex128 each) inside
// the 256 MB Infinity Cache (it also bounds the scratch at 192 MB).  The scratch form is bound by what ONE compute unit can pull
// through its L1 -- a rotation step moves 5 N^2 x 16 B per bin -- not by the number of resident bins: 95 or 512 concurrent bins of
// N = 256 take the same 4.1-4.2 s for 1025 bins (profiles/r03_pinv_bench.txt)
inline int pv_grid(int K, int N, bool in_lds)
{
  int g = 512;
  if (!in_lds) {
    const long per = 2L * N * (N | 1) * (long)sizeof(cd2);
    const long fit = (192L << 20) / per;
    g = (int)(fit < 32 ? 32 : (fit > 512 ? 512 : fit));
  }
  return K < g ? K : g;
}

}  // namespace

extern "C" {

long btk_mvdr_pinv_scratch_bytes(int K, int N)
{
  if (K < 1 || N < 1 || pv_in_lds(N)) return 0;
  return (long)pv_grid(K, N, false) * 2 * N * (N | 1) * (long)sizeof(cd2);
}

// Asynchronous form: everything on `stream`, nothing allocated, no host synchronisation.  identity_count [dev int] is incremented
// once per bin that ended with the identity (zero it first); scratch [dev] btk_mvdr_pinv_scratch_bytes(K, N) bytes (may be null
// when that is 0, i.e. N <= 64).
int btk_mvdr_pinv_fallback_async(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                                 const int* fail_flags, int* identity_count, void* scratch, void* stream)
{
  if (!R || !wq || !W || !fail_flags || !identity_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback_async: null argument");
  if (K < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_pinv_fallback_async: bad sizes");
  const bool in_lds = pv_in_lds(N);
  if (!in_lds && !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback_async: N = %d needs a scratch buffer", N);
  const int np = (N + 1) / 2;
  const int nt = in_lds ? 256 : 1024;
  int tpp = 1;                                                              // lanes per column pair: power of two, <= 64, <= threads / pairs
  while (tpp * 2 <= 64 && tpp * 2 * np <= nt && tpp * 2 <= N) tpp *= 2;
  const size_t lds = pv_small_lds(N) + (in_lds ? (size_t)2 * N * (N | 1) * sizeof(cd2) : 0);
  auto kern = in_lds ? mvdr_pinv_kernel<true, 256> : mvdr_pinv_kernel<false, 1024>;
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)pv_grid(K, N, in_lds)), dim3(nt), lds, as_stream(stream), static_cast<const float2*>(R),
                     static_cast<const float2*>(wq), static_cast<float2*>(W), K, N, first_bin, threshold, fail_flags, identity_count,
                     static_cast<cd2*>(scratch), tpp);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

// calc_mvdr_weights for the flagged bins (beamformer.cc:2372-2397) -- the blocking form the node layers call: allocates what the
// kernel needs, runs it, hands *identity_count [host] back.  Synchronises the stream.
int btk_mvdr_pinv_fallback(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                           const int* fail_flags, int* identity_count, void* stream)
{
  if (!R || !wq || !W || !fail_flags) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback: null argument");
  if (K < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_pinv_fallback: bad sizes");
  hipStream_t st = as_stream(stream);
  const long sb = btk_mvdr_pinv_scratch_bytes(K, N);
  char* buf = nullptr;
  BTK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&buf), (size_t)sb + 16));
  int* cnt = reinterpret_cast<int*>(buf + sb);
  int rc = BTK_OK, h = 0;
  hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int), st);
  if (e == hipSuccess) rc = btk_mvdr_pinv_fallback_async(R, wq, W, K, N, first_bin, threshold, fail_flags, cnt, sb ? buf : nullptr, stream);
  if (e == hipSuccess && rc == BTK_OK) e = hipMemcpyAsync(&h, cnt, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && rc == BTK_OK) e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  if (rc != BTK_OK) return rc;
  if (e != hipSuccess) return btk_set_error(BTK_ERR_HIP, "btk_mvdr_pinv_fallback: %s", hipGetErrorString(e));
  if (identity_count) *identity_count = h;
  return BTK_OK;
}

}  // extern "C"
