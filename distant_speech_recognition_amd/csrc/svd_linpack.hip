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
#include <cstdlib>
#include "btk_internal.h"
#include "linpack_f32.h"
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace {

// srotg for the wavefront walk below.  Two thirds of a rotation step's cycles were its four IEEE divisions: the compiler's
// sequence for `/` (v_div_scale x 2, v_rcp, v_div_fmas, v_div_fixup around six fused multiply-adds) spends five quarter-rate
// instructions per quotient.  Between operands of moderate size that sequence IS the six multiply-adds -- the scale instructions
// pass their operands through, v_div_fmas is a plain fma, v_div_fixup returns its first operand -- so for |sa|, |sb| in
// [2^-60, 2^60] (then every divisor and quotient of the step stays well inside the normal range) the same multiply-adds are issued
// directly, the reciprocal and its refinement shared by the two quotients of a divisor: the same correctly rounded quotients
// (the bit-for-bit tests against the compiled reference run through here), 2 quarter-rate instructions instead of 20.
// Anything else -- zeros, tiny or huge values -- takes lpk::rotg as it is.
__device__ __forceinline__ void div_pair(float a, float b, float d, float& qa, float& qb)
{
  float r = __builtin_amdgcn_rcpf(d);
  const float nd = -d;
  r = fmaf(fmaf(nd, r, 1.0f), r, r);
  float q = a * r;
  q = fmaf(fmaf(nd, q, a), r, q);
  qa = fmaf(fmaf(nd, q, a), r, q);
  q = b * r;
  q = fmaf(fmaf(nd, q, b), r, q);
  qb = fmaf(fmaf(nd, q, b), r, q);
}
__device__ __forceinline__ bool moderate(float x)              // 2^-60 <= |x| < 2^61
{
  return (((unsigned)__float_as_int(x) >> 23) & 0xffu) - 67u <= 120u;
}
__device__ __forceinline__ void rotg_dev(float& sa, float sb, float& c, float& s)
{
  if (!(moderate(sa) && moderate(sb))) { lpk::rotg(sa, sb, c, s); return; }
  const float asa = fabsf(sa), asb = fabsf(sb);
  const float roe = (asb < asa) ? sa : sb;
  const float scale = asa + asb;
  float qa, qb;
  div_pair(sa, sb, scale, qa, qb);
  float r = scale * sqrtf(qa * qa + qb * qb);
  r = ((roe < 0.0f) ? -1.0f : 1.0f) * r;
  div_pair(sa, sb, r, c, s);
  sa = r;
}

// lpk::qr_iterate (linpack_c.cc:9909-10190) walked by ONE WAVEFRONT: the same float32 operations in the same order -- the results
// are bit for bit those of the serial body (tests/test_gpu_linpack_rule.py against the g++ build and the compiled reference) --,
// arranged for the latency of the recurrence, which is what a batch of bins costs (every bin is resident at once):
//  * the two scans that open every pass (the last negligible e above l, the last negligible s) test 64 entries per step, one per
//    lane, and take the first hit by ballot: each entry's test is independent of the others, only the choice is ordered;
//  * the rotation chases (deflation, split, shifted QR step) run lane-uniform with the two values a step hands to the next one kept
//    in registers and the next step's fresh s / e entries loaded one step ahead: no LDS round trip inside the dependent chain of
//    srotg's four divisions and square root; stores are fire and forget (LDS executes a wavefront's accesses in order).
// s, e: LDS, m entries each.  All 64 lanes call it with the same arguments; every lane returns INFO.
__device__ int qr_iterate_wave(int m, float* s, float* e)
{
  const int lane = (int)(threadIdx.x & 63);
  const bool wr = lane == 0;
  const int maxit = 30;
  const int mm = m;
  int iter = 0, info = 0;
  for (;;) {
    if (m == 0) break;
    if (maxit <= iter) { info = m; break; }
    // l: the largest l in [1, m - 1] whose e(l) is negligible beside its two neighbours on the diagonal, else 0
    int l = 0;
    for (int base = m - 1; base >= 1; base -= 64) {
      const int li = base - lane;
      bool hit = false;
      if (li >= 1) {
        const float test = fabsf(s[li - 1]) + fabsf(s[li]);
        const float ztest = test + fabsf(e[li - 1]);
        hit = ztest == test;
      }
      const unsigned long long b = __ballot(hit);
      if (b) { l = base - (__ffsll((long long)b) - 1); break; }
    }
    if (l > 0 && wr) e[l - 1] = 0.0f;
    int kase;
    if (l == m - 1) kase = 4;
    else {
      // ls: the largest ls in [l + 1, m] whose s(ls) is negligible beside the e entries next to it, else l
      int ls = l;
      for (int base = m; base >= l + 1; base -= 64) {
        const int i = base - lane;
        bool hit = false;
        if (i >= l + 1) {
          float test = 0.0f;
          if (i != m) test = test + fabsf(e[i - 1]);
          if (i != l + 1) test = test + fabsf(e[i - 2]);
          const float ztest = test + fabsf(s[i - 1]);
          hit = ztest == test;
        }
        const unsigned long long b = __ballot(hit);
        if (b) { ls = base - (__ffsll((long long)b) - 1); break; }
      }
      if (ls != l && wr) s[ls - 1] = 0.0f;
      if (ls == l) kase = 3;
      else if (ls == m) kase = 1;
      else { kase = 2; l = ls; }
    }
    l = l + 1;
    float cs, sn;
    if (kase == 1) {
      // deflate negligible s(m).  The C++ source counts kk from 1 where the Fortran counts from l (linpack_c.cc:10039-10041):
      // k runs from m - 2 + l down to l, m - 1 steps; lpk::qr_iterate keeps that, so does this
      float f = e[m - 2];
      if (wr) e[m - 2] = 0.0f;
      const int k0 = m - 2 + l;
      float s_k = s[k0 - 1], e_k = (k0 != l) ? e[k0 - 2] : 0.0f;            // s[k - 1], e[k - 2] of the first step
      for (int k = k0; k >= l; --k) {
        const float s_n = (k - 1 >= l) ? s[k - 2] : 0.0f;                  // the next step's entries, one step ahead
        const float e_n = (k - 1 > l) ? e[k - 3] : 0.0f;
        float t1 = s_k;
        rotg_dev(t1, f, cs, sn);
        if (wr) s[k - 1] = t1;
        if (k != l) { f = -sn * e_k; if (wr) e[k - 2] = cs * e_k; }
        s_k = s_n; e_k = e_n;
      }
    } else if (kase == 2) {                              // split at negligible s(l): k = l up to m
      float f = e[l - 2];
      if (wr) e[l - 2] = 0.0f;
      float s_k = s[l - 1], e_k = e[l - 1];
      for (int k = l; k <= m; ++k) {
        const float s_n = (k < m) ? s[k] : 0.0f;
        const float e_n = (k < m) ? e[k] : 0.0f;
        float t1 = s_k;
        rotg_dev(t1, f, cs, sn);
        if (wr) s[k - 1] = t1;
        f = -sn * e_k;
        if (wr) e[k - 1] = cs * e_k;
        s_k = s_n; e_k = e_n;
      }
    } else if (kase == 3) {                              // one shifted QR step
      const float sM = s[m - 1], sM1 = s[m - 2], eM1 = e[m - 2], sL = s[l - 1], eL = e[l - 1];
      const float scale = lpk::r4max(fabsf(sM), lpk::r4max(fabsf(sM1), lpk::r4max(fabsf(eM1), lpk::r4max(fabsf(sL), fabsf(eL)))));
      const float sm = sM / scale, smm1 = sM1 / scale, emm1 = eM1 / scale, sl = sL / scale, el = eL / scale;
      const float b = ((smm1 + sm) * (smm1 - sm) + emm1 * emm1) / 2.0f;
      const float c = (sm * emm1) * (sm * emm1);
      float shift = 0.0f;
      if (b != 0.0f || c != 0.0f) {
        shift = sqrtf(b * b + c);
        if (b < 0.0f) shift = -shift;
        shift = c / (b + shift);
      }
      float f = (sl + sm) * (sl - sm) + shift;
      float g = sl * el;
      // s_a = s[k - 1], e_a = e[k - 1] (what the step before left there), s_b = s[k], e_b = e[k] (fresh)
      float s_a = sL, e_a = eL, s_b = s[l], e_b = e[l];
      for (int k = l; k <= m - 1; ++k) {
        const float s_n = (k + 1 < m) ? s[k + 1] : 0.0f;                   // the next step's fresh entries, one step ahead
        const float e_n = (k + 1 < m) ? e[k + 1] : 0.0f;
        rotg_dev(f, g, cs, sn);
        if (k != l && wr) e[k - 2] = f;
        f = cs * s_a + sn * e_a;
        e_a = cs * e_a - sn * s_a;
        g = sn * s_b;
        s_b = cs * s_b;
        rotg_dev(f, g, cs, sn);
        if (wr) s[k - 1] = f;
        f = cs * e_a + sn * s_b;
        s_b = -sn * e_a + cs * s_b;
        g = sn * e_b;
        e_b = cs * e_b;
        s_a = s_b; e_a = e_b; s_b = s_n; e_b = e_n;
      }
      if (wr) { s[m - 1] = s_a; e[m - 1] = e_a; e[m - 2] = f; }
      iter = iter + 1;
    } else {                                             // convergence
      float a = s[l - 1];
      if (a < 0.0f) { a = -a; if (wr) s[l - 1] = a; }
      while (l != mm) {
        const float nx = s[l];
        if (nx <= a) break;
        if (wr) { s[l - 1] = nx; s[l] = a; }
        l = l + 1;
      }
      iter = 0;
      m = m - 1;
    }
  }
  return info;
}

// lpk::nrm2 (scnrm2's scaled sum of squares, blas1_c.cc:1551-1660) by one wavefront, same roundings: a component either raises
// the scale (rare: the running maximum) or adds (t / scale)^2 to the sum -- the division, the widening and the square are each
// component's own and run one per lane, 64 at a time; what stays serial is the float64 addition into the float32 sum, in component
// order (two v_readlane + convert, add, convert per component instead of a whole IEEE division).  This was the bulk of the
// bidiagonalisation phase: thread 0 summing 2 (n - l) components at every one of the 2 n Householder steps.
// x: LDS.  Every lane of the first wavefront calls it with the same arguments and gets the norm.
__device__ float nrm2_wave(int n, const lpk::cf* x)
{
  if (n < 1) return 0.0f;
  const int lane = (int)(threadIdx.x & 63);
  const float* xf = reinterpret_cast<const float*>(x);
  const int nc = 2 * n;
  float scale = 0.0f, ssq = 1.0f;
  for (int base = 0; base < nc; base += 64) {
    const int c = base + lane;
    const float v = c < nc ? xf[c] : 0.0f;
    const float t = lpk::r4abs(v);
    unsigned long long nz = __ballot(v != 0.0f);               // components still to be taken, in lane order
    while (nz) {
      const unsigned long long up = __ballot(scale < t) & nz;  // ... those that would raise the scale as it is now
      const unsigned long long run = up ? (nz & ((1ull << (__ffsll((long long)up) - 1)) - 1ull)) : nz;
      if (run) {
        const double q = (double)(t / scale);
        const double qq = q * q;
        const int lo = __double2loint(qq), hi = __double2hiint(qq);
        unsigned long long r = run;
        while (r) {
          const int i = __ffsll((long long)r) - 1;
          r &= r - 1ull;
          const double qi = __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
          ssq = (float)((double)ssq + qi);
        }
        nz &= ~run;
      }
      if (up) {
        const int j = __ffsll((long long)up) - 1;
        const float tj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), j));
        const double q = (double)(scale / tj);
        ssq = (float)(1.0 + (double)ssq * (q * q));
        scale = tj;
        nz &= ~(1ull << j);
      }
    }
  }
  return scale * sqrtf(ssq);
}

constexpr int LP_ROWS_PER_THREAD = 2;              // row_sums through the LDS tile: rows of a matrix / threads of its workgroup
constexpr int LP_TILE_MAX_N = 512;                 // ... which is used up to this many rows (tile: n x 9 complex)

struct WgCtx {
  int qw;          // the wavefront that walks the QR iteration: workgroups that share a CU take different ones (a workgroup's
  int* info_slot;  // wavefronts sit on different SIMDs; the iteration keeps one SIMD busy for the whole of its run)
  __device__ int tid() const { return (int)threadIdx.x; }
  __device__ int nthreads() const { return (int)blockDim.x; }
  __device__ void barrier() const { __syncthreads(); }
  __device__ int qr(int m, float* s, float* e) const
  {
    if ((int)(threadIdx.x >> 6) == qw) {
      const int info = qr_iterate_wave(m, s, e);
      if ((threadIdx.x & 63) == 0) *info_slot = info;
    }
    __syncthreads();
    return *info_slot;
  }
  __device__ float nrm2(int n, const lpk::cf* x) const { return threadIdx.x < 64 ? nrm2_wave(n, x) : 0.0f; }
  // the row sums of the row step.  A thread per row reads ITS row along j: fine in LDS; on a matrix in global memory every
  // wavefront load touches 64 cache lines (and 256 rows x 128 B is the whole L1), so there the rows go through an LDS tile of
  // RT_J columns -- loaded as 64-byte row segments, eight lanes each -- and the thread of row i adds its RT_J terms from the
  // tile: the same terms in the same order.  tile == nullptr: the plain loop (matrix in LDS).
  lpk::cf* tile;
  static constexpr int RT_J = 8;
  __device__ void row_sums(const lpk::cf* x, int ld, int i0, int n, int p, const lpk::cf* ev, lpk::cf* work) const
  {
    if (!tile) { lpk::row_sums(*this, x, ld, i0, n, p, ev, work); return; }
    const int t = (int)threadIdx.x, nth = (int)blockDim.x;
    const int jl = t & (RT_J - 1), rl = t / RT_J, rstep = nth / RT_J;
    lpk::cf acc[LP_ROWS_PER_THREAD];
#pragma unroll
    for (int u = 0; u < LP_ROWS_PER_THREAD; ++u) acc[u] = lpk::mk(0.0f, 0.0f);
    for (int jc = i0; jc < p; jc += RT_J) {
      for (int r = i0 + rl; r < n; r += rstep) {
        const int j = jc + jl;
        tile[(r - i0) * (RT_J + 1) + jl] = j < p ? x[(long)r * ld + j] : lpk::mk(0.0f, 0.0f);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < LP_ROWS_PER_THREAD; ++u) {
        const int i = i0 + t + u * nth;
        if (i < n) {
          const lpk::cf* tr = tile + (i - i0) * (RT_J + 1);
          for (int jj = 0; jj < RT_J && jc + jj < p; ++jj) {
            const lpk::cf ej = ev[jc + jj];
            if (lpk::cabs1(ej) != 0.0f) acc[u] = lpk::cadd(acc[u], lpk::cmul(ej, tr[jj]));
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < LP_ROWS_PER_THREAD; ++u) { const int i = i0 + t + u * nth; if (i < n) work[i] = acc[u]; }
    __syncthreads();
  }
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
// PHASE 0: the whole decomposition in one launch.  PHASE 1: the reduction only -- the bidiagonal goes to `se` [K][2][m] -- and
// PHASE 2 (qr_phase_kernel below): the QR iteration on it.  The matrix is needed by the reduction only, the iteration is what takes
// long: a batch whose matrices fit LDS singly but not all at once reduces them in a few short LDS rounds (PHASE 1) and then
// iterates on all bins at once, instead of reducing out of L2 / HBM at the latency of global memory.
template <bool IN_LDS, int PHASE>
__global__ __launch_bounds__(LP_THREADS)
void csvdc_values_kernel(const float2* __restrict__ A, int n, int p, float* __restrict__ s_out, float* __restrict__ e_out,
                         int* __restrict__ info_out, float2* __restrict__ scratch, float threshold, int* __restrict__ rule_flags,
                         int skip_dc, int k_offset, int kper, float* __restrict__ se)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using lpk::cf;
  const int k = blockIdx.x, tid = threadIdx.x, m = lp_m(n, p);
  if (skip_dc && (kper > 0 ? (k % kper) == 0 : (k + k_offset) == 0)) {
    if (PHASE == 1) return;                                    // (the second phase writes the outputs of a skipped bin)
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
  cx.qw = (int)(blockIdx.x % (blockDim.x >> 6));
  cx.info_slot = flag + 2;
  cx.tile = (!IN_LDS && n <= LP_TILE_MAX_N) ? reinterpret_cast<cf*>(smem + small) : nullptr;
  if (PHASE == 1) {
    lpk::csvdc_reduce(cx, x, ld, n, p, w, s, e);
    for (int i = tid; i < 2 * m; i += (int)blockDim.x) se[(long)k * 2 * m + i] = s[i];          // (e follows s in LDS)
    return;
  }
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

// PHASE 2: the QR iteration, one wavefront per bin, FOUR BINS PER WORKGROUP.  A bin's iteration is one wavefront's serial run that
// keeps its SIMD busy (two such wavefronts on a SIMD take 1.7 - 2 x as long: profiles/r06_csvdc_forms.txt); the hardware deals the
// wavefronts of ONE workgroup out over the four SIMDs of its CU, but stacks single-wavefront workgroups as it likes -- so the four
// bins of a workgroup are on four SIMDs by construction, and the LDS share asked for (lp_spread_lds) keeps the number of
// workgroups per CU at ceil(K / 1024).  The bidiagonal of `se` goes to LDS; outputs as csvdc_values_kernel's.
__global__ __launch_bounds__(1024)
void qr_phase_kernel(const float* __restrict__ se, int K, int n, int p, float* __restrict__ s_out, float* __restrict__ e_out,
                     int* __restrict__ info_out, float threshold, int* __restrict__ rule_flags, int skip_dc, int k_offset, int kper)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), m = lp_m(n, p);
  const int k = (int)blockIdx.x * (int)(blockDim.x >> 6) + wave;
  if (k >= K) return;                                          // (wavefronts are independent: no workgroup barrier below)
  if (skip_dc && (kper > 0 ? (k % kper) == 0 : (k + k_offset) == 0)) {
    if (lane == 0) { if (info_out) info_out[k] = 0; if (rule_flags) rule_flags[k] = 0; }
    for (int i = lane; i < m; i += 64) { if (s_out) s_out[(long)k * m + i] = 0.f; if (e_out) e_out[(long)k * m + i] = 0.f; }
    return;
  }
  float* s = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * m;
  float* e = s + m;
  for (int i = lane; i < 2 * m; i += 64) s[i] = se[(long)k * 2 * m + i];
  const int info = qr_iterate_wave(m, s, e);                   // (LDS executes a wavefront's accesses in order)
  if (lane == 0) {
    if (info_out) info_out[k] = info;
    if (rule_flags) {
      int bad = info != 0;
      for (int i = 0; i < p && i < m && !bad; ++i) bad = fabsf(s[i]) < threshold;
      rule_flags[k] = bad;
    }
  }
  for (int i = lane; i < m; i += 64) { if (s_out) s_out[(long)k * m + i] = s[i]; if (e_out) e_out[(long)k * m + i] = e[i]; }
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

// Workgroups per CU capped by the LDS share each one asks for (used or not): the dispatcher stacks small workgroups on a CU as long
// as they fit, and the iteration wants them dealt out evenly (qr_phase_kernel).
inline size_t lp_spread_lds(int nwg, size_t need)
{
  const size_t per_cu = ((size_t)nwg + 255) / 256;
  const size_t want = ((size_t)160 * 1024 / (per_cu ? per_cu : 1)) & ~(size_t)511;
  return want > need ? want : need;
}

// BTK_CSVDC_SPLIT=1 (diagnostics; measured and not the default, profiles/r06_csvdc_forms.txt): two launches -- the reduction
// (csvdc_values_kernel PHASE 1: matrices in LDS when one fits, in as many rounds as the batch needs, else in the global scratch
// copy) leaves the bidiagonals in `se`, qr_phase_kernel iterates on all bins at once.  It loses: the reduction, not the iteration,
// is the larger half at 128 channels and above, and LDS rounds of it (one 132 KB matrix per CU at 128 channels) take longer than
// the whole batch side by side out of L2.
inline bool lp_split()
{
  static const int on = getenv("BTK_CSVDC_SPLIT") && atoi(getenv("BTK_CSVDC_SPLIT")) == 1;
  return on;
}
inline long lp_se_bytes(int K, int n, int p) { return lp_split() ? (((long)sizeof(float) * 2 * K * lp_m(n, p) + 255) & ~255L) : 0; }
inline long lp_mat_scratch_bytes(int K, int n, int p)
{
  if (lp_split()) return lp_fits_lds(n, p) ? 0 : (long)sizeof(float2) * K * n * p;
  return lp_in_lds(K, n, p) ? 0 : (long)sizeof(float2) * K * n * p;
}

int launch_values(const void* A, int K, int n, int p, float* s, float* e, int* info, void* scratch, float threshold,
                  int* rule_flags, int skip_dc, int k_offset, int kper, hipStream_t st)
{
  if (!A) return btk_set_error(BTK_ERR_PARAMETER, "btk_csvdc_values: null argument");
  if (K < 1 || n < 1 || p < 1 || n > 2048 || p > 2048) return btk_set_error(BTK_ERR_DIMENSION, "btk_csvdc_values: bad sizes K=%d n=%d p=%d", K, n, p);
  const bool split = lp_split();
  const bool in_lds = split ? lp_fits_lds(n, p) : lp_in_lds(K, n, p);
  if ((!in_lds || split) && !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_csvdc_values: %d matrices of %d x %d need a scratch buffer (btk_csvdc_scratch_bytes)", K, n, p);
  const size_t small = (lp_small_lds(n, p) + 15) & ~(size_t)15;
  const size_t lds = small + (in_lds ? lp_mat_lds(n, p) : (n <= LP_TILE_MAX_N ? sizeof(float2) * (size_t)n * (WgCtx::RT_J + 1) : 0));
  if (split) {
    float* se = static_cast<float*>(scratch);
    float2* mats = reinterpret_cast<float2*>(static_cast<char*>(scratch) + lp_se_bytes(K, n, p));
    auto kern = in_lds ? csvdc_values_kernel<true, 1> : csvdc_values_kernel<false, 1>;
    if (lds > 64 * 1024)
      BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)K), dim3((unsigned)lp_threads(n, p)), lds, st, static_cast<const float2*>(A), n, p, s, e, info,
                       mats, threshold, rule_flags, skip_dc, k_offset, kper, se);
    BTK_HIP_CHECK(hipGetLastError());
    static const int qrw = getenv("BTK_CSVDC_QRW") ? atoi(getenv("BTK_CSVDC_QRW")) : 4;          // bins (wavefronts) per workgroup
    const int nwg = (K + qrw - 1) / qrw;
    const size_t lds_q = lp_spread_lds(nwg, sizeof(float) * 2 * qrw * (size_t)lp_m(n, p));
    if (lds_q > 64 * 1024)
      BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
    hipLaunchKernelGGL(qr_phase_kernel, dim3((unsigned)nwg), dim3(64 * qrw), lds_q, st, se, K, n, p, s, e, info, threshold, rule_flags, skip_dc, k_offset, kper);
    BTK_HIP_CHECK(hipGetLastError());
    return BTK_OK;
  }
  auto kern = in_lds ? csvdc_values_kernel<true, 0> : csvdc_values_kernel<false, 0>;
  if (lds > 64 * 1024)
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)K), dim3((unsigned)lp_threads(n, p)), lds, st, static_cast<const float2*>(A), n, p, s, e, info,
                     static_cast<float2*>(scratch), threshold, rule_flags, skip_dc, k_offset, kper, static_cast<float*>(nullptr));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // namespace

extern "C" {

long btk_csvdc_scratch_bytes(int K, int n, int p)
{
  if (K < 1 || n < 1 || p < 1) return 0;
  return lp_mat_scratch_bytes(K, n, p) + lp_se_bytes(K, n, p);
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
