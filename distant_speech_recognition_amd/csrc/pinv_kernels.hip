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
// (odd pitch) in LDS while they fit (N <= 70: 2 N (N | 1) complex128 + 3.6 KB below 160 KB); a bin whose sweeps have not converged
// after 60 rounds is counted (counts[1]) and still answered from the factors it has.
//
// Above that (round 4) the SVD is not needed to apply the reference's rule: "some singular value below the threshold" makes the
// caller take the identity, and otherwise the pseudo-inverse IS the inverse.  mvdr_gj_kernel inverts the float32-rounded matrix in
// place (Gauss-Jordan with row pivoting, float64, the matrix in a per-workgroup slice of a global scratch), gets 1 / sigma_min =
// || R^-1 ||_2 from a power iteration on R^-H R^-1 and forms the weights from the inverse: N^3 complex multiply-adds over a 1 MB
// matrix (N = 256) instead of ~10 sweeps of N^2 / 2 rotations over two of them -- 1025 bins of N = 256 in ~0.1 s instead of 4.1 s
// (profiles/r04_pinv_bench.txt).
#include "btk_internal.h"
#include <map>
#include <mutex>

namespace {

thread_local int g_last_not_converged = 0;

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
    bool converged = false;
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
      if (!rot) { converged = true; break; }
    }
    if (!converged && tid == 0) atomicAdd(identity_count + 1, 1);         // counts[1]: answered from factors that still rotate
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

// ---- N beyond the LDS form: explicit inverse + || R^-1 ||_2 (see the header).  A column-major [N][LD] in the workgroup's
// scratch slice; LDS: f / x [N], rk / y [N], dv [N], tv [N] complex128, reduction cells, pivot rows.
template <int NT>
__global__ __launch_bounds__(NT)
void mvdr_gj_kernel(const float2* __restrict__ R, const float2* __restrict__ Dq, float2* __restrict__ W, int K, int N,
                    int first_bin, float threshold, const int* __restrict__ fail_flags, int* __restrict__ identity_count,
                    cd2* __restrict__ scratch)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LD = N | 1;
  cd2* f = reinterpret_cast<cd2*>(smem);
  cd2* rk = f + N;
  cd2* dv = rk + N;
  cd2* tv = dv + N;
  double* red = reinterpret_cast<double*>(tv + N);                        // [2 NW]
  int* ired = reinterpret_cast<int*>(red + 2 * NW);                       // [NW]
  int* ipiv = ired + NW;                                                  // [N]
  cd2* A = scratch + (size_t)blockIdx.x * N * LD;

  // y_i = sum_j A[i][j] x_j : 16 rows x 4 column groups per wavefront, 256-byte row segments
  auto matvec = [&](const cd2* x, cd2* y) {
    const int il = lane & 15, jg = lane >> 4;
    for (int i0 = wave * 16; i0 < N; i0 += NW * 16) {
      const int i = i0 + il;
      double sr = 0, si = 0;
      if (i < N)
        for (int j = jg; j < N; j += 4) { const cd2 v = cmul(A[(size_t)j * LD + i], x[j]); sr += v.x; si += v.y; }
      sr += __shfl_xor(sr, 16); si += __shfl_xor(si, 16);
      sr += __shfl_xor(sr, 32); si += __shfl_xor(si, 32);
      if (i < N && jg == 0) y[i] = cd2{sr, si};
    }
  };
  // z_j = sum_i conj(A[i][j]) y_i : one wavefront per (contiguous) column
  auto matvec_h = [&](const cd2* y, cd2* z) {
    for (int j = wave; j < N; j += NW) {
      double sr = 0, si = 0;
      for (int i = lane; i < N; i += 64) { const cd2 v = cmulc(A[(size_t)j * LD + i], y[i]); sr += v.x; si += v.y; }
      sr = group_sum(sr, 64); si = group_sum(si, 64);
      if (lane == 0) z[j] = cd2{sr, si};
    }
  };
  // || v ||^2 of an LDS vector, known to every thread afterwards
  auto norm2 = [&](const cd2* v) {
    double s2 = 0;
    for (int i = tid; i < N; i += NT) s2 += v[i].x * v[i].x + v[i].y * v[i].y;
    s2 = group_sum(s2, 64);
    __syncthreads();
    if (lane == 0) red[wave] = s2;
    __syncthreads();
    s2 = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) s2 += red[w];
    return s2;
  };

  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    if (!fail_flags[k] || k + first_bin == 0) continue;                    // uniform; global bin 0 keeps its all-ones weight (:2369-2371)
    const float2* Rk = R + (size_t)k * N * N;
    __syncthreads();
    for (int i = wave; i < N; i += NW)                                     // row i of R (contiguous) -> A[.][i]
      for (int j = lane; j < N; j += 64) { const float2 r = Rk[(size_t)i * N + j]; A[(size_t)j * LD + i] = cd2{(double)r.x, (double)r.y}; }
    for (int i = tid; i < N; i += NT) { const float2 d = Dq[(size_t)k * N + i]; dv[i] = cd2{(double)d.x, (double)d.y}; }
    __syncthreads();

    bool singular = false;
    for (int s = 0; s < N; s++) {
      // pivot: the largest entry of column s at or below the diagonal (ties: the first)
      double best = -1.0; int bi = s;
      for (int i = s + tid; i < N; i += NT) {
        const cd2 a = A[(size_t)s * LD + i];
        const double mg = a.x * a.x + a.y * a.y;
        if (mg > best) { best = mg; bi = i; }
      }
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) { red[wave] = best; ired[wave] = bi; }
      __syncthreads();
      best = red[0]; bi = ired[0];
#pragma unroll
      for (int w = 1; w < NW; w++) if (red[w] > best || (red[w] == best && ired[w] < bi)) { best = red[w]; bi = ired[w]; }
      if (!(best > 0.0) || !(best < 1.0e300)) { singular = true; break; }  // uniform: a zero (or non-finite) column
      const int p = bi;
      if (tid == 0) ipiv[s] = p;
      if (p != s)
        for (int j = tid; j < N; j += NT) { const cd2 t = A[(size_t)j * LD + s]; A[(size_t)j * LD + s] = A[(size_t)j * LD + p]; A[(size_t)j * LD + p] = t; }
      __syncthreads();
      const cd2 piv = A[(size_t)s * LD + s];
      const double pm = piv.x * piv.x + piv.y * piv.y;
      const cd2 ip = cd2{piv.x / pm, -piv.y / pm};
      for (int i = tid; i < N; i += NT) {
        f[i] = A[(size_t)s * LD + i];
        rk[i] = (i == s) ? ip : cmul(A[(size_t)i * LD + s], ip);
      }
      __syncthreads();
      // row s <- rk; every other row i: B[i][j] <- (j == s ? 0 : B[i][j]) - f[i] rk[j]
      for (int j = wave; j < N; j += NW) {
        const cd2 r = rk[j];
        cd2* col = A + (size_t)j * LD;
        for (int i = lane; i < N; i += 64) {
          cd2 a = col[i];
          if (j == s) a = cd2{0.0, 0.0};
          const cd2 fi = f[i];
          col[i] = (i == s) ? r : cd2{a.x - (fi.x * r.x - fi.y * r.y), a.y - (fi.x * r.y + fi.y * r.x)};
        }
      }
      __syncthreads();
    }
    bool ident = singular;
    if (!singular) {
      // the row swaps factored P R: R^-1 = (P R)^-1 P, i.e. the column swaps in reverse order
      for (int s = N - 1; s >= 0; s--) {
        const int p = ipiv[s];
        if (p != s) {
          for (int i = tid; i < N; i += NT) { const cd2 t = A[(size_t)s * LD + i]; A[(size_t)s * LD + i] = A[(size_t)p * LD + i]; A[(size_t)p * LD + i] = t; }
          __syncthreads();
        }
      }
      // 1 / sigma_min^2 = the largest eigenvalue of R^-H R^-1: power iteration from a fixed pseudo-random start
      for (int i = tid; i < N; i += NT) {
        unsigned h = (unsigned)i * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        f[i] = cd2{(double)(h & 0xffff) / 32768.0 - 1.0, (double)(h >> 16) / 32768.0 - 1.0};
      }
      __syncthreads();
      // The iterate approaches 1 / sigma_min^2 from BELOW, i.e. sigma_min is over-estimated until it has converged; a plateau on
      // a non-dominant component or a cluster of small singular values could hide one that is under the threshold.  So the early
      // stop (relative gain <= 1e-7 after 8 steps, at most 40) only decides bins whose estimate is clear of the threshold; an
      // estimate within a factor 4 above it keeps iterating -- up to 400 steps, gain <= 1e-12 -- before the bin is called regular.
      double mu = 0.0;
      int limit = 40;
      double gain = 1.0e-7;
      for (int it = 0; it < limit; it++) {
        const double n2 = norm2(f);
        const double sc = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
        for (int i = tid; i < N; i += NT) f[i] = cd2{f[i].x * sc, f[i].y * sc};
        __syncthreads();
        matvec(f, rk);                                                     // y = R^-1 x
        __syncthreads();
        const double mu_new = norm2(rk);                                   // -> 1 / sigma_min^2 from below
        bool done = it >= 8 && mu_new <= mu * (1.0 + gain);
        mu = mu_new;
        if (!(mu < 1.0e300)) break;                                        // uniform
        const double est = mu > 0.0 ? 1.0 / sqrt(mu) : 0.0;
        if ((done || it + 1 == limit) && limit == 40 && (float)est >= threshold && est < 4.0 * (double)threshold) {
          limit = 400; gain = 1.0e-12; done = false;                       // close call: do not stop on the loose criterion
        }
        if (done) break;
        matvec_h(rk, f);                                                   // x = R^-H y
        __syncthreads();
      }
      const double smin = (mu > 0.0 && mu < 1.0e300) ? 1.0 / sqrt(mu) : 0.0;
      ident = (float)smin < threshold;                                     // beamformer.cc:262-266: some singular value below dThreshold
    }
    __syncthreads();
    if (ident) {                                                           // "ret = false" -> gsl_matrix_complex_set_identity(invR)
      for (int i = tid; i < N; i += NT) tv[i] = dv[i];
      if (tid == 0) atomicAdd(identity_count, 1);
    } else {
      matvec_h(dv, tv);                                                    // t = pinv(R)^H d
    }
    __syncthreads();
    // lam = zdotc(t, d); w = t / (N lam)
    double lr = 0, li = 0;
    for (int i = tid; i < N; i += NT) { const cd2 x = cmulc(tv[i], dv[i]); lr += x.x; li += x.y; }
    lr = group_sum(lr, 64); li = group_sum(li, 64);
    if (lane == 0) { red[2 * wave] = lr; red[2 * wave + 1] = li; }
    __syncthreads();
    lr = 0; li = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { lr += red[2 * w]; li += red[2 * w + 1]; }
    const double nr = lr * N, ni = li * N, den = nr * nr + ni * ni;
    for (int i = tid; i < N; i += NT) {
      const cd2 t = tv[i];
      W[(size_t)k * N + i] = make_float2((float)((t.x * nr + t.y * ni) / den), (float)((t.y * nr - t.x * ni) / den));
    }
  }
}

inline size_t pv_small_lds(int N) { return sizeof(double) * (((N + 1) & ~1) + 34) + sizeof(cd2) * 3 * N + 16; }
inline bool pv_in_lds(int N) { return pv_small_lds(N) + (size_t)2 * N * (N | 1) * sizeof(cd2) <= 160 * 1024 - 512; }
inline size_t gj_lds(int N) { return sizeof(cd2) * 4 * N + sizeof(double) * 32 + sizeof(int) * (16 + N) + 16; }
// persistent workgroups: at most 512; in the scratch form no more than keep their matrices (N (N | 1) complex128 each: 1 MB at
// N = 256) inside 192 MB -- the Infinity Cache holds 256
inline int pv_grid(int K, int N, bool in_lds)
{
  int g = 512;
  if (!in_lds) {
    const long per = (long)N * (N | 1) * (long)sizeof(cd2);
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
  return (long)pv_grid(K, N, false) * N * (N | 1) * (long)sizeof(cd2);
}

// Asynchronous form: everything on `stream`, nothing allocated, no host synchronisation.  identity_count [dev int[2]] (zero it
// first): [0] is incremented once per bin that ended with the identity, [1] once per bin whose Jacobi sweeps did not converge
// (LDS form); scratch [dev] btk_mvdr_pinv_scratch_bytes(K, N) bytes (may be null when that is 0, i.e. the LDS form).
int btk_mvdr_pinv_fallback_async(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                                 const int* fail_flags, int* identity_count, void* scratch, void* stream)
{
  if (!R || !wq || !W || !fail_flags || !identity_count) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback_async: null argument");
  if (K < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_pinv_fallback_async: bad sizes");
  const bool in_lds = pv_in_lds(N);
  if (!in_lds && !scratch) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback_async: N = %d needs a scratch buffer", N);
  if (!in_lds) {
    if (gj_lds(N) > 150 * 1024) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_pinv_fallback_async: N = %d is beyond this kernel", N);
    const size_t lds = gj_lds(N);
    auto kern = mvdr_gj_kernel<1024>;
    BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)pv_grid(K, N, false)), dim3(1024), lds, as_stream(stream), static_cast<const float2*>(R),
                       static_cast<const float2*>(wq), static_cast<float2*>(W), K, N, first_bin, threshold, fail_flags, identity_count,
                       static_cast<cd2*>(scratch));
    BTK_HIP_CHECK(hipGetLastError());
    return BTK_OK;
  }
  const int np = (N + 1) / 2;
  const int nt = 256;
  int tpp = 1;                                                              // lanes per column pair: power of two, <= 64, <= threads / pairs
  while (tpp * 2 <= 64 && tpp * 2 * np <= nt && tpp * 2 <= N) tpp *= 2;
  const size_t lds = pv_small_lds(N) + (size_t)2 * N * (N | 1) * sizeof(cd2);
  auto kern = mvdr_pinv_kernel<true, 256>;
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)pv_grid(K, N, true)), dim3(nt), lds, as_stream(stream), static_cast<const float2*>(R),
                     static_cast<const float2*>(wq), static_cast<float2*>(W), K, N, first_bin, threshold, fail_flags, identity_count,
                     static_cast<cd2*>(scratch), tpp);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

// calc_mvdr_weights for the flagged bins (beamformer.cc:2372-2397) -- the blocking form the node layers call: runs the kernel and
// hands *identity_count [host] back.  Synchronises the stream.  The scratch (up to 192 MB at N = 256) is kept per device between
// calls -- this sits on the calc_mvdr_weights path of every weight update -- and grows only.
int btk_mvdr_pinv_fallback(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                           const int* fail_flags, int* identity_count, void* stream)
{
  if (!R || !wq || !W || !fail_flags) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback: null argument");
  if (K < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_pinv_fallback: bad sizes");
  hipStream_t st = as_stream(stream);
  const long sb = btk_mvdr_pinv_scratch_bytes(K, N);
  struct Cache { char* buf = nullptr; size_t bytes = 0; };
  static std::mutex mtx;
  static std::map<int, Cache> cache;
  std::lock_guard<std::mutex> lock(mtx);
  int devid = 0;
  BTK_HIP_CHECK(hipGetDevice(&devid));
  Cache& c = cache[devid];
  const size_t need = (size_t)sb + 16;
  if (c.bytes < need) {
    if (c.buf) { BTK_HIP_CHECK(hipStreamSynchronize(st)); (void)hipFree(c.buf); c.buf = nullptr; c.bytes = 0; }
    BTK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c.buf), need));
    c.bytes = need;
  }
  int* cnt = reinterpret_cast<int*>(c.buf + (c.bytes - 16));
  int h[2] = {0, 0};
  BTK_HIP_CHECK(hipMemsetAsync(cnt, 0, 2 * sizeof(int), st));
  const int rc = btk_mvdr_pinv_fallback_async(R, wq, W, K, N, first_bin, threshold, fail_flags, cnt, sb ? c.buf : nullptr, stream);
  if (rc != BTK_OK) return rc;
  BTK_HIP_CHECK(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
  BTK_HIP_CHECK(hipStreamSynchronize(st));
  if (identity_count) *identity_count = h[0];
  g_last_not_converged = h[1];
  return BTK_OK;
}

// bins of this thread's last btk_mvdr_pinv_fallback call whose SVD iteration stopped at its sweep limit (their weights come from
// factors that were still rotating; 0 in every test of this repository)
int btk_mvdr_pinv_not_converged(void) { return g_last_not_converged; }

}  // extern "C"
