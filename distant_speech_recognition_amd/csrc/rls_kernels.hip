// RLS sidelobe cancellers in GSC configuration (gfx950), float64 recursion.
//
// Two reference variants share one kernel:
//   mode 0  SubbandGSCRLS::next + update_active_weight_vector2_   (beamformer/beamformer.cc:1514-1645)
//   mode 1  SubbandGSCRLSBeamformer.__iter__                       (lib/pybeamformer.py:817-898)
//
// Algebra.  The reference keeps an (N-1)x(N-1) precision matrix Pz and N-1 active weights and forms
// Z = B^H x (mode 0) or Z = B^T x (mode 1) per frame and bin.  The blocking matrix has orthonormal columns that
// span the complement of conj(wq) (mode 0) resp. conj(vs) (mode 1) -- B^T v = 0 in both --, so with
//      mode 0:  P = B Pz B^H,        w = wl = B wa         (column)
//      mode 1:  P = conj(B) Pz B^T,  w = u  = wa^H B^T     (row)
// every quantity of the recursion lives in N dimensions and the blocking matrix is never read:
//      Pz Z -> P x,   Z^H Pz -> x^H P,   Z^H Pz Z -> x^H P x,   wa^H Z -> w^H x resp. u x,   |wa| = |w|,
//      Pz_0 = c I  ->  P_0 = c (I - v v^H / |v|^2)   with v = conj(wq) (mode 0: calc_blocking_matrix_ builds B with
//      B^T wq = 0, beamformer.cc:373-454) resp. v = vs (mode 1).
// It is the SAME recursion in another basis (outputs agree to rounding); btk_rls_* keep P [N][N] and w [N] in
// complex128.  float64 because the conventional RLS update amplifies rounding by mu^-t.
//
// Mapping.  One (stream, bin) per group of 4 NP threads (NP = N rounded up to 4..64): thread (r, c) owns the row
// slice P[r][cW..cW+W) AND the column slice P[cW..cW+W)[r] (W = NP/4), so both P x and x^H P are W multiply-adds
// plus a DPP quad reduction, and the rank-1 update is elementwise on both copies.  Vectors that every thread
// needs (P x, x^H P, w) go through LDS; x_t is staged in 16-frame tiles (128-byte rows) with register prefetch.
#include "btk_internal.h"
#include <cstdlib>

namespace {

constexpr int RTB = 16;                 // frames per LDS tile
constexpr int RLD = RTB + 1;            // padded row (float2 units)

struct zd { double x, y; };
__device__ __forceinline__ zd zmk(double x, double y) { zd r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ zd zadd(zd a, zd b) { return zmk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ zd zsub(zd a, zd b) { return zmk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ zd zconj(zd a) { return zmk(a.x, -a.y); }
__device__ __forceinline__ zd zscale(zd a, double s) { return zmk(a.x * s, a.y * s); }
__device__ __forceinline__ zd zmul(zd a, zd b) { return zmk(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)); }
// c + a b
__device__ __forceinline__ zd zfma(zd a, zd b, zd c)
{
  c.x = fma(a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
  c.y = fma(a.x, b.y, c.y); c.y = fma(a.y, b.x, c.y);
  return c;
}
// c + conj(a) b
__device__ __forceinline__ zd zfmac(zd a, zd b, zd c)
{
  c.x = fma(a.x, b.x, c.x); c.x = fma(a.y, b.y, c.x);
  c.y = fma(a.x, b.y, c.y); c.y = fma(-a.y, b.x, c.y);
  return c;
}
__device__ __forceinline__ zd zinv(zd a) { const double d = 1.0 / fma(a.x, a.x, a.y * a.y); return zmk(a.x * d, -a.y * d); }

template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v)
{
  v += dpp_d<0xB1>(v);                  // quad_perm [1,0,3,2]
  v += dpp_d<0x4E>(v);                  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ zd quad_sum(zd v) { return zmk(quad_sum(v.x), quad_sum(v.y)); }

struct RlsParams {
  int mode;
  double mu, gamma, reg, init_load, alpha2, max_norm, beta, sil_thresh;     // mode 1
  int copt;
  long min_frames;
  double diag_w, alpha;                                                      // mode 0
  int qctype, normalize, update;
};

// stream_state[s] = { E_avg, unused, isamp, ttl_updates } (doubles, in/out); ctrl[s][t] = 1 adapt / 0 hold.
// One wavefront per stream: E_avg is a linear recurrence independent of the gate, scanned 64 frames at a time in float64.
__global__ __launch_bounds__(64)
void rls_control_kernel(const float* __restrict__ energy, long T, double beta, double sil_thresh,
                        double* __restrict__ stream_state, float* __restrict__ ctrl)
{
  const int s = blockIdx.x, lane = threadIdx.x;
  double* st = stream_state + 4 * (long)s;
  double E = st[0];
  long ttl = (long)st[3];
  const float* e = energy + (long)s * T;
  float* c = ctrl + (long)s * T;
  for (long t0 = 0; t0 < T; t0 += 64) {
    const long t = t0 + lane;
    const bool ok = t < T;
    const double en = ok ? (double)e[t] : 0.0;
    double a = ok ? beta : 1.0, b = ok ? (1.0 - beta) * en : 0.0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double a2 = __shfl_up(a, d, 64), b2 = __shfl_up(b, d, 64);
      if (lane >= d) { b = fma(a, b2, b); a *= a2; }
    }
    const double Et = fma(a, E, b);
    double Eprev = __shfl_up(Et, 1, 64);
    if (lane == 0) Eprev = E;
    const bool adapt = ok && en > Eprev / sil_thresh;          // pybeamformer.py:827
    if (ok) c[t] = adapt ? 1.f : 0.f;
    ttl += __popcll(__ballot(adapt));
    const int last = (T - t0) >= 64 ? 63 : (int)(T - t0) - 1;
    E = __shfl(Et, last, 64);
  }
  if (lane == 0) { st[0] = E; st[2] = (double)((long)st[2] + T); st[3] = (double)ttl; }
}

// P = p0 (I - v v^H / |v|^2) (mode 1) or p0 (I - conj(v) v^T / |v|^2) (mode 0), w = 0
__global__ void rls_init_kernel(const zd* __restrict__ V, int per_stream, int conj_v, double p0, int K, int N,
                                zd* __restrict__ P, zd* __restrict__ Wst)
{
  const int k = blockIdx.x, s = blockIdx.y;
  const zd* v = V + ((long)(per_stream ? s : 0) * K + k) * N;
  double vv = 0.0;
  for (int i = 0; i < N; i++) vv += v[i].x * v[i].x + v[i].y * v[i].y;
  const double iv = vv > 0.0 ? 1.0 / vv : 0.0;
  zd* Pk = P + ((long)s * K + k) * N * N;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
    const int i = e / N, j = e % N;
    zd q = zscale(conj_v ? zmul(zconj(v[i]), v[j]) : zmul(v[i], zconj(v[j])), -iv);
    if (i == j) q.x += 1.0;
    Pk[e] = zscale(q, p0);
  }
  for (int i = threadIdx.x; i < N; i += blockDim.x) Wst[((long)s * K + k) * N + i] = zmk(0.0, 0.0);
}

template <int NP>
__global__ __launch_bounds__((4 * NP < 64) ? 64 : 4 * NP)
void rls_bin_kernel(const float2* __restrict__ X, const zd* __restrict__ V /* [Sw][K][N] */, int per_stream,
                    float2* __restrict__ Y, int K, int N, long T_stride, long T,
                    const float* __restrict__ ctrl, const double* __restrict__ state_before, RlsParams p,
                    zd* __restrict__ Pst /* [S][K][N][N] */, zd* __restrict__ Wst /* [S][K][N] */)
{
  constexpr int W = NP / 4;
  constexpr int TPB = 4 * NP;                              // threads per bin
  constexpr int NT = TPB < 64 ? 64 : TPB;
  constexpr int BPW = NT / TPB;                            // bins per workgroup
  constexpr int LPT = (NP * RTB) / TPB;                    // tile elements loaded per thread (= 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per bin: xt[2][NP][RLD] float2 | vv[NP] avec[NP] bvec[NP] wvec[NP] nvec[NP] zd | yout[RTB] float2
  constexpr int BIN_BYTES = 2 * NP * RLD * 8 + 5 * NP * 16 + RTB * 8;
  const int tid = threadIdx.x;
  const int sub = tid / TPB, ti = tid % TPB;
  const int r = ti >> 2, c = ti & 3;
  char* base = smem + sub * BIN_BYTES;
  float2* xt = reinterpret_cast<float2*>(base);
  zd* vvec = reinterpret_cast<zd*>(base + 2 * NP * RLD * 8);
  zd* avec = vvec + NP;
  zd* bvec = avec + NP;
  zd* wvec = bvec + NP;
  zd* nvec = wvec + NP;
  float2* yout = reinterpret_cast<float2*>(nvec + NP);

  const int s = blockIdx.y;
  const int k = blockIdx.x * BPW + sub;
  const bool kvalid = k < K;
  const int kk = kvalid ? k : K - 1;
  const long sk = (long)s * K + kk;
  const zd* v = V + ((long)(per_stream ? s : 0) * K + kk) * N;
  zd* Pk = Pst + sk * N * N;
  zd* wk = Wst + sk * N;
  const float2* xk = X + sk * N * T_stride;

  // ---- state
  zd Prow[W], Pcol[W];
#pragma unroll
  for (int q = 0; q < W; q++) {
    const int j = c * W + q;
    const bool ok = r < N && j < N;
    Prow[q] = ok ? Pk[(long)r * N + j] : zmk(0.0, 0.0);
    Pcol[q] = ok ? Pk[(long)j * N + r] : zmk(0.0, 0.0);
  }
  zd w_r = r < N ? wk[r] : zmk(0.0, 0.0);
  if (c == 0) { vvec[r] = r < N ? v[r] : zmk(0.0, 0.0); wvec[r] = w_r; }
  const long isamp0 = (long)state_before[4 * s + 2];
  const double inv_mu = 1.0 / p.mu;

  float2 pre[LPT];
  auto prefetch = [&](long t0) {
#pragma unroll
    for (int q = 0; q < LPT; q++) {
      const int e = ti + q * TPB;                          // e = n * RTB + f
      const int n = e / RTB, f = e % RTB;
      const long t = t0 + f;
      pre[q] = (n < N && t < T) ? xk[(long)n * T_stride + t] : make_float2(0.f, 0.f);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < LPT; q++) {
      const int e = ti + q * TPB;
      xt[(buf * NP + e / RTB) * RLD + (e % RTB)] = pre[q];
    }
  };
  prefetch(0);
  stage(0);
  __syncthreads();
  double vv = 0.0;
#pragma unroll
  for (int q = 0; q < W; q++) { const zd a = vvec[c * W + q]; vv = fma(a.x, a.x, fma(a.y, a.y, vv)); }
  vv = quad_sum(vv);
  const double inv_vv = vv > 0.0 ? 1.0 / vv : 0.0;

  int buf = 0;
  for (long t0 = 0; t0 < T; t0 += RTB, buf ^= 1) {
    if (t0 + RTB < T) prefetch(t0 + RTB);
    const int nt = (T - t0 < RTB) ? (int)(T - t0) : RTB;
    for (int tt = 0; tt < nt; tt++) {
      // WG-uniform: every bin of the workgroup walks through the same barriers; `commit` masks the state change
      const bool adapt = (p.mode == 0) ? (p.update != 0) : (ctrl[(long)s * T + t0 + tt] != 0.f);
      const bool commit = (p.mode == 1) || k > 0;                            // beamformer.cc:1589 starts at bin 1
      zd xs[W];
#pragma unroll
      for (int q = 0; q < W; q++) {
        const float2 xf = xt[(buf * NP + c * W + q) * RLD + tt];
        xs[q] = zmk((double)xf.x, (double)xf.y);
      }
      // upper branch Yc = v^H x and canceller output with the current weights
      zd pYc = zmk(0.0, 0.0), pWx = zmk(0.0, 0.0);
      double pnw = 0.0;
#pragma unroll
      for (int q = 0; q < W; q++) {
        const zd vq = vvec[c * W + q], wq = wvec[c * W + q];
        pYc = zfmac(vq, xs[q], pYc);
        pWx = (p.mode == 1) ? zfma(wq, xs[q], pWx) : zfmac(wq, xs[q], pWx);
        const zd d = zsub(vq, wq);
        pnw = fma(d.x, d.x, fma(d.y, d.y, pnw));
      }
      const zd Yc = quad_sum(pYc);
      zd Wx = quad_sum(pWx);
      zd y;
      if (p.mode == 0) {
        y = (k == 0) ? Yc : zsub(Yc, Wx);                                    // beamformer.cc:1540-1558
        if (p.normalize && k > 0) {                                          // calc_gsc_output :1229-1237
          const double nrm = sqrt(quad_sum(pnw));
          y = zscale(y, 1.0 / (nrm * (double)N));
        }
      } else {
        y = Yc;
      }
      if (adapt) {
        zd pa = zmk(0.0, 0.0), pb = zmk(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < W; q++) {
          pa = zfma(Prow[q], xs[q], pa);                                     // (P x)_r
          pb = zfmac(xs[q], Pcol[q], pb);                                    // (x^H P)_r
        }
        const zd a_r = quad_sum(pa), b_r = quad_sum(pb);
        if (c == 0) { avec[r] = a_r; bvec[r] = b_r; }
        __syncthreads();
        zd pip = zmk(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < W; q++)
          pip = (p.mode == 1) ? zfmac(xs[q], avec[c * W + q], pip) : zfma(bvec[c * W + q], xs[q], pip);
        const zd ip = quad_sum(pip);
        zd inv;
        if (p.mode == 1) inv = zinv(zmk(p.mu + ip.x, ip.y));                 // pybeamformer.py:840
        else inv = zscale(zinv(zmk(fma(ip.x, inv_mu, 1.0), ip.y * inv_mu)), inv_mu);   // beamformer.cc:1598-1606
        if (!commit) inv = zmk(0.0, 0.0);
        const double sc = commit ? inv_mu : 1.0;
        const zd g_r = zmul(a_r, inv);
        zd prr = zmk(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < W; q++) {
          const zd g_q = zmul(avec[c * W + q], inv);
          const zd wq = wvec[c * W + q];
          // P <- (P - g temp) / mu on both copies (same arithmetic per element)
          const zd e1 = zfma(zmk(-g_r.x, -g_r.y), bvec[c * W + q], Prow[q]);
          Prow[q] = zscale(e1, sc);
          const zd e2 = zfma(zmk(-g_q.x, -g_q.y), b_r, Pcol[q]);
          Pcol[q] = zscale(e2, sc);
          // regularisation mat-vec with the OLD weights: mode 0 (P wl)_r, mode 1 (P conj(u))_r
          prr = zfma(Prow[q], (p.mode == 1) ? zconj(wq) : wq, prr);
        }
        const zd rr = quad_sum(prr);
        zd wn;
        if (p.mode == 1) {
          const zd ep = zsub(Yc, Wx);                                        // :845
          wn = zadd(w_r, zmul(zscale(zconj(g_r), p.gamma), ep));             // :846
          if (p.reg > 0.0) wn = zsub(wn, zscale(zconj(rr), p.reg));          // :848-849
        } else {
          const zd epA = zconj(y);                                           // :1620
          wn = zadd(zsub(w_r, zscale(rr, p.diag_w)), zmul(g_r, epA));        // :1622-1630
        }
        if (c == 0) nvec[r] = wn;
        __syncthreads();
        double pn = 0.0;
#pragma unroll
        for (int q = 0; q < W; q++) { const zd nq = nvec[c * W + q]; pn = fma(nq.x, nq.x, fma(nq.y, nq.y, pn)); }
        const double n2 = quad_sum(pn);
        if (p.mode == 0) {
          if (p.qctype == 1 || (p.qctype == 2 && n2 >= p.alpha)) wn = zscale(wn, p.alpha / sqrt(n2));   // :1631-1641
        } else if (p.copt > 0) {
          const bool quad = (p.copt == 1 || p.copt == 3) && n2 > p.alpha2;   // :853-866
          if (__syncthreads_or(quad ? 1 : 0)) {
            zd pva = zmk(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < W; q++) pva = zfma(Prow[q], zconj(nvec[c * W + q]), pva);
            const zd va_r = quad_sum(pva);
            if (c == 0) avec[r] = va_r;
            __syncthreads();
            double paa = 0.0, pbb = 0.0;
#pragma unroll
            for (int q = 0; q < W; q++) {
              const zd vq = avec[c * W + q], nq = nvec[c * W + q];
              paa = fma(vq.x, vq.x, fma(vq.y, vq.y, paa));
              pbb += vq.x * nq.x - vq.y * nq.y;                              // Re(conj(va) . waK), waK = conj(waHK)
            }
            const double a = quad_sum(paa), b = -2.0 * quad_sum(pbb), cq = n2 - p.alpha2;
            const double arg = b * b - 4.0 * a * cq;
            const double betaK = (arg > 0.0) ? -(b + sqrt(arg)) / (2.0 * a) : -b / (2.0 * a);
            if (quad) wn = zsub(wn, zscale(zconj(va_r), betaK));
          }
          if (p.copt >= 2 && n2 > p.max_norm) {                              // :867-870
            wn = zscale(wn, sqrt(p.max_norm / n2));
            const double p0 = 1.0 / p.init_load;
            const zd v_r = vvec[r];
#pragma unroll
            for (int q = 0; q < W; q++) {
              const int j = c * W + q;
              const zd vq = vvec[j];
              zd e1 = zscale(zmul(v_r, zconj(vq)), -inv_vv);
              zd e2 = zscale(zmul(vq, zconj(v_r)), -inv_vv);
              if (j == r) { e1.x += 1.0; e2.x += 1.0; }
              const bool ok = r < N && j < N;
              Prow[q] = ok ? zscale(e1, p0) : zmk(0.0, 0.0);
              Pcol[q] = ok ? zscale(e2, p0) : zmk(0.0, 0.0);
            }
          }
        }
        if (commit) w_r = wn;
        if (c == 0) wvec[r] = w_r;                                           // old wvec was last read before the nvec barrier
        __syncthreads();
        if (p.mode == 1) {
          zd pw2 = zmk(0.0, 0.0);
#pragma unroll
          for (int q = 0; q < W; q++) pw2 = zfma(wvec[c * W + q], xs[q], pw2);
          Wx = quad_sum(pw2);
        }
      }
      if (p.mode == 1 && isamp0 + t0 + tt >= p.min_frames) y = zsub(Yc, Wx);   // pybeamformer.py:894-897
      if (ti == 0) yout[tt] = make_float2((float)y.x, (float)y.y);
    }
    __syncthreads();
    if (kvalid && ti < nt) Y[sk * T_stride + t0 + ti] = yout[ti];
    if (t0 + RTB < T) stage(buf ^ 1);
    // ---- once per tile: P <- Q P Q with Q = I - n n^H, n the blocked direction.  In exact arithmetic P n = 0 and
    // n^H P = 0 for ever; in floating point the component along n is multiplied by 1/mu per frame (nothing in the
    // recursion damps it), so it is removed before it can matter.  (The reference cannot leak: it works in N-1
    // dimensions.)  P -= (P v) v^H/|v|^2 + v (v^H P)/|v|^2 - v (v^H P v) v^H/|v|^4, v = vs (mode 1) / conj(wq) (mode 0)
    {
      zd pa = zmk(0.0, 0.0), pb = zmk(0.0, 0.0);
#pragma unroll
      for (int q = 0; q < W; q++) {
        zd vq = vvec[c * W + q];
        if (p.mode == 0) vq = zconj(vq);
        pa = zfma(Prow[q], vq, pa);
        pb = zfmac(vq, Pcol[q], pb);
      }
      const zd a_r = quad_sum(pa), b_r = quad_sum(pb);
      if (c == 0) { avec[r] = a_r; bvec[r] = b_r; }
      __syncthreads();
      zd ps = zmk(0.0, 0.0);
#pragma unroll
      for (int q = 0; q < W; q++) {
        zd vq = vvec[c * W + q];
        if (p.mode == 0) vq = zconj(vq);
        ps = zfmac(vq, avec[c * W + q], ps);
      }
      const zd sv = zscale(quad_sum(ps), inv_vv * inv_vv);
      zd v_r = vvec[r];
      if (p.mode == 0) v_r = zconj(v_r);
#pragma unroll
      for (int q = 0; q < W; q++) {
        zd vq = vvec[c * W + q];
        if (p.mode == 0) vq = zconj(vq);
        // row copy: element (r, j)
        zd d1 = zadd(zmul(a_r, zconj(vq)), zmul(v_r, bvec[c * W + q]));
        d1 = zsub(zscale(d1, inv_vv), zmul(zmul(v_r, zconj(vq)), sv));
        Prow[q] = zsub(Prow[q], d1);
        // column copy: element (i, r)
        zd d2 = zadd(zmul(avec[c * W + q], zconj(v_r)), zmul(vq, b_r));
        d2 = zsub(zscale(d2, inv_vv), zmul(zmul(vq, zconj(v_r)), sv));
        Pcol[q] = zsub(Pcol[q], d2);
      }
    }
    __syncthreads();
  }

  if (kvalid && r < N) {
#pragma unroll
    for (int q = 0; q < W; q++) {
      const int j = c * W + q;
      if (j < N) Pk[(long)r * N + j] = Prow[q];
    }
    if (c == 0) wk[r] = w_r;
  }
}

template <int NP>
int launch_rls(const float2* X, const zd* V, int per_stream, float2* Y, int S, int K, int N, long T_stride, long T,
               const float* ctrl, const double* state_before, const RlsParams& p, zd* P, zd* Wst, hipStream_t st)
{
  constexpr int TPB = 4 * NP;
  constexpr int NT = TPB < 64 ? 64 : TPB;
  constexpr int BPW = NT / TPB;
  constexpr int BIN_BYTES = 2 * NP * RLD * 8 + 5 * NP * 16 + RTB * 8;
  const size_t lds = (size_t)BPW * BIN_BYTES;
  hipLaunchKernelGGL(rls_bin_kernel<NP>, dim3((unsigned)((K + BPW - 1) / BPW), (unsigned)S), dim3(NT), lds, st,
                     X, V, per_stream, Y, K, N, T_stride, T, ctrl, state_before, p, P, Wst);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}


}  // namespace

extern "C" {

long btk_rls_workspace_bytes(int S, long T)
{
  return (long)sizeof(float) * 2 * S * T + (long)sizeof(double) * 4 * S + 64;
}

int btk_rls_init(int mode, const void* v, int per_stream, double p0, int S, int K, int N, void* P_state, void* w_state,
                 void* stream)
{
  if (mode != 0 && mode != 1) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_init: mode must be 0 or 1");
  if (!v || !P_state || !w_state) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_init: null argument");
  if (S <= 0 || K <= 0 || N < 2) return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_init: bad sizes S=%d K=%d N=%d", S, K, N);
  hipLaunchKernelGGL(rls_init_kernel, dim3((unsigned)K, (unsigned)S), dim3(256), 0, as_stream(stream),
                     static_cast<const zd*>(v), per_stream, mode == 0 ? 1 : 0, p0, K, N, static_cast<zd*>(P_state), static_cast<zd*>(w_state));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_rls_process(int mode, const double* params /* host, 10 doubles */, const void* v, int per_stream,
                    const void* X, void* Y, int S, int M, int N, long T_stride, long T,
                    void* P_state, void* w_state, double* stream_state, void* workspace, void* stream)
{
  if (!params || !v || !X || !Y || !P_state || !w_state || !stream_state || !workspace)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_process: null argument");
  if (mode != 0 && mode != 1) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_process: mode must be 0 or 1");
  if (S <= 0 || N < 2 || M < 2 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_process: bad sizes S=%d N=%d M=%d T=%ld", S, N, M, T);
  if (N > 64) return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_process: N=%d > 64 channels not supported", N);
  if (T == 0) return BTK_OK;
  const int K = M / 2 + 1;
  RlsParams p = {};
  p.mode = mode;
  if (mode == 1) {
    p.beta = params[0]; p.gamma = params[1]; p.mu = params[2]; p.init_load = params[3]; p.reg = params[4];
    p.sil_thresh = params[5]; p.copt = (int)params[6]; p.alpha2 = params[7]; p.max_norm = params[8];
    p.min_frames = (long)params[9];
  } else {
    p.mu = params[0]; p.diag_w = params[1]; p.qctype = (int)params[2]; p.alpha = params[3];
    p.normalize = params[4] != 0.0; p.update = params[5] != 0.0;
  }
  if (!(p.mu > 0.0)) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_process: mu must be > 0");
  hipStream_t st = as_stream(stream);
  float* energy = static_cast<float*>(workspace);
  float* ctrl = energy + (long)S * T;
  double* state_before = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ctrl + (long)S * T) + 63) & ~(uintptr_t)63);
  BTK_HIP_CHECK(hipMemcpyAsync(state_before, stream_state, sizeof(double) * 4 * S, hipMemcpyDeviceToDevice, st));
  if (mode == 1) {
    const int rc = btk_frame_energy(X, S, M, N, T_stride, T, energy, T, stream);
    if (rc != BTK_OK) return rc;
    hipLaunchKernelGGL(rls_control_kernel, dim3((unsigned)S), dim3(64), 0, st, energy, T, p.beta, p.sil_thresh, stream_state, ctrl);
    BTK_HIP_CHECK(hipGetLastError());
  }
  const float2* Xp = static_cast<const float2*>(X);
  const zd* V = static_cast<const zd*>(v);
  float2* Yp = static_cast<float2*>(Y);
  zd* P = static_cast<zd*>(P_state);
  zd* Wst = static_cast<zd*>(w_state);
  if (N <= 4)       return launch_rls<4>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  else if (N <= 8)  return launch_rls<8>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  else if (N <= 16) return launch_rls<16>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  else if (N <= 32) return launch_rls<32>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  return launch_rls<64>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
}

}  // extern "C"
