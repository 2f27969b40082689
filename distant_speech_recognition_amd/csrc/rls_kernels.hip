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
                                zd* __restrict__ P, zd* __restrict__ Wst, const zd* __restrict__ CX, int NC)
{
  const int k = blockIdx.x, s = blockIdx.y;
  const zd* v = V + ((long)(per_stream ? s : 0) * K + k) * N;
  const zd* cx = (NC > 1) ? CX + ((long)(per_stream ? s : 0) * K + k) * (NC - 1) * N : nullptr;
  double vv = 0.0;
  for (int i = 0; i < N; i++) vv += v[i].x * v[i].x + v[i].y * v[i].y;
  const double iv = vv > 0.0 ? 1.0 / vv : 0.0;
  zd* Pk = P + ((long)s * K + k) * N * N;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
    const int i = e / N, j = e % N;
    zd q = zscale(conj_v ? zmul(zconj(v[i]), v[j]) : zmul(v[i], zconj(v[j])), -iv);
    for (int d = 0; d + 1 < NC; d++) q = zsub(q, zmul(cx[d * N + i], zconj(cx[d * N + j])));     // further blocked directions (NC > 1)
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



// ------------------------------------------------------------------------------------------------
// Round 3: more than 64 channels, and more than one constraint (SubbandGSCRLSBeamformer(..., Nc), lib/pybeamformer.py:784-797; the
// C++ SubbandGSCRLS after calc_gsc_weights_2 / _n).  One 256-thread workgroup per (stream, bin); P lives in LDS as the PACKED lower
// triangle of a Hermitian matrix (N (N + 1) / 2 complex128: 81 KB at N = 100, 132 KB at N = 128).  The recursion keeps P Hermitian
// in exact arithmetic -- x^H P = (P x)^H and the gain denominator mu + x^H P x is real -- so one matrix-vector product a = P x per
// frame serves both sides of the reference's update and P <- (P - a a^H / den) / mu touches every stored entry once (the register
// kernel above carries row AND column copies and reproduces the reference's two products separately; the two agree to rounding).
// With NC > 1 constraints the blocking matrix spans the complement of conj(v) AND of NC - 1 further orthonormal directions c_j
// (btk_nlms_constraint_vectors, as in the NLMS canceller): conj(B) B^T = I - v v^H / |v|^2 - sum_j c_j c_j^H is the initial
// projector, the projector of the norm-reset branch, and what the per-tile leak removal projects with, direction by direction.
struct RlsLds {
  zd* P;        // packed lower triangle: (i, j <= i) at i (i + 1) / 2 + j
  zd *xv, *av, *wv, *nv, *vv, *dv, *cx;      // [N] each (dv: the blocked direction, v in mode 1, conj(v) in mode 0), cx [NC-1][N]
  double* red;  // [4 waves][8]
  float2* xt;   // [2][N][RLD]
};

__device__ __forceinline__ zd pk_get(const zd* P, int i, int j) { return (j <= i) ? P[i * (i + 1) / 2 + j] : zconj(P[j * (j + 1) / 2 + i]); }

// sums of up to 6 doubles over the NT threads; every thread gets the totals (two barriers; none in a one-wave workgroup)
template <int NV, int NT>
__device__ __forceinline__ void block_sums(double (&v)[NV], double* red)
{
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    double x = v[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if constexpr (NT == 64) v[q] = x;
    else if ((tid & 63) == 0) red[(tid >> 6) * 8 + q] = x;
  }
  if constexpr (NT > 64) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; q++) {
      double x = red[q];
#pragma unroll
      for (int w = 1; w < NT / 64; w++) x += red[8 * w + q];
      v[q] = x;
    }
    __syncthreads();
  }
}

// PG (round 4, 128 < N <= 256): the Hermitian-packed matrix no longer fits the LDS (N (N + 1) / 2 complex128 = 526 KB at N = 256), so
// the recursion works on the exported state itself -- the FULL [N][N] array in global memory, both triangles kept.  Every access is
// then a sweep along rows with consecutive lanes on consecutive columns (P_ij is read as conj(P_ji)), and the two triangles stay
// exact conjugates of each other because (i, j) and (j, i) are updated with the same products (den_inv and sc are real).

// out_i = sum_j P_ij in_j (CONJ_IN: in_j conjugated), Hermitian packed; two lanes per row (even / odd j), N <= NT / 2
template <bool CONJ_IN, bool PG = false>
__device__ __forceinline__ void pk_matvec(const zd* P, const zd* in, zd* out, int N)
{
  const int tid = threadIdx.x, i = tid >> 1, h = tid & 1;
  zd acc = zmk(0.0, 0.0);
  if (i < N) {
    for (int j = h; j < N; j += 2) {
      const zd pij = PG ? zconj(P[(long)j * N + i]) : pk_get(P, i, j);
      acc = zfma(pij, CONJ_IN ? zconj(in[j]) : in[j], acc);
    }
  }
  acc.x += __shfl_xor(acc.x, 1, 64); acc.y += __shfl_xor(acc.y, 1, 64);
  if (i < N && h == 0) out[i] = acc;
  __syncthreads();
}

// P <- sc (P - g a^H)  on the stored triangle (g = a den_inv with a real den_inv keeps it Hermitian); two lanes per row
template <bool PG = false>
__device__ __forceinline__ void pk_rank1(zd* P, const zd* a, double den_inv, double sc, int N)
{
  const int tid = threadIdx.x, i = tid >> 1, h = tid & 1;
  if constexpr (PG) {
    for (int r = tid >> 6; r < N; r += blockDim.x >> 6) {               // one wavefront per row, lanes along the columns
      const zd gr = zscale(a[r], den_inv);
      zd* row = P + (long)r * N;
      for (int c = tid & 63; c < N; c += 64) {
        zd e = zscale(zsub(row[c], zmul(gr, zconj(a[c]))), sc);
        if (c == r) e.y = 0.0;
        row[c] = e;
      }
    }
    __syncthreads();
    return;
  }
  if (i < N) {
    const zd gi = zscale(a[i], den_inv);
    zd* row = P + i * (i + 1) / 2;
    for (int j = h; j <= i; j += 2) {
      zd e = zscale(zsub(row[j], zmul(gi, zconj(a[j]))), sc);
      // the diagonal of a Hermitian matrix is real: an imaginary rounding residue there is divided by mu every frame and nothing
      // in the recursion damps it (0.97^-1200 = 7e15: it reached O(1) after 1 200 frames before this line existed)
      if (j == i) e.y = 0.0;
      row[j] = e;
    }
  }
  __syncthreads();
}

// P <- p0 (I - sum_d n_d n_d^H) with the orthonormal directions n_0 = vdir / |vdir| and c_j
template <bool PG = false>
__device__ __forceinline__ void pk_set_projector(zd* P, const zd* vdir, double inv_vv, const zd* cx, int NC, double p0, int N)
{
  const int tid = threadIdx.x, i = tid >> 1, h = tid & 1;
  if constexpr (PG) {
    for (int r = tid >> 6; r < N; r += blockDim.x >> 6)
      for (int c = tid & 63; c < N; c += 64) {
        zd q = zscale(zmul(vdir[r], zconj(vdir[c])), -inv_vv);
        for (int d = 0; d + 1 < NC; d++) q = zsub(q, zmul(cx[d * N + r], zconj(cx[d * N + c])));
        if (r == c) { q.x += 1.0; q.y = 0.0; }
        P[(long)r * N + c] = zscale(q, p0);
      }
    __syncthreads();
    return;
  }
  if (i < N) {
    zd* row = P + i * (i + 1) / 2;
    for (int j = h; j <= i; j += 2) {
      zd q = zscale(zmul(vdir[i], zconj(vdir[j])), -inv_vv);
      for (int d = 0; d + 1 < NC; d++) q = zsub(q, zmul(cx[d * N + i], zconj(cx[d * N + j])));
      if (i == j) { q.x += 1.0; q.y = 0.0; }
      row[j] = zscale(q, p0);
    }
  }
  __syncthreads();
}

template <int NT, bool PG = false>
__global__ __launch_bounds__(NT)
void rls_packed_kernel(const float2* __restrict__ X, const zd* __restrict__ V, int per_stream, const zd* __restrict__ CX, int NC,
                       float2* __restrict__ Y, int K, int N, long T_stride, long T, const float* __restrict__ ctrl,
                       const double* __restrict__ state_before, RlsParams p, zd* __restrict__ Pst, zd* __restrict__ Wst,
                       int rtb /* frames per snapshot tile: 16, double-buffered; 8, single buffer, when the LDS is short (N > 112) */)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int s = blockIdx.y, k = blockIdx.x;
  const int rld = rtb + 1, nbuf = rtb == RTB ? 2 : 1;
  const long sk = (long)s * K + k;
  const int NP = PG ? 0 : N * (N + 1) / 2;
  RlsLds L;
  L.P = reinterpret_cast<zd*>(smem);
  L.xv = L.P + NP; L.av = L.xv + N; L.wv = L.av + N; L.nv = L.wv + N; L.vv = L.nv + N; L.dv = L.vv + N; L.cx = L.dv + N;
  L.red = reinterpret_cast<double*>(L.cx + (NC > 1 ? (NC - 1) * N : 0));
  L.xt = reinterpret_cast<float2*>(L.red + 72);                       // (8 cells per wavefront, up to 8 wavefronts + spare)
  float2* yout = L.xt + nbuf * N * rld;
  const zd* v = V + ((long)(per_stream ? s : 0) * K + k) * N;
  zd* Pk = Pst + sk * N * N;
  zd* wk = Wst + sk * N;
  const float2* xk = X + sk * N * T_stride;

  if constexpr (PG) {
    L.P = Pk;                                              // the state itself; its diagonal is real by contract
    for (int n = tid; n < N; n += NT) Pk[(long)n * N + n].y = 0.0;
  }
  for (int e = tid; e < NP; e += NT) {                     // lower triangle of the exported [N][N] state
    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > e) i--;
    while ((i + 1) * (i + 2) / 2 <= e) i++;
    const int j = e - i * (i + 1) / 2;
    L.P[e] = Pk[(long)i * N + j];
    if (i == j) L.P[e].y = 0.0;
  }
  // v as it enters the projector: mode 1 v = vs; mode 0 the blocked direction is conj(wq)
  for (int n = tid; n < N; n += NT) { L.vv[n] = v[n]; L.dv[n] = (p.mode == 0) ? zconj(v[n]) : v[n]; L.wv[n] = wk[n]; }
  for (int e = tid; e < (NC - 1) * N; e += NT) L.cx[e] = CX[((long)(per_stream ? s : 0) * K + k) * (NC - 1) * N + e];
  __syncthreads();
  double vs2[1] = {0.0};
  for (int n = tid; n < N; n += NT) vs2[0] += L.vv[n].x * L.vv[n].x + L.vv[n].y * L.vv[n].y;
  block_sums<1, NT>(vs2, L.red);
  const double inv_vv = vs2[0] > 0.0 ? 1.0 / vs2[0] : 0.0;
  // the direction the projector removes, as a vector in LDS: mode 1 vs, mode 0 conj(wq); kept in nv during the leak step only
  const long isamp0 = (long)state_before[4 * s + 2];
  const double inv_mu = 1.0 / p.mu;
  const bool commit = (p.mode == 1) || k > 0;               // beamformer.cc:1589 starts at bin 1

  auto stage = [&](int buf, long t0) {
    for (int e = tid; e < N * rtb; e += NT) {
      const int n = e / rtb, f = e % rtb;
      const long t = t0 + f;
      L.xt[(buf * N + n) * rld + f] = (t < T) ? xk[(long)n * T_stride + t] : make_float2(0.f, 0.f);
    }
  };
  if (nbuf == 2) stage(0, 0);
  __syncthreads();
  int buf = 0;
  for (long t0 = 0; t0 < T; t0 += rtb, buf = (nbuf == 2) ? buf ^ 1 : 0) {
    if (nbuf == 2) { if (t0 + rtb < T) stage(buf ^ 1, t0 + rtb); }      // the other buffer: last read a tile ago
    else { stage(0, t0); __syncthreads(); }                                 // one buffer: the previous tile ended with a barrier
    const int nt = (T - t0 < rtb) ? (int)(T - t0) : rtb;
    for (int tt = 0; tt < nt; tt++) {
      const bool adapt = (p.mode == 0) ? (p.update != 0) : (ctrl[(long)s * T + t0 + tt] != 0.f);
      for (int n = tid; n < N; n += NT) { const float2 xf = L.xt[(buf * N + n) * rld + tt]; L.xv[n] = zmk((double)xf.x, (double)xf.y); }
      __syncthreads();
      // Yc = v^H x, canceller output with the current weights, |v - w|^2 (mode 0 normalisation)
      double r5[5] = {0, 0, 0, 0, 0};
      for (int n = tid; n < N; n += NT) {
        const zd x = L.xv[n], vq = L.vv[n], wq = L.wv[n];
        const zd a = zfmac(vq, x, zmk(0, 0));
        const zd b = (p.mode == 1) ? zfma(wq, x, zmk(0, 0)) : zfmac(wq, x, zmk(0, 0));
        const zd d = zsub(vq, wq);
        r5[0] += a.x; r5[1] += a.y; r5[2] += b.x; r5[3] += b.y; r5[4] += d.x * d.x + d.y * d.y;
      }
      block_sums<5, NT>(r5, L.red);
      const zd Yc = zmk(r5[0], r5[1]);
      zd Wx = zmk(r5[2], r5[3]);
      zd y;
      if (p.mode == 0) {
        y = (k == 0) ? Yc : zsub(Yc, Wx);                                    // beamformer.cc:1540-1558
        if (p.normalize && k > 0) y = zscale(y, 1.0 / (sqrt(r5[4]) * (double)N));   // calc_gsc_output :1229-1237
      } else {
        y = Yc;
      }
      if (adapt) {                                                           // (workgroup-uniform)
        pk_matvec<false, PG>(L.P, L.xv, L.av, N);                                // a = P x;  x^H P = a^H
        double ipr[1] = {0.0};
        for (int n = tid; n < N; n += NT) { const zd x = L.xv[n], a = L.av[n]; ipr[0] += x.x * a.x + x.y * a.y; }   // Re(x^H a)
        block_sums<1, NT>(ipr, L.red);
        // mode 1: g = a / (mu + x^H P x) (pybeamformer.py:840); mode 0: g = a / (mu (1 + x^H P x / mu)) (beamformer.cc:1598-1606)
        double den_inv = 1.0 / (p.mu + ipr[0]);
        if (!commit) den_inv = 0.0;
        pk_rank1<PG>(L.P, L.av, den_inv, commit ? inv_mu : 1.0, N);              // P <- (P - g a^H) / mu
        // regularisation mat-vec with the OLD weights and the NEW P: mode 0 (P wl), mode 1 (P conj(u))
        const bool need_rr = (p.mode == 1) ? (p.reg > 0.0) : (p.diag_w != 0.0);
        if (need_rr) { if (p.mode == 1) pk_matvec<true, PG>(L.P, L.wv, L.nv, N); else pk_matvec<false, PG>(L.P, L.wv, L.nv, N); }
        double n2s[1] = {0.0};
        for (int n = tid; n < N; n += NT) {
          const zd g = zscale(L.av[n], den_inv), w_n = L.wv[n];
          const zd rr = need_rr ? L.nv[n] : zmk(0.0, 0.0);
          zd wn;
          if (p.mode == 1) {
            const zd ep = zsub(Yc, Wx);                                      // :845
            wn = zadd(w_n, zmul(zscale(zconj(g), p.gamma), ep));             // :846
            if (p.reg > 0.0) wn = zsub(wn, zscale(zconj(rr), p.reg));        // :848-849
          } else {
            wn = zadd(zsub(w_n, zscale(rr, p.diag_w)), zmul(g, zconj(y)));   // :1620-1630
          }
          L.xv[n] = wn;                                                      // x is dead from here on: xv holds the candidate
          n2s[0] += wn.x * wn.x + wn.y * wn.y;
        }
        __syncthreads();                                                     // (nv was read above, xv written: order both)
        block_sums<1, NT>(n2s, L.red);
        double n2 = n2s[0];
        double scale_w = 1.0;
        if (p.mode == 0) {
          if (p.qctype == 1 || (p.qctype == 2 && n2 >= p.alpha)) scale_w = p.alpha / sqrt(n2);       // :1631-1641
        } else if (p.copt > 0) {
          const bool quad = (p.copt == 1 || p.copt == 3) && n2 > p.alpha2;   // :853-866
          if (quad) {
            pk_matvec<true, PG>(L.P, L.xv, L.nv, N);                             // va = P conj(waHK)
            double q2[2] = {0.0, 0.0};
            for (int n = tid; n < N; n += NT) {
              const zd va = L.nv[n], nq = L.xv[n];
              q2[0] += va.x * va.x + va.y * va.y;
              q2[1] += va.x * nq.x - va.y * nq.y;                            // Re(conj(va) . waK), waK = conj(waHK)
            }
            block_sums<2, NT>(q2, L.red);
            const double a = q2[0], b = -2.0 * q2[1], cq = n2 - p.alpha2;
            const double arg = b * b - 4.0 * a * cq;
            const double betaK = (arg > 0.0) ? -(b + sqrt(arg)) / (2.0 * a) : -b / (2.0 * a);
            for (int n = tid; n < N; n += NT) L.xv[n] = zsub(L.xv[n], zscale(zconj(L.nv[n]), betaK));
            __syncthreads();
          }
          if (p.copt >= 2 && n2 > p.max_norm) {                              // :867-870 (n2 of the unconstrained candidate)
            scale_w = sqrt(p.max_norm / n2);
            pk_set_projector<PG>(L.P, L.dv, inv_vv, L.cx, NC, 1.0 / p.init_load, N);
          }
        }
        if (commit) for (int n = tid; n < N; n += NT) L.wv[n] = zscale(L.xv[n], scale_w);
        __syncthreads();
        if (p.mode == 1) {                                                   // output with the updated weights
          for (int n = tid; n < N; n += NT) { const float2 xf = L.xt[(buf * N + n) * rld + tt]; L.xv[n] = zmk((double)xf.x, (double)xf.y); }
          __syncthreads();
          double w2[2] = {0.0, 0.0};
          for (int n = tid; n < N; n += NT) { const zd t = zfma(L.wv[n], L.xv[n], zmk(0, 0)); w2[0] += t.x; w2[1] += t.y; }
          block_sums<2, NT>(w2, L.red);
          Wx = zmk(w2[0], w2[1]);
        }
      }
      if (p.mode == 1 && isamp0 + t0 + tt >= p.min_frames) y = zsub(Yc, Wx);   // pybeamformer.py:894-897
      if (tid == 0) yout[tt] = make_float2((float)y.x, (float)y.y);
      __syncthreads();
    }
    if (tid < nt) Y[sk * T_stride + t0 + tid] = yout[tid];
    // ---- once per tile: P <- Q P Q, Q = I - n n^H for every blocked direction n (v / |v| resp. conj(wq) / |wq|, then the c_j):
    // in exact arithmetic P n = 0 for ever; in floating point that component is multiplied by 1 / mu per frame
    for (int d = 0; d < NC; d++) {
      for (int n = tid; n < N; n += NT) {
        L.xv[n] = (d == 0) ? zscale(L.dv[n], sqrt(inv_vv)) : L.cx[(d - 1) * N + n];
      }
      __syncthreads();
      pk_matvec<false, PG>(L.P, L.xv, L.av, N);                                  // a = P n
      double sv[1] = {0.0};
      for (int n = tid; n < N; n += NT) { const zd nd = L.xv[n], a = L.av[n]; sv[0] += nd.x * a.x + nd.y * a.y; }   // n^H P n (real)
      block_sums<1, NT>(sv, L.red);
      const int i = tid >> 1, h = tid & 1;
      if constexpr (PG) {
        for (int r = tid >> 6; r < N; r += NT >> 6) {
          zd* row = L.P + (long)r * N;
          const zd ar = L.av[r], nr = L.xv[r];
          for (int c = tid & 63; c < N; c += 64) {
            const zd ac = L.av[c], nc = L.xv[c];
            zd dlt = zadd(zmul(ar, zconj(nc)), zmul(nr, zconj(ac)));
            dlt = zsub(dlt, zscale(zmul(nr, zconj(nc)), sv[0]));
            zd e = zsub(row[c], dlt);
            if (c == r) e.y = 0.0;
            row[c] = e;
          }
        }
      } else if (i < N) {
        zd* row = L.P + i * (i + 1) / 2;
        const zd ai = L.av[i], ni = L.xv[i];
        for (int j = h; j <= i; j += 2) {
          const zd aj = L.av[j], nj = L.xv[j];
          zd dlt = zadd(zmul(ai, zconj(nj)), zmul(ni, zconj(aj)));
          dlt = zsub(dlt, zscale(zmul(ni, zconj(nj)), sv[0]));
          zd e = zsub(row[j], dlt);
          if (j == i) e.y = 0.0;                                              // (see pk_rank1)
          row[j] = e;
        }
      }
      __syncthreads();
    }
  }
  // export: both triangles of P [N][N], w [N]
  if constexpr (!PG) { for (int e = tid; e < N * N; e += NT) { const int i = e / N, j = e % N; Pk[e] = pk_get(L.P, i, j); } }
  for (int n = tid; n < N; n += NT) wk[n] = L.wv[n];
}

inline size_t rls_packed_lds(int N, int NC, int rtb, bool pg = false)
{
  return sizeof(zd) * ((pg ? 0 : (size_t)N * (N + 1) / 2) + 6 * (size_t)N + (size_t)(NC > 1 ? NC - 1 : 0) * N) + sizeof(double) * 72 +
         sizeof(float2) * ((rtb == RTB ? 2 : 1) * (size_t)N * (rtb + 1) + RTB);
}
inline int rls_packed_rtb(int N, int NC) { return rls_packed_lds(N, NC, RTB) <= 160 * 1024 - 256 ? RTB : 8; }


template <int NT, bool PG = false>
int launch_rls_packed(const float2* X, const zd* V, int per_stream, const zd* cx, int NC, float2* Y, int S, int K, int N, long T_stride, long T,
                      const float* ctrl, const double* state_before, const RlsParams& p, zd* P, zd* W, int rtb, size_t lds, hipStream_t st)
{
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rls_packed_kernel<NT, PG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((rls_packed_kernel<NT, PG>), dim3((unsigned)K, (unsigned)S), dim3(NT), lds, st, X, V, per_stream, cx, NC, Y, K, N, T_stride, T,
                     ctrl, state_before, p, P, W, rtb);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // namespace

extern "C" {

long btk_rls_workspace_bytes(int S, long T)
{
  return (long)sizeof(float) * 2 * S * T + (long)sizeof(double) * 4 * S + 64;
}

int btk_rls_init_nc(int mode, const void* v, int per_stream, const void* cx, int NC, double p0, int S, int K, int N, void* P_state,
                    void* w_state, void* stream)
{
  if (mode != 0 && mode != 1) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_init: mode must be 0 or 1");
  if (!v || !P_state || !w_state) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_init: null argument");
  if (S <= 0 || K <= 0 || N < 2) return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_init: bad sizes S=%d K=%d N=%d", S, K, N);
  if (NC < 1 || NC >= N || (NC > 1 && !cx)) return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_init: NC=%d constraints with N=%d channels", NC, N);
  hipLaunchKernelGGL(rls_init_kernel, dim3((unsigned)K, (unsigned)S), dim3(256), 0, as_stream(stream),
                     static_cast<const zd*>(v), per_stream, mode == 0 ? 1 : 0, p0, K, N, static_cast<zd*>(P_state), static_cast<zd*>(w_state),
                     static_cast<const zd*>(cx), NC);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_rls_init(int mode, const void* v, int per_stream, double p0, int S, int K, int N, void* P_state, void* w_state,
                 void* stream)
{
  return btk_rls_init_nc(mode, v, per_stream, nullptr, 1, p0, S, K, N, P_state, w_state, stream);
}

int btk_rls_process(int mode, const double* params /* host, 10 doubles */, const void* v, int per_stream,
                    const void* X, void* Y, int S, int M, int N, long T_stride, long T,
                    void* P_state, void* w_state, double* stream_state, void* workspace, void* stream)
{
  return btk_rls_process_nc(mode, params, v, per_stream, nullptr, 1, X, Y, S, M, N, T_stride, T, P_state, w_state, stream_state, workspace, stream);
}

int btk_rls_process_nc(int mode, const double* params /* host, 10 doubles */, const void* v, int per_stream, const void* cx, int NC,
                       const void* X, void* Y, int S, int M, int N, long T_stride, long T,
                       void* P_state, void* w_state, double* stream_state, void* workspace, void* stream)
{
  if (!params || !v || !X || !Y || !P_state || !w_state || !stream_state || !workspace)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_process: null argument");
  if (mode != 0 && mode != 1) return btk_set_error(BTK_ERR_PARAMETER, "btk_rls_process: mode must be 0 or 1");
  if (S <= 0 || N < 2 || M < 2 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_process: bad sizes S=%d N=%d M=%d T=%ld", S, N, M, T);
  if (NC < 1 || NC >= N || (NC > 1 && !cx)) return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_process: NC=%d constraints with N=%d channels", NC, N);
  // register kernel: N <= 64 with one constraint; packed-Hermitian LDS kernel: anything else that fits the LDS (N <= 128)
  const bool packed = N > 64 || NC > 1 || btk_switches().rls_packed;
  // ... and up to N = 256 (two lanes per row in a 512-thread workgroup) on the precision matrix in global memory, where it is exported anyway
  const bool pglobal = packed && (N > 128 || rls_packed_lds(N, NC, rls_packed_rtb(N, NC)) > 160 * 1024 - 256);
  if (pglobal && (N > 256 || rls_packed_lds(N, NC, 8, true) > 160 * 1024 - 256))
    return btk_set_error(BTK_ERR_DIMENSION, "btk_rls_process: N=%d channels (NC=%d) exceed this kernel (N <= 256)", N, NC);
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
  if (pglobal) {
    const int rtb = rls_packed_lds(N, NC, RTB, true) <= 160 * 1024 - 256 ? RTB : 8;
    return launch_rls_packed<512, true>(Xp, V, per_stream, static_cast<const zd*>(cx), NC, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, rtb,
                                        rls_packed_lds(N, NC, rtb, true), st);
  }
  if (packed) {
    const int rtb = rls_packed_rtb(N, NC);
    const size_t lds = rls_packed_lds(N, NC, rtb);
    // two lanes per matrix row: a 64-thread (one wave, barrier-free sums) workgroup up to N = 32, 128 threads up to 64 -- small
    // arrays then keep 4x / 2x as many bins resident per CU as the 256-thread form
    if (N <= 32)      return launch_rls_packed<64>(Xp, V, per_stream, static_cast<const zd*>(cx), NC, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, rtb, lds, st);
    else if (N <= 64) return launch_rls_packed<128>(Xp, V, per_stream, static_cast<const zd*>(cx), NC, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, rtb, lds, st);
    return launch_rls_packed<256>(Xp, V, per_stream, static_cast<const zd*>(cx), NC, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, rtb, lds, st);
  }
  if (N <= 4)       return launch_rls<4>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  else if (N <= 8)  return launch_rls<8>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  else if (N <= 16) return launch_rls<16>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  else if (N <= 32) return launch_rls<32>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
  return launch_rls<64>(Xp, V, per_stream, Yp, S, K, N, T_stride, T, ctrl, state_before, p, P, Wst, st);
}

}  // extern "C"
