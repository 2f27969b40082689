// linpack_f32.h -- the singular values and the convergence flag of the reference's float32 SVD, rounding for rounding.
//
// The reference's pseudoinverse() (beamformer/beamformer.cc:232-289) calls J. Burkardt's C++ LINPACK csvdc
// (matrix/linpack_c.cc:9516-10193, BLAS-1 helpers matrix/blas1_c.cc) on the float32-rounded matrix and reports failure when
//   * csvdc returns INFO != 0 (its shifted-QR iteration did not deflate a singular value within 30 sweeps), or
//   * a singular value is below the threshold;
// SubbandMVDR::calc_mvdr_weights (beamformer.cc:2379-2384) then replaces the inverse by the identity, i.e. the bin becomes
// delay-and-sum.  Whether the float32 iteration converges is a property of the float32 ROUNDINGS, not of the matrix: on the
// 256-microphone diffuse model of BASELINE config C5 it fails on about half the bins.  Reproducing that decision therefore
// needs the same arithmetic in the same order: float32 products and sums without contraction, the float64 detours the C++
// source takes (pow(float, int), abs(complex<float>) == hypotf == sqrt of a float64 sum of squares in glibc 2.35,
// complex / complex == libgcc's __divsc3 evaluated in float64), and the serial summation order of every dot product.
//
// What is NOT needed: U and V.  Their rotations never feed back into s and e, so job = 0 gives the same s, e and INFO as
// the reference's job = 11 -- a converging bin keeps the engine's Cholesky answer, a failing one takes the identity.
//
// One body, two builds: csvdc_values<Ctx>() below is the kernel body (csrc/svd_linpack.hip, one workgroup per matrix:
// Ctx::tid / nthreads / barrier are the workgroup's) and, with a one-thread context, plain serial C++ that g++ compiles
// (tests/cpp/linpack_host.cc) so that the CPU suite can compare it bit for bit with the reference's compiled csvdc
// (oracle/_ref) without a GPU.  Every loop over threads is over independent columns / rows; each column's (row's) own sum
// runs serially inside one thread, in the reference's order.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LPK_FN __host__ __device__ inline
#define LPK_MEMFN __host__ __device__ inline
#else
#define LPK_FN static inline
#define LPK_MEMFN inline
#endif
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace lpk {

struct cf { float re, im; };

LPK_FN cf mk(float a, float b) { cf z; z.re = a; z.im = b; return z; }
// r4_abs (blas1_c.cc:1402): keeps the sign of a negative zero, like the reference
LPK_FN float r4abs(float x) { return (0.0f <= x) ? x : -x; }
LPK_FN float cabs1(cf z) { return r4abs(z.re) + r4abs(z.im); }                          // blas1_c.cc:5
// abs(complex<float>) (glibc hypotf) and cabs2 (blas1_c.cc:56, pow(float, int) is float64): float64 sum of exact squares
LPK_FN float hyp(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }
LPK_FN cf cmul(cf a, cf b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
LPK_FN cf cadd(cf a, cf b) { return mk(a.re + b.re, a.im + b.im); }
LPK_FN cf cneg(cf a) { return mk(-a.re, -a.im); }
LPK_FN cf conj(cf a) { return mk(a.re, -a.im); }
// complex<float> / complex<float>: libgcc's __divsc3 works in the next wider type
LPK_FN cf cdiv(cf p, cf q)
{
  const double a = p.re, b = p.im, c = q.re, d = q.im;
  const double den = c * c + d * d;
  return mk((float)((a * c + b * d) / den), (float)((b * c - a * d) / den));
}
// csign2 (blas1_c.cc:851): |z1| * (z2 / |z2|), a real scale on each component
LPK_FN cf csign2(cf z1, cf z2)
{
  const float a2 = hyp(z2.re, z2.im);
  if (a2 == 0.0f) return mk(0.0f, 0.0f);
  const float a1 = hyp(z1.re, z1.im);
  return mk(a1 * (z2.re / a2), a1 * (z2.im / a2));
}
// one component of scnrm2's scaled sum of squares (blas1_c.cc:1551-1660)
LPK_FN void nrm2_acc(float& scale, float& ssq, float v)
{
  if (v != 0.0f) {
    const float t = r4abs(v);
    if (scale < t) {
      const double q = (double)(scale / t);
      ssq = (float)(1.0 + (double)ssq * (q * q));
      scale = t;
    } else {
      const double q = (double)(t / scale);
      ssq = (float)((double)ssq + q * q);
    }
  }
}
LPK_FN float nrm2(int n, const cf* x)
{
  if (n < 1) return 0.0f;
  float scale = 0.0f, ssq = 1.0f;
  for (int i = 0; i < n; ++i) { nrm2_acc(scale, ssq, x[i].re); nrm2_acc(scale, ssq, x[i].im); }
  return scale * sqrtf(ssq);
}
// srotg (linpack_c.cc:10796): *sa <- r; the z the reference leaves in *sb is never read by csvdc
LPK_FN void rotg(float& sa, float sb, float& c, float& s)
{
  const float roe = (r4abs(sb) < r4abs(sa)) ? sa : sb;
  const float scale = r4abs(sa) + r4abs(sb);
  float r;
  if (scale == 0.0f) { c = 1.0f; s = 0.0f; r = 0.0f; }
  else {
    const float qa = sa / scale, qb = sb / scale;
    r = scale * sqrtf(qa * qa + qb * qb);
    r = ((roe < 0.0f) ? -1.0f : 1.0f) * r;
    c = sa / r;
    s = sb / r;
  }
  sa = r;
}
LPK_FN float r4max(float x, float y) { return (y < x) ? x : y; }

// "Transform S and E so that they are real" (linpack_c.cc:9869-9905).  sc / ec: the complex bidiagonal, m entries each;
// s / e receive the real parts (the imaginary parts are exact zeros from here on: every later operation is real * complex).
LPK_FN void realify(int m, cf* sc, cf* ec, float* s, float* e)
{
  for (int i = 0; i < m; ++i) {
    if (cabs1(sc[i]) != 0.0f) {
      const cf t = mk(hyp(sc[i].re, sc[i].im), 0.0f);
      const cf r = cdiv(sc[i], t);
      sc[i] = t;
      if (i < m - 1) ec[i] = cdiv(ec[i], r);
    }
    if (i == m - 1) break;
    if (cabs1(ec[i]) != 0.0f) {
      const cf t = mk(hyp(ec[i].re, ec[i].im), 0.0f);
      const cf r = cdiv(t, ec[i]);
      ec[i] = t;
      sc[i + 1] = cmul(sc[i + 1], r);
    }
  }
  for (int i = 0; i < m; ++i) { s[i] = sc[i].re; e[i] = ec[i].re; }
}

// The main iteration of csvdc on the real bidiagonal (linpack_c.cc:9909-10190), indices as in the source (1-based l, m).
// Returns INFO; s holds the singular values, descending where converged.
LPK_FN int qr_iterate(int m, float* s, float* e)
{
  const int maxit = 30;
  const int mm = m;
  int iter = 0, info = 0;
  for (;;) {
    if (m == 0) break;
    if (maxit <= iter) { info = m; break; }
    int l = 0, kase;
    for (int ll = 1; ll <= m; ++ll) {
      l = m - ll;
      if (l == 0) break;
      const float test = fabsf(s[l - 1]) + fabsf(s[l]);
      const float ztest = test + fabsf(e[l - 1]);
      if (ztest == test) { e[l - 1] = 0.0f; break; }
    }
    if (l == m - 1) kase = 4;
    else {
      const int lp1 = l + 1, mp1 = m + 1;
      int ls = 0;
      for (int lls = lp1; lls <= mp1; ++lls) {
        ls = m - lls + lp1;
        if (ls == l) break;
        float test = 0.0f;
        if (ls != m) test = test + fabsf(e[ls - 1]);
        if (ls != l + 1) test = test + fabsf(e[ls - 2]);
        const float ztest = test + fabsf(s[ls - 1]);
        if (ztest == test) { s[ls - 1] = 0.0f; break; }
      }
      if (ls == l) kase = 3;
      else if (ls == m) kase = 1;
      else { kase = 2; l = ls; }
    }
    l = l + 1;
    float cs, sn;
    if (kase == 1) {                                     // deflate negligible s(m)
      const int mm1 = m - 1;
      float f = e[m - 2];
      e[m - 2] = 0.0f;
      for (int kk = 1; kk <= mm1; ++kk) {
        const int k = mm1 - kk + l;
        float t1 = s[k - 1];
        rotg(t1, f, cs, sn);
        s[k - 1] = t1;
        if (k != l) { f = -sn * e[k - 2]; e[k - 2] = cs * e[k - 2]; }
      }
    } else if (kase == 2) {                              // split at negligible s(l)
      float f = e[l - 2];
      e[l - 2] = 0.0f;
      for (int k = l; k <= m; ++k) {
        float t1 = s[k - 1];
        rotg(t1, f, cs, sn);
        s[k - 1] = t1;
        f = -sn * e[k - 1];
        e[k - 1] = cs * e[k - 1];
      }
    } else if (kase == 3) {                              // one shifted QR step
      const float scale = r4max(fabsf(s[m - 1]), r4max(fabsf(s[m - 2]), r4max(fabsf(e[m - 2]), r4max(fabsf(s[l - 1]), fabsf(e[l - 1])))));
      const float sm = s[m - 1] / scale, smm1 = s[m - 2] / scale, emm1 = e[m - 2] / scale, sl = s[l - 1] / scale, el = e[l - 1] / scale;
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
      for (int k = l; k <= m - 1; ++k) {
        rotg(f, g, cs, sn);
        if (k != l) e[k - 2] = f;
        f = cs * s[k - 1] + sn * e[k - 1];
        e[k - 1] = cs * e[k - 1] - sn * s[k - 1];
        g = sn * s[k];
        s[k] = cs * s[k];
        rotg(f, g, cs, sn);
        s[k - 1] = f;
        f = cs * e[k - 1] + sn * s[k];
        s[k] = -sn * e[k - 1] + cs * s[k];
        g = sn * e[k];
        e[k] = cs * e[k];
      }
      e[m - 2] = f;
      iter = iter + 1;
    } else {                                             // convergence
      if (s[l - 1] < 0.0f) s[l - 1] = -s[l - 1];
      while (l != mm) {
        if (s[l] <= s[l - 1]) break;
        const float t = s[l - 1]; s[l - 1] = s[l]; s[l] = t;
        l = l + 1;
      }
      iter = 0;
      m = m - 1;
    }
  }
  return info;
}

// Column j of x, rows [i0, i1): y_i += t v_i (caxpy) and sum_i conj(v_i) y_i (cdotc, i ascending).  Element for element the
// plain loops; written in batches of eight rows -- the eight loads first -- because the rows are `ld` elements apart in memory the
// compiler cannot tell apart: a load-modify-store loop otherwise waits for every element's round trip (a matrix in L2) in turn.
LPK_FN void axpy_rows(cf* y, int ld, int i0, int i1, cf t, const cf* v)
{
  int i = i0;
  for (; i + 8 <= i1; i += 8) {
    cf a[8];
    for (int u = 0; u < 8; ++u) a[u] = y[(long)(i + u) * ld];
    for (int u = 0; u < 8; ++u) y[(long)(i + u) * ld] = cadd(a[u], cmul(t, v[i + u]));
  }
  for (; i < i1; ++i) y[(long)i * ld] = cadd(y[(long)i * ld], cmul(t, v[i]));
}
LPK_FN cf dot_rows(const cf* y, int ld, int i0, int i1, const cf* v)
{
  cf dot = mk(0.0f, 0.0f);
  int i = i0;
  for (; i + 8 <= i1; i += 8) {
    cf a[8];
    for (int u = 0; u < 8; ++u) a[u] = y[(long)(i + u) * ld];
    for (int u = 0; u < 8; ++u) dot = cadd(dot, cmul(conj(v[i + u]), a[u]));
  }
  for (; i < i1; ++i) dot = cadd(dot, cmul(conj(v[i]), y[(long)i * ld]));
  return dot;
}

// work_i = sum_{j >= i0} e_j x(i, j) for the rows i >= i0, j ascending, zero e_j skipped (linpack_c.cc:9806-9816: one caxpy per
// column, so each row's sum runs over j); a thread per row
template <class Ctx>
LPK_FN void row_sums(const Ctx& cx, const cf* x, int ld, int i0, int n, int p, const cf* ev, cf* work)
{
  for (int i = i0 + cx.tid(); i < n; i += cx.nthreads()) {
    cf acc = mk(0.0f, 0.0f);
    const cf* xr = x + (long)i * ld;
    for (int j = i0; j < p; ++j) {
      const cf ej = ev[j];
      if (cabs1(ej) != 0.0f) acc = cadd(acc, cmul(ej, xr[j]));
    }
    work[i] = acc;
  }
  cx.barrier();
}

// Work arrays of one matrix (LDS in the kernel).  Sizes: col, work: n; ev: p; sc, ec: max(n + 1, p) + 1; t: 2.
struct Work { cf *col, *ev, *work, *sc, *ec, *t; int* flag; };

// csvdc with job = 0 on the n x p matrix x (element (i, j) at x[i * ld + j], destroyed), linpack_c.cc:9676-9867 for the
// reduction.  s, e: min(n + 1, p) floats each (thread 0 writes them); returns INFO on thread 0 (other threads: undefined).
// csvdc_reduce: the Householder reduction to the real bidiagonal (s, e; thread 0 writes them, complete behind the final barrier);
// csvdc_values = csvdc_reduce + the QR iteration on it.
template <class Ctx>
LPK_FN void csvdc_reduce(Ctx& cx, cf* x, int ld, int n, int p, Work w, float* s, float* e)
{
  const int tid = cx.tid(), nth = cx.nthreads();
  const int nct = (n - 1 < p) ? n - 1 : p;
  const int nrt0 = (p - 2 < n) ? p - 2 : n;
  const int nrt = nrt0 > 0 ? nrt0 : 0;
  const int lu = nct > nrt ? nct : nrt;
  for (int L = 0; L < lu; ++L) {                         // L = l - 1
    const bool colstep = L < nct;
    if (colstep) {
      for (int i = L + tid; i < n; i += nth) w.col[i] = x[(long)i * ld + L];
      cx.barrier();
      const float nrm_col = cx.nrm2(n - L, w.col + L);       // (every thread calls; thread 0 gets the value)
      if (tid == 0) {
        cf sl = mk(nrm_col, 0.0f);
        int scaled = 0;
        if (cabs1(sl) != 0.0f) {
          if (cabs1(w.col[L]) != 0.0f) sl = csign2(sl, w.col[L]);
          w.t[0] = cdiv(mk(1.0f, 0.0f), sl);
          scaled = 1;
        }
        w.sc[L] = cneg(sl);
        w.flag[0] = scaled;
      }
      cx.barrier();
      if (w.flag[0]) {
        const cf t = w.t[0];
        for (int i = L + tid; i < n; i += nth) {
          cf v = cmul(t, w.col[i]);
          if (i == L) v = cadd(mk(1.0f, 0.0f), v);
          w.col[i] = v;
        }
      }
      cx.barrier();
    }
    // columns j > L: the Householder reflection (each thread its own columns, serial over rows), then row L into e
    {
      const bool reflect = colstep && w.flag[0];
      for (int j = L + 1 + tid; j < p; j += nth) {
        if (reflect) {
          const cf dot = dot_rows(x + j, ld, L, n, w.col);
          const cf t = cdiv(cneg(dot), w.col[L]);
          if (cabs1(t) != 0.0f) axpy_rows(x + j, ld, L, n, t, w.col);
        }
        w.ev[j] = conj(x[(long)L * ld + j]);
      }
    }
    cx.barrier();
    if (L < nrt) {
      const float nrm_row = cx.nrm2(p - L - 1, w.ev + L + 1);
      if (tid == 0) {
        cf el = mk(nrm_row, 0.0f);
        int scaled = 0;
        if (cabs1(el) != 0.0f) {
          if (cabs1(w.ev[L + 1]) != 0.0f) el = csign2(el, w.ev[L + 1]);
          w.t[0] = cdiv(mk(1.0f, 0.0f), el);
          scaled = 1;
        }
        el = cneg(conj(el));
        w.ec[L] = el;
        w.flag[0] = scaled;
        w.flag[1] = (L + 1 < n && cabs1(el) != 0.0f) ? 1 : 0;
      }
      cx.barrier();
      if (w.flag[0]) {
        const cf t = w.t[0];
        for (int j = L + 1 + tid; j < p; j += nth) {
          cf v = cmul(t, w.ev[j]);
          if (j == L + 1) v = cadd(mk(1.0f, 0.0f), v);
          w.ev[j] = v;
        }
      }
      cx.barrier();
      if (w.flag[1]) {
        // work_i = sum_j e_j x(i, j), j ascending (one caxpy per column in the source: the order of each row's sum is j)
        cx.row_sums(x, ld, L + 1, n, p, w.ev, w.work);        // (ends with a barrier)
        const cf e1 = w.ev[L + 1];
        for (int j = L + 1 + tid; j < p; j += nth) {
          const cf c = conj(cdiv(cneg(w.ev[j]), e1));
          if (cabs1(c) != 0.0f) axpy_rows(x + j, ld, L + 1, n, c, w.work);
        }
      }
      cx.barrier();
    }
  }
  const int m = (p < n + 1) ? p : n + 1;
  if (tid == 0) {
    if (nct < p) w.sc[nct] = x[(long)nct * ld + nct];
    if (n < m) w.sc[m - 1] = mk(0.0f, 0.0f);
    if (nrt + 1 < m) w.ec[nrt] = x[(long)nrt * ld + (m - 1)];
    w.ec[m - 1] = mk(0.0f, 0.0f);
    realify(m, w.sc, w.ec, s, e);
  }
  cx.barrier();
}

template <class Ctx>
LPK_FN int csvdc_values(Ctx& cx, cf* x, int ld, int n, int p, Work w, float* s, float* e)
{
  csvdc_reduce(cx, x, ld, n, p, w, s, e);
  // the serial recurrence: one thread in the host build; in the kernel the first wavefront walks it together (Ctx::qr)
  const int info = cx.qr((p < n + 1) ? p : n + 1, s, e);
  cx.barrier();
  return info;
}

struct SerialCtx {
  LPK_MEMFN int tid() const { return 0; }
  LPK_MEMFN int nthreads() const { return 1; }
  LPK_MEMFN void barrier() const {}
  LPK_MEMFN int qr(int m, float* s, float* e) const { return qr_iterate(m, s, e); }
  LPK_MEMFN float nrm2(int n, const cf* x) const { return lpk::nrm2(n, x); }
  LPK_MEMFN void row_sums(const cf* x, int ld, int i0, int n, int p, const cf* ev, cf* work) const { lpk::row_sums(*this, x, ld, i0, n, p, ev, work); }
};

}  // namespace lpk
