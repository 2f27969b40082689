// BeamformerWeights with the reference's accessors (reference beamformer/beamformer.h:53-67): names, return types, aliasing.
// Host code only (weight design runs in float64 on the host) -- no GPU needed.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include "beamformer/beamformer.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main()
{
  const unsigned M = 64, N = 4;
  gsl_vector* delays = gsl_vector_calloc(N);
  for (unsigned c = 0; c < N; c++) gsl_vector_set(delays, c, 1.0e-4 * c);
  BeamformerWeights w(M, N, false, 1);
  CHECK(!w.isHalfBandShift() && w.fftLen() == M && w.chanN() == N && w.NC() == 1);
  w.calcMainlobe(16000.0f, delays, true);
  // the reference's signatures: arrays of per-bin gsl objects
  gsl_vector_complex** wq = w.wq();
  gsl_matrix_complex** B = w.B();
  gsl_vector_complex** wa = w.wa();
  gsl_vector_complex** ta = w.arrayManifold();
  gsl_vector_complex** csd = w.CSDs();
  gsl_vector_complex* wp1 = w.wp1();
  CHECK(wq[3]->size == N && ta[3]->size == N && wa[3]->size == N - 1 && B[3]->size1 == N && B[3]->size2 == N - 1);
  CHECK(csd[M - 1]->size == N * N && wp1->size == M);
  CHECK(w.wq_f(5) == wq[5] && w.wl_f(5)->size == N && w.B_f(5) == B[5]);
  // |wq_k[c]| = 1 / N, and ta == wq after calcMainlobe
  for (unsigned c = 0; c < N; c++) {
    const gsl_complex z = gsl_vector_complex_get(wq[7], c), t = gsl_vector_complex_get(ta[7], c);
    CHECK(std::fabs(std::hypot(GSL_REAL(z), GSL_IMAG(z)) - 1.0 / N) < 1e-12);
    CHECK(GSL_REAL(z) == GSL_REAL(t) && GSL_IMAG(z) == GSL_IMAG(t));
  }
  // wq^T B = 0 through the accessors (the reference's calc_blocking_matrix_ builds the complement of conj(wq), beamformer.cc:373-454)
  for (unsigned j = 0; j < N - 1; j++) {
    double re = 0, im = 0;
    for (unsigned c = 0; c < N; c++) {
      const gsl_complex a = gsl_vector_complex_get(wq[9], c), b = gsl_matrix_complex_get(B[9], c, j);
      re += GSL_REAL(a) * GSL_REAL(b) - GSL_IMAG(a) * GSL_IMAG(b);
      im += GSL_REAL(a) * GSL_IMAG(b) + GSL_IMAG(a) * GSL_REAL(b);
    }
    CHECK(std::fabs(re) < 1e-12 && std::fabs(im) < 1e-12);
  }
  // the views alias the object's storage: a write through wq() is what the object computes with
  gsl_vector_complex_set(wq[2], 1, gsl_complex_rect(0.25, -0.5));
  CHECK(w.wq_v[2 * N + 1] == std::complex<double>(0.25, -0.5));
  gsl_vector_complex* a = gsl_vector_complex_calloc(N - 1);
  gsl_vector_complex_set(a, 0, gsl_complex_rect(0.1, 0.2));
  w.calcSidelobeCancellerU_f(9, a);
  CHECK(GSL_REAL(gsl_vector_complex_get(wa[9], 0)) == 0.1 && GSL_IMAG(gsl_vector_complex_get(wa[9], 0)) == 0.2);
  double nl = 0;
  for (unsigned c = 0; c < N; c++) nl += std::norm(w.wl_v[9 * N + c]);
  CHECK(nl > 0);                                                   // wl = B wa
  // halfBandShift: every one of the M bins has its own vector, the partner of bin k is bin M-1-k (beamformer.cc:515-527)
  BeamformerWeights h(M, N, true, 1);
  CHECK(h.isHalfBandShift());
  h.calcMainlobe(16000.0f, delays, true);
  for (unsigned c = 0; c < N; c++) {
    const gsl_complex z0 = gsl_vector_complex_get(h.wq()[0], c), z1 = gsl_vector_complex_get(h.wq()[M - 1], c);
    CHECK(std::fabs(GSL_REAL(z0) - GSL_REAL(z1)) < 1e-15 && std::fabs(GSL_IMAG(z0) + GSL_IMAG(z1)) < 1e-15);
  }
  gsl_vector_complex_free(a); gsl_vector_free(delays);
  printf("ok\n");
  return 0;
}
