// pinv_host.hip -- the reference's pseudo-inverse semantics on the host: btk_pinv for the NC > 2 LCMV constraint Gram matrix
// (a handful of NC x NC matrices per design) and as the checker of the batched GPU solve (pinv_kernels.hip), which serves the
// bins the Cholesky kernel (mvdr_kernels.hip) flags; btk_mvdr_pinv_fallback_host is that solve bin by bin on one host thread
// (15 ms per 64 x 64, 1.7 s per 256 x 256 matrix) -- kept for tests and A/B timing only, no product path calls it.
//
// Reference: pseudoinverse() (beamformer/beamformer.cc:232-289) casts the matrix to complex<float>, takes its SVD with
// LINPACK csvdc (matrix/linpack_c.cc:9516), replaces singular values below dThreshold by 0 (AND reports failure, upon
// which calc_mvdr_weights substitutes the identity, :2381-2383), inverts the others and forms V S^-1 U^H.
// The pseudo-inverse of a matrix is unique, so any accurate SVD reproduces it: here a one-sided Jacobi (Hestenes) SVD of
// the float32-rounded matrix in float64 arithmetic -- what the reference's float32 Householder/QR iteration approximates.
#include "btk_internal.h"
#include <cmath>
#include <complex>
#include <vector>

namespace {
typedef std::complex<double> cd;

// A [M][N] row-major (M >= N handled directly; M < N through the conjugate transpose).  Returns the number of singular
// values below the threshold; invA [N][M].
int pinv_jacobi(const cd* A, int M, int N, double threshold, cd* invA)
{
  if (M < N) {                                   // pinv(A) = pinv(A^H)^H
    std::vector<cd> At((size_t)N * M), it((size_t)M * N);
    for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) At[(size_t)j * M + i] = std::conj(A[(size_t)i * N + j]);
    const int nz = pinv_jacobi(At.data(), N, M, threshold, it.data());
    for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) invA[(size_t)j * M + i] = std::conj(it[(size_t)i * N + j]);
    return nz;
  }
  // G = A (columns rotated until mutually orthogonal), V accumulates the rotations: A V = G, A = G V^H
  std::vector<cd> G((size_t)M * N), V((size_t)N * N, cd(0, 0));
  for (size_t i = 0; i < (size_t)M * N; i++) G[i] = cd((double)(float)A[i].real(), (double)(float)A[i].imag());
  for (int j = 0; j < N; j++) V[(size_t)j * N + j] = cd(1, 0);
  const double eps = 1e-15;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < N - 1; p++)
      for (int q = p + 1; q < N; q++) {
        double alpha = 0, beta = 0; cd gamma(0, 0);
        for (int i = 0; i < M; i++) {
          const cd gp = G[(size_t)i * N + p], gq = G[(size_t)i * N + q];
          alpha += std::norm(gp); beta += std::norm(gq); gamma += std::conj(gp) * gq;
        }
        const double ag = std::abs(gamma);
        if (ag <= eps * std::sqrt(alpha * beta) || ag == 0.0) continue;
        rotated = true;
        const cd ph = gamma / ag;                                  // g_q <- conj(ph) g_q makes the inner product real
        const double zeta = (beta - alpha) / (2.0 * ag);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < M; i++) {
          const cd gp = G[(size_t)i * N + p], gq = std::conj(ph) * G[(size_t)i * N + q];
          G[(size_t)i * N + p] = c * gp - s * gq;
          G[(size_t)i * N + q] = s * gp + c * gq;
        }
        for (int i = 0; i < N; i++) {
          const cd vp = V[(size_t)i * N + p], vq = std::conj(ph) * V[(size_t)i * N + q];
          V[(size_t)i * N + p] = c * vp - s * vq;
          V[(size_t)i * N + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  int below = 0;
  std::vector<double> inv_s2(N);
  for (int k = 0; k < N; k++) {
    double s2 = 0;
    for (int i = 0; i < M; i++) s2 += std::norm(G[(size_t)i * N + k]);
    const double sv = std::sqrt(s2);
    if ((float)sv < (float)threshold) { below++; inv_s2[k] = 0.0; }          // s[k] = 0 (beamformer.cc:262-266)
    else inv_s2[k] = 1.0 / s2;                                               // V (1/s) U^H = V conj(G)^T / s^2
  }
  for (int j = 0; j < N; j++)
    for (int i = 0; i < M; i++) {
      cd x(0, 0);
      for (int k = 0; k < N; k++) x += V[(size_t)j * N + k] * inv_s2[k] * std::conj(G[(size_t)i * N + k]);
      invA[(size_t)j * M + i] = x;
    }
  return below;
}
}  // namespace

extern "C" {

int btk_pinv(const double* A, int M, int N, float threshold, double* invA, int* below_threshold)
{
  if (!A || !invA) return btk_set_error(BTK_ERR_PARAMETER, "btk_pinv: null argument");
  if (M < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_pinv: bad sizes %d x %d", M, N);
  const int nz = pinv_jacobi(reinterpret_cast<const cd*>(A), M, N, threshold, reinterpret_cast<cd*>(invA));
  if (below_threshold) *below_threshold = nz;
  return BTK_OK;
}

// calc_mvdr_weights for the flagged bins, literally (beamformer.cc:2372-2397): invR = pinv(R_k) or the identity when
// pseudoinverse() reports failure; tmpH = invR^H d; w = tmpH / (N d^H... zdotc(tmpH, d)).  Synchronises the stream.
int btk_mvdr_pinv_fallback_host(const void* R, const void* wq, void* W, int K, int N, int first_bin, float threshold,
                           const int* fail_flags, int* identity_count, void* stream)
{
  if (!R || !wq || !W || !fail_flags) return btk_set_error(BTK_ERR_PARAMETER, "btk_mvdr_pinv_fallback_host: null argument");
  if (K < 1 || N < 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_mvdr_pinv_fallback_host: bad sizes");
  hipStream_t st = as_stream(stream);
  std::vector<int> flags(K);
  BTK_HIP_CHECK(hipMemcpyAsync(flags.data(), fail_flags, sizeof(int) * K, hipMemcpyDeviceToHost, st));
  BTK_HIP_CHECK(hipStreamSynchronize(st));
  int nident = 0;
  std::vector<float> r((size_t)2 * N * N), d((size_t)2 * N), w((size_t)2 * N);
  std::vector<cd> Rk((size_t)N * N), inv((size_t)N * N), t(N);
  for (int k = 0; k < K; k++) {
    if (!flags[k] || k + first_bin == 0) continue;                 // global bin 0 keeps its all-ones weight (:2369-2371)
    BTK_HIP_CHECK(hipMemcpy(r.data(), static_cast<const float*>(R) + (size_t)k * 2 * N * N, sizeof(float) * r.size(), hipMemcpyDeviceToHost));
    BTK_HIP_CHECK(hipMemcpy(d.data(), static_cast<const float*>(wq) + (size_t)k * 2 * N, sizeof(float) * d.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)N * N; i++) Rk[i] = cd(r[2 * i], r[2 * i + 1]);
    const int below = pinv_jacobi(Rk.data(), N, N, threshold, inv.data());
    if (below > 0) {                                               // "ret = false" -> gsl_matrix_complex_set_identity(invR)
      nident++;
      std::fill(inv.begin(), inv.end(), cd(0, 0));
      for (int i = 0; i < N; i++) inv[(size_t)i * N + i] = cd(1, 0);
    }
    cd lam(0, 0);
    for (int i = 0; i < N; i++) {                                  // tmpH = invR^H d
      cd acc(0, 0);
      for (int j = 0; j < N; j++) acc += std::conj(inv[(size_t)j * N + i]) * cd(d[2 * j], d[2 * j + 1]);
      t[i] = acc;
      lam += std::conj(acc) * cd(d[2 * i], d[2 * i + 1]);          // zdotc(tmpH, d)
    }
    const cd norm = lam * (double)N;
    for (int i = 0; i < N; i++) { const cd v = t[i] / norm; w[2 * i] = (float)v.real(); w[2 * i + 1] = (float)v.imag(); }
    BTK_HIP_CHECK(hipMemcpy(static_cast<float*>(W) + (size_t)k * 2 * N, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
  }
  if (identity_count) *identity_count = nident;
  return BTK_OK;
}

}  // extern "C"
