// btk_api.hip -- C-ABI plumbing of libbtkhip: error state, device selection, filter-bank plans and
// the host-side (float64) weight design that runs once per look direction.
#include "btk_internal.h"
#include <cstdlib>
#include <cmath>
#include <complex>
#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string g_last_error;
using cd = std::complex<double>;
}

const btk_switches_t& btk_switches()
{
  static const btk_switches_t sw = [] {
    auto flag = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
    btk_switches_t s;
    s.disable_analysis512 = flag("BTK_DISABLE_ANALYSIS512"); s.disable_synthesis512 = flag("BTK_DISABLE_SYNTHESIS512");
    s.disable_fast = flag("BTK_DISABLE_FAST"); s.disable_fused = flag("BTK_DISABLE_FUSED");
    s.nlms_v1 = flag("BTK_NLMS_V1"); s.wpe_noskip = flag("BTK_WPE_NOSKIP"); s.wpe_timing = flag("BTK_WPE_TIMING"); s.syn_narrow = flag("BTK_SYN_NARROW"); s.rls_packed = flag("BTK_RLS_PACKED"); s.wpe_herk_blocks = flag("BTK_WPE_HERK_BLOCKS"); s.wpe_solve_panel = flag("BTK_WPE_SOLVE_PANEL"); s.wpe_solve_reg = flag("BTK_WPE_SOLVE_REG"); s.wpe_predict_valu = flag("BTK_WPE_PREDICT_VALU"); s.wpe_lagprod_f32 = flag("BTK_WPE_LAGPROD_F32"); { const char* e = getenv("BTK_WPE_LAGPROD_WAVES"); s.wpe_lagprod_waves = e ? atoi(e) : 2; }
    s.nlms_alt = num("BTK_NLMS_ALT", 0); s.mvdr_reg_min = num("BTK_MVDR_REG_MIN", 64); s.fused_var = num("BTK_FUSED_VAR", -1);
    s.pf_jb = num("BTK_PF_JB", 0);
    s.pf_tpw = num("BTK_PF_TPW", 0);
    s.pf_mfma_min = num("BTK_PF_MFMA_MIN", 8);
    return s;
  }();                                                   // C++11 magic static: initialised once, thread-safe
  return sw;
}

int btk_set_error(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

extern "C" {

const char* btk_last_error(void) { return g_last_error.c_str(); }
int btk_version(void) { return 100; }

int btk_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int btk_set_device(int device)
{
  BTK_HIP_CHECK(hipSetDevice(device));
  return BTK_OK;
}

int btk_synchronize(void* stream)
{
  BTK_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  return BTK_OK;
}

// ---------------------------------------------------------------------------------------------
// OverSampledDFTFilterBank ctor (reference modulated/modulated.cc:232-268)
int btk_fb_create(btk_fb_t** out, int M, int m, int r, int dct, int synthesis, const double* prototype)
{
  if (!out || !prototype) return btk_set_error(BTK_ERR_PARAMETER, "btk_fb_create: null argument");
  if (M < 64 || M > 2048 || (M & (M - 1)))
    return btk_set_error(BTK_ERR_PARAMETER, "M=%d must be a power of two in [64, 2048]", M);
  if (m < 1 || m > 16) return btk_set_error(BTK_ERR_PARAMETER, "m=%d out of range [1,16]", m);
  if (r < 0 || (M >> r) < 4) return btk_set_error(BTK_ERR_PARAMETER, "r=%d leaves a frame shift < 4", r);
  btk_fb* fb = new btk_fb();
  fb->M = M; fb->m = m; fb->r = r; fb->R = 1 << r; fb->D = M / fb->R; fb->K = M / 2 + 1;
  fb->dct = dct; fb->synthesis = synthesis ? 1 : 0; fb->gain_factor = 1;
  fb->kx0 = 0; fb->kx1 = fb->K;
  fb->laN = 0;
  switch (dct) {                                    // modulated.cc:246-264
    case 1: fb->pd = m * fb->R - 1; break;
    case 2:
      if (synthesis) fb->pd = m * fb->R / 2;
      else { fb->pd = m * fb->R - 1; fb->laN = m * fb->R / 2 - 1; }
      break;
    default: fb->pd = 2 * m - 1; break;
  }
  std::vector<float> hp((size_t)m * M);
  for (size_t i = 0; i < hp.size(); i++) hp[i] = (float)prototype[i];
  std::vector<float2> tw(M);
  for (int j = 0; j < M; j++) {
    const double a = 2.0 * M_PI * (double)j / (double)M;
    tw[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  fb->d_proto = nullptr; fb->d_tw = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&fb->d_proto), sizeof(float) * hp.size());
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&fb->d_tw), sizeof(float2) * M);
  if (e == hipSuccess) e = hipMemcpy(fb->d_proto, hp.data(), sizeof(float) * hp.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(fb->d_tw, tw.data(), sizeof(float2) * M, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (fb->d_proto) (void)hipFree(fb->d_proto);
    if (fb->d_tw) (void)hipFree(fb->d_tw);
    delete fb;
    return btk_set_error(e == hipErrorOutOfMemory ? BTK_ERR_ALLOCATION : BTK_ERR_HIP,
                         "btk_fb_create: %s", hipGetErrorString(e));
  }
  *out = fb;
  return BTK_OK;
}

void btk_fb_destroy(btk_fb_t* fb)
{
  if (!fb) return;
  (void)hipFree(fb->d_proto);
  (void)hipFree(fb->d_tw);
  delete fb;
}

int btk_fb_processing_delay(const btk_fb_t* fb) { return fb ? fb->pd : -1; }
int btk_fb_lookahead(const btk_fb_t* fb) { return fb ? fb->laN : -1; }

// frames emitted before jiterator_error: the look-ahead consumes laN blocks (modulated.cc:425-439),
// every remaining input block yields a frame, then processing_delay zero-padded frames (:440-466).
long btk_fb_analysis_num_frames(const btk_fb_t* fb, long nsamples)
{
  if (!fb || nsamples < 0) return -1;
  const long nblk = (nsamples + fb->D - 1) / fb->D;
  // a source that ends while the look-ahead blocks are being skipped sets is_end_ before the first frame
  // (update_buffer_, modulated.cc:423-438): no frame at all, not even the zero-padded ones
  if (nblk < fb->laN) return 0;
  return nblk - fb->laN + fb->pd;
}

long btk_fb_synthesis_num_blocks(const btk_fb_t* fb, long nframes)
{
  if (!fb) return -1;
  return nframes > fb->pd ? nframes - fb->pd : 0;
}

// ---------------------------------------------------------------------------------------------
// Host-side weight design (float64).  These are the engine's own implementations of the
// reference's one-off weight computations; the streaming work stays on the GPU.

// BeamformerWeights::calcMainlobe, halfBandShift == false (beamformer/beamformer.cc:527-553)
int btk_weights_mainlobe(int M, int N, float samplerate, const double* delays, double* wq_out)
{
  if (M < 2 || N < 1 || !delays || !wq_out) return btk_set_error(BTK_ERR_PARAMETER, "btk_weights_mainlobe: bad argument");
  cd* wq = reinterpret_cast<cd*>(wq_out);
  const int half = M / 2;
  const double invN = (double)N;
  for (int c = 0; c < N; c++) wq[c] = cd(1.0, 0.0) / invN;
  for (int k = 1; k < half; k++)
    for (int c = 0; c < N; c++) {
      const double ph = -2.0 * M_PI * k * delays[c] * samplerate / M;
      wq[(size_t)k * N + c] = std::polar(1.0, ph) / invN;
      wq[(size_t)(M - k) * N + c] = std::polar(1.0, -ph) / invN;
    }
  for (int c = 0; c < N; c++) {
    const double ph = -M_PI * samplerate * delays[c];
    wq[(size_t)half * N + c] = std::polar(1.0, ph) / invN;
  }
  return BTK_OK;
}

// BeamformerWeights::calcMainlobe with halfBandShift_ == true (beamformer.cc:515-527): bin k sits at (k + 0.5) fs / M, all M bins are
// filled, and the conjugate partner of bin k is bin M - 1 - k (not M - k).
int btk_weights_mainlobe_halfband(int M, int N, float samplerate, const double* delays, double* wq_out)
{
  if (M < 2 || N < 1 || !delays || !wq_out) return btk_set_error(BTK_ERR_PARAMETER, "btk_weights_mainlobe_halfband: bad argument");
  cd* wq = reinterpret_cast<cd*>(wq_out);
  const int half = M / 2;
  const double dN = (double)N;
  const float fshift = 0.5f;
  for (int k = 0; k < half; k++)
    for (int c = 0; c < N; c++) {
      const double ph = -2.0 * M_PI * (fshift + k) * samplerate * delays[c] / M;
      wq[(size_t)k * N + c] = std::polar(1.0, ph) / dN;
      wq[(size_t)(M - 1 - k) * N + c] = std::polar(1.0, -ph) / dN;
    }
  return BTK_OK;
}

// calc_blocking_matrix_ (beamformer.cc:373-454): classical Gram-Schmidt over the first N-NC
// columns of the projector I - conj(a) a^T / |a|^2.
int btk_weights_blocking_matrix(const double* a_in, int N, int NC, double* B_out)
{
  const int bs = N - NC;
  if (bs <= 0) return btk_set_error(BTK_ERR_DIMENSION, "The number of sensors %d > the number of constraints %d", N, NC);
  const cd* a = reinterpret_cast<const cd*>(a_in);
  cd* B = reinterpret_cast<cd*>(B_out);
  double nrm2 = 0.0;
  for (int i = 0; i < N; i++) nrm2 += std::norm(a[i]);
  std::vector<cd> col(N);
  for (int j = 0; j < bs; j++) {
    const cd aj = a[j] / nrm2;
    for (int i = 0; i < N; i++) col[i] = (i == j ? cd(1.0, 0.0) : cd(0.0, 0.0)) - std::conj(a[i]) * aj;
    for (int q = 0; q < j; q++) {
      cd ip(0.0, 0.0);
      for (int i = 0; i < N; i++) ip += std::conj(B[(size_t)i * bs + q]) * col[i];
      for (int i = 0; i < N; i++) col[i] -= ip * B[(size_t)i * bs + q];
    }
    double nn = 0.0;
    for (int i = 0; i < N; i++) nn += std::norm(col[i]);
    nn = std::sqrt(nn);
    for (int i = 0; i < N; i++) B[(size_t)i * bs + j] = col[i] / nn;
  }
  return BTK_OK;
}

// calcSidelobeCancellerU_f (beamformer.cc:752-767)
int btk_weights_sidelobe(const double* B_in, const double* wa_in, int N, int NC, double* wl_out)
{
  const int bs = N - NC;
  if (bs <= 0) return btk_set_error(BTK_ERR_DIMENSION, "btk_weights_sidelobe: N=%d NC=%d", N, NC);
  const cd* B = reinterpret_cast<const cd*>(B_in);
  const cd* wa = reinterpret_cast<const cd*>(wa_in);
  cd* wl = reinterpret_cast<cd*>(wl_out);
  for (int i = 0; i < N; i++) {
    cd acc(0.0, 0.0);
    for (int j = 0; j < bs; j++) acc += B[(size_t)i * bs + j] * wa[j];
    wl[i] = acc;
  }
  return BTK_OK;
}

// Effective weights of SubbandGSC::next: bin 0 uses wq alone (beamformer.cc:1288-1291), bins
// 1..M/2 use wq - wl with the optional w/(|w| N) normalisation of calc_gsc_output (:1228-1237).
int btk_weights_gsc_effective(const double* wq_in, const double* wl_in, int M, int N, int normalize, float* w_out)
{
  if (!wq_in || !w_out) return btk_set_error(BTK_ERR_PARAMETER, "btk_weights_gsc_effective: null argument");
  const cd* wq = reinterpret_cast<const cd*>(wq_in);
  const cd* wl = reinterpret_cast<const cd*>(wl_in);
  const int K = M / 2 + 1;
  std::vector<cd> w(N);
  for (int k = 0; k < K; k++) {
    for (int c = 0; c < N; c++)
      w[c] = (k == 0 || !wl) ? wq[(size_t)k * N + c] : wq[(size_t)k * N + c] - wl[(size_t)k * N + c];
    if (normalize && k > 0 && wl) {
      double nn = 0.0;
      for (int c = 0; c < N; c++) nn += std::norm(w[c]);
      nn = std::sqrt(nn);
      for (int c = 0; c < N; c++) w[c] /= (nn * N);
    }
    for (int c = 0; c < N; c++) {
      w_out[2 * ((size_t)k * N + c)] = (float)w[c].real();
      w_out[2 * ((size_t)k * N + c) + 1] = (float)w[c].imag();
    }
  }
  return BTK_OK;
}

// u = wa^H B^T and back (B^T has orthonormal rows, so wa^H = u conj(B)); see nlms_kernels.hip
int btk_nlms_wa_to_u(const double* waH_in, const double* B_in, int N, double* u_out)
{
  if (N < 2) return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_wa_to_u: N=%d", N);
  const cd* wa = reinterpret_cast<const cd*>(waH_in);
  const cd* B = reinterpret_cast<const cd*>(B_in);
  cd* u = reinterpret_cast<cd*>(u_out);
  const int bs = N - 1;
  for (int n = 0; n < N; n++) {
    cd acc(0.0, 0.0);
    for (int i = 0; i < bs; i++) acc += wa[i] * B[(size_t)n * bs + i];
    u[n] = acc;
  }
  return BTK_OK;
}

// The Nc - 1 orthonormal vectors that, with vs / |vs|, span the orthogonal complement of span(conj(B)) (B [N][N-NC] from
// btk_weights_blocking_matrix(vs, N, NC)): conj(B) B^T = I - vs vs^H / |vs|^2 - sum_j c_j c_j^H.  Gram-Schmidt on the
// columns of that residual projector, largest column first.  cx [NC-1][N] complex128.
int btk_nlms_constraint_vectors(const double* vs_in, const double* B_in, int N, int NC, double* cx_out)
{
  if (!vs_in || !B_in || !cx_out) return btk_set_error(BTK_ERR_PARAMETER, "btk_nlms_constraint_vectors: null argument");
  if (N < 2 || NC < 1 || NC >= N) return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_constraint_vectors: N=%d NC=%d", N, NC);
  const cd* vs = reinterpret_cast<const cd*>(vs_in);
  const cd* B = reinterpret_cast<const cd*>(B_in);
  cd* cx = reinterpret_cast<cd*>(cx_out);
  const int bs = N - NC;
  double vv = 0.0;
  for (int i = 0; i < N; i++) vv += std::norm(vs[i]);
  if (!(vv > 0.0)) return btk_set_error(BTK_ERR_NUMERIC, "btk_nlms_constraint_vectors: zero array manifold");
  std::vector<cd> R((size_t)N * N);                        // residual projector, Hermitian
  for (int a = 0; a < N; a++)
    for (int b = 0; b < N; b++) {
      cd acc = (a == b ? cd(1, 0) : cd(0, 0)) - vs[a] * std::conj(vs[b]) / vv;
      for (int i = 0; i < bs; i++) acc -= std::conj(B[(size_t)a * bs + i]) * B[(size_t)b * bs + i];
      R[(size_t)a * N + b] = acc;
    }
  std::vector<char> used(N, 0);
  for (int j = 0; j < NC - 1; j++) {
    int best = -1; double bn = -1.0;
    std::vector<cd> v(N), bv(N);
    for (int col = 0; col < N; col++) {
      if (used[col]) continue;
      for (int a = 0; a < N; a++) v[a] = R[(size_t)a * N + col];
      for (int q = 0; q < j; q++) {                       // orthogonalise against the vectors found so far
        cd ip(0, 0);
        for (int a = 0; a < N; a++) ip += std::conj(cx[(size_t)q * N + a]) * v[a];
        for (int a = 0; a < N; a++) v[a] -= cx[(size_t)q * N + a] * ip;
      }
      double nn = 0.0;
      for (int a = 0; a < N; a++) nn += std::norm(v[a]);
      if (nn > bn) { bn = nn; best = col; bv = v; }
    }
    if (best < 0 || !(bn > 1e-20)) return btk_set_error(BTK_ERR_NUMERIC, "btk_nlms_constraint_vectors: rank of the residual projector < NC - 1");
    used[best] = 1;
    const double inv = 1.0 / std::sqrt(bn);
    for (int a = 0; a < N; a++) cx[(size_t)j * N + a] = bv[a] * inv;
  }
  return BTK_OK;
}

int btk_nlms_u_to_wa_nc(const double* u_in, const double* B_in, int N, int NC, double* waH_out)
{
  if (N < 2 || NC < 1 || NC >= N) return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_u_to_wa_nc: N=%d NC=%d", N, NC);
  const cd* u = reinterpret_cast<const cd*>(u_in);
  const cd* B = reinterpret_cast<const cd*>(B_in);
  cd* wa = reinterpret_cast<cd*>(waH_out);
  const int bs = N - NC;
  for (int i = 0; i < bs; i++) {
    cd acc(0.0, 0.0);
    for (int n = 0; n < N; n++) acc += u[n] * std::conj(B[(size_t)n * bs + i]);
    wa[i] = acc;
  }
  return BTK_OK;
}

int btk_nlms_u_to_wa(const double* u_in, const double* B_in, int N, double* waH_out)
{
  if (N < 2) return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_u_to_wa: N=%d", N);
  const cd* u = reinterpret_cast<const cd*>(u_in);
  const cd* B = reinterpret_cast<const cd*>(B_in);
  cd* wa = reinterpret_cast<cd*>(waH_out);
  const int bs = N - 1;
  for (int i = 0; i < bs; i++) {
    cd acc(0.0, 0.0);
    for (int n = 0; n < N; n++) acc += u[n] * std::conj(B[(size_t)n * bs + i]);
    wa[i] = acc;
  }
  return BTK_OK;
}

}  // extern "C"

// LCMV quiescent weights with NC = 2 constraints (look direction + one null):
// BeamformerWeights::calcMainlobe2 / calcMainlobeN (reference beamformer/beamformer.cc:572-721) with
// calc_null_beamformer_ (:299-363) and the thresholded closed-form 2x2 inverse calc_inverse_22mat_ (:181-221).
// The reference's treatment of bin M/2 (:692-703) is reproduced as written, see the comment there.
namespace {
void lcmv_solve2(cd* wt, const cd* wj, int N)
{
  cd g00(0, 0), g01(0, 0), g10(0, 0), g11(0, 0);
  for (int i = 0; i < N; i++) {
    g00 += std::conj(wt[i]) * wt[i]; g01 += std::conj(wt[i]) * wj[i];
    g10 += std::conj(wj[i]) * wt[i]; g11 += std::conj(wj[i]) * wj[i];
  }
  cd det = g00 * g11 - g01 * g10;
  if (std::abs(det) < 1.0e-7) { g00 += 0.01; g11 += 0.01; det = g00 * g11 - g01 * g10; }
  const cd v0 = g11 / det, v1 = -(g10 / det);                // first column of the inverse
  for (int i = 0; i < N; i++) wt[i] = wt[i] * v0 + wj[i] * v1;
}
}  // namespace

extern "C" int btk_weights_mainlobe_2(int M, int N, float samplerate, const double* delaysT, const double* delaysI, double* wq_out)
{
  if (N < 2) return btk_set_error(BTK_ERR_DIMENSION, "The number of channels must be > 2 but it is %d\n", N);
  int rc = btk_weights_mainlobe(M, N, samplerate, delaysT, wq_out);
  if (rc) return rc;
  cd* wq = reinterpret_cast<cd*>(wq_out);
  const int half = M / 2;
  std::vector<cd> wj(N);
  for (int c = 0; c < N; c++) wq[c] = cd(1.0 / N, 0.0);
  for (int k = 1; k < half; k++) {
    cd* vec = wq + (size_t)k * N;
    for (int c = 0; c < N; c++) {
      vec[c] *= (double)N;
      wj[c] = std::polar(1.0, -2.0 * M_PI * k * samplerate * delaysI[c] / M);
    }
    lcmv_solve2(vec, wj.data(), N);
  }
  cd* vec = wq + (size_t)half * N;                          // bin M/2: literal reference behaviour (:692-703)
  for (int c = 0; c < N; c++) {
    vec[c] = std::polar(1.0, -M_PI * samplerate * delaysI[c]) / (double)N;
    lcmv_solve2(vec, wj.data(), N);                         // wj still holds bin M/2-1
  }
  return BTK_OK;
}

// LCMV quiescent weights with NC >= 2 constraints (look direction + NC-1 nulls): BeamformerWeights::calcMainlobeN
// (reference beamformer/beamformer.cc:600-721).  NC = 2 is btk_weights_mainlobe_2 (closed-form 2x2 inverse); for
// NC > 2 the reference inverts the NC x NC Gram matrix C^H C with pseudoinverse(invMat, invMat) (:352-355): float32
// SVD, singular values below 1e-8 zeroed, the return value ignored -- btk_pinv (pinv_host.hip) has exactly that rule.
namespace {
bool lcmv_solve_n(cd* wt, const std::vector<std::vector<cd> >& wj, int N, int NC)
{
  std::vector<cd> G((size_t)NC * NC), inv((size_t)NC * NC, cd(0, 0));
  auto col = [&](int j, int i) -> cd { return j == 0 ? wt[i] : wj[j - 1][i]; };
  for (int a = 0; a < NC; a++)
    for (int b = 0; b < NC; b++) {
      cd acc(0, 0);
      for (int i = 0; i < N; i++) acc += std::conj(col(a, i)) * col(b, i);
      G[(size_t)a * NC + b] = acc;
    }
  if (btk_pinv(reinterpret_cast<const double*>(G.data()), NC, NC, 1.0e-8f, reinterpret_cast<double*>(inv.data()), nullptr) != BTK_OK)
    return false;
  // wt <- C inv g, g = e_0
  std::vector<cd> out(N);
  for (int i = 0; i < N; i++) {
    cd acc(0, 0);
    for (int j = 0; j < NC; j++) acc += col(j, i) * inv[(size_t)j * NC + 0];
    out[i] = acc;
  }
  for (int i = 0; i < N; i++) wt[i] = out[i];
  return true;
}
}  // namespace

extern "C" int btk_weights_mainlobe_n(int M, int N, float samplerate, const double* delaysT, const double* delaysIs, int NC,
                                      double* wq_out)
{
  if (NC < 2 || NC > N)
    return btk_set_error(BTK_ERR_DIMENSION, "1 < the number of constraints %d <= the number of sensors %d.\n", NC, N);
  if (NC == 2) return btk_weights_mainlobe_2(M, N, samplerate, delaysT, delaysIs, wq_out);
  int rc = btk_weights_mainlobe(M, N, samplerate, delaysT, wq_out);
  if (rc) return rc;
  cd* wq = reinterpret_cast<cd*>(wq_out);
  const int half = M / 2;
  std::vector<std::vector<cd> > wj(NC - 1, std::vector<cd>(N));
  for (int c = 0; c < N; c++) wq[c] = cd(1.0 / N, 0.0);
  for (int k = 1; k < half; k++) {
    cd* vec = wq + (size_t)k * N;
    for (int c = 0; c < N; c++) {
      vec[c] *= (double)N;
      for (int n = 0; n < NC - 1; n++) wj[n][c] = std::polar(1.0, -2.0 * M_PI * k * samplerate * delaysIs[(size_t)n * N + c] / M);
    }
    if (!lcmv_solve_n(vec, wj, N, NC)) return btk_set_error(BTK_ERR_NUMERIC, "calc_null_beamformer_() failed\n");
  }
  cd* vec = wq + (size_t)half * N;                          // bin M/2: literal reference behaviour (:692-703)
  for (int c = 0; c < N; c++) {
    vec[c] = std::polar(1.0, -M_PI * samplerate * delaysIs[(size_t)(NC - 2) * N + c]) / (double)N;   // the last n wins
    if (!lcmv_solve_n(vec, wj, N, NC)) return btk_set_error(BTK_ERR_NUMERIC, "calc_null_beamformer_() failed\n");
  }
  return BTK_OK;
}

