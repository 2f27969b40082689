// pf_kernels.hip -- Zelinski post-filter for gfx950.
//
// Replaces ZelinskiFilter_f / ZelinskiFilter / ZelinskiPostFilter::next
// (reference postfilter/postfilter.cc:57-219, 424-491).
//
// The reference keeps N(N-1)/2 cross spectral densities phi_ij per bin and updates each of them
// per frame (O(N^2)); only their SUM and the sum of the auto spectral densities enter the gain:
//     W = clamp( f(sum_{i<j} phi_ij) / sum_i phi_ii * 2/(N-1), 1e-4, 1 ),  f = |.| or max(Re .,0)
// Every phi obeys the same first-order recursion with the same alpha, so the sums obey it too:
//     Phi_t = a_t Phi_{t-1} + b_t c_t,   c_t = sum_{i<j} x'_i conj(x'_j),   x'_i = conj(d_i) x_i
//     Psi_t = a_t Psi_{t-1} + b_t e_t,   e_t = sum_i |x'_i|^2
// (a_t, b_t) = (0, 1) while frame_no_ <= 0, i.e. for the first two frames (postfilter.cc:460-463),
// (alpha, 1-alpha) afterwards.  c_t is an O(N) prefix-sum form: sum_j conj(x'_j) sum_{i<j} x'_i.
//
//   1. bf_apply_stats_kernel : one pass over X (lanes own frames, channels in registers):
//                              y_t = w^H x_t, c_t, e_t  -- the snapshot is read ONCE for beamformer
//                              and post-filter together.
//   2. zelinski_iir_kernel   : one wavefront per (stream, bin) row; 64-frame chunks scanned with a
//                              Hillis-Steele linear-recurrence scan; gain applied to Y in place.
#include "btk_internal.h"
#include <cstdlib>

namespace {

constexpr int PF_NT = 256;
constexpr int PF_UNROLL = 16;

__global__ __launch_bounds__(PF_NT)
void bf_apply_stats_kernel(const float2* __restrict__ W, long w_stream_stride,
                           const float2* __restrict__ Dv /* alignment vector d, same layout as W */,
                           const float2* __restrict__ X, float2* __restrict__ Y,
                           float2* __restrict__ Cc, float* __restrict__ Ee,
                           int K, int N, long T_stride, long T)
{
  const int k = blockIdx.y, s = blockIdx.z;
  const long t = (long)blockIdx.x * PF_NT + threadIdx.x;
  if (t >= T) return;
  const float2* w = W + s * w_stream_stride + (long)k * N;
  const float2* d = Dv + s * w_stream_stride + (long)k * N;
  const float2* x = X + ((long)s * K + k) * N * T_stride + t;
  float yr = 0.f, yi = 0.f, pr = 0.f, pi = 0.f, cr = 0.f, ci = 0.f, e = 0.f;
  auto step = [&](int n, float2 v) {
    const float2 wn = w[n], dn = d[n];
    yr = fmaf(wn.x, v.x, fmaf(wn.y, v.y, yr));                  // conj(w) x
    yi = fmaf(wn.x, v.y, fmaf(-wn.y, v.x, yi));
    const float ar = fmaf(dn.x, v.x, dn.y * v.y);               // x' = conj(d) x
    const float ai = fmaf(dn.x, v.y, -dn.y * v.x);
    cr = fmaf(pr, ar, fmaf(pi, ai, cr));                        // prefix * conj(x')
    ci = fmaf(pi, ar, fmaf(-pr, ai, ci));
    pr += ar; pi += ai;
    e = fmaf(ar, ar, fmaf(ai, ai, e));
  };
  // the loads of PF_UNROLL channels are issued before the first of them is consumed (as bf_apply_kernel does): the prefix
  // sums make every step depend on the previous one, and hipcc keeps load and use together otherwise
  int n = 0;
  for (; n + PF_UNROLL <= N; n += PF_UNROLL) {
    float2 v[PF_UNROLL];
#pragma unroll
    for (int u = 0; u < PF_UNROLL; u++) v[u] = btk_ld<true>(x + (long)(n + u) * T_stride);
#pragma unroll
    for (int u = 0; u < PF_UNROLL; u++) step(n + u, v[u]);
  }
  for (; n < N; n++) step(n, x[(long)n * T_stride]);
  const long o = ((long)s * K + k) * T_stride + t;
  Y[o] = make_float2(yr, yi);
  Cc[o] = make_float2(cr, ci);
  Ee[o] = e;
}

// Inclusive wave-wide scan of the affine maps v -> a v + b[i] (one multiplier, NB offsets) in the DPP network: four row_shr steps
// inside the 16-lane rows, then the row totals travel with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).
// Lanes without a source keep the identity map (`old` operand), so no lane tests are needed.  The shuffle form of the same
// scan was 30 dependent ds_bpermute round trips per 64-frame chunk.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float identity, float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int NB, int CTRL, int ROW_MASK>
__device__ __forceinline__ void affine_scan_step(float& a, float (&b)[NB])
{
  const float a2 = dpp_or<CTRL, ROW_MASK>(1.f, a);
  float p[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) p[i] = dpp_or<CTRL, ROW_MASK>(0.f, b[i]);
#pragma unroll
  for (int i = 0; i < NB; i++) b[i] = fmaf(a, p[i], b[i]);     // (this o earlier)(v) = a (a2 v + p) + b
  a *= a2;
}
template <int NB>
__device__ __forceinline__ void affine_scan(float& a, float (&b)[NB])
{
  affine_scan_step<NB, 0x111, 0xF>(a, b);      // row_shr:1
  affine_scan_step<NB, 0x112, 0xF>(a, b);      // row_shr:2
  affine_scan_step<NB, 0x114, 0xF>(a, b);      // row_shr:4
  affine_scan_step<NB, 0x118, 0xF>(a, b);      // row_shr:8
  affine_scan_step<NB, 0x142, 0xA>(a, b);      // row_bcast:15 -> rows 1, 3
  affine_scan_step<NB, 0x143, 0xC>(a, b);      // row_bcast:31 -> rows 2, 3
}

// One wavefront per (s,k).  frame_base = number of frames this post-filter has already produced
// (ZelinskiPostFilter::frame_no_ + 1).
__global__ __launch_bounds__(64)
void zelinski_iir_kernel(float2* __restrict__ Y, const float2* __restrict__ Cc, const float* __restrict__ Ee,
                         int K, int N, long T_stride, long T, float alpha, int type, int min_frames,
                         long frame_base, float2* __restrict__ Phi /* [S][K] */, float* __restrict__ Psi /* [S][K] */,
                         float* __restrict__ Wlast /* [S][K] */)
{
  const int k = blockIdx.x, s = blockIdx.y;
  const int lane = threadIdx.x;
  const long row = ((long)s * K + k);
  float2 phi_c = Phi[row];
  float psi_c = Psi[row];
  float wlast = Wlast[row];
  const float scale = 2.0f / ((float)N - 1.0f);
  // the statistics (and the beamformer output the gain scales) of chunk i + 1 are requested before chunk i is scanned: the
  // 24 dependent cross-lane steps of a scan and the latency of a chunk's loads no longer add up
  const bool scales = type != 0;
  // the 64-frame scan chunks sit on multiples of 64 of the STREAM's frame counter, not of this launch: a launch that starts
  // inside a chunk (history restarted mid-stream by a weight change) scans the same chunks whether the stream is processed in
  // one launch or block by block (the blocks of the node layer end on multiples of 64), hence the same roundings
  const long ph = frame_base & 63;
  const long tf = lane - ph;
  float2 cn = (tf >= 0 && tf < T) ? Cc[row * T_stride + tf] : make_float2(0.f, 0.f);
  float en = (tf >= 0 && tf < T) ? Ee[row * T_stride + tf] : 0.f;
  float2 yn = (scales && tf >= 0 && tf < T) ? Y[row * T_stride + tf] : make_float2(0.f, 0.f);
  for (long t0 = -ph; t0 < T; t0 += 64) {
    const long t = t0 + lane;
    const bool ok = t >= 0 && t < T;
    const long g = frame_base + t;                              // global frame index; frame_no_ before increment = g-1
    float a = ok ? ((g >= 2) ? alpha : 0.f) : 1.f;              // identity element beyond the end
    const float bsc = (g >= 2 && alpha > 0.f) ? 1.f - alpha : 1.f;
    if (alpha <= 0.f) a = ok ? 0.f : 1.f;                       // calc_CSD_: alpha <= 0 -> no memory
    const float2 c = cn, y = yn;
    const float e = en;
    {
      const long tn = t + 64;
      const bool okn = tn >= 0 && tn < T;
      cn = okn ? Cc[row * T_stride + tn] : make_float2(0.f, 0.f);
      en = okn ? Ee[row * T_stride + tn] : 0.f;
      yn = (scales && okn) ? Y[row * T_stride + tn] : make_float2(0.f, 0.f);
    }
    float bb[3] = {ok ? bsc * c.x : 0.f, ok ? bsc * c.y : 0.f, ok ? bsc * e : 0.f};
    affine_scan<3>(a, bb);                                      // inclusive scan of the affine maps v -> a v + b
    const float phr = fmaf(a, phi_c.x, bb[0]), phim = fmaf(a, phi_c.y, bb[1]), ps = fmaf(a, psi_c, bb[2]);
    if (ok) {
      const bool apply = (g - 1) >= (long)min_frames;           // frame_no_ (= g-1, pre-increment) < min_frames -> NO_USE_POST_FILTER
      const int pft = apply ? type : 0;
      float num = (pft & 1) ? fmaxf(phr, 0.f) : sqrtf(phr * phr + phim * phim);
      float Wf = (num / ps) * scale;
      if (Wf >= 1.0f) Wf = 1.0f;
      if (Wf < 1.0e-4f) Wf = 1.0e-4f;
      if (apply && type != 0)                                   // NO_USE_POST_FILTER: the CSDs are just updated (postfilter.cc:197-199)
        Y[row * T_stride + t] = make_float2(Wf * y.x, Wf * y.y);
      wlast = Wf;
    }
    // carry = state after the last valid frame of the chunk
    const int last = (T - t0) >= 64 ? 63 : (int)(T - t0) - 1;
    phi_c = make_float2(__shfl(phr, last, 64), __shfl(phim, last, 64));
    psi_c = __shfl(ps, last, 64);
    wlast = __shfl(wlast, last, 64);
  }
  if (lane == 0) { Phi[row] = phi_c; Psi[row] = psi_c; Wlast[row] = wlast; }
}

// ------------------------------------------------------------------------------------------------
// McCowan / Lefkimmiatis post-filters (postfilter/postfilter.cc:496-1190).
//
// Their gains need per-pair weights: clean PSD  sum_{i<j} (phi_ij - R_ij (phi_ii + phi_jj)/2) / (1 - R_ij)   (:798-829)
//                                    noise PSD  sum_{i<j} ((phi_ii + phi_jj)/2 - phi_ij) / (1 - R_ij)         (:1041-1077)
// with the coherence R_ij clipped at threshold_of_Rij_.  Every phi obeys the same first-order recursion, and the
// sums are LINEAR in the phi, so -- as for Zelinski -- only the recursively averaged sums are state:
//     U_t = a_t U_{t-1} + b_t u_t,   u_t = sum_{i<=j} Cs[j][i] x'_i conj(x'_j)
//     V_t = a_t V_{t-1} + b_t v_t,   v_t = sum_{i<=j} Cv[j][i] x'_i conj(x'_j)
// with per-bin coefficient matrices built once from R (pf_coherence_coeff_kernel): off-diagonal 1/(1-R_ij) resp.
// -1/(1-R_ij), diagonal = the collected (phi_ii + phi_jj)/2 terms.  McCowan: W = g(U) / Psi * 2/(N-1) -- the
// Zelinski formula with Phi replaced by U, so btk_zelinski_process is reused.  Lefkimmiatis: W = g(U)/(g(U) + g(V)/L).
//
//   3. pf_coherence_coeff_kernel : R [K][N][N] -> Cs, Cv [K][N][N] (row j holds i <= j), float64 arithmetic
//   4. bf_apply_stats2_kernel    : y_t, e_t and the quadratic forms u_t (v_t); lanes own frames, 16 rows of C at a
//                                  time in registers, C entries are wave-uniform (scalar loads), outer sums in float64
//   5. lefkimmiatis_iir_kernel   : the same chunked linear-recurrence scan as zelinski_iir_kernel on (U, V)
__global__ __launch_bounds__(64)
void pf_coherence_coeff_kernel(const float2* __restrict__ R, float threshold, int N,
                               float2* __restrict__ Cs, float2* __restrict__ Cv /* nullable */)
{
  const int k = blockIdx.x;
  const float2* Rk = R + (long)k * N * N;
  const double thr = (double)threshold;
  for (int j = threadIdx.x; j < N; j += 64) {
    double dsr = 0.0, dsi = 0.0, dvr = 0.0, dvi = 0.0;
    for (int i = 0; i < N; i++) {
      if (i == j) continue;
      const int a = i < j ? i : j, b = i < j ? j : i;              // only the upper triangle R[a][b], a < b, is read
      const float2 rf = Rk[(long)a * N + b];
      // clean PSD rule (:813-815)
      double rr = rf.x, ri = rf.y;
      if (rr > thr && ri <= 0.0) { rr = thr; ri = 0.0; }
      double dr = 1.0 - rr, di = -ri, dn = dr * dr + di * di;
      const double csr = dr / dn, csi = -di / dn;                  // 1 / (1 - R)
      // - 0.5 R / (1 - R) goes to both diagonal entries of the pair
      dsr -= 0.5 * (csr * rr - csi * ri); dsi -= 0.5 * (csr * ri + csi * rr);
      if (i < j) Cs[((long)k * N + j) * N + i] = make_float2((float)csr, (float)csi);
      else Cs[((long)k * N + j) * N + i] = make_float2(0.f, 0.f);
      if (Cv) {
        // noise PSD rule (:1057-1062)
        double qr = rf.x, qi = rf.y;
        if (qr > thr) { qr = thr; qi = 0.0; }
        else if (qr == 1.0) { qr = 0.99; qi = 0.0; }
        dr = 1.0 - qr; di = -qi; dn = dr * dr + di * di;
        const double cvr = dr / dn, cvi = -di / dn;
        dvr += 0.5 * cvr; dvi += 0.5 * cvi;
        if (i < j) Cv[((long)k * N + j) * N + i] = make_float2((float)-cvr, (float)-cvi);
        else Cv[((long)k * N + j) * N + i] = make_float2(0.f, 0.f);
      }
    }
    Cs[((long)k * N + j) * N + j] = make_float2((float)dsr, (float)dsi);
    if (Cv) Cv[((long)k * N + j) * N + j] = make_float2((float)dvr, (float)dvi);
  }
}

template <int NQ, int PF_JB>
__global__ __launch_bounds__(PF_NT)
void bf_apply_stats2_kernel(const float2* __restrict__ W, long w_stream_stride, const float2* __restrict__ Dv,
                            const float2* __restrict__ X, float2* __restrict__ Y,
                            const float2* __restrict__ Cs, const float2* __restrict__ Cv,
                            float2* __restrict__ U, float2* __restrict__ V, float* __restrict__ Ee,
                            int K, int N, long T_stride, long T)
{
  const int k = blockIdx.y, s = blockIdx.z;
  const long t = (long)blockIdx.x * PF_NT + threadIdx.x;
  const float2* cs = Cs + (long)k * N * N;
  const float2* cv = (NQ == 2) ? Cv + (long)k * N * N : nullptr;
  // A diffuse-field coherence matrix is real (sinc of the distances), and so are the pair weights built from it: half of the
  // multiply-adds of the complex form would multiply zeros.  One scan of the bin's coefficients decides per workgroup.
  int im = 0;
  for (int idx = threadIdx.x; idx < N * N; idx += PF_NT) {
    im |= (cs[idx].y != 0.f);
    if (NQ == 2) im |= (cv[idx].y != 0.f);
  }
  const bool complex_c = __syncthreads_or(im) != 0;
  if (t >= T) return;
  const float2* w = W + s * w_stream_stride + (long)k * N;
  const float2* d = Dv + s * w_stream_stride + (long)k * N;
  const float2* x = X + ((long)s * K + k) * N * T_stride + t;
  float yr = 0.f, yi = 0.f, e = 0.f;
#pragma unroll 8
  for (int n = 0; n < N; n++) {
    const float2 v = x[(long)n * T_stride];
    const float2 wn = w[n], dn = d[n];
    yr = fmaf(wn.x, v.x, fmaf(wn.y, v.y, yr));
    yi = fmaf(wn.x, v.y, fmaf(-wn.y, v.x, yi));
    const float ar = fmaf(dn.x, v.x, dn.y * v.y), ai = fmaf(dn.x, v.y, -dn.y * v.x);
    e = fmaf(ar, ar, fmaf(ai, ai, e));
  }
  double ur = 0.0, ui = 0.0, vr = 0.0, vi = 0.0;
  for (int jb = 0; jb < N; jb += PF_JB) {
    float ar[PF_JB], ai[PF_JB], br[PF_JB], bi[PF_JB];
#pragma unroll
    for (int jj = 0; jj < PF_JB; jj++) { ar[jj] = ai[jj] = 0.f; br[jj] = bi[jj] = 0.f; }
    const int iend = (jb + PF_JB < N) ? jb + PF_JB : N;
    if (complex_c) {
      for (int i = 0; i < iend; i++) {
        const float2 xv = x[(long)i * T_stride];
        const float2 dn = d[i];
        const float xr = fmaf(dn.x, xv.x, dn.y * xv.y), xi = fmaf(dn.x, xv.y, -dn.y * xv.x);   // x'_i = conj(d_i) x_i
#pragma unroll
        for (int jj = 0; jj < PF_JB; jj++) {
          const int j = (jb + jj < N) ? jb + jj : N - 1;            // clamped rows are never used
          const float2 c = cs[(long)j * N + i];                     // wave-uniform
          ar[jj] = fmaf(c.x, xr, fmaf(-c.y, xi, ar[jj]));
          ai[jj] = fmaf(c.x, xi, fmaf(c.y, xr, ai[jj]));
          if (NQ == 2) {
            const float2 c2 = cv[(long)j * N + i];
            br[jj] = fmaf(c2.x, xr, fmaf(-c2.y, xi, br[jj]));
            bi[jj] = fmaf(c2.x, xi, fmaf(c2.y, xr, bi[jj]));
          }
        }
      }
    } else {
      for (int i = 0; i < iend; i++) {
        const float2 xv = x[(long)i * T_stride];
        const float2 dn = d[i];
        const float xr = fmaf(dn.x, xv.x, dn.y * xv.y), xi = fmaf(dn.x, xv.y, -dn.y * xv.x);
#pragma unroll
        for (int jj = 0; jj < PF_JB; jj++) {
          const int j = (jb + jj < N) ? jb + jj : N - 1;
          const float c = cs[(long)j * N + i].x;
          ar[jj] = fmaf(c, xr, ar[jj]);
          ai[jj] = fmaf(c, xi, ai[jj]);
          if (NQ == 2) {
            const float c2 = cv[(long)j * N + i].x;
            br[jj] = fmaf(c2, xr, br[jj]);
            bi[jj] = fmaf(c2, xi, bi[jj]);
          }
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < PF_JB; jj++) {
      const int j = jb + jj;
      if (j < N) {
        const float2 xv = x[(long)j * T_stride];
        const float2 dn = d[j];
        const double xr = fmaf(dn.x, xv.x, dn.y * xv.y), xi = fmaf(dn.x, xv.y, -dn.y * xv.x);
        ur += xr * (double)ar[jj] + xi * (double)ai[jj];          // conj(x'_j) a_j
        ui += xr * (double)ai[jj] - xi * (double)ar[jj];
        if (NQ == 2) {
          vr += xr * (double)br[jj] + xi * (double)bi[jj];
          vi += xr * (double)bi[jj] - xi * (double)br[jj];
        }
      }
    }
  }
  const long o = ((long)s * K + k) * T_stride + t;
  Y[o] = make_float2(yr, yi);
  U[o] = make_float2((float)ur, (float)ui);
  if (NQ == 2) V[o] = make_float2((float)vr, (float)vi);
  Ee[o] = e;
}

// ------------------------------------------------------------------------------------------------
// 4b. bf_apply_stats2_mfma_kernel (N = 16, 32, 48, 64): the inner products a_j = sum_{i<=j} C[j][i] x'_i of the quadratic forms
// are a matrix product per bin -- C (N x N, lower triangle, constant) times the snapshots x' (N x frames) -- and run on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation, as the fmaf chains of the VALU kernel).
//   rows of a 16x16 block = j, columns = 16 consecutive frames, the contraction runs over the channels i, four per
//   instruction; real and imaginary parts of x' are two B operands against the same A (a real C, the diffuse-field case: two
//   instructions per four channels and block; a complex C adds the two cross products).  Only the 10 of 16 blocks on or below
//   the diagonal are computed (N = 64).
// The contraction order is free, so the channel a lane feeds into step m is chosen as i(m, g) = 16 (m / 4) + 4 g + m % 4 with
// g = lane / 16: these are exactly the rows j = 16 jb + 4 g + v the lane receives in accumulator register v of block jb
// (m = 4 jb + v).  A lane therefore already holds x'_j for every a_j it gets back: the outer sums u = sum_j conj(x'_j) a_j
// (float64, as in the VALU kernel) need no exchange but two adds across the four lane groups, and the snapshot goes HBM ->
// registers -> matrix core without touching LDS.  LDS holds the bin's coefficients, transposed and split ([i][j] real /
// imaginary planes, row stride N + 4: the four rows an operand read touches fall on disjoint bank halves), upper triangle zeroed.
// A workgroup = MF_NW wavefronts = one (stream, bin) and MF_TPW x MF_NW tiles of 16 frames (the coefficients are staged once per
// workgroup); the other wavefronts of a SIMD load their tiles while one feeds the matrix core.
constexpr int MF_TPW = 8;      // tiles per wavefront: 4 / 8 / 16 / 32 -> 2.61 / 2.51 / 2.54 / 2.55 ms (one form), 4.42 / 4.11 / 4.08 / 4.11 (two); BTK_PF_TPW overrides
// wavefronts per workgroup (the coefficient planes allow two workgroups per CU): one form = 8, four wavefronts per SIMD in 128
// VGPRs (2.46 ms against 2.69 with 4 at the C0 shape); two forms = 4, two per SIMD in 172 VGPRs (4.0 ms; with 8 the kernel spills,
// 4.85 ms, one form after the other 7.1 ms, and six wavefronts land 2-2-1-1 on the SIMDs, 5.9 ms)
constexpr int mf_nw(int nq, int nb = 4) { return (nq == 1 && nb <= 4) ? 8 : 4; }     // more than 64 channels: 4 wavefronts, 256 VGPRs each

// Round 3: any channel count.  NB = ceil(NR / 16) blocks; the channels NR .. 16 NB - 1 are padding: their weights, alignment entries
// and coefficient rows / columns are zero in LDS, so x' = conj(d) x, y, e and every a_j they touch vanish by themselves; only the
// snapshot loads of the last block need care (a clamped per-lane row instead of the wave-uniform row pointer) to stay inside X.
template <int NQ, int NB /* ceil(NR / 16) */, bool PAD /* NR < 16 NB */>
__global__ __launch_bounds__(64 * mf_nw(NQ, NB), mf_nw(NQ, NB) / 2)       // (HIP: the second number is wavefronts per SIMD)
void bf_apply_stats2_mfma_kernel(const float2* __restrict__ W, long w_stream_stride, const float2* __restrict__ Dv,
                                 const float2* __restrict__ X, float2* __restrict__ Y,
                                 const float2* __restrict__ Cs, const float2* __restrict__ Cv,
                                 float2* __restrict__ U, float2* __restrict__ V, float* __restrict__ Ee,
                                 int K, long T_stride, long T, int tpw, int NR /* real channel count, 16 (NB - 1) < NR <= 16 NB */)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int N = 16 * NB, LD = N + 4, M4 = N / 4, MF_NW = mf_nw(NQ, NB);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctr = reinterpret_cast<float*>(smem);                  // [NQ][N][LD] real parts, [i][j]
  float* cti = ctr + NQ * N * LD;                               // [NQ][N][LD] imaginary parts
  float2* ws = reinterpret_cast<float2*>(cti + NQ * N * LD);    // [N] beamformer weights
  float2* ds = ws + N;                                          // [N] alignment vector
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.y, s = blockIdx.z;
  const int n = lane & 15, g = lane >> 4;

  int im = 0;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const float2* c = (q == 0 ? Cs : Cv) + (long)k * NR * NR;
    for (int idx = tid; idx < N * N; idx += 64 * MF_NW) {
      const int j = idx / N, i = idx % N;                       // C[j][i], i <= j used (the reference reads nothing else of a row either)
      float2 v = (j < NR && i <= j) ? c[j * NR + i] : make_float2(0.f, 0.f);
      im |= (v.y != 0.f);
      ctr[(q * N + i) * LD + j] = v.x;
      cti[(q * N + i) * LD + j] = v.y;
    }
  }
  if (tid < N) {
    ws[tid] = tid < NR ? W[s * w_stream_stride + (long)k * NR + tid] : make_float2(0.f, 0.f);
    ds[tid] = tid < NR ? Dv[s * w_stream_stride + (long)k * NR + tid] : make_float2(0.f, 0.f);
  }
  const bool complex_c = __syncthreads_or(im) != 0;

  const float2* xk = X + ((long)s * K + k) * NR * T_stride;
  const long f0 = (long)blockIdx.x * (MF_NW * tpw * 16);
  auto chan = [&](int m) { return 16 * (m >> 2) + 4 * g + (m & 3); };
  float2 xv[M4];
  long t0 = f0 + __builtin_amdgcn_readfirstlane(wave) * 16;
#pragma unroll 1
  for (int q = 0; q < tpw && t0 < T; q++, t0 += MF_NW * 16) {
    {
      // row pointers are wave-uniform (SGPR pair), the lane adds one 32-bit byte offset (its group's 4 rows and its frame,
      // clamped into the block for a ragged last tile: those columns are never stored): global_load with an SGPR base from
      // inline asm -- hipcc builds per-lane 64-bit addresses for the same loads.  The loads are invisible to hipcc's
      // counters, hence the explicit wait below.
      const long tl = (t0 + n < T) ? n : (T - 1 - t0);
      const unsigned vb = ((unsigned)(4 * g) * (unsigned)T_stride + (unsigned)tl) * 8u;
      v2f raw[M4];
#pragma unroll
      for (int m = 0; m < M4; m++) {
        if (PAD && m >= M4 - 4) {
          // last block of a padded channel count: the lane's own row, clamped to the last real channel (its x' is zeroed by d = 0)
          const int ch = chan(m) < NR ? chan(m) : NR - 1;
          const float2 v = xk[(long)ch * T_stride + t0 + tl];
          raw[m] = v2f{v.x, v.y};
        } else {
          const float2* rowp = xk + (long)(16 * (m >> 2) + (m & 3)) * T_stride + t0;
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(raw[m]) : "v"(vb), "s"(rowp) : "memory");
        }
      }
      // every loaded value is an in/out operand of the wait (of an empty asm right behind it from the fifth on; volatile asms keep
      // their order): its register carries it across, hipcc has no use of it that could be scheduled earlier
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) :: "memory");
        else asm volatile("" : "+v"(raw[4 * c]), "+v"(raw[4 * c + 1]), "+v"(raw[4 * c + 2]), "+v"(raw[4 * c + 3]) :: "memory");
      }
#pragma unroll
      for (int m = 0; m < M4; m++) xv[m] = make_float2(raw[m].x, raw[m].y);
    }
    // (the weights and coefficients in LDS do not change from tile to tile: an opaque zero in their addresses keeps hipcc
    //  from hoisting the loop-invariant LDS reads out of the tile loop and spilling them)
    int lo = 0;
    asm volatile("" : "+s"(lo));
    // y = w^H x, x' = conj(d) x, e = sum |x'|^2 over this lane group's channels
    float yr = 0.f, yi = 0.f, e = 0.f;
#pragma unroll
    for (int m = 0; m < M4; m++) {
      const int i = chan(m) + lo;
      const float2 wn = ws[i], dn = ds[i], v = xv[m];
      yr = fmaf(wn.x, v.x, fmaf(wn.y, v.y, yr));
      yi = fmaf(wn.x, v.y, fmaf(-wn.y, v.x, yi));
      const float ar = fmaf(dn.x, v.x, dn.y * v.y), ai = fmaf(dn.x, v.y, -dn.y * v.x);
      e = fmaf(ar, ar, fmaf(ai, ai, e));
      xv[m] = make_float2(ar, ai);
      // four channels' weights in flight, not all of them: the sums are pinned here, or hipcc parks the weights in scratch
      // and forms y after the matrix instructions
      if ((m & 3) == 3) asm volatile("" : "+v"(yr), "+v"(yi), "+v"(e));
    }
    double ur = 0.0, ui = 0.0, vr = 0.0, vi = 0.0;
#pragma unroll
    for (int jb = 0; jb < NB; jb++) {
      v4f are[NQ], aim[NQ];
#pragma unroll
      for (int f = 0; f < NQ; f++) { are[f] = v4f{0.f, 0.f, 0.f, 0.f}; aim[f] = v4f{0.f, 0.f, 0.f, 0.f}; }
      const int mend = 4 * (jb + 1);                           // channels i < 16 (jb + 1): the blocks right of the diagonal are zero
      if (!complex_c) {
#pragma unroll
        for (int m = 0; m < mend; m++) {
          const int off = chan(m) * LD + jb * 16 + n + lo;
#pragma unroll
          for (int f = 0; f < NQ; f++) {
            const float a = ctr[f * N * LD + off];
            are[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[m].x, are[f], 0, 0, 0);
            aim[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[m].y, aim[f], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < mend; m++) {
          const int off = chan(m) * LD + jb * 16 + n + lo;
#pragma unroll
          for (int f = 0; f < NQ; f++) {
            const float a = ctr[f * N * LD + off], b = cti[f * N * LD + off];
            are[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[m].x, are[f], 0, 0, 0);
            aim[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[m].y, aim[f], 0, 0, 0);
            are[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, -xv[m].y, are[f], 0, 0, 0);
            aim[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, xv[m].x, aim[f], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const double xr = xv[4 * jb + v].x, xi = xv[4 * jb + v].y;         // x'_j of accumulator row v (see the channel map)
        ur = fma(xr, (double)are[0][v], fma(xi, (double)aim[0][v], ur));           // += conj(x'_j) a_j, two chained multiply-adds per part
        ui = fma(xr, (double)aim[0][v], fma(-xi, (double)are[0][v], ui));
        if (NQ == 2) {
          vr = fma(xr, (double)are[NQ - 1][v], fma(xi, (double)aim[NQ - 1][v], vr));
          vi = fma(xr, (double)aim[NQ - 1][v], fma(-xi, (double)are[NQ - 1][v], vi));
        }
      }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh *= 2) {
      yr += __shfl_xor(yr, sh, 64); yi += __shfl_xor(yi, sh, 64); e += __shfl_xor(e, sh, 64);
      ur += __shfl_xor(ur, sh, 64); ui += __shfl_xor(ui, sh, 64);
      if (NQ == 2) { vr += __shfl_xor(vr, sh, 64); vi += __shfl_xor(vi, sh, 64); }
    }
    const long t = t0 + n;
    if (g == 0 && t < T) {
      const long o = ((long)s * K + k) * T_stride + t;
      Y[o] = make_float2(yr, yi);
      U[o] = make_float2((float)ur, (float)ui);
      if (NQ == 2) V[o] = make_float2((float)vr, (float)vi);
      Ee[o] = e;
    }
  }
}

template <int NQ, int NB, bool PAD>
int launch_stats2_mfma(const float2* W, long wss, const float2* D, const float2* X, float2* Y, const float2* Cs, const float2* Cv,
                       float2* U, float2* V, float* E, int S, int K, long T_stride, long T, hipStream_t st, int NR)
{
  constexpr int N = 16 * NB;
  const size_t lds = sizeof(float) * 2 * NQ * N * (N + 4) + sizeof(float2) * 2 * N;
  auto kern = bf_apply_stats2_mfma_kernel<NQ, NB, PAD>;
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  constexpr int MF_NW = mf_nw(NQ, NB);
  const int tpw = btk_switches().pf_tpw > 0 ? btk_switches().pf_tpw : MF_TPW;
  const long per_wg = (long)MF_NW * tpw * 16;
  hipLaunchKernelGGL(kern, dim3((unsigned)((T + per_wg - 1) / per_wg), (unsigned)K, (unsigned)S), dim3(64 * MF_NW), lds, st,
                     W, wss, D, X, Y, Cs, Cv, U, V, E, K, T_stride, T, tpw, NR);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

// One wavefront per (s,k): W = g(U) / (g(U) + g(V) / Lambda_k), Lambda_k = 1 for k < fbinX1 (postfilter.cc:1122-1135)
__global__ __launch_bounds__(64)
void lefkimmiatis_iir_kernel(float2* __restrict__ Y, const float2* __restrict__ Uc, const float2* __restrict__ Vc,
                             const float2* __restrict__ Lambda /* [K] */, int fbinX1,
                             int K, long T_stride, long T, float alpha, int type, int min_frames, long frame_base,
                             float2* __restrict__ Us /* [S][K] */, float2* __restrict__ Vs /* [S][K] */,
                             float* __restrict__ Wlast)
{
  const int k = blockIdx.x, s = blockIdx.y;
  const int lane = threadIdx.x;
  const long row = ((long)s * K + k);
  float2 u_c = Us[row], v_c = Vs[row];
  float wlast = Wlast[row];
  float lam = 1.f;
  if (k >= fbinX1) { const float2 L = Lambda[k]; lam = (type & 1) ? L.x : sqrtf(L.x * L.x + L.y * L.y); }
  for (long t0 = -(frame_base & 63); t0 < T; t0 += 64) {       // chunks on multiples of 64 of the stream's frame counter (see zelinski_iir_kernel)
    const long t = t0 + lane;
    const bool ok = t >= 0 && t < T;
    const long g = frame_base + t;
    float a = ok ? ((g >= 2) ? alpha : 0.f) : 1.f;
    const float bsc = (g >= 2 && alpha > 0.f) ? 1.f - alpha : 1.f;
    if (alpha <= 0.f) a = ok ? 0.f : 1.f;
    const float2 cu = ok ? Uc[row * T_stride + t] : make_float2(0.f, 0.f);
    const float2 cvv = ok ? Vc[row * T_stride + t] : make_float2(0.f, 0.f);
    float bb[4] = {ok ? bsc * cu.x : 0.f, ok ? bsc * cu.y : 0.f, ok ? bsc * cvv.x : 0.f, ok ? bsc * cvv.y : 0.f};
    affine_scan<4>(a, bb);
    const float u0 = fmaf(a, u_c.x, bb[0]), u1 = fmaf(a, u_c.y, bb[1]), v0 = fmaf(a, v_c.x, bb[2]), v1 = fmaf(a, v_c.y, bb[3]);
    if (ok) {
      const float gs = (type & 1) ? u0 : sqrtf(u0 * u0 + u1 * u1);
      const float gv = (type & 1) ? v0 : sqrtf(v0 * v0 + v1 * v1);
      float Wf = gs / (gs + gv / lam);
      if (Wf > 1.0f) Wf = 1.0f;
      if (Wf < 1.0e-4f) Wf = 1.0e-4f;
      if ((g - 1) >= (long)min_frames) {
        const float2 y = Y[row * T_stride + t];
        Y[row * T_stride + t] = make_float2(Wf * y.x, Wf * y.y);
      }
      wlast = Wf;
    }
    const int last = (T - t0) >= 64 ? 63 : (int)(T - t0) - 1;
    u_c = make_float2(__shfl(u0, last, 64), __shfl(u1, last, 64));
    v_c = make_float2(__shfl(v0, last, 64), __shfl(v1, last, 64));
    wlast = __shfl(wlast, last, 64);
  }
  if (lane == 0) { Us[row] = u_c; Vs[row] = v_c; Wlast[row] = wlast; }
}

}  // namespace

extern "C" {

int btk_bf_apply_stats(const void* W, const void* D, int per_stream_weights, const void* X, void* Y,
                       void* C, float* E, int S, int K, int N, long T_stride, long T, void* stream)
{
  if (!W || !D || !X || !Y || !C || !E) return btk_set_error(BTK_ERR_PARAMETER, "btk_bf_apply_stats: null argument");
  if (S <= 0 || K <= 0 || N <= 0 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_bf_apply_stats: bad sizes S=%d K=%d N=%d T=%ld", S, K, N, T);
  if (T == 0) return BTK_OK;
  const long wss = per_stream_weights ? (long)K * N : 0;
  dim3 grid((unsigned)((T + PF_NT - 1) / PF_NT), (unsigned)K, (unsigned)S);
  hipLaunchKernelGGL(bf_apply_stats_kernel, grid, dim3(PF_NT), 0, as_stream(stream),
                     static_cast<const float2*>(W), wss, static_cast<const float2*>(D),
                     static_cast<const float2*>(X), static_cast<float2*>(Y), static_cast<float2*>(C), E,
                     K, N, T_stride, T);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_zelinski_process(void* Y, const void* C, const float* E, int S, int K, int N, long T_stride, long T,
                         double alpha, int type, int min_frames, long frames_done,
                         void* phi_state, float* psi_state, float* w_last, void* stream)
{
  if (!Y || !C || !E || !phi_state || !psi_state || !w_last)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_zelinski_process: null argument");
  if (N <= 1) return btk_set_error(BTK_ERR_DIMENSION, "The number of channels %d is <= 1 ", N);
  if (S <= 0 || K <= 0 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_zelinski_process: bad sizes S=%d K=%d T=%ld", S, K, T);
  if (T == 0) return BTK_OK;
  hipLaunchKernelGGL(zelinski_iir_kernel, dim3((unsigned)K, (unsigned)S), dim3(64), 0, as_stream(stream),
                     static_cast<float2*>(Y), static_cast<const float2*>(C), E, K, N, T_stride, T,
                     (float)alpha, type, min_frames, frames_done, static_cast<float2*>(phi_state), psi_state, w_last);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_pf_coherence_coeffs(const void* R, float threshold, int K, int N, void* Cs, void* Cv, void* stream)
{
  if (!R || !Cs) return btk_set_error(BTK_ERR_PARAMETER, "construct/set a noise coherence matrix");
  if (K <= 0 || N <= 1) return btk_set_error(BTK_ERR_DIMENSION, "btk_pf_coherence_coeffs: bad sizes K=%d N=%d", K, N);
  hipLaunchKernelGGL(pf_coherence_coeff_kernel, dim3((unsigned)K), dim3(64), 0, as_stream(stream),
                     static_cast<const float2*>(R), threshold, N, static_cast<float2*>(Cs), static_cast<float2*>(Cv));
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_bf_apply_stats2(const void* W, const void* D, int per_stream_weights, const void* X, void* Y,
                        const void* Cs, const void* Cv, void* U, void* V, float* E,
                        int S, int K, int N, long T_stride, long T, void* stream)
{
  if (!W || !D || !X || !Y || !Cs || !U || !E) return btk_set_error(BTK_ERR_PARAMETER, "btk_bf_apply_stats2: null argument");
  if ((Cv == nullptr) != (V == nullptr)) return btk_set_error(BTK_ERR_PARAMETER, "btk_bf_apply_stats2: Cv and V go together");
  if (S <= 0 || K <= 0 || N <= 1 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_bf_apply_stats2: bad sizes S=%d K=%d N=%d T=%ld", S, K, N, T);
  if (T == 0) return BTK_OK;
  const long wss = per_stream_weights ? (long)K * N : 0;
  dim3 grid((unsigned)((T + PF_NT - 1) / PF_NT), (unsigned)K, (unsigned)S);
  // rows of C per register block: 16 rows x two forms overflow the SGPR file (the C entries are scalar loads) and
  // N <= 8 wastes half of a 16-row block; measured in profiles/pf_ab.py.  BTK_PF_JB overrides (benchmarking only).
  const int jb_env = btk_switches().pf_jb;
  // The coefficient products on the matrix cores (BTK_PF_JB set = the VALU kernel, for the A/B in profiles/): any N up to 64 (two
  // forms in one pass) or 128 (one form per pass), padded to a multiple of 16 inside the kernel.  Below BTK_PF_MFMA_MIN channels
  // (default 8: at N = 8 the matrix cores still win, 0.55 against 0.58 ms; smaller arrays are mostly padding) the VALU kernel stays.
  if (!jb_env && N >= btk_switches().pf_mfma_min && N <= 128 && (long)N * T_stride * 8 < (1L << 31)) {
    const float2 *Wp = static_cast<const float2*>(W), *Dp = static_cast<const float2*>(D), *Xp = static_cast<const float2*>(X);
    const float2 *Csp = static_cast<const float2*>(Cs), *Cvp = static_cast<const float2*>(Cv);
    float2 *Yp = static_cast<float2*>(Y), *Up = static_cast<float2*>(U), *Vp = static_cast<float2*>(V);
    hipStream_t st = as_stream(stream);
#define BTK_PF_CASE(NQ_, NB_) case NB_: return (N % 16) ? launch_stats2_mfma<NQ_, NB_, true>(Wp, wss, Dp, Xp, Yp, Csp, Cvp, Up, Vp, E, S, K, T_stride, T, st, N) \
                                                         : launch_stats2_mfma<NQ_, NB_, false>(Wp, wss, Dp, Xp, Yp, Csp, Cvp, Up, Vp, E, S, K, T_stride, T, st, N);
    if (Cv && N > 64) {
      // two forms of more than 64 channels: their coefficient planes do not fit the LDS together -- one pass per form (the second
      // recomputes y and e: the same values), still 1.8 x the VALU kernel at N = 100 (profiles/r03_pf_ab.txt)
      const int rc = btk_bf_apply_stats2(W, D, per_stream_weights, X, Y, Cs, nullptr, U, nullptr, E, S, K, N, T_stride, T, stream);
      if (rc != BTK_OK) return rc;
      return btk_bf_apply_stats2(W, D, per_stream_weights, X, Y, Cv, nullptr, V, nullptr, E, S, K, N, T_stride, T, stream);
    }
    if (Cv) {
      switch ((N + 15) / 16) { BTK_PF_CASE(2, 1) BTK_PF_CASE(2, 2) BTK_PF_CASE(2, 3) BTK_PF_CASE(2, 4) }
    } else {
      switch ((N + 15) / 16) { BTK_PF_CASE(1, 1) BTK_PF_CASE(1, 2) BTK_PF_CASE(1, 3) BTK_PF_CASE(1, 4) BTK_PF_CASE(1, 5) BTK_PF_CASE(1, 6) BTK_PF_CASE(1, 7) BTK_PF_CASE(1, 8) }
    }
#undef BTK_PF_CASE
  }
  const int jb = jb_env ? jb_env : (Cv ? 8 : (N <= 8 ? 8 : 16));
  if (Cv && jb == 16)
    hipLaunchKernelGGL((bf_apply_stats2_kernel<2, 16>), grid, dim3(PF_NT), 0, as_stream(stream),
                       static_cast<const float2*>(W), wss, static_cast<const float2*>(D), static_cast<const float2*>(X),
                       static_cast<float2*>(Y), static_cast<const float2*>(Cs), static_cast<const float2*>(Cv),
                       static_cast<float2*>(U), static_cast<float2*>(V), E, K, N, T_stride, T);
  else if (Cv)
    hipLaunchKernelGGL((bf_apply_stats2_kernel<2, 8>), grid, dim3(PF_NT), 0, as_stream(stream),
                       static_cast<const float2*>(W), wss, static_cast<const float2*>(D), static_cast<const float2*>(X),
                       static_cast<float2*>(Y), static_cast<const float2*>(Cs), static_cast<const float2*>(Cv),
                       static_cast<float2*>(U), static_cast<float2*>(V), E, K, N, T_stride, T);
  else if (jb == 8)
    hipLaunchKernelGGL((bf_apply_stats2_kernel<1, 8>), grid, dim3(PF_NT), 0, as_stream(stream),
                       static_cast<const float2*>(W), wss, static_cast<const float2*>(D), static_cast<const float2*>(X),
                       static_cast<float2*>(Y), static_cast<const float2*>(Cs), static_cast<const float2*>(nullptr),
                       static_cast<float2*>(U), static_cast<float2*>(nullptr), E, K, N, T_stride, T);
  else
    hipLaunchKernelGGL((bf_apply_stats2_kernel<1, 16>), grid, dim3(PF_NT), 0, as_stream(stream),
                       static_cast<const float2*>(W), wss, static_cast<const float2*>(D), static_cast<const float2*>(X),
                       static_cast<float2*>(Y), static_cast<const float2*>(Cs), static_cast<const float2*>(nullptr),
                       static_cast<float2*>(U), static_cast<float2*>(nullptr), E, K, N, T_stride, T);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

int btk_lefkimmiatis_process(void* Y, const void* U, const void* V, const void* lambda, int fbinX1,
                             int S, int K, int N, long T_stride, long T, double alpha, int type, int min_frames,
                             long frames_done, void* u_state, void* v_state, float* w_last, void* stream)
{
  if (!Y || !U || !V || !lambda || !u_state || !v_state || !w_last)
    return btk_set_error(BTK_ERR_PARAMETER, "btk_lefkimmiatis_process: null argument");
  if (N <= 1) return btk_set_error(BTK_ERR_DIMENSION, "The number of channels %d is <= 1 ", N);
  if (S <= 0 || K <= 0 || T < 0 || T_stride < T || fbinX1 < 0)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_lefkimmiatis_process: bad sizes S=%d K=%d T=%ld", S, K, T);
  if (T == 0) return BTK_OK;
  hipLaunchKernelGGL(lefkimmiatis_iir_kernel, dim3((unsigned)K, (unsigned)S), dim3(64), 0, as_stream(stream),
                     static_cast<float2*>(Y), static_cast<const float2*>(U), static_cast<const float2*>(V),
                     static_cast<const float2*>(lambda), fbinX1, K, T_stride, T, (float)alpha, type, min_frames,
                     frames_done, static_cast<float2*>(u_state), static_cast<float2*>(v_state), w_last);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // extern "C"
