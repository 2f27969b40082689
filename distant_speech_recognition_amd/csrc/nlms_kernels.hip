// nlms_kernels.hip -- adaptive GSC canceller (leaky power-normalised NLMS) for gfx950.
//
// Replaces SubbandGSCLMSBeamformer.__iter__ (reference lib/pybeamformer.py:659-734), a per-frame,
// per-bin numpy loop, with three kernels per block of T frames:
//
//   1. nlms_energy_kernel   energy[s][t] = |X_0^H X_0| / M over all M bins of channel 0
//                           (update_snapshot_array(chan_no=0), pybeamformer.py:263-277,:665)
//   2. nlms_control_kernel  the scalar recurrences of one stream: gamma halving (:668-670), the
//                           silence gate energy > E_avg / sil_thresh (:672), E_avg update (:731)
//   3. nlms_bin_kernel      one wavefront (or lane group) per (stream, bin): sequential in t,
//                           channels across lanes, x_t staged through LDS in 16-frame tiles.
//
// Algebra used by kernel 3 (Nc = 1).  The reference keeps wa (N-1 values) and forms Z = B^T x
// (an (N-1)xN matvec per frame and bin).  With u = wa^H B^T (N values) every quantity it needs is
// O(N):   wa^H Z = u x,   |wa|^2 = |u|^2   (B^T has orthonormal rows),
//         conj(Z)^T B^T = (Q x)^H  with  Q = conj(B B^H) = I - vs vs^H / |vs|^2,  Q x = x - vs Yc / |vs|^2
// because B spans the complement of conj(vs) (calc_blocking_matrix, pybeamformer.py:309-341) and
// vs^H x = Yc is the upper-branch output.  The update wa^H += a e conj(Z)^T - a l wa^H becomes
//         u <- (1 - a l) u + a e (Q x)^H
// which is the SAME recursion in another basis: outputs are identical up to rounding, the
// blocking matrix never has to be read, and the kernel stays HBM-bound (8 K N bytes per frame).
// wa is recovered on demand as wa^H = u conj(B) (host side, btk_nlms_u_to_wa).
#include "btk_internal.h"
#include "fft_packed.h"
#include <cstdlib>

namespace {

constexpr int TB = 16;                 // frames per LDS tile (128-byte rows)
constexpr int LDW = TB + 1;            // padded row length (float2 units): conflict-free column reads

// |X_0^H X_0| / M of channel 0 per frame (MultiChannelSource.update_snapshot_array, lib/pybeamformer.py:263-277).  A workgroup owns 64
// frames; its four wavefronts take the bins k = q, q + 4, ... (rows 2 MB apart: every load is its own DRAM page, so the loads of eight
// bins are in flight before the first is used) and their partial sums are added in a fixed order.  Round 4's one-thread-per-frame loop
// over all 257 bins ran 0.33-0.64 ms per 32 x 4096 frames on two latency-bound wavefronts per SIMD; this form streams.
__global__ __launch_bounds__(256)
void nlms_energy_kernel(const float2* __restrict__ X, int K, int N, int M, long T_stride, long T,
                        float* __restrict__ energy /* [S][e_stride] */, long e_stride)
{
  __shared__ float part[4][64];
  const int s = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long t = (long)blockIdx.x * 64 + lane;
  const bool ok = t < T;
  const float2* x = X + (long)s * K * N * T_stride + (ok ? t : 0);   // channel 0
  const long kstep = (long)N * T_stride;
  float acc = 0.f;
  int k = q;
  for (; k + 28 < K; k += 32) {
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = x[(long)(k + 4 * j) * kstep];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float p = fmaf(v[j].x, v[j].x, v[j].y * v[j].y);
      acc += (k + 4 * j == 0 || k + 4 * j == K - 1) ? p : 2.f * p;   // mirror bins M-k carry |X_k|^2 again
    }
  }
  for (; k < K; k += 4) {
    const float2 v = x[(long)k * kstep];
    const float p = fmaf(v.x, v.x, v.y * v.y);
    acc += (k == 0 || k == K - 1) ? p : 2.f * p;
  }
  part[q][lane] = acc;
  __syncthreads();
  if (q == 0 && ok) energy[(long)s * e_stride + t] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)M;
}

struct NlmsParams {
  float beta, init_gamma, reg, energy_floor, sil_thresh, max_wa_l2norm;
  int min_frames, slowdown_after;
};

// stream_state[s] = { E_avg, gamma, isamp, ttl_updates } (doubles, in/out)
// One wavefront per stream.  E_avg obeys the linear recurrence E_t = beta E_{t-1} + (1-beta) en_t whatever the gate
// decides, so 64 frames at a time are scanned as affine maps (float64, like the reference's Python floats); the gate of
// frame t compares with E_{t-1}; the step size follows the deterministic halving schedule (gamma *= 0.5 whenever
// isamp > 0 and isamp % slowdown_after == 0, pybeamformer.py:668-670), i.e. a power of two of the block's first gamma.
__global__ __launch_bounds__(64)
void nlms_control_kernel(const float* __restrict__ energy, long T, NlmsParams p,
                         double* __restrict__ stream_state, float* __restrict__ ctrl /* [S][T] */)
{
  const int s = blockIdx.x, lane = threadIdx.x;
  double* st = stream_state + 4 * (long)s;
  double E = st[0];
  const double gamma0 = st[1];
  const long isamp0 = (long)st[2];
  long ttl = (long)st[3];
  const float* e = energy + (long)s * T;
  float* c = ctrl + (long)s * T;
  const double beta = (double)p.beta, omb = 1.0 - (double)p.beta, sil = (double)p.sil_thresh;
  const long sa = p.slowdown_after;
  // halvings applied before frame isamp0 are already in gamma0: multiples of sa in [1, isamp0 - 1]
  const long h0 = isamp0 > 0 ? (isamp0 - 1) / sa : 0;
  // scan chunks on multiples of 64 of the stream's sample counter (not of this launch): a stream processed block by block (blocks
  // ending on multiples of 64) and in one launch round alike, also when a recursion restarts mid-stream
  for (long t0 = -(isamp0 & 63); t0 < T; t0 += 64) {
    const long t = t0 + lane;
    const bool ok = t >= 0 && t < T;
    const double en = ok ? (double)e[t] : 0.0;
    double a = ok ? beta : 1.0, b = ok ? omb * en : 0.0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double a2 = __shfl_up(a, d, 64), b2 = __shfl_up(b, d, 64);
      if (lane >= d) { b = fma(a, b2, b); a *= a2; }
    }
    const double Et = fma(a, E, b);                            // E after frame t
    double Eprev = __shfl_up(Et, 1, 64);
    if (lane == 0) Eprev = E;
    const bool adapt = ok && en > Eprev / sil;                 // :672, :690
    const long isamp = isamp0 + t;
    const long h = isamp > 0 ? isamp / sa : 0;                 // multiples of sa in [1, isamp]
    const double gamma = ldexp(gamma0, -(int)(h - h0));
    if (ok) c[t] = adapt ? (float)gamma : 0.f;
    ttl += __popcll(__ballot(adapt));
    const int last = (T - t0) >= 64 ? 63 : (int)(T - t0) - 1;
    E = __shfl(Et, last, 64);
  }
  if (lane == 0) {
    const long isamp_end = isamp0 + T;
    // gamma as the sequential loop leaves it: halvings for every frame processed so far, i.e. multiples of sa in [1, isamp_end - 1]
    const long h_end = isamp_end > 0 ? (isamp_end - 1) / sa : 0;
    st[0] = E; st[1] = ldexp(gamma0, -(int)(h_end - h0)); st[2] = (double)isamp_end; st[3] = (double)ttl;
  }
}

template <int GROUP>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
  for (int d = GROUP / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// GROUP lanes cooperate on one bin (GROUP = 64 covers N <= 64*CPL); a wavefront holds 64/GROUP bins.
template <int GROUP, int CPL>
__global__ __launch_bounds__(64)
void nlms_bin_kernel(const float2* __restrict__ X, const float2* __restrict__ VS /* [K][N] */,
                     float2* __restrict__ Y, int K, int N, long T_stride, long T,
                     const float* __restrict__ ctrl,
                     const double* __restrict__ stream_state_before /* isamp at block start, [S][4] */,
                     NlmsParams p, float2* __restrict__ U /* [S][K][N] in/out */,
                     float* __restrict__ sigma2 /* [S][K] in/out */)
{
  constexpr int BPW = 64 / GROUP;                       // bins per wavefront
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);      // [BPW*NR][LDW], NR = N rounded to GROUP*CPL
  const int lane = threadIdx.x;
  const int s = blockIdx.y;
  const int kb = blockIdx.x * BPW;
  const int gl = lane % GROUP, gi = lane / GROUP;       // lane in group, group (local bin) index
  const int k = kb + gi;
  const bool kvalid = k < K;
  const int NR = GROUP * CPL;
  const long isamp0 = (long)stream_state_before[4 * (long)s + 2];

  // per-lane constants and state
  float2 vs[CPL], u[CPL];
  float vvp = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; c++) {
    const int n = gl + GROUP * c;
    const bool ok = kvalid && n < N;
    vs[c] = ok ? VS[(long)k * N + n] : make_float2(0.f, 0.f);
    u[c] = ok ? U[((long)s * K + k) * N + n] : make_float2(0.f, 0.f);
    vvp += vs[c].x * vs[c].x + vs[c].y * vs[c].y;
  }
  const float vv = group_sum<GROUP>(vvp);
  const float inv_vv = vv > 0.f ? 1.f / vv : 0.f;
  float sig = kvalid ? sigma2[(long)s * K + k] : 1.f;

  // tile loader: rows = BPW*NR (row r -> local bin r / NR, channel r % NR), TB float2 per row;
  // 16 consecutive lanes read one 128-byte row segment.
  constexpr int ROWS_PER_PASS = 64 / TB;                // 4 rows per wave-load
  constexpr int NPASS = CPL * 64 / ROWS_PER_PASS;       // BPW*NR rows / 4 rows per pass
  float2 pre[NPASS];
  float creg = 0.f;                                     // ctrl[t0 + lane%TB] of the staged tile
  const int lrow = lane / TB, lcol = lane % TB;

  auto prefetch = [&](long t0) {
    const long t = t0 + lcol;
#pragma unroll
    for (int q = 0; q < NPASS; q++) {
      const int r = q * ROWS_PER_PASS + lrow;
      const int bl = r / NR, n = r % NR;
      const int kk = kb + bl;
      float2 v = make_float2(0.f, 0.f);
      if (kk < K && n < N && t < T)
        v = X[(((long)s * K + kk) * N + n) * T_stride + t];
      pre[q] = v;
    }
    creg = t < T ? ctrl[(long)s * T + t] : 0.f;
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < NPASS; q++) tile[(q * ROWS_PER_PASS + lrow) * LDW + lcol] = pre[q];
  };

  prefetch(0);
  for (long t0 = 0; t0 < T; t0 += TB) {
    __syncthreads();                                    // previous tile fully consumed
    commit();
    const float ctile = creg;
    __syncthreads();
    if (t0 + TB < T) prefetch(t0 + TB);                 // loads fly while this tile is processed
    float2 yout = make_float2(0.f, 0.f);
    const int nsteps = (T - t0) < TB ? (int)(T - t0) : TB;
    for (int tt = 0; tt < nsteps; tt++) {
      const long t = t0 + tt;
      float2 x[CPL];
      float ycr = 0.f, yci = 0.f, pr = 0.f, pi = 0.f, xx = 0.f, uu = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        x[c] = tile[(gi * NR + gl + GROUP * c) * LDW + tt];
        // Yc = conj(vs) . x ; p = u . x
        ycr = fmaf(vs[c].x, x[c].x, fmaf(vs[c].y, x[c].y, ycr));
        yci = fmaf(vs[c].x, x[c].y, fmaf(-vs[c].y, x[c].x, yci));
        pr = fmaf(u[c].x, x[c].x, fmaf(-u[c].y, x[c].y, pr));
        pi = fmaf(u[c].x, x[c].y, fmaf(u[c].y, x[c].x, pi));
        xx = fmaf(x[c].x, x[c].x, fmaf(x[c].y, x[c].y, xx));
        uu = fmaf(u[c].x, u[c].x, fmaf(u[c].y, u[c].y, uu));
      }
      ycr = group_sum<GROUP>(ycr); yci = group_sum<GROUP>(yci);
      pr = group_sum<GROUP>(pr);   pi = group_sum<GROUP>(pi);
      xx = group_sum<GROUP>(xx);   uu = group_sum<GROUP>(uu);

      const long isamp = isamp0 + t;
      float se = (isamp > 0) ? fmaf(sig, p.beta, (1.f - p.beta) * xx) : xx;          // :682-685
      if (se < p.energy_floor) se = p.energy_floor;                                  // :687-688
      const float gam = __shfl(ctile, tt, 64);                                        // 0 => no adaptation
      float pnr = pr, pni = pi;
      if (gam > 0.f) {                                                               // :690-720
        const float er = ycr - pr, ei = yci - pi;                                    // epa
        const float a = gam / se;
        const float c1 = p.reg > 0.f ? 1.f - a * p.reg : 1.f;
        const float c2r = a * er, c2i = a * ei;
        const float gg = xx - (ycr * ycr + yci * yci) * inv_vv;                      // |Q x|^2
        const float nrm = c1 * c1 * uu + (c2r * c2r + c2i * c2i) * gg + 2.f * c1 * (c2r * pr + c2i * pi);
        const float cK = nrm > p.max_wa_l2norm ? sqrtf(p.max_wa_l2norm / nrm) : 1.f;
        const float sr = ycr * inv_vv, si = yci * inv_vv;                            // Yc / |vs|^2
#pragma unroll
        for (int c = 0; c < CPL; c++) {
          // q = x - vs * (Yc/|vs|^2);  u <- cK (c1 u + c2 conj(q))
          const float qr = x[c].x - (vs[c].x * sr - vs[c].y * si);
          const float qi = x[c].y - (vs[c].x * si + vs[c].y * sr);
          const float nr = c1 * u[c].x + (c2r * qr + c2i * qi);
          const float ni = c1 * u[c].y + (c2i * qr - c2r * qi);
          u[c] = make_float2(cK * nr, cK * ni);
        }
        sig = se;
        pnr = cK * (c1 * pr + c2r * gg);
        pni = cK * (c1 * pi + c2i * gg);
      }
      const bool active = isamp >= p.min_frames;                                     // :723-726
      const float outr = active ? ycr - pnr : ycr, outi = active ? yci - pni : yci;
      if (GROUP >= TB) {
        if (gl == tt) yout = make_float2(outr, outi);     // lane tt of the group keeps frame tt
      } else {
        // small groups: 8 lanes cannot hold 16 frames, write the frame directly
        if (gl == 0 && kvalid) Y[((long)s * K + k) * T_stride + t] = make_float2(outr, outi);
      }
    }
    if (GROUP >= TB && kvalid && gl < nsteps)
      Y[((long)s * K + k) * T_stride + t0 + gl] = yout;
  }

  // write the state back
  if (kvalid) {
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int n = gl + GROUP * c;
      if (n < N) U[((long)s * K + k) * N + n] = u[c];
    }
    if (gl == 0) sigma2[(long)s * K + k] = sig;
  }
}

// packed-f32 complex accumulation forms of the recursion (next to acc_conjw_z / acc_conjw_conjz of fft_packed.h)
__device__ __forceinline__ void acc_w_z(f2& A, f2 w, f2 z)              // A += w z
{
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(A) : "v"(w), "v"(z));                       // (w.x z.x, w.x z.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(A) : "v"(w), "v"(z));        // (-w.y z.y, w.y z.x)
}
__device__ __forceinline__ void acc_w_conjz(f2& A, f2 w, f2 z)          // A += w conj(z)
{
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(A) : "v"(w), "v"(z));        // (w.x z.x, -w.x z.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(A) : "v"(w), "v"(z));                       // (w.y z.y, w.y z.x)
}

// ---- v2: row-DPP reductions, several bins per wavefront, float4 tile loads --------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over aligned groups of GROUP lanes, result in every lane of the group
template <int GROUP> __device__ __forceinline__ float group_sum2(float v)
{
  v += dpp_f<0xB1>(v);                                  // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                                  // quad_perm [2,3,0,1]
  if (GROUP >= 8) v += dpp_f<0x141>(v);                 // row_half_mirror
  if (GROUP >= 16) v += dpp_f<0x140>(v);                // row_mirror
  if (GROUP >= 32) v += __shfl_xor(v, 16, 64);
  if (GROUP >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// GROUP lanes x CPL channels per lane cooperate on one bin; a wavefront carries 64/GROUP bins, so the
// per-step scalar arithmetic of the recursion is shared by 64/GROUP bins and the reductions stay inside
// DPP rows.  TBF frames per LDS tile (rows of TBF*8 bytes read as float4 pairs).  Requires even T_stride.
// NC > 1 (Nc constraints, pybeamformer.py:309-341 with Nc > 1): the reference's blocking matrix keeps the FIRST N - Nc
// Gram-Schmidt columns of the same projector, so conj(B) B^T = I - vs vs^H / |vs|^2 - sum_j c_j c_j^H with Nc - 1 further
// orthonormal vectors c_j (CX [K][NC-1][N], from btk_nlms_constraint_vectors).  Only Q x and |Q x|^2 change:
//     Q x = x - vs Yc / |vs|^2 - sum_j c_j (c_j^H x),   |Q x|^2 = |x|^2 - |Yc|^2 / |vs|^2 - sum_j |c_j^H x|^2;
// u Q x = u x still holds because u stays in the row space of B^T.
// NPF LDS tiles are fetched together: a tile of 8 frames takes 64 bytes of each of its 64 CPL snapshot rows, and visiting
// thousands of rows 64 bytes at a time runs HBM at 2.8-3.6 TB/s whatever the kernel does with the data
// (profiles/ubench/strided_rows.hip); requesting both halves of every 128-byte line back to back reads at 5.4-5.9 TB/s.
// NPF = 2 keeps the 8-frame LDS tile (two wavefronts per SIMD) and holds the second half in registers until its turn.
// UNR / MINW (round 6, diagnostics): the steps of a tile unrolled UNR-fold and the register budget of MINW wavefronts per SIMD.  The
// default form <16, 4, 8, 1, 2> holds two prefetched tiles in registers (249 VGPRs: TWO wavefronts per SIMD, 2 048 on the chip);
// <.., NPF = 1, UNR = 4, MINW = 3> fits 166 VGPRs without spills (three per SIMD) and is slower per wavefront (BTK_NLMS_ALT=7).
template <int GROUP, int CPL, int TBF, int NC = 1, int NPF = 1, int UNR = TBF, int MINW = 1>
__global__ __launch_bounds__(64, MINW)
void nlms_bin2_kernel(const float2* __restrict__ X, const float2* __restrict__ VS, float2* __restrict__ Y,
                      int K, int N, long T_stride, long T, const float* __restrict__ ctrl,
                      const double* __restrict__ stream_state_before, NlmsParams p,
                      float2* __restrict__ U, float* __restrict__ sigma2, const float2* __restrict__ CX = nullptr)
{
  constexpr int NX = NC > 1 ? NC - 1 : 1;
  constexpr int BPW = 64 / GROUP;
  constexpr int NR = GROUP * CPL;
  // LDS tile [64 CPL rows][TBF frames], unpadded (round 6), the frame slot of a row XOR-ed with bits of the row index: column reads
  // (the 64 lanes of a step read 64 different rows at one frame) and the loader's row writes both spread over all banks (4 passes
  // for 64 8-byte accesses, the minimum) as with the 9-slot rows of rounds 2-5, in 16 KB instead of 18 (ten workgroups per CU by
  // LDS; what bounds the residency of the default form is its 249 VGPRs, see btk_nlms_process_nc).
  constexpr int SH = TBF >= 16 ? 0 : (TBF == 8 ? 1 : 2);  // slot = t ^ ((row >> SH) & (TBF - 1)): with the row pitch of 2 TBF dwords
  constexpr bool FX1 = GROUP >= 16 || CPL == 1;           // the swizzle of a lane's rows does not depend on the channel index
  constexpr int LPR = TBF / 2;                          // lanes per row (float4 = 2 frames)
  constexpr int RPP = 64 / LPR;                         // rows per wave-load
  constexpr int NPASS = 64 * CPL / RPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);      // [64*CPL rows][TBF], swizzled
  const int lane = threadIdx.x;
  const int s = blockIdx.y;
  const int kb = blockIdx.x * BPW;
  const int gl = lane % GROUP, gi = lane / GROUP;
  const int k = kb + gi;
  const bool kvalid = k < K;
  const long isamp0 = (long)stream_state_before[4 * (long)s + 2];
  const int rd_row = gi * NR + gl;                      // this lane's row of channel c: rd_row + GROUP c
  const int rd_fx = (rd_row >> SH) & (TBF - 1);

  f2 vs[CPL], u[CPL];
  float vvp = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; c++) {
    const int n = gl + GROUP * c;
    const bool ok = kvalid && n < N;
    const float2 a = ok ? VS[(long)k * N + n] : make_float2(0.f, 0.f);
    const float2 b = ok ? U[((long)s * K + k) * N + n] : make_float2(0.f, 0.f);
    vs[c] = f2{a.x, a.y};
    u[c] = f2{b.x, b.y};
    vvp += a.x * a.x + a.y * a.y;
  }
  const float vv = group_sum2<GROUP>(vvp);
  const float inv_vv = vv > 0.f ? 1.f / vv : 0.f;
  float sig = kvalid ? sigma2[(long)s * K + k] : 1.f;
  f2 cx[NX][CPL];
  if constexpr (NC > 1) {
#pragma unroll
    for (int jx = 0; jx < NX; jx++)
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int n = gl + GROUP * c;
        const float2 a = (kvalid && n < N) ? CX[((long)k * NX + jx) * N + n] : make_float2(0.f, 0.f);
        cx[jx][c] = f2{a.x, a.y};
      }
  }

  float4 pre[NPF][NPASS];
  float creg[NPF];
  const int lrow = lane / LPR, lc4 = lane % LPR;
  auto prefetch = [&](long tp) {
    // whole tiles of whole bins with N == NR channels (the C0 shape): rows kb N .. (kb + BPW) N - 1 of the stream are consecutive
    // and every load is in range -- one uniform test instead of three per load
    if (N == NR && kb + BPW <= K && tp + NPF * TBF <= T) {
      const float2* base = X + (((long)s * K + kb) * N + lrow) * T_stride + tp + 2 * lc4;
#pragma unroll
      for (int q = 0; q < NPASS; q++) {
#pragma unroll
        for (int h = 0; h < NPF; h++)
          pre[h][q] = btk_ld<false>(reinterpret_cast<const float4*>(base + (long)(q * RPP) * T_stride + h * TBF));
      }
#pragma unroll
      for (int h = 0; h < NPF; h++) creg[h] = ctrl[(long)s * T + tp + h * TBF + (lane % TBF)];
      return;
    }
#pragma unroll
    for (int q = 0; q < NPASS; q++) {
      const int r = q * RPP + lrow;
      const int bl = r / NR, n = r % NR;
      const int kk = kb + bl;
      const float2* row = X + (((long)s * K + kk) * N + n) * T_stride;
#pragma unroll
      for (int h = 0; h < NPF; h++) {                    // the tiles of a row back to back: one 128-byte line per NPF = 2
        const long t = tp + h * TBF + 2 * lc4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < K && n < N && t < T) {
          if (t + 1 < T) v = *reinterpret_cast<const float4*>(row + t);
          else { const float2 a = row[t]; v = make_float4(a.x, a.y, 0.f, 0.f); }
        }
        pre[h][q] = v;
      }
    }
#pragma unroll
    for (int h = 0; h < NPF; h++) {
      const long tc = tp + h * TBF + (lane % TBF);
      creg[h] = tc < T ? ctrl[(long)s * T + tc] : 0.f;
    }
  };
  auto commit = [&](const float4 (&pr)[NPASS]) {
#pragma unroll
    for (int q = 0; q < NPASS; q++) {
      const int r = q * RPP + lrow;
      const int f = (r >> SH) & (TBF - 1);
      float2* d = tile + r * TBF;
      d[(2 * lc4) ^ f] = make_float2(pr[q].x, pr[q].y);
      d[(2 * lc4 + 1) ^ f] = make_float2(pr[q].z, pr[q].w);
    }
  };

  prefetch(0);
  for (long tp = 0; tp < T; tp += NPF * TBF) {
#pragma unroll
  for (int h = 0; h < NPF; h++) {
    const long t0 = tp + h * TBF;
    if (h > 0 && t0 >= T) break;
    __syncthreads();
    commit(pre[h]);
    const float ctile = creg[h];
    __syncthreads();
    if (h == NPF - 1 && tp + NPF * TBF < T) prefetch(tp + NPF * TBF);
    float2 yout = make_float2(0.f, 0.f);
    float uu = 0.f;
    const int nsteps = (T - t0) < TBF ? (int)(T - t0) : TBF;
    // every tile runs all TBF steps: beyond the end of the block the tile holds zeros and the step size read from ctrl is 0
    // (no update, nothing stored), so the steps need no branch
#pragma unroll UNR
    for (int tt = 0; tt < TBF; tt++) {
      {
        // complex arithmetic in packed float32 (v_pk_fma_f32 with op_sel / neg modifiers): 5 packed FMAs per channel for the
        // five sums, 5 for the update -- half the instructions of the scalar form, and the recursion is issue-bound
        f2 x[CPL];
        f2 yc = {0.f, 0.f}, pp = {0.f, 0.f}, xx2 = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CPL; c++) {
          const int fx = FX1 ? rd_fx : (((rd_row + GROUP * c) >> SH) & (TBF - 1));
          const float2 xv = tile[(rd_row + GROUP * c) * TBF + (tt ^ fx)];
          x[c] = f2{xv.x, xv.y};
          acc_conjw_z(yc, vs[c], x[c]);                              // Yc = vs^H x
          acc_w_z(pp, u[c], x[c]);                                   // p = u x
          xx2 = __builtin_elementwise_fma(x[c], x[c], xx2);
        }
        const float ycr = group_sum2<GROUP>(yc.x), yci = group_sum2<GROUP>(yc.y);
        const float pr = group_sum2<GROUP>(pp.x), pi = group_sum2<GROUP>(pp.y);
        const float xx = group_sum2<GROUP>(xx2.x + xx2.y);
        float dxr[NX], dxi[NX], dd = 0.f;                             // d_j = c_j^H x
        if constexpr (NC > 1) {
#pragma unroll
          for (int jx = 0; jx < NX; jx++) {
            f2 dj = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CPL; c++) acc_conjw_z(dj, cx[jx][c], x[c]);
            dxr[jx] = group_sum2<GROUP>(dj.x); dxi[jx] = group_sum2<GROUP>(dj.y);
            dd = fmaf(dxr[jx], dxr[jx], fmaf(dxi[jx], dxi[jx], dd));
          }
        }
        // |u|^2: summed from u at the first step of a tile, carried as cK^2 nrm (the same expansion the clip uses) after
        // an update inside it -- rounding drift is bounded to TBF steps
        if (tt == 0) {
          f2 uu2 = {0.f, 0.f};
#pragma unroll
          for (int c = 0; c < CPL; c++) uu2 = __builtin_elementwise_fma(u[c], u[c], uu2);
          uu = group_sum2<GROUP>(uu2.x + uu2.y);
        }

        const long isamp = isamp0 + t0 + tt;
        float se = (isamp > 0) ? fmaf(sig, p.beta, (1.f - p.beta) * xx) : xx;
        if (se < p.energy_floor) se = p.energy_floor;
        const float gam = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ctile), tt));
        float pnr = pr, pni = pi;
        if (gam > 0.f) {
          const float er = ycr - pr, ei = yci - pi;
          const float a = gam * __builtin_amdgcn_rcpf(se);            // 1-ulp reciprocal: tolerance is 1e-4
          const float c1 = p.reg > 0.f ? 1.f - a * p.reg : 1.f;
          const float c2r = a * er, c2i = a * ei;
          const float gg = xx - (ycr * ycr + yci * yci) * inv_vv - dd;
          const float nrm = c1 * c1 * uu + (c2r * c2r + c2i * c2i) * gg + 2.f * c1 * (c2r * pr + c2i * pi);
          const float cK = nrm > p.max_wa_l2norm ? __builtin_amdgcn_sqrtf(p.max_wa_l2norm * __builtin_amdgcn_rcpf(nrm)) : 1.f;
          const f2 ns = {-ycr * inv_vv, -yci * inv_vv};                // q = x - vs Yc / |vs|^2
          const f2 k1 = {cK * c1, cK * c1}, k2 = {cK * c2r, cK * c2i}; // u <- cK (c1 u + c2 conj(q))
#pragma unroll
          for (int c = 0; c < CPL; c++) {
            f2 q = x[c];
            acc_w_z(q, ns, vs[c]);
            if constexpr (NC > 1) {
#pragma unroll
              for (int jx = 0; jx < NX; jx++) acc_w_z(q, f2{-dxr[jx], -dxi[jx]}, cx[jx][c]);
            }
            f2 un = k1 * u[c];
            acc_w_conjz(un, k2, q);
            u[c] = un;
          }
          sig = se;
          uu = cK * cK * nrm;
          pnr = cK * (c1 * pr + c2r * gg);
          pni = cK * (c1 * pi + c2i * gg);
        }
        const bool active = isamp >= p.min_frames;
        const float outr = active ? ycr - pnr : ycr, outi = active ? yci - pni : yci;
        if (GROUP >= TBF) {
          if (gl == tt) yout = make_float2(outr, outi);
        } else {
          if (gl == 0 && kvalid && tt < nsteps) Y[((long)s * K + k) * T_stride + t0 + tt] = make_float2(outr, outi);
        }
      }
    }
    if (GROUP >= TBF && kvalid && gl < nsteps)
      Y[((long)s * K + k) * T_stride + t0 + gl] = yout;
  }
  }
  if (kvalid) {
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int n = gl + GROUP * c;
      if (n < N) U[((long)s * K + k) * N + n] = make_float2(u[c].x, u[c].y);
    }
    if (gl == 0) sigma2[(long)s * K + k] = sig;
  }
}

template <int GROUP, int CPL, int TBF, int NC = 1, int NPF = 1, int UNR = TBF, int MINW = 1>
int launch_bin2(const float2* X, const float2* VS, float2* Y, int S, int K, int N, long T_stride, long T,
                const float* ctrl, const double* state_before, NlmsParams p, float2* U, float* sigma2, hipStream_t st,
                const float2* CX = nullptr)
{
  constexpr int BPW = 64 / GROUP;
  const size_t lds = sizeof(float2) * (size_t)64 * CPL * TBF;
  dim3 grid((unsigned)((K + BPW - 1) / BPW), (unsigned)S);
  hipLaunchKernelGGL((nlms_bin2_kernel<GROUP, CPL, TBF, NC, NPF, UNR, MINW>), grid, dim3(64), lds, st, X, VS, Y, K, N, T_stride, T,
                     ctrl, state_before, p, U, sigma2, CX);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

template <int GROUP, int CPL>
int launch_bin(const float2* X, const float2* VS, float2* Y, int S, int K, int N, long T_stride, long T,
               const float* ctrl, const double* state_before, NlmsParams p, float2* U, float* sigma2, hipStream_t st)
{
  constexpr int BPW = 64 / GROUP;
  const int NR = GROUP * CPL;
  const size_t lds = sizeof(float2) * (size_t)BPW * NR * LDW;
  dim3 grid((unsigned)((K + BPW - 1) / BPW), (unsigned)S);
  hipLaunchKernelGGL((nlms_bin_kernel<GROUP, CPL>), grid, dim3(64), lds, st, X, VS, Y, K, N, T_stride, T,
                     ctrl, state_before, p, U, sigma2);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // namespace

extern "C" {

// energy[s][t] = |X_0^H X_0| / M of channel 0 over all M bins: MultiChannelSource.update_snapshot_array
// (pybeamformer.py:263-277) as used by the NLMS gate (:665) and the covariance gates (:978, :1080, :1131).
int btk_frame_energy(const void* X, int S, int M, int N, long T_stride, long T, float* energy, long e_stride, void* stream)
{
  if (!X || !energy) return btk_set_error(BTK_ERR_PARAMETER, "btk_frame_energy: null argument");
  if (S <= 0 || N <= 0 || T < 0 || T_stride < T || e_stride < T) return btk_set_error(BTK_ERR_DIMENSION, "btk_frame_energy: bad sizes");
  if (T == 0) return BTK_OK;
  hipLaunchKernelGGL(nlms_energy_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)S), dim3(256), 0, as_stream(stream),
                     static_cast<const float2*>(X), M / 2 + 1, N, M, T_stride, T, energy, e_stride);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

long btk_nlms_workspace_bytes(int S, long T)
{
  // energy [S][T] + ctrl [S][T] floats + a copy of the stream state [S][4] doubles
  return (long)sizeof(float) * 2 * S * T + (long)sizeof(double) * 4 * S + 64;
}

int btk_nlms_process(const float* params /* host, 8 floats */, const void* vs, const void* X, void* Y,
                     int S, int M, int N, long T_stride, long T,
                     void* u_state, float* sigma2, double* stream_state, void* workspace, void* stream)
{
  return btk_nlms_process_nc(params, vs, nullptr, 1, X, Y, S, M, N, T_stride, T, u_state, sigma2, stream_state, workspace, stream);
}

int btk_nlms_process_nc(const float* params /* host, 8 floats */, const void* vs, const void* cextra, int NC, const void* X, void* Y,
                        int S, int M, int N, long T_stride, long T,
                        void* u_state, float* sigma2, double* stream_state, void* workspace, void* stream)
{
  if (!params || !vs || !X || !Y || !u_state || !sigma2 || !stream_state || !workspace || (NC > 1 && !cextra))
    return btk_set_error(BTK_ERR_PARAMETER, "btk_nlms_process: null argument");
  if (NC < 1 || NC > 8 || NC >= N)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_process: %d constraints (1..8, < N = %d channels) not supported", NC, N);
  if (S <= 0 || N < 2 || M < 2 || T < 0 || T_stride < T)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_process: bad sizes S=%d N=%d M=%d T=%ld", S, N, M, T);
  if (N > 256) return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_process: N=%d > 256 channels not supported", N);
  if (T == 0) return BTK_OK;
  const int K = M / 2 + 1;
  NlmsParams p;
  p.beta = params[0]; p.init_gamma = params[1]; p.reg = params[2]; p.energy_floor = params[3];
  p.sil_thresh = params[4]; p.max_wa_l2norm = params[5];
  p.min_frames = (int)params[6]; p.slowdown_after = (int)params[7];
  if (p.slowdown_after < 1) return btk_set_error(BTK_ERR_PARAMETER, "slowdown_after must be >= 1");
  hipStream_t st = as_stream(stream);
  float* energy = static_cast<float*>(workspace);
  float* ctrl = energy + (long)S * T;
  double* state_before = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ctrl + (long)S * T) + 63) & ~(uintptr_t)63);
  const float2* Xp = static_cast<const float2*>(X);

  hipLaunchKernelGGL(nlms_energy_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)S), dim3(256), 0, st,
                     Xp, K, N, M, T_stride, T, energy, T);
  BTK_HIP_CHECK(hipMemcpyAsync(state_before, stream_state, sizeof(double) * 4 * S, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(nlms_control_kernel, dim3((unsigned)S), dim3(64), 0, st, energy, T, p, stream_state, ctrl);
  BTK_HIP_CHECK(hipGetLastError());

  const float2* VS = static_cast<const float2*>(vs);
  float2* Yp = static_cast<float2*>(Y);
  float2* U = static_cast<float2*>(u_state);
  const bool v1 = btk_switches().nlms_v1;                                        // A/B switches of profiles/ (btk_internal.h)
  const int alt = btk_switches().nlms_alt;
  const bool vec_ok = (T_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  if (NC > 1) {
    if (!vec_ok) return btk_set_error(BTK_ERR_DIMENSION, "btk_nlms_process: Nc > 1 needs an even T_stride and a 16-byte aligned X");
    const float2* CXp = static_cast<const float2*>(cextra);
#define BTK_NLMS_NC(G, C, TB_)                                                                                                      \
    switch (NC) {                                                                                                                    \
      case 2: return launch_bin2<G, C, TB_, 2>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);          \
      case 3: return launch_bin2<G, C, TB_, 3>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);          \
      case 4: return launch_bin2<G, C, TB_, 4>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);          \
      case 5: return launch_bin2<G, C, TB_, 5>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);          \
      case 6: return launch_bin2<G, C, TB_, 6>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);          \
      case 7: return launch_bin2<G, C, TB_, 7>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);          \
      default: return launch_bin2<G, C, TB_, 8>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st, CXp);         \
    }
    if (N <= 16) { BTK_NLMS_NC(16, 1, 16) }
    else if (N <= 64) { BTK_NLMS_NC(16, 4, 8) }
    else { BTK_NLMS_NC(64, 4, 8) }
#undef BTK_NLMS_NC
  }
  if (vec_ok && !v1) {
    if (N <= 8)        return launch_bin2<8, 1, 16>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
    else if (N <= 16)  return launch_bin2<16, 1, 16>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
    else if (N <= 32)  return launch_bin2<16, 2, 16>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
    else if (N <= 64) {
      if (alt == 1) return launch_bin2<32, 2, 16>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
      if (alt == 2) return launch_bin2<8, 8, 8>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
      if (alt == 3) return launch_bin2<16, 4, 16>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
      if (alt == 4) return launch_bin2<32, 2, 16>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
      if (alt == 6) return launch_bin2<16, 4, 8>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
      // One wavefront per workgroup, every wavefront walks all T frames: a launch costs (rounds of resident wavefronts) x (a
      // wavefront's walk).  The two-tile form (6.49 -> 6.05 ms at 32 streams in round 2) has 249 VGPRs: two wavefronts per SIMD,
      // 2 048 on the chip -- and the C0 launch has 32 streams x 65 bin groups = 2 080: the last 32 run as a second round (5.3 ms
      // where 31 streams take 3.85, profiles/r06_nlms_residency.txt).  BTK_NLMS_ALT=7 is the form that fits three per SIMD (one
      // tile in registers, steps unrolled 4-fold, 166 VGPRs, no spills): 14-24 % slower per wavefront, 5.8 ms at 32 streams, and
      // not bit-identical to the default form (other fused multiply-add contractions) -- measured, not selected.
      if (alt == 7) return launch_bin2<16, 4, 8, 1, 1, 4, 3>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
      return launch_bin2<16, 4, 8, 1, 2>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
    }
    else if (N <= 128) return launch_bin2<32, 4, 8, 1, 2>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
    return launch_bin2<64, 4, 8, 1, 2>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
  }
  if (N <= 8)        return launch_bin<8, 1>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
  else if (N <= 16)  return launch_bin<16, 1>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
  else if (N <= 32)  return launch_bin<32, 1>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
  else if (N <= 64)  return launch_bin<64, 1>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
  else if (N <= 128) return launch_bin<64, 2>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
  return launch_bin<64, 4>(Xp, VS, Yp, S, K, N, T_stride, T, ctrl, state_before, p, U, sigma2, st);
}

}  // extern "C"
