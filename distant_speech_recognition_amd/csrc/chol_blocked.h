// chol_blocked.h -- blocked left-looking complex Cholesky A = L L^H of one Hermitian matrix per 256-thread workgroup, with the two
// triangular solves (gfx950).  Shared by the WPE normal equations (wpe_kernels.hip, P = C x lags) and the MVDR solve of arrays too
// large for an LDS-resident matrix (mvdr_kernels.hip, N > 136).
//   mat   [P][P] row-major in global memory (L2-resident), lower triangle used and overwritten by L
//   rhs   [P] in LDS: right-hand side in, solution of (L L^H) x = rhs out
//   red   [512] floats of LDS scratch, panel [P][CH_LD] float2 of LDS
// 16-column panels: updated from the already factored columns on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32), factored in
// LDS -- the 16 x 16 diagonal block by one wavefront in registers with SGPR broadcasts --, written back once; forward substitution
// rides along, back substitution re-reads the panels in reverse order.  Returns false as soon as a pivot is <= pivot_floor (rhs is
// then partly substituted: callers that need it reload it).
#pragma once
#include <hip/hip_runtime.h>

namespace cholb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CH_NB = 16;              // panel width
constexpr int CH_LD = CH_NB + 1;       // padded row (float2): conflict-free row-per-lane access

__device__ __forceinline__ float2 cmul_conj_b(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// value of lane `l` (compile-time) broadcast to the wavefront through an SGPR
__device__ __forceinline__ float lane_value(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }


inline size_t lds_bytes(int P) { return sizeof(float2) * (size_t)((P + 1) & ~1) + sizeof(float) * 512 + sizeof(float2) * (size_t)P * CH_LD; }

// mark(i): phase timing hook (i = 0 panel load, 1 MFMA update, 2 diagonal block, 3 rows below the block, 4 write-back + forward
// substitution); wrot rotates the single-wavefront phases over the four wavefronts (= SIMDs) of the workgroup
template <class Mark>
__device__ __forceinline__ bool solve(float2* __restrict__ mat, int P, float2* rhs, float* red, float2* panel, float pivot_floor,
                                      int wrot, Mark&& mark)
{
  const int tid = threadIdx.x;
  bool bad = false;
  const int lane = tid & 63, wave = tid >> 6;
  // the single-wavefront phases rotate over the four wavefronts (= SIMDs) with the panel and the workgroup: with every workgroup
  // using its wavefront 0, the four resident workgroups of a CU queue up on one SIMD while three idle
  const int mi = lane & 15, mk = lane >> 4;                       // MFMA operand coordinates of this lane
  for (int jb = 0; jb < P && !bad; jb += CH_NB) {
    const int nb = (P - jb < CH_NB) ? P - jb : CH_NB;
    const int rows = P - jb;
    // ---- panel <- A[jb.., jb..jb+nb)
    for (int idx = tid; idx < rows * CH_NB; idx += 256) {
      const int r = idx / CH_NB, cc = idx % CH_NB;
      panel[r * CH_LD + cc] = (cc < nb) ? mat[(long)(jb + r) * P + jb + cc] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    mark(0);                                                         // panel load
    // ---- left-looking update on the matrix cores: panel[r][cc] -= sum_{q<jb} L[jb+r][q] conj(L[jb+cc][q]).
    //      A wavefront owns 16-row blocks; v_mfma_f32_16x16x4_f32 (exact fp32) takes A[i][k] from lane i + 16 k and B[k][j] from
    //      lane j + 16 k, so a lane loads four consecutive columns of its row (q0 + 4 mk ..) for both operands and the four
    //      k-steps of a 16-column chunk pair lane group mk with column q0 + 4 mk + step -- any pairing sums the same products.
    //      Re(a conj b) = ar br + ai bi, Im = ai br - ar bi: four MFMAs per step.
    if (jb > 0) {
      const int nrb = (rows + 15) / 16;
      const float2* brow = mat + (long)(jb + mi) * P;               // the panel's own rows jb + j (valid while j < nb)
      const bool bok = mi < nb;
      for (int rb = wave; rb < nrb; rb += 4) {
        const int r = rb * 16 + mi;
        const bool aok = r < rows;
        const float2* arow = mat + (long)(jb + (aok ? r : 0)) * P;
        f32x4 cr = {0.f, 0.f, 0.f, 0.f}, ci = {0.f, 0.f, 0.f, 0.f};
        float2 av[4], bv[4], an[4], bn[4];
        auto ld = [&](float2 (&a)[4], float2 (&b)[4], int q0) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            a[e] = aok ? arow[q0 + 4 * mk + e] : make_float2(0.f, 0.f);
            b[e] = bok ? brow[q0 + 4 * mk + e] : make_float2(0.f, 0.f);
          }
        };
        ld(av, bv, 0);
        for (int q0 = 0; q0 < jb; q0 += 16) {
          if (q0 + 16 < jb) ld(an, bn, q0 + 16);                   // the next chunk's loads fly under this chunk's MFMAs
#pragma unroll
          for (int e = 0; e < 4; e++) {
            cr = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e].x, bv[e].x, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e].y, bv[e].y, cr, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e].y, bv[e].x, ci, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[e].x, bv[e].y, ci, 0, 0, 0);
          }
#pragma unroll
          for (int e = 0; e < 4; e++) { av[e] = an[e]; bv[e] = bn[e]; }
        }
        // D[i][j]: register v of lane l holds row 4 (l / 16) + v, column l % 16
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int rr = rb * 16 + 4 * mk + v;
          if (rr < rows) {
            float2 t = panel[rr * CH_LD + mi];
            panel[rr * CH_LD + mi] = make_float2(t.x - cr[v], t.y - ci[v]);
          }
        }
      }
    }
    __syncthreads();
    mark(1);                                                         // MFMA update
    // ---- the 16 x 16 diagonal block: one wavefront, column by column (LDS accesses of one wavefront stay in program order)
    if (wave == ((wrot + (jb >> 4)) & 3)) {
      // lane r < 16 holds row r of the block in registers; the pivot and the column entries L[c2][cc] travel through SGPRs
      // (v_readlane): the serial chain is register arithmetic, not LDS round trips (19 k -> a few k cycles per panel)
      int okflag = 1;
      const int r = lane & 15;
      float2 row[CH_NB];
#pragma unroll
      for (int c2 = 0; c2 < CH_NB; c2++) row[c2] = panel[r * CH_LD + c2];
#pragma unroll
      for (int cc = 0; cc < CH_NB; cc++) {
        if (cc < nb && okflag) {
          const float piv = lane_value(row[cc].x, cc);
          if (!(piv > pivot_floor)) { okflag = 0; }
          else {
            const float d = sqrtf(piv), inv = 1.0f / d;
            if (r == cc) row[cc] = make_float2(d, 0.f);
            else if (r > cc) row[cc] = make_float2(row[cc].x * inv, row[cc].y * inv);
#pragma unroll
            for (int c2 = cc + 1; c2 < CH_NB; c2++) {
              if (c2 < nb) {
                const float2 lc = make_float2(lane_value(row[cc].x, c2), lane_value(row[cc].y, c2));    // L[c2][cc]
                if (r >= c2) { const float2 t = cmul_conj_b(row[cc], lc); row[c2].x -= t.x; row[c2].y -= t.y; }
              }
            }
          }
        }
      }
      if (lane < nb) {
#pragma unroll
        for (int c2 = 0; c2 < CH_NB; c2++) panel[lane * CH_LD + c2] = row[c2];
      }
      if (lane == 0) red[0] = okflag ? 1.f : 0.f;
    }
    __syncthreads();
    if (red[0] == 0.f) { bad = true; break; }
    mark(2);                                                         // diagonal block
    // ---- rows below the block: x L11^H = a, one thread per row, no barriers (L11 entries are LDS broadcasts)
    for (int r = nb + tid; r < rows; r += 256) {
      float2 x[CH_NB];
#pragma unroll
      for (int cc = 0; cc < CH_NB; cc++) {
        if (cc < nb) {
          float2 v = panel[r * CH_LD + cc];
#pragma unroll
          for (int c2 = 0; c2 < cc; c2++) {
            const float2 t = cmul_conj_b(x[c2], panel[cc * CH_LD + c2]);
            v.x -= t.x; v.y -= t.y;
          }
          const float inv = 1.0f / panel[cc * CH_LD + cc].x;
          x[cc] = make_float2(v.x * inv, v.y * inv);
          panel[r * CH_LD + cc] = x[cc];
        }
      }
    }
    __syncthreads();
    mark(3);                                                         // rows below the block
    if (bad) break;
    // ---- write the factored panel back, forward substitution for its columns: L y = r
    for (int idx = tid; idx < rows * CH_NB; idx += 256) {
      const int r = idx / CH_NB, cc = idx % CH_NB;
      if (cc < nb && cc <= r) mat[(long)(jb + r) * P + jb + cc] = panel[r * CH_LD + cc];
    }
    if (wave == ((wrot + (jb >> 4) + 1) & 3)) {
      // L11 y = r for the panel's own entries: lane r holds y_r and row r of L11, column by column through SGPR broadcasts
      const int r = lane & 15;
      float2 lr[CH_NB];
#pragma unroll
      for (int c2 = 0; c2 < CH_NB; c2++) lr[c2] = panel[r * CH_LD + c2];
      float2 y = (lane < nb) ? rhs[jb + lane] : make_float2(0.f, 0.f);
#pragma unroll
      for (int c2 = 0; c2 < CH_NB; c2++) {
        if (c2 < nb) {
          const float dinv = 1.0f / lane_value(lr[c2].x, c2);
          const float2 yc = make_float2(lane_value(y.x, c2) * dinv, lane_value(y.y, c2) * dinv);
          if (r == c2) y = yc;
          else if (r > c2) { y.x -= lr[c2].x * yc.x - lr[c2].y * yc.y; y.y -= lr[c2].x * yc.y + lr[c2].y * yc.x; }
        }
      }
      if (lane < nb) rhs[jb + lane] = y;
    }
    __syncthreads();
    for (int r = nb + tid; r < rows; r += 256) {
      float2 y = rhs[jb + r];
      for (int cc = 0; cc < nb; cc++) {
        const float2 l = panel[r * CH_LD + cc], yy = rhs[jb + cc];
        y.x -= l.x * yy.x - l.y * yy.y;
        y.y -= l.x * yy.y + l.y * yy.x;
      }
      rhs[jb + r] = y;
    }
    __syncthreads();
    mark(4);                                                         // write-back + forward substitution
  }
  if (bad) return false;
  // ---- back substitution L^H g = y, panels in reverse order
  const int npan = (P + CH_NB - 1) / CH_NB;
  for (int pb = npan - 1; pb >= 0; pb--) {
    const int jb = pb * CH_NB;
    const int nb = (P - jb < CH_NB) ? P - jb : CH_NB;
    const int rows = P - jb;
    __syncthreads();
    for (int idx = tid; idx < rows * CH_NB; idx += 256) {
      const int r = idx / CH_NB, cc = idx % CH_NB;
      panel[r * CH_LD + cc] = (cc < nb && cc <= r) ? mat[(long)(jb + r) * P + jb + cc] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    // contributions of the rows below the diagonal block: sum_r conj(L[r][cc]) g[r], r >= nb
    float2 part[CH_NB];
#pragma unroll
    for (int cc = 0; cc < CH_NB; cc++) part[cc] = make_float2(0.f, 0.f);
    for (int r = nb + tid; r < rows; r += 256) {
      const float2 gr = rhs[jb + r];
#pragma unroll
      for (int cc = 0; cc < CH_NB; cc++) {
        const float2 l = panel[r * CH_LD + cc];
        part[cc].x += l.x * gr.x + l.y * gr.y;                     // conj(l) * g
        part[cc].y += l.x * gr.y - l.y * gr.x;
      }
    }
    // reduce the 16 partial sums over the workgroup: wave shuffles, then 4 waves through LDS
#pragma unroll
    for (int cc = 0; cc < CH_NB; cc++) {
      float px = part[cc].x, py = part[cc].y;
      for (int o = 32; o > 0; o >>= 1) { px += __shfl_xor(px, o, 64); py += __shfl_xor(py, o, 64); }
      if ((tid & 63) == 0) { red[((tid >> 6) * CH_NB + cc) * 2] = px; red[((tid >> 6) * CH_NB + cc) * 2 + 1] = py; }
    }
    __syncthreads();
    if (wave == ((wrot + pb) & 3)) {
      // L11^H g = z for the panel's own entries: lane cc holds z_cc and column cc of L11
      const int cc = lane & 15;
      float2 lc[CH_NB];
#pragma unroll
      for (int c2 = 0; c2 < CH_NB; c2++) lc[c2] = panel[c2 * CH_LD + cc];            // L[c2][cc] (zero above the diagonal)
      float2 z = make_float2(0.f, 0.f);
      if (lane < nb) {
        z = rhs[jb + lane];
        for (int wv = 0; wv < 4; wv++) { z.x -= red[(wv * CH_NB + lane) * 2]; z.y -= red[(wv * CH_NB + lane) * 2 + 1]; }
      }
#pragma unroll
      for (int c2 = CH_NB - 1; c2 >= 0; c2--) {
        if (c2 < nb) {
          const float dinv = 1.0f / lane_value(lc[c2].x, c2);
          const float2 gg = make_float2(lane_value(z.x, c2) * dinv, lane_value(z.y, c2) * dinv);
          if (cc == c2) z = gg;
          else if (cc < c2) { z.x -= lc[c2].x * gg.x + lc[c2].y * gg.y; z.y -= lc[c2].x * gg.y - lc[c2].y * gg.x; }     // conj(l) g
        }
      }
      if (lane < nb) rhs[jb + lane] = z;
    }
    __syncthreads();
  }
  return true;
}

}  // namespace cholb
