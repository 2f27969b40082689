// chol_reg.h -- register-resident right-looking blocked complex Cholesky solve of one Hermitian system per 512-thread workgroup (gfx950).
// Round 4.  The panel solver of chol_blocked.h keeps the matrix in global memory / L2 and is bound by the issue slots of its CU (four
// systems per CU, 2.5 M cycles each at P = 264: panel loads, operand loads of the left-looking update and a back substitution that
// re-reads every panel).  Here the whole lower triangle lives in the ACCUMULATOR registers of the matrix cores for the lifetime of a
// system: 16 x 16 complex tiles, eight VGPRs each, dealt round-robin to the eight wavefronts (P + 1 <= 272: 153 tiles, <= 20 per
// wavefront = 160 VGPRs).  Step k of the right-looking factorisation:
//   (a) the owners of the tiles of block column k write them to an LDS panel;
//   (b) EVERY wavefront factors the diagonal block (one row per lane in lanes 0..15, pivots and column entries broadcast through SGPRs)
//       and, in the same instruction stream, solves x L11^H = a for its share of the rows below (one row per lane in lanes 16..): the
//       right-looking column step "scale column cc, subtract its multiple from the columns to the right" is the same for a row of the
//       block and a row below it.  The block is factored eight times over -- in lanes that would idle -- and nobody waits for a
//       single wavefront to finish it (the first version did: 13 k + 6 k cycles per step, now 5 k);
//   (c) every wavefront subtracts L21_i L21_j^H from the trailing tiles it owns: v_mfma_f32_16x16x4_f32, both operands straight from
//       the panel in LDS, the accumulator IS the matrix tile; the owners of the column tiles take L21 back into their registers.
// The right-hand side rides along as an extra ROW of the matrix (the last row of the last tile row: [A r; r^H .] factors into
// [L 0; y^H .] with L y = r), so forward substitution costs nothing; back substitution walks the block rows in reverse with the tiles
// still in registers (partial products per tile, summed in a fixed order: bit-reproducible).  Global traffic per system: the lower
// triangle read once, the solution written once -- L never leaves the chip.
#pragma once
#include <hip/hip_runtime.h>
#include "fft_packed.h"

namespace cholr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTH = 512, NWAVE = 8;
constexpr int NT_MAX = 17;                                          // tile rows of 16: P + 1 <= 272
constexpr int NTILE_MAX = NT_MAX * (NT_MAX + 1) / 2;                // 153
constexpr int NS = (NTILE_MAX + NWAVE - 1) / NWAVE;                 // 20 tile slots per wavefront
constexpr int LD = 17;                                              // float2 row pitch of panel / diagonal-block rows (34 banks: see (c))
constexpr int ROWS = 16 * NT_MAX;
constexpr int P_MAX = ROWS - 1;                                     // 271

// LDS carve-up (float2 units unless noted)
constexpr int LDP = 34;                                             // row pitch of the L21 panel: [16 L | pad | 16 -L | pad]  (68 banks: see (c))
constexpr int NEG = 17;                                             // offset of the negated copy within a panel row
constexpr int OFF_COL = 0;                                          // [ROWS][LD]: block column k as the tile owners hand it over
constexpr int OFF_PANEL = OFF_COL + ROWS * LD;                      // [ROWS][LDP]: L21 and -L21; the back substitution reuses it for its partial sums
constexpr int OFF_DSAVE = OFF_PANEL + ROWS * LDP;                   // [NT_MAX][16][LD]: L11^-H of every diagonal block
constexpr int OFF_YV = OFF_DSAVE + NT_MAX * 16 * LD;                // [ROWS]
constexpr int OFF_XV = OFF_YV + ROWS;                               // [ROWS]
constexpr int OFF_DIAG = OFF_XV + ROWS;                             // float [ROWS] (+ 16 floats of reduction scratch, + the pivot-failure flag)
inline size_t lds_bytes() { return sizeof(float2) * (size_t)OFF_DIAG + sizeof(float) * (ROWS + 32); }

__device__ __forceinline__ float2 cmul_conj_b(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ float lane_value(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }

// mat [P][P] row-major (lower triangle read, never written); rhs_conj(p) = conj(r[p]); diag(p) = the (real) diagonal entry to use.
// On success the solution of A x = r is in xv[0..P) (LDS, returned pointer) after a workgroup barrier.  Returns nullptr when a pivot
// is <= pivot_floor.  mark(i): phase-timing hook (0 load, 1 panel write, 2 block + row solves, 3 trailing update, 4 back substitution).
template <class RhsConj, class Diag, class Mark>
__device__ __forceinline__ const float2* solve(const float2* __restrict__ mat, int P, RhsConj&& rhs_conj, Diag&& diag, float pivot_floor,
                                               char* smem, Mark&& mark)
{
  float2* const lds = reinterpret_cast<float2*>(smem);
  float2* const col = lds + OFF_COL;
  float2* const pan = lds + OFF_PANEL;
  float2* const dsave = lds + OFF_DSAVE;
  float2* const yv = lds + OFF_YV;
  float2* const xv = lds + OFF_XV;
  int* const bad = reinterpret_cast<int*>(reinterpret_cast<float*>(lds + OFF_DIAG) + ROWS + 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 15, mg = lane >> 4;
  const int NT = (P + 1 + 15) >> 4;                                 // tile rows; the right-hand side is row RH, rows P .. RH-1 are identity padding
  const int RH = 16 * NT - 1;
  const int ntiles = NT * (NT + 1) / 2;

  // tile slots of this wavefront: slot s <-> tile u = 8 s + wave of the lower triangle enumerated COLUMN by column from the last one
  // (column j = NT - 1 - c holds c + 1 tiles): the tiles still alive at step k (j > k) are then a prefix of the enumeration, dealt round
  // robin -- every wavefront owns ceil or floor of (alive / 8) of them at every step (the row-major deal was 8 % off that).
  //   (packed (ti + 1) << 8 | (tj + 1), 0 = no tile; unpacked through an opaque copy at every use so that the compiler keeps ONE scalar per
  //   slot instead of hoisting a dozen per-lane addresses per slot out of the step loop: 20 slots x that is the whole register file)
  int tij[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int u = NWAVE * s + wave;
    int c = 0;
    while ((c + 1) * (c + 2) / 2 <= u) c++;
    const int j = NT - 1 - c, i = j + (u - c * (c + 1) / 2);
    tij[s] = (u < ntiles) ? (((i + 1) << 8) | (j + 1)) : 0;
  }
  auto tile_of = [&](int s, int& i, int& j) {
    int v = tij[s];
    asm volatile("" : "+s"(v));
    i = (v >> 8) - 1; j = (v & 255) - 1;
  };
  if (tid == 0) *bad = 0;                                           // (published by the first barrier of step 0)
  f32x4 re[NS], im[NS];
  // ---- load: register v of lane (mg, mi) of a tile = row 4 mg + v, column mi.  Every lane loads from a valid address and selects
  //      afterwards (no divergent branches around the loads), eight tiles' loads in flight.
  //      A branch per slot ends the basic block and the loads of slot s + 1 then wait for those of slot s (one global round trip per slot,
  //      20 per system): slots are taken four at a time, and a group whose last slot holds a tile -- all but the last group of a wavefront
  //      -- runs without branches, sixteen loads in flight.
  auto load_slot = [&](int s, int ti, int tj) {
    const int q = 16 * tj + mi;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int r = 16 * ti + 4 * mg + v;
      const bool low = r < P && q < r;
      int idx = low ? r * P + q : 0;                              // (P <= 271: 32 bits)
      asm volatile("" : "+v"(idx));                               // (opaque: keeps the load unconditional)
      const float2 a = mat[idx];
      const float dg = (r < P) ? diag(r) : ((r < RH) ? 1.f : 0.f);
      re[s][v] = low ? a.x : ((q == r) ? dg : 0.f);
      im[s][v] = low ? a.y : 0.f;
    }
  };
  constexpr int LG = 4;
#pragma unroll
  for (int s0 = 0; s0 < NS; s0 += LG) {
#pragma unroll
    for (int s = s0; s < s0 + LG && s < NS; s++) { re[s] = f32x4{0.f, 0.f, 0.f, 0.f}; im[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int slast = (s0 + LG - 1 < NS) ? s0 + LG - 1 : NS - 1;
    int tl_i, tl_j;
    tile_of(slast, tl_i, tl_j);
    if (tl_i >= 0) {                                                // tiles are dealt in order: every slot of the group holds one
#pragma unroll
      for (int s = s0; s < s0 + LG && s < NS; s++) {
        int ti, tj;
        tile_of(s, ti, tj);
        load_slot(s, ti, tj);
      }
    } else {
#pragma unroll
      for (int s = s0; s < s0 + LG && s < NS; s++) {
        int ti, tj;
        tile_of(s, ti, tj);
        if (ti >= 0) load_slot(s, ti, tj);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // the right-hand-side row: row 15 of the tiles of the last tile row
#pragma unroll
  for (int s = 0; s < NS; s++) {
    int ti, tj;
    tile_of(s, ti, tj);
    if (ti == NT - 1 && mg == 3) {
      const int q = 16 * tj + mi;
      const float2 a = (q < P) ? rhs_conj(q) : make_float2(0.f, 0.f);
      re[s][3] = a.x; im[s][3] = a.y;
    }
  }
  mark(0);

  for (int k = 0; k < NT; k++) {
    float2* const dk = dsave + k * (16 * LD);
    // the tiles still alive (j > k) are the slots u < Tk of the enumeration, block column k the NT - k tiles after them (diagonal first)
    const int Tk = (NT - 1 - k) * (NT - k) / 2;
    // ---- (a) block column k -> staging buffer (not the panel: slower wavefronts may still be reading L21 of step k - 1 from it)
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int u = NWAVE * s + wave;
      if (u >= Tk && u < Tk + NT - k) {
        const int ti = k + (u - Tk);
#pragma unroll
        for (int v = 0; v < 4; v++) col[(16 * ti + 4 * mg + v) * LD + mi] = make_float2(re[s][v], im[s][v]);
      }
    }
    __syncthreads();
    mark(1);
    // ---- (b) diagonal block (lanes 0..15) + rows below (lanes 16..63, 48 per wavefront), one instruction stream, in as few wavefronts
    //      as hold the rows (each pays the whole stream whatever its lanes carry; with <= 4 of them every one has a SIMD to itself).  The
    //      next wavefront carries the 16 rows of the IDENTITY in its lanes 16..31: solved like any other row they become L11^-H, which
    //      turns the back substitution's 16 dependent steps per block into one matrix-vector product.
    //      Rows r < c2 of the block pick up meaningless updates in their upper-triangle entries; nothing reads those.
    {
      const int nb = (k == NT - 1) ? 15 : 16;                       // the right-hand-side row takes part as a row, never as a pivot
      const int nbelow = 16 * (NT - 1 - k);
      const int nw = (nbelow + 47) / 48;                            // wavefronts 0 .. nw-1 carry rows, wavefront nw the identity (nw <= 6)
      if (wave <= nw) {
        const int lr = lane - 16;
        const bool isrow = wave < nw && lane >= 16 && wave * 48 + lr < nbelow;
        const bool isid = wave == nw && lane >= 16 && lane < 32;
        const int grow = (lane < 16) ? 16 * k + lane : 16 * (k + 1) + wave * 48 + lr;
        f2 row[16];
        const bool real = lane < 16 || isrow;
        int srow = real ? grow : 16 * k;                              // every lane reads a valid row and selects afterwards; the row index goes
        asm volatile("" : "+v"(srow));                                // through an opaque copy, or hipcc turns the selects back into branches
        const float2* src = col + srow * LD;                          // around the reads
#pragma unroll
        for (int c2 = 0; c2 < 16; c2++) {
          const float2 t = src[c2];
          row[c2] = f2{real ? t.x : ((isid && lr == c2) ? 1.f : 0.f), real ? t.y : 0.f};
        }
        bool ok = true;
#pragma unroll
        for (int cc = 0; cc < 16; cc++) {
          if (cc < nb) {
            const float piv = lane_value(row[cc].x, cc);
            ok = ok && (piv > pivot_floor);
            const float inv = __builtin_amdgcn_rsqf(piv), d = piv * inv;    // (v_rsq_f32: 1 ulp; the IEEE sqrt + divide expansions are 30 instructions
                                                                            //  per column on the one chain everything waits for)
            row[cc] = (lane == cc) ? f2{d, 0.f} : row[cc] * inv;
#pragma unroll
            for (int c2 = cc + 1; c2 < 16; c2++) {
              const f2 lc = f2{lane_value(row[cc].x, c2), lane_value(row[cc].y, c2)};     // L[c2][cc]
              // row[c2] -= row[cc] conj(lc) = (a.x lx + a.y ly, a.y lx - a.x ly)
              row[c2] = fms_ks<0>(lc, row[cc], row[c2]);               // - lx a
              row[c2] = fma_ib_ks<1>(lc, row[cc], row[c2]);            // + i ly a
            }
          }
        }
        if (!ok) *bad = 1;
        if (isid) {
#pragma unroll
          for (int c2 = 0; c2 < 16; c2++) dk[lr * LD + c2] = make_float2(row[c2].x, row[c2].y);       // L11^-H, row lr
        }
        if (isrow) {
#pragma unroll
          for (int c2 = 0; c2 < 16; c2++) {
            pan[grow * LDP + c2] = make_float2(row[c2].x, row[c2].y);
            pan[grow * LDP + NEG + c2] = make_float2(-row[c2].x, -row[c2].y);
          }
        }
        if (k == NT - 1 && wave == nw && lane == 15) {
          // y^H: the right-hand-side row of the last block (its 15 pivot columns)
#pragma unroll
          for (int c2 = 0; c2 < 16; c2++) yv[16 * k + c2] = (c2 < 15) ? make_float2(row[c2].x, -row[c2].y) : make_float2(0.f, 0.f);
        }
      }
    }
    __syncthreads();
    if (*bad) return nullptr;
    mark(2);
    // ---- (c) trailing update on the matrix cores; the owners of the column tiles take L21 back.
    //      v_mfma_f32_16x16x4_f32: A[i][kk] from lane i + 16 kk, B[kk][j] from lane j + 16 kk.  Lane (mg, mi) supplies column mg + 4 e of
    //      row mi of the panel's tile row in step e, for both operands (any pairing of the 16 columns with the (e, kk) steps sums the
    //      same products); with the 68-bank row pitch a half-wavefront's 8-byte reads cover the 64 banks exactly once.
    //      C -= A B^H:  re += (-ar) br + (-ai) bi,  im += (-ai) br + ar bi -- the negated operands are READ (the panel holds L21 and
    //      -L21), not computed: a vector instruction between two matrix instructions costs the pipe about ten cycles on this chip
    //      (profiles/ubench/mfma_war.hip), and the first version, with eight sign flips per tile, kept it half busy.
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int u = NWAVE * s + wave;
      if (u < Tk) {
        int ti, tj;
        tile_of(s, ti, tj);
        const float2* pa = pan + (16 * ti + mi) * LDP + mg;
        const float2* pb = pan + (16 * tj + mi) * LDP + mg;
        float ax[4];
        float2 n[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { ax[e] = pa[4 * e].x; n[e] = pa[NEG + 4 * e]; b[e] = pb[4 * e]; }
        // (inline assembly pins the accumulator: through the builtin hipcc writes the result to OTHER registers than the tile's and
        //  pays for the joins of the three ways through this loop body with copies and spills)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\t"
                       "v_mfma_f32_16x16x4_f32 %1, %3, %4, %1\n\t"
                       "v_mfma_f32_16x16x4_f32 %0, %3, %5, %0\n\t"
                       "v_mfma_f32_16x16x4_f32 %1, %6, %5, %1"
                       : "+v"(re[s]), "+v"(im[s]) : "v"(n[e].x), "v"(n[e].y), "v"(b[e].x), "v"(b[e].y), "v"(ax[e]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);                            // (one tile's operands at a time: twenty tiles' worth would not fit)
    }
    // the owners of block column k's tiles take L21 back (a loop of its own: as a third way through the loop above it made hipcc spill)
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int u = NWAVE * s + wave;
      if (u > Tk && u < Tk + NT - k) {
        const int ti = k + (u - Tk);
        float2 a[4];
#pragma unroll
        for (int v = 0; v < 4; v++) a[v] = pan[(16 * ti + 4 * mg + v) * LDP + mi];
        re[s] = f32x4{a[0].x, a[1].x, a[2].x, a[3].x};
        im[s] = f32x4{a[0].y, a[1].y, a[2].y, a[3].y};
      }
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");             // (the hazard recogniser does not see inside the assembly: 18 wait states between
                                                                    //  the last matrix instruction and the first vector read of its result)
    mark(3);
  }

  // ---- back substitution L^H x = y.  y^H is the right-hand-side row: row 15 of the tiles of the last tile row (registers; the last
  //      block's own 15 entries were written by step (b)).
#pragma unroll
  for (int s = 0; s < NS; s++) {
    int ti, tj;
    tile_of(s, ti, tj);
    if (ti == NT - 1 && tj >= 0 && tj < NT - 1 && mg == 3) yv[16 * tj + mi] = make_float2(re[s][3], -im[s][3]);
  }
  __syncthreads();
  float2* const part = pan;                                      // [NT_MAX][NT_MAX][16]: tile (i, k)'s  sum_r conj(L[r][c]) x_i[r]
  for (int k = NT - 1; k >= 0; k--) {
    {
      // x_k = L11^-H z,  z = y_k - sum_i (tile (i, k))^H x_i, in every wavefront (identical results; nobody waits for one wavefront to
      // publish x).  Lane (mg, mi): z_mi with the partial sums i = k + 1 + mg, + 4, ... added across the four lane groups, then row mi
      // of L11^-H (upper triangular) times z, the z entries broadcast through SGPRs.
      const int nb = (k == NT - 1) ? 15 : 16;
      const float2* dk = dsave + k * (16 * LD);
      float zx = 0.f, zy = 0.f;
      if (mg == 0) { const float2 y = yv[16 * k + mi]; zx = y.x; zy = y.y; }
      for (int i = k + 1 + mg; i < NT; i += 4) { const float2 p = part[(i * NT_MAX + k) * 16 + mi]; zx -= p.x; zy -= p.y; }
      zx += __shfl_xor(zx, 16, 64); zy += __shfl_xor(zy, 16, 64);
      zx += __shfl_xor(zx, 32, 64); zy += __shfl_xor(zy, 32, 64);
      f2 acc0 = f2{0.f, 0.f}, acc1 = f2{0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nb) {
          const float2 xr = dk[mi * LD + c];                          // L11^-H [mi][c] (zero for c < mi)
          const f2 xm = f2{xr.x, xr.y};
          const f2 zc = f2{lane_value(zx, c), lane_value(zy, c)};
          if (c & 1) { acc1 = fma_ks<0>(zc, xm, acc1); acc1 = fma_ib_ks<1>(zc, xm, acc1); }
          else       { acc0 = fma_ks<0>(zc, xm, acc0); acc0 = fma_ib_ks<1>(zc, xm, acc0); }
        }
      }
      const f2 x = acc0 + acc1;
      if (lane < 16) xv[16 * k + lane] = (lane < nb) ? make_float2(x.x, x.y) : make_float2(0.f, 0.f);     // (eight identical copies land on the same words)
    }
    if (k > 0) {
#pragma unroll
      for (int s = 0; s < NS; s++) {
        int ti, tj;
        tile_of(s, ti, tj);
        if (ti == k && tj >= 0 && tj < k) {
          float px = 0.f, py = 0.f;
#pragma unroll
          for (int v = 0; v < 4; v++) {
            const float2 x = xv[16 * k + 4 * mg + v];
            float lr = re[s][v], li = im[s][v];
            asm volatile("" : "+v"(lr), "+v"(li));                  // (opaque copies: hipcc otherwise builds packed-math operand pairs of all 160
                                                                    //  tile registers ahead of the loop and spills them)
            px = fmaf(lr, x.x, fmaf(li, x.y, px));                  // conj(l) x
            py = fmaf(lr, x.y, fmaf(-li, x.x, py));
          }
          px += __shfl_xor(px, 16, 64); py += __shfl_xor(py, 16, 64);
          px += __shfl_xor(px, 32, 64); py += __shfl_xor(py, 32, 64);
          if (mg == 0) part[(k * NT_MAX + tj) * 16 + mi] = make_float2(px, py);
        }
      }
    }
    __syncthreads();
  }
  mark(4);
  return xv;
}

}  // namespace cholr
