// fb_fused_big.hip -- fused analysis bank -> fixed-weight beamformer for the large geometries M = 1024 / 2048 (m = 4, r = 1), gfx950.
//
// Reference: OverSampledDFTAnalysisBank::next x N channels (modulated/modulated.cc:375-409) feeding SubbandDS / GSC / MVDR::next
// with static weights (beamformer/beamformer.cc:1267-1311, 2537-2587) -- BASELINE's superdirective 256-mic / 2048-bin array and the
// 64-mic / 1024-bin MVDR once its weights are designed.  Same contract as analysis512_bfz_kernel (fb_analysis512.hip): one
// workgroup owns a (stream, 8-frame tile[, channel group]) and walks over its channels; the N x K snapshot block never reaches HBM
// (N (4 D + 8 K) + 8 K (N + 1) bytes per frame become 4 D N + 8 K: 5.25 MB -> 1.05 MB per frame at 256 x 2048).
//
// What changes against M = 512 is the transform.  The beamformer sum lives in the Z domain on the FFT lanes (two accumulators per
// Z bin), so a lane may hold no more than 16 bins or the accumulators alone fill the register file -- the round-3 attempt (the
// generic two-pass kernel with 32-point passes and a per-channel Hermitian pass) spilled ~400 registers and ran 3.4-4.6 x slower than
// the staged pair.  Here the NF = M/2-point complex FFT of a frame runs on LPF = NF / 16 lanes (64 at M = 2048: one frame per
// wavefront; 32 at M = 1024) as THREE in-register passes 16 x Q x 16, Q = NF / 256:
//   n = LPF a + l, l = 16 b + j                 (a < 16, b < Q, j < 16)
//   1a  lane (b, j)   : B[c]          = sum_a z[LPF a + l] W_16^{a c}                                   (radix 16)
//   1b  lane (jg, c)  : A[c + 16 d]   = sum_b W_LPF^{b c} B_b[c] W_Q^{b d}   for its 16 / Q values of j   (radix Q, twiddle folded)
//   2   lane k1       : Z[k1 + LPF k2] = sum_j W_NF^{j k1} A_j[k1] W_16^{j k2}                          (radix 16, twiddle folded)
// with the folded-constant butterflies of fft_packed.h (a twiddle is one rotation FMA, its cosine rides in the butterfly) and no
// workgroup barrier inside the transform.  The hand-over 1b -> 2 is a wave-private LDS exchange (17-column padded rows); the hand-over
// 1a -> 1b trades the lane's ROW (b = lane bits 4, 5) against two bits of the register index, which gfx950 does in registers:
// v_permlane32_swap / v_permlane16_swap (rows4 / rows2 below; the first version went through LDS here too: 16 ds_write_b64 +
// 16 ds_read_b64 a lane, and LDS stores cost three times a load).  Per channel a lane spends 64 (4-tap polyphase) + 72 + 12 Q + 94 (FFT)
// + 66 (two complex multiply-adds per bin) packed instructions and 8 Q row swaps.
// The polyphase stage is the register-window form of the M = 512 kernel (15 eight-byte loads straight from HBM / L2 for 16 outputs,
// issued one channel ahead), weights are staged as (w[q], w[NF - q]) pairs through a double-buffered LDS region.
// One stream of a large array has few tiles (512 frames = 64): the channels are then split over CG workgroups per tile whose
// partial sums go to a scratch block and are added by a second, tiny kernel in a fixed order (bit-reproducible, no atomics).
#include "btk_internal.h"
#include "fft_packed.h"
#include <type_traits>

namespace {

constexpr int B_MT = 4, B_TT = 8;
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int LOG2M> struct BG {
  static constexpr int M = 1 << LOG2M, NF = M / 2, Q = NF / 256, LPF = NF / 16;
  static constexpr int NT = B_TT * LPF;                  // 256 / 512 threads: thread = pair-index class in the polyphase stage
  static constexpr int FPWV = 64 / LPF;                  // frames per wavefront: 2 / 1
  static constexpr int FRS = 272 * Q;                    // float2 per frame region: Q x (16 rows x 17) for the exchanges (>= NF)
  static constexpr int FRO = NF + 4;                     // frame pitch of the output staging (8 frames on disjoint banks)
  static constexpr int WSTRB = NF + 16;                  // float4 per channel in the weight-pair table (entry NF = bin NF)
  static constexpr int NWP = (WSTRB + NT - 1) / NT;      // weight pairs per thread and channel
};

// W [Sw][K][N] -> Wq [Sw][N][WSTRB] float4: entry i < NF = (w[i], w[(NF - i) & (NF - 1)]), entry NF = (w[NF], 0, 0), the rest 0
__global__ void big_pair_weights_kernel(const float2* __restrict__ W, float4* __restrict__ Wq, int K, int N, int Sw, int NF, int wstr)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Sw * N * wstr) return;
  const int e = (int)(i % wstr);
  const int n = (int)((i / wstr) % N);
  const long s = i / ((long)N * wstr);
  const float2* Ws = W + s * (long)K * N;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e <= NF) {
    const float2 a = Ws[(long)e * N + n];
    const float2 bq = (e < NF) ? Ws[(long)((NF - e) & (NF - 1)) * N + n] : make_float2(0.f, 0.f);
    o = make_float4(a.x, a.y, bq.x, bq.y);
  }
  Wq[i] = o;
}

// Y[s][k][t] = sum over the CG partial blocks P[g][s][k][t] (rows tcount wide), in the order g = 0, 1, ...
__global__ void big_reduce_kernel(const float2* __restrict__ P, float2* __restrict__ Y, int CG, long rows, long tcount, long T_stride)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * tcount) return;
  const long row = i / tcount, t = i % tcount;
  float2 a = P[i];
  for (int g = 1; g < CG; g++) { const float2 b = P[(long)g * rows * tcount + i]; a.x += b.x; a.y += b.y; }
  Y[row * T_stride + t] = a;
}

// Register <-> lane-row exchanges of gfx950 (a row = 16 lanes): v_permlane32_swap trades rows 2, 3 of its first operand with rows 0, 1 of
// its second, v_permlane16_swap the odd rows of the first with the even rows of the second.  rows4(): slot s of lane row p receives x_p of
// lane row s (a 4 x 4 transpose of register index against row index, four instructions per four registers); rows2(): the same 2 x 2
// inside each pair of rows.  They carry pass 1a's result to pass 1b without the LDS round trip (16 ds_write_b64 + 16 ds_read_b64 a lane).
__device__ __forceinline__ void swap32(f2& a, f2& b)                         // both components
{
  const float ax = a.x, ay = a.y, bx = b.x, by = b.y;                        // (a bit cast applied to a vector ELEMENT reads element 0 with this compiler: go through scalars)
  const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(ax), __float_as_uint(bx), false, false);
  const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(ay), __float_as_uint(by), false, false);
  a = f2{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b = f2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
__device__ __forceinline__ void swap16(f2& a, f2& b)
{
  const float ax = a.x, ay = a.y, bx = b.x, by = b.y;                        // (a bit cast applied to a vector ELEMENT reads element 0 with this compiler: go through scalars)
  const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(ax), __float_as_uint(bx), false, false);
  const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(ay), __float_as_uint(by), false, false);
  a = f2{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b = f2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
__device__ __forceinline__ void rows4(f2* u) { swap32(u[0], u[2]); swap32(u[1], u[3]); swap16(u[0], u[1]); swap16(u[2], u[3]); }
__device__ __forceinline__ void rows2(f2* u) { swap16(u[0], u[1]); }

// PT = float: un-normalised float samples; PT = short: the 16-bit PCM itself, widened in registers (btk_fb_analysis_bf_i16: half the
// bytes of the dominant stream, the same bits out)
template <int LOG2M, int VAR, typename PT = float>     // VAR & 4: pass 1a -> 1b through rows4 / rows2 instead of LDS; VAR & 1: the window loads of the next channel are issued unconditionally and interleaved with the transform; VAR & 2: see the polyphase stage
__global__ __launch_bounds__(BG<LOG2M>::NT, (LOG2M == 10) ? 2 : 1)
void analysis_bfz_big_kernel(const PT* __restrict__ pcm, long nsamples, long pcm_stride,
                             const float* __restrict__ proto, const float2* __restrict__ twg,
                             int laN, float gain, int N, int K, const float4* __restrict__ Wq, long w_stream_stride,
                             float2* __restrict__ Yout, long T_stride, long part_stride /* float2 between channel-group blocks (0: one group) */,
                             long t0, long tcount, int ntiles, int tiles_per_xcd, int S, int CG)
{
  using G = BG<LOG2M>;
  constexpr int M = G::M, NF = G::NF, Q = G::Q, LPF = G::LPF, NT = G::NT, FRS = G::FRS, FRO = G::FRO, WSTRB = G::WSTRB, NWP = G::NWP;
  constexpr int R = 2, D = M / R, TT = B_TT;
  constexpr int SPAN = (TT - 1) * D + B_MT * M;
  constexpr int NWG = TT + (B_MT - 1) * R + 1;                              // 15 window rows: 8 frames x 2 pair indices from one sweep
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f2* fbuf = reinterpret_cast<f2*>(smem);                                    // [TT][FRS]
  f4* wq = reinterpret_cast<f4*>(smem + sizeof(f2) * TT * FRS);              // [2][WSTRB] weight pairs of a channel

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bI = blockIdx.x;
  const int xcd = bI & 7, j0 = bI >> 3;
  const int cg = j0 % CG, j1 = j0 / CG;
  const int s = j1 / tiles_per_xcd;
  const int tile = xcd * tiles_per_xcd + j1 % tiles_per_xcd;
  if (s >= S || tile >= ntiles) return;
  const long tt0 = (long)tile * TT;
  const int nper = (N + CG - 1) / CG;
  const int nbeg = cg * nper, nend = (nbeg + nper < N) ? nbeg + nper : N;    // this workgroup's channels

  const long g0 = (t0 + tt0 + laN + 1) * (long)D - (long)B_MT * M;
  constexpr bool I16 = sizeof(PT) == 2;
  const bool vec_ok = ((pcm_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & (2 * sizeof(PT) - 1)) == 0);
  const bool inb = vec_ok && g0 >= 0 && g0 + SPAN <= nsamples;
  const float4* wts = Wq + (long)s * w_stream_stride;

  // ---- per-thread constants
  // polyphase: thread = class n0, owns the pair indices n0 and n0 + NT (their windows are the same samples one frame apart)
  const int n0 = tid;
  const int woff = M / 2 - 2 - 2 * n0;                                       // first sample of window row 0 relative to g0
  f2 h[2][B_MT];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int k = 0; k < B_MT; k++) { const float2 t = *reinterpret_cast<const float2*>(proto + 2 * (n0 + q * NT) + M * k); h[q][k] = f2{t.x, t.y}; }
  // FFT lanes: frame fw of the wavefront, lane l of the frame
  const int fw = lane / LPF, l = lane % LPF;
  const int frame = wave * G::FPWV + fw;
  f2* fb = fbuf + frame * FRS;
  constexpr bool SWAP = (VAR & 4) != 0;
  constexpr bool TWL = SWAP && Q == 4;                                       // the 12 twiddles of a lane live in LDS (24 registers the M = 2048 kernel does not have)
  constexpr int NS = (SWAP && !TWL) ? 16 / Q : 1;                            // SWAP: the lane of row b' holds c = Q sg + b' for sg < 16 / Q, so it needs their twiddles
  f2 tw1b[NS * (Q > 1 ? Q - 1 : 1)];                                         // W_LPF^{b c} as (cos, tan); c = l & 15 (pass 1b lane of the LDS form)
#pragma unroll
  for (int sg = 0; sg < NS; sg++)
#pragma unroll
    for (int b = 1; b < Q; b++) {
      const int c = SWAP ? Q * sg + (l >> 4) : (l & 15);
      const float2 t = twg[(32 * b * c) & (M - 1)];
      tw1b[sg * (Q - 1) + b - 1] = tw_tangent(t.x, t.y);
    }
  f2* twl = reinterpret_cast<f2*>(wq + 2 * WSTRB);                           // TWL: [16 / Q][Q - 1][LPF]
  if constexpr (TWL) {
    for (int i = tid; i < (16 / Q) * (Q - 1) * LPF; i += NT) {
      const int li = i % LPF, e = i / LPF, sg = e / (Q - 1), b = e % (Q - 1) + 1;
      const float2 t = twg[(32 * b * (Q * sg + (li >> 4))) & (M - 1)];
      twl[i] = tw_tangent(t.x, t.y);                                         // (read after the first channel's barriers)
    }
  }
  f2 tw2[15];                                                                // W_NF^{j k1}, k1 = l, j = 1..15, as (cos, tan)
#pragma unroll
  for (int j = 1; j < 16; j++) { const float2 t = twg[(2 * j * l) & (M - 1)]; tw2[j - 1] = tw_tangent(t.x, t.y); }
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int k = 0; k < B_MT; k++) asm volatile("" : "+v"(h[q][k]));          // retire these loads before the channel loop (see fb_analysis512.hip)
#pragma unroll
  for (int j = 0; j < 15; j++) asm volatile("" : "+v"(tw2[j]));
#pragma unroll
  for (int b = 0; b < NS * (Q > 1 ? Q - 1 : 1); b++) asm volatile("" : "+v"(tw1b[b]));
  const f2 k_hc = f2{0.70710678118654752f, 0.92387953251128674f}, k_t1 = f2{0.41421356237309503f, 0.41421356237309503f};

  f2 accA[16], accB[16];
#pragma unroll
  for (int k2 = 0; k2 < 16; k2++) { accA[k2] = f2{0.f, 0.f}; accB[k2] = f2{0.f, 0.f}; }
  float2 accN = make_float2(0.f, 0.f);

  float2 win[NWG];
  f4 wpre[NWP];
  // window rows of channel n: interior tiles take them unguarded (the loop of an interior tile holds no bounds test)
  auto wload = [&](int n, auto fast) {
    const PT* src = pcm + ((long)s * N + n) * pcm_stride;
    if constexpr (decltype(fast)::value) {
      // buffer loads: the channel's row base is a scalar resource, the row a scalar offset, the thread's part one 32-bit register --
      // no vector address arithmetic per load (plain pointers cost the loop 32 v_add_co / v_addc per channel)
      // (int16: a TYPED resource -- 16_16 SSCALED -- whose loads deliver the two samples as floats: btk_internal.h)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<PT*>(src + g0), 0, 0x7fffffff, I16 ? BTK_RSRC_I16X2_SSCALED : 0x00020000);
      const unsigned vo = (unsigned)woff * (unsigned)sizeof(PT);             // woff >= 0: M / 2 - 2 - 2 n0, n0 < M / 4
#pragma unroll
      for (int i = 0; i < NWG; i++) {
        if constexpr (I16) {
          const btk_f2v t = btk_buffer_load_i16x2_f32(rs, (int)vo, i * D * 2, 0);
          win[i] = make_float2(t.x, t.y);
        } else {
          const u2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, i * D * 4, 0);
          win[i] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NWG; i++) {
        const long g = g0 + woff + (long)i * D;
        win[i].x = (g >= 0 && g < nsamples) ? (float)src[g] : 0.0f;
        win[i].y = (g + 1 >= 0 && g + 1 < nsamples) ? (float)src[g + 1] : 0.0f;
      }
    }
  };
  auto wfetch = [&](int n) {
#pragma unroll
    for (int q = 0; q < NWP; q++) {
      const int e = tid + q * NT;
      if ((WSTRB % NT) == 0 || e < WSTRB) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wts + (long)n * WSTRB), 0, WSTRB * 16, 0x00020000);
        const u4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)tid * 16u, q * NT * 16, 0);
        wpre[q] = f4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
      }
    }
  };
  auto wstage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NWP; q++) {
      const int e = tid + q * NT;
      if ((WSTRB % NT) == 0 || e < WSTRB) wq[buf * WSTRB + e] = wpre[q];
    }
  };

  auto channels = [&](auto fast) {
    if (nbeg < nend) { wload(nbeg, fast); wfetch(nbeg); wstage(0); }
    for (int n = nbeg; n < nend; n++) {
      const int wbuf = (n - nbeg) & 1;
#if !defined(BTK_BIG_ABLATE) || !(BTK_BIG_ABLATE & 3)                        // ablation builds (profiles/; results WRONG by design): 1 = no barriers in the channel loop, 2 = no barrier A
      __syncthreads();                                                       // A: frames and weight buffer of channel n - 1 are consumed
#endif
      // ---- polyphase: z = (h.x x.y, h.y x.x) summed over the taps; tap k of index n0 + q NT, frame g uses row g + 2 (3 - k) + (1 - q)
      if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(1);          // (between its two barriers a wavefront holds the others up: fb_analysis512.hip PRIO)
      {
        f2 po[2][TT];
#pragma unroll
        for (int k = 0; k < B_MT; k++)
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int g = 0; g < TT; g++) {
              const float2 xw = win[g + R * (B_MT - 1 - k) + (1 - q)];
              const f2 x = f2{xw.x, xw.y};
              if (k == 0) po[q][g] = pk_mul_xswap(h[q][k], x);
              else pk_fma_xswap(po[q][g], h[q][k], x);
            }
#pragma unroll
        for (int g = 0; g < TT; g++)
#pragma unroll
          for (int q = 0; q < 2; q++) fbuf[g * FRS + n0 + q * NT] = po[q][g];
      }
      if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(0);
#if !defined(BTK_BIG_ABLATE) || !(BTK_BIG_ABLATE & 1)
      __syncthreads();                                                       // B: frames written
#endif
      constexpr bool SPREAD = (VAR & 1) != 0 && decltype(fast)::value;
      if constexpr (SPREAD) { wload(n + 1 < nend ? n + 1 : n, fast); wfetch(n + 1 < nend ? n + 1 : n); }   // same basic block as the transform
      else if (n + 1 < nend) { wload(n + 1, fast); wfetch(n + 1); }          // land under the transform

      // ---- wave-private NF-point FFT of this lane's frame, result in registers
      f2 v[16];
      {
        const int b = l >> 4, j = l & 15;                                    // pass 1a
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = fb[LPF * a + l];
        dft16t(v, k_hc, k_t1);
        if constexpr (!SWAP) {
#pragma unroll
          for (int c = 0; c < 16; c++) fb[b * 272 + 17 * j + c] = v[c];
        }
      }
      if constexpr (Q > 1) {                                                 // pass 1b: radix Q over b, twiddles W_LPF^{b c} folded
        const int jg = l >> 4, c = l & 15;                                   // (SWAP: jg is the lane's row b', c its j)
#pragma unroll
        for (int jl = 0; jl < 16 / Q; jl++) {
          if constexpr (SWAP) {                                              // v[Q jl + s] <- B_{b = s}[c = Q jl + b'] of this lane's j
            if constexpr (Q == 4) rows4(v + jl * Q); else rows2(v + jl * Q);
          } else {
            const int j = jg + Q * jl;
#pragma unroll
            for (int b = 0; b < Q; b++) v[jl * Q + b] = fb[b * 272 + 17 * j + c];
          }
        }
#pragma unroll
        for (int jl = 0; jl < 16 / Q; jl++) {
          f2* u = v + jl * Q;
          f2 twv[Q - 1];
#pragma unroll
          for (int b = 0; b < Q - 1; b++) twv[b] = TWL ? twl[(jl * (Q - 1) + b) * LPF + l] : tw1b[(SWAP ? jl * (Q - 1) : 0) + b];
          const f2* tw = twv;
          if constexpr (Q == 2) {
            const f2 r1 = fma_ib_kv<1>(tw[0], u[1], u[1]);
            const f2 o0 = fma_kv<0>(tw[0], r1, u[0]), o1 = fms_kv<0>(tw[0], r1, u[0]);
            u[0] = o0; u[1] = o1;
          } else {
            const f2 r1 = fma_ib_kv<1>(tw[0], u[1], u[1]), r2 = fma_ib_kv<1>(tw[1], u[2], u[2]), r3 = fma_ib_kv<1>(tw[2], u[3], u[3]);
            const f2 s02 = fma_kv<0>(tw[1], r2, u[0]), d02 = fms_kv<0>(tw[1], r2, u[0]);
            const f2 m1 = mul_kv<0>(tw[0], r1);
            const f2 s13 = fma_kv<0>(tw[2], r3, m1), td = fms_kv<0>(tw[2], r3, m1);
            u[0] = s02 + s13; u[1] = add_ib(d02, td); u[2] = s02 - s13; u[3] = sub_ib(d02, td);
          }
        }
#pragma unroll
        for (int jl = 0; jl < 16 / Q; jl++) {
#pragma unroll
          for (int d = 0; d < Q; d++) {
            if constexpr (SWAP) fb[17 * ((Q * jl + jg) + 16 * d) + c] = v[jl * Q + d];      // A_j[c' + 16 d], c' = Q jl + b', j = this lane's
            else fb[17 * (c + 16 * d) + (jg + Q * jl)] = v[jl * Q + d];
          }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = fb[17 * l + j];
      } else {
        // (Q = 1 would read the transposed 16 x 16 block here; M = 512 has its own kernel)
      }
      const f4* wl = wq + wbuf * WSTRB + l;
      f4 wg[2][4];
#pragma unroll
      for (int q = 0; q < 4; q++) wg[0][q] = wl[q * LPF];
      dft16t_tw(v, tw2, k_hc, k_t1);                                         // v[k2] = Z[l + LPF k2]

      // ---- A[q] += conj(w[q]) Z[q],  B'[q] += conj(w[(NF - q) & (NF - 1)]) conj(Z[q])
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (g < 3) {
#pragma unroll
          for (int q = 0; q < 4; q++) wg[(g + 1) & 1][q] = wl[((g + 1) * 4 + q) * LPF];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k2 = g * 4 + q;
          const f4 w4 = wg[g & 1][q];
          acc_conjw_z(accA[k2], w4.xy, v[k2]);
          acc_conjw_conjz(accB[k2], w4.zw, v[k2]);
        }
      }
      {
        const float r = v[0].x - v[0].y;                                     // bin NF (lanes l == 0): X = gain (Z0.re - Z0.im)
        const f4 wN = wq[wbuf * WSTRB + NF];
        accN.x = fmaf(wN.x, r, accN.x);
        accN.y = fmaf(-wN.y, r, accN.y);
      }
      if constexpr (SPREAD) {
#pragma unroll
        for (int i = 0; i < NWG + NWP; i++) {                                // one load after every three LDS instructions of the transform
          __builtin_amdgcn_sched_group_barrier(0x080, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      if (n + 1 < nend) wstage(wbuf ^ 1);                                    // the other buffer was last read one channel ago (barrier A)
    }
  };
  if (inb) channels(std::true_type{});
  else channels(std::false_type{});

  __syncthreads();
  // ---- once per tile: B[k] = B'[(NF - k) & (NF - 1)] through the frame region, Hermitian post-pass, transposed store
  {
    const float hg = 0.5f * gain;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) fb[l + LPF * k2] = accB[k2];
    f2 yv[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
      const int k = l + LPF * k2;
      const f2 Bk = fb[(NF - k) & (NF - 1)];
      const float2 w = twg[k];
      const float2 c1 = make_float2(1.f + w.y, -w.x), c2 = make_float2(1.f - w.y, w.x);
      const f2 a = accA[k2];
      yv[k2] = f2{hg * ((c1.x * a.x - c1.y * a.y) + (c2.x * Bk.x - c2.y * Bk.y)),
                  hg * ((c1.x * a.y + c1.y * a.x) + (c2.x * Bk.y + c2.y * Bk.x))};
    }
    __syncthreads();                                                         // every frame's partner reads are done: restage with the output pitch
    f2* fo = fbuf + frame * FRO;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) fo[l + LPF * k2] = yv[k2];
    if (l == 0) reinterpret_cast<float2*>(wq)[frame] = make_float2(gain * accN.x, gain * accN.y);      // weights are dead
  }
  __syncthreads();
  {
    const int f = tid % TT, kq = tid / TT;                                   // NT / TT = LPF bin columns
    if (tt0 + f < tcount) {
      float2* Y = Yout + (long)cg * part_stride;
      float2* yo = Y + (long)s * K * T_stride + tt0 + f;
      const f2* zf = fbuf + f * FRO;
#pragma unroll 4
      for (int it = 0; it < 16; it++) { const f2 y = zf[kq + LPF * it]; yo[(long)(kq + LPF * it) * T_stride] = make_float2(y.x, y.y); }
      if (kq == 0) yo[(long)NF * T_stride] = reinterpret_cast<const float2*>(wq)[f];
    }
  }
}

// channel groups per tile: enough workgroups for the chip (about two per CU of work items), never fewer than 8 channels per group
// (C5 block, 64 tiles: 8 groups 0.279 ms, 4 groups -- one round of 256 workgroups -- 0.286, 16 groups 0.298)
inline int big_cg(int S, int N, long tcount)
{
  const long tiles = (tcount + B_TT - 1) / B_TT * (long)S;
  int cg = 1;
  while (cg < 16 && tiles * cg < 384 && N / (2 * cg) >= 8) cg *= 2;
  return cg;
}

template <int LOG2M, typename PT = float>
int launch_big(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, const float2* W, int per_stream,
               void* scratch, float2* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  using G = BG<LOG2M>;
  const int K = fb->K, Sw = per_stream ? S : 1;
  const float gain = fb->gain_factor > 0 ? (float)fb->gain_factor : 1.0f;
  float4* Wq = static_cast<float4*>(scratch);
  const long nw = (long)Sw * N * G::WSTRB;
  hipLaunchKernelGGL(big_pair_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, W, Wq, K, N, Sw, G::NF, G::WSTRB);
  const int CG = big_cg(S, N, tcount);
  float2* part = reinterpret_cast<float2*>(Wq + nw);
  const long rows = (long)S * K;
  const int ntiles = (int)((tcount + B_TT - 1) / B_TT);
  const int tiles_per_xcd = (ntiles + 7) / 8;
  const long nblocks = (long)8 * tiles_per_xcd * S * CG;
  const size_t lds = sizeof(f2) * B_TT * G::FRS + sizeof(f4) * 2 * G::WSTRB + sizeof(f2) * 16 * G::LPF;   // (+ the pass-1b twiddle table of the row-swap form)
  // VAR bits (BTK_FUSED_VAR forces a combination; profiles/r04_fused_big_ab.txt, profiles/r05_fused_big_variants.txt):
  //   1  window loads interleaved with the transform     M = 2048 -0.7 %, M = 1024 +2 % alone, -1 % on top of 4
  //   2  polyphase stage at wave priority 1              M = 2048 -0.6 ... -1.0 %
  //   4  pass 1a -> 1b through v_permlane32/16_swap instead of LDS (round 5): M = 1024 -7.5 %, M = 2048 -2 %, bit-identical
  const int var = btk_switches().fused_var >= 0 ? (btk_switches().fused_var & 7) : 7;
  using KernT = decltype(&analysis_bfz_big_kernel<LOG2M, 7, PT>);
  KernT kern = analysis_bfz_big_kernel<LOG2M, 7, PT>;                       // (the int16 entry has the production form only)
  if constexpr (sizeof(PT) == 4) {
    static const KernT kerns[8] = {analysis_bfz_big_kernel<LOG2M, 0>, analysis_bfz_big_kernel<LOG2M, 1>, analysis_bfz_big_kernel<LOG2M, 2>, analysis_bfz_big_kernel<LOG2M, 3>,
                                   analysis_bfz_big_kernel<LOG2M, 4>, analysis_bfz_big_kernel<LOG2M, 5>, analysis_bfz_big_kernel<LOG2M, 6>, analysis_bfz_big_kernel<LOG2M, 7>};
    kern = kerns[var];
  }
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(G::NT), lds, st, pcm, nsamples, pcm_stride, fb->d_proto, fb->d_tw, fb->laN, gain,
                     N, K, Wq, per_stream ? (long)N * G::WSTRB : 0L, CG > 1 ? part : Y, CG > 1 ? tcount : T_stride,
                     CG > 1 ? rows * tcount : 0L, t0, tcount, ntiles, tiles_per_xcd, S, CG);
  BTK_HIP_CHECK(hipGetLastError());
  if (CG > 1) {
    const long n = rows * tcount;
    hipLaunchKernelGGL(big_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, Y, CG, rows, tcount, T_stride);
    BTK_HIP_CHECK(hipGetLastError());
  }
  return BTK_OK;
}

}  // namespace

// bytes of scratch btk_fb_analysis_bf needs for these geometries (weight pairs + the partial blocks of a channel-split launch); 0 = not covered
long btk_big_analysis_bf_scratch_bytes(const btk_fb* fb, int S, int N, int per_stream, long tcount)
{
  if (fb->m != B_MT || fb->R != 2 || (fb->M != 1024 && fb->M != 2048) || fb->kx0 != 0 || fb->kx1 != fb->K) return 0;
  const long wstr = fb->M / 2 + 16;
  const int CG = big_cg(S, N, tcount);
  return (long)sizeof(float4) * (per_stream ? S : 1) * N * wstr + (CG > 1 ? (long)sizeof(float2) * CG * S * fb->K * tcount : 0) + 16;
}

int btk_big_analysis_bf_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                            int per_stream, void* scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  if (btk_big_analysis_bf_scratch_bytes(fb, S, N, per_stream, tcount) == 0) return 0;
  const float2* Wp = static_cast<const float2*>(W);
  float2* Yp = static_cast<float2*>(Y);
  int rc;
  if (fb->M == 1024) rc = launch_big<10>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, scratch, Yp, T_stride, t0, tcount, st);
  else rc = launch_big<11>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, scratch, Yp, T_stride, t0, tcount, st);
  return rc == BTK_OK ? 1 : rc;
}

// the same from 16-bit PCM (btk_fb_analysis_bf_i16)
int btk_big_analysis_bf_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                                int per_stream, void* scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  if (btk_big_analysis_bf_scratch_bytes(fb, S, N, per_stream, tcount) == 0) return 0;
  const float2* Wp = static_cast<const float2*>(W);
  float2* Yp = static_cast<float2*>(Y);
  int rc;
  if (fb->M == 1024) rc = launch_big<10, short>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, scratch, Yp, T_stride, t0, tcount, st);
  else rc = launch_big<11, short>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, scratch, Yp, T_stride, t0, tcount, st);
  return rc == BTK_OK ? 1 : rc;
}
