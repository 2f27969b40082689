// fb_fast.hip -- register-FFT filter-bank kernels for every power-of-two M in {256, 512, 1024, 2048}, m = 4.
//
// Generalisation of the schedule tuned in fb_analysis512.hip (reference modulated/modulated.cc:375-409,
// 553-612): persistent tiles with register prefetch, sliding-window polyphase, the M/2-point complex FFT as
// two in-register passes NF = P1 x P2 (P in {8,16,32}) with padded conflict-free LDS exchanges, wavefront-
// private frames (no workgroup barrier inside the FFT), Hermitian pass fused into the store.
//
//   n = P2 r + j   (r < P1, j < P2)            kappa = k1 + P1 k2   (k1 < P1, k2 < P2)
//   pass 1: lane j   : A[j][k1]  = sum_r z[P2 r + j] W_P1^{r k1};   A'[j][k1] = A[j][k1] W_NF^{j k1}
//   pass 2: lane k1  : Z[k1 + P1 k2] = sum_j A'[j][k1] W_P2^{j k2}
// LDS frame layout: input z[n] at r (P2+1) + j; A' and Z at j (P1+1) + k1 == (kappa / P1)(P1+1) + kappa % P1.
#include "btk_internal.h"
#include "fft_lds.h"
#include "fft_packed.h"
#include <cstdlib>

namespace {

constexpr int F_NT = 256, F_MT = 4;

__device__ __forceinline__ void f_dft4p(float2& a0, float2& a1, float2& a2, float2& a3)
{
  const float2 s02 = caddf(a0, a2), d02 = csubf(a0, a2);
  const float2 s13 = caddf(a1, a3), d13 = cmul_i<+1>(csubf(a1, a3));
  a0 = caddf(s02, s13); a1 = caddf(d02, d13); a2 = csubf(s02, s13); a3 = csubf(d02, d13);
}

// positive-exponent DFTs on register arrays: v[k] <- sum_r v[r] e^{+j 2 pi r k / P}
__device__ __forceinline__ void f_dft8p(float2 (&v)[8])
{
  constexpr float H = 0.70710678118654752f;
  float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  f_dft4p(e0, e1, e2, e3);
  f_dft4p(o0, o1, o2, o3);
  o1 = cmulf(o1, make_float2(H, H));
  o2 = cmul_i<+1>(o2);
  o3 = cmulf(o3, make_float2(-H, H));
  v[0] = caddf(e0, o0); v[4] = csubf(e0, o0);
  v[1] = caddf(e1, o1); v[5] = csubf(e1, o1);
  v[2] = caddf(e2, o2); v[6] = csubf(e2, o2);
  v[3] = caddf(e3, o3); v[7] = csubf(e3, o3);
}

__device__ __forceinline__ void f_dft16p(float2 (&v)[16])
{
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  float2 t[4][4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    float2 x0 = v[b], x1 = v[4 + b], x2 = v[8 + b], x3 = v[12 + b];
    f_dft4p(x0, x1, x2, x3);
    t[b][0] = x0; t[b][1] = x1; t[b][2] = x2; t[b][3] = x3;
  }
  t[1][1] = cmulf(t[1][1], make_float2(C1, S1));
  t[1][2] = cmulf(t[1][2], make_float2(H, H));
  t[1][3] = cmulf(t[1][3], make_float2(S1, C1));
  t[2][1] = cmulf(t[2][1], make_float2(H, H));
  t[2][2] = cmul_i<+1>(t[2][2]);
  t[2][3] = cmulf(t[2][3], make_float2(-H, H));
  t[3][1] = cmulf(t[3][1], make_float2(S1, C1));
  t[3][2] = cmulf(t[3][2], make_float2(-H, H));
  t[3][3] = cmulf(t[3][3], make_float2(-C1, -S1));
#pragma unroll
  for (int c = 0; c < 4; c++) {
    float2 y0 = t[0][c], y1 = t[1][c], y2 = t[2][c], y3 = t[3][c];
    f_dft4p(y0, y1, y2, y3);
    v[c] = y0; v[c + 4] = y1; v[c + 8] = y2; v[c + 12] = y3;
  }
}

__device__ __forceinline__ void f_dft32p(float2 (&v)[32])
{
  // r = 2a + b: two 16-point DFTs (even/odd), odd half times W32^c, radix-2 combine
  constexpr float CW[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f,
                            0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                            -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
  constexpr float SW[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f,
                            0.70710678118654752f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f,
                            1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
  float2 e[16], o[16];
#pragma unroll
  for (int a = 0; a < 16; a++) { e[a] = v[2 * a]; o[a] = v[2 * a + 1]; }
  f_dft16p(e);
  f_dft16p(o);
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const float2 t = (c == 0) ? o[0] : cmulf(o[c], make_float2(CW[c], SW[c]));
    v[c] = caddf(e[c], t);
    v[c + 16] = csubf(e[c], t);
  }
}

template <int P> __device__ __forceinline__ void f_dftp(float2 (&v)[P]);
template <> __device__ __forceinline__ void f_dftp<8>(float2 (&v)[8]) { f_dft8p(v); }
template <> __device__ __forceinline__ void f_dftp<16>(float2 (&v)[16]) { f_dft16p(v); }
template <> __device__ __forceinline__ void f_dftp<32>(float2 (&v)[32]) { f_dft32p(v); }

template <int LOG2M> struct FG {
  static constexpr int M = 1 << LOG2M, NF = M / 2;
  static constexpr int P1 = (LOG2M >= 10) ? 32 : 16;
  static constexpr int P2 = NF / P1;                          // 8, 16, 16, 32
  static constexpr int LOG2P1 = (LOG2M >= 10) ? 5 : 4;
  static constexpr int FP1 = 64 / P2, FP2 = 64 / P1;          // frames per wave per pass round
  static constexpr int FPW = FP1 > FP2 ? FP1 : FP2;           // frames per wave: 8, 4, 4, 2
  static constexpr int TT = 4 * FPW;                          // frames per workgroup tile: 32, 16, 16, 8
  static constexpr int LA = P2 + 1, LB = P1 + 1;
  static constexpr int FRS0 = (P1 * LA > P2 * LB) ? P1 * LA : P2 * LB;
  static constexpr int FRS = FRS0 | 1;                        // odd float2 stride: frames land on different banks
  static constexpr int NSPLIT = NF < F_NT ? F_NT / NF : 1;    // frame groups when a tile has fewer pair indices than threads
  static constexpr int NOWN = NF > F_NT ? NF / F_NT : 1;      // pair indices per thread
  static constexpr int FPT = TT / NSPLIT;                     // frames per (thread, pair index)
  static constexpr int KQ = F_NT / TT;                        // bin lanes in the store pass
};

// wave-private FFT of this wave's FPW frames (positive exponent); CONJ: forward transform through conjugation.
// Packed float32 arithmetic (fft_packed.h): the +-i products and twiddles ride on op_sel / neg modifiers.
template <int LOG2M, bool CONJ>
__device__ __forceinline__ void wave_fft(float2* __restrict__ frames /* this wave's first frame */, const float2* __restrict__ twj, int lane)
{
  using G = FG<LOG2M>;
  constexpr int P1 = G::P1, P2 = G::P2, LA = G::LA, LB = G::LB, FRS = G::FRS;
  const f2* twq = reinterpret_cast<const f2*>(twj);
#pragma unroll
  for (int rd = 0; rd < G::FPW / G::FP1; rd++) {               // pass 1: lane = (frame, j)
    const int fl = lane / P2, j = lane % P2;
    f2* fb = reinterpret_cast<f2*>(frames) + (rd * G::FP1 + fl) * FRS;
    f2 v[P1];
#pragma unroll
    for (int r = 0; r < P1; r++) { const f2 z = fb[r * LA + j]; v[r] = CONJ ? f2{z.x, -z.y} : z; }
    dftq<P1>(v);
#pragma unroll
    for (int k1 = 1; k1 < P1; k1++) v[k1] = cmulv(v[k1], twq[k1 * P2 + j]);
#pragma unroll
    for (int k1 = 0; k1 < P1; k1++) fb[j * LB + k1] = v[k1];
  }
#pragma unroll
  for (int rd = 0; rd < G::FPW / G::FP2; rd++) {               // pass 2: lane = (frame, k1)
    const int fl = lane / P1, k1 = lane % P1;
    f2* fb = reinterpret_cast<f2*>(frames) + (rd * G::FP2 + fl) * FRS;
    f2 v[P2];
#pragma unroll
    for (int jp = 0; jp < P2; jp++) v[jp] = fb[jp * LB + k1];
    dftq<P2>(v);
#pragma unroll
    for (int k2 = 0; k2 < P2; k2++) fb[k2 * LB + k1] = CONJ ? f2{v[k2].x, -v[k2].y} : v[k2];
  }
}

template <int LOG2M> __device__ __forceinline__ int zidx(int kappa)
{
  using G = FG<LOG2M>;
  return (kappa >> G::LOG2P1) * G::LB + (kappa & (G::P1 - 1));
}

// Polyphase ownership.  With D = M / R the windows of the sample-pair indices n and n + M/(2 GP) (GP = 2 for R >= 2)
// are the same LDS words shifted by CG = R / GP frames, so a thread owns GP such indices of NCL classes for FPT frames
// and reads every word once: NW = FPT + (m-1) R + (GP-1) CG words for GP FPT outputs per class.
//   V[i] = xs[(M - 2 - 2 n0 - (GP-1) M/GP) + (f0 + i) D];  index n0 + q NPG, frame f0 + g, tap k uses
//   V[g + R (m-1-k) + (GP-1-q) CG]   (= xs[f D + m M - 2 - 2 n - M k], modulated.cc:380-392)
template <int LOG2M, int R, bool PAIR = true> struct PP {
  using G = FG<LOG2M>;
  static constexpr int GP = (PAIR && R >= 2) ? 2 : 1;
  static constexpr int NPG = G::NF / GP;                        // classes
  static constexpr int NCL = NPG > F_NT ? NPG / F_NT : 1;       // classes per thread
  static constexpr int NSP = NPG < F_NT ? F_NT / NPG : 1;       // frame groups when there are fewer classes than threads
  static constexpr int FPT = G::TT / NSP;
  static constexpr int CG = R / GP;
  static constexpr int NW = FPT + (F_MT - 1) * R + (GP - 1) * CG;
};

template <int LOG2M, int R, bool PAIR>
__device__ __forceinline__ void pp_taps(const float* __restrict__ proto, int tid, float2 (&h)[PP<LOG2M, R, PAIR>::NCL][PP<LOG2M, R, PAIR>::GP][F_MT])
{
  using P = PP<LOG2M, R, PAIR>;
  constexpr int M = 1 << LOG2M;
  const int n0 = (P::NPG >= F_NT) ? tid : (tid % P::NPG);
#pragma unroll
  for (int c = 0; c < P::NCL; c++)
#pragma unroll
    for (int q = 0; q < P::GP; q++)
#pragma unroll
      for (int k = 0; k < F_MT; k++)
        h[c][q][k] = *reinterpret_cast<const float2*>(proto + 2 * (n0 + c * F_NT + q * P::NPG) + M * k);
}

template <int LOG2M, int R, bool PAIR>
__device__ __forceinline__ void pp_pull(const float* __restrict__ xs, int tid, float2 (&win)[PP<LOG2M, R, PAIR>::NCL][PP<LOG2M, R, PAIR>::NW])
{
  using P = PP<LOG2M, R, PAIR>;
  constexpr int M = 1 << LOG2M, D = M / R;
  const int n0 = (P::NPG >= F_NT) ? tid : (tid % P::NPG);
  const int f0 = (P::NPG >= F_NT) ? 0 : (tid / P::NPG) * P::FPT;
#pragma unroll
  for (int c = 0; c < P::NCL; c++) {
    const float* wbase = xs + (M - 2 - 2 * (n0 + c * F_NT) - (P::GP - 1) * (M / P::GP)) + f0 * D;
#pragma unroll
    for (int i = 0; i < P::NW; i++) win[c][i] = *reinterpret_cast<const float2*>(wbase + i * D);
  }
}

template <int LOG2M, int R, bool PAIR>
__device__ __forceinline__ void pp_emit(float2* __restrict__ fbuf, int tid, const float2 (&win)[PP<LOG2M, R, PAIR>::NCL][PP<LOG2M, R, PAIR>::NW],
                                        const float2 (&h)[PP<LOG2M, R, PAIR>::NCL][PP<LOG2M, R, PAIR>::GP][F_MT])
{
  using P = PP<LOG2M, R, PAIR>;
  using G = FG<LOG2M>;
  const int n0 = (P::NPG >= F_NT) ? tid : (tid % P::NPG);
  const int f0 = (P::NPG >= F_NT) ? 0 : (tid / P::NPG) * P::FPT;
#pragma unroll
  for (int c = 0; c < P::NCL; c++)
#pragma unroll
    for (int q = 0; q < P::GP; q++) {
      const int nn = n0 + c * F_NT + q * P::NPG;
      const int zoff = (nn / G::P2) * G::LA + (nn % G::P2);
#pragma unroll
      for (int g = 0; g < P::FPT; g++) {
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int k = 0; k < F_MT; k++) {
          const float2 x = win[c][g + R * (F_MT - 1 - k) + (P::GP - 1 - q) * P::CG];
          p0 = fmaf(h[c][q][k].x, x.y, p0);
          p1 = fmaf(h[c][q][k].y, x.x, p1);
        }
        fbuf[(f0 + g) * G::FRS + zoff] = make_float2(p0, p1);
      }
    }
}

constexpr int F_RUN = 8;       // consecutive tiles a workgroup walks through

// ------------------------------------------------------------------------------------------------ analysis
// PT = short (round 6): the 16-bit PCM the float samples were read from -- four samples per typed buffer load, widened by the load
// unit (btk_internal.h: btk_buffer_load_i16x4_f32), the same bits as the float entry
template <int LOG2M, int R, bool SHARD, typename PT = float>      // SHARD: only the bins [k0, k1) are stored (the full range needs no per-bin test)
__global__ __launch_bounds__(F_NT, 2)
void fast_analysis_kernel(const PT* __restrict__ pcm, long nsamples, long pcm_stride,
                          const float* __restrict__ proto, const float2* __restrict__ twg,
                          int laN, float gain, int N, int K, float2* __restrict__ X,
                          long T_stride, long t0, long tcount, int ntiles, int nruns, int nchan, int k0, int k1)
{
  // X [S][K][N][T_stride] holds the bins [k0, k1) of the plan (K = k1 - k0; the whole range for an unsharded plan)
  using G = FG<LOG2M>;
  constexpr int M = G::M, NF = G::NF, TT = G::TT, FRS = G::FRS, P2 = G::P2;
  constexpr int D = M / R;
  constexpr int SPAN = (TT - 1) * D + F_MT * M;
  constexpr int FB_BYTES = TT * FRS * 8;
  constexpr int REG_U = ((SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES + 15) & ~15;
  constexpr int NV4 = (SPAN / 4 + F_NT - 1) / F_NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float2* fbuf = reinterpret_cast<float2*>(smem);
  float2* twj = reinterpret_cast<float2*>(smem + REG_U);        // [P1][P2] W_NF^{j k1}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int chan = (slot / nruns) * 8 + xcd;
  const int run = slot % nruns;
  if (chan >= nchan) return;
  const int tile_first = run * F_RUN;
  const int tile_end = (tile_first + F_RUN < ntiles) ? tile_first + F_RUN : ntiles;

  for (int i = tid; i < NF; i += F_NT) twj[i] = twg[(2 * (i % P2) * (i / P2)) & (M - 1)];     // i = k1*P2 + j

  const PT* src = pcm + (long)chan * pcm_stride;
  constexpr bool I16 = sizeof(PT) == 2;
  const bool vec_ok = ((pcm_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & (I16 ? 7 : 15)) == 0) && (!I16 || nsamples < (1L << 30));
  const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<PT*>(src), 0, 0x7fffffff, I16 ? BTK_RSRC_I16X4_SSCALED : 0);
  float4 pre[NV4];
  auto fetch = [&](int tile) {
    const long g0 = (t0 + (long)tile * TT + laN + 1) * (long)D - (long)F_MT * M;
    if (vec_ok && g0 >= 0 && g0 + SPAN <= nsamples) {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * F_NT) * 4;
        if (l < SPAN) {
          if constexpr (I16) { const btk_f4v w = btk_buffer_load_i16x4_f32(rs16, (int)((g0 + l) * 2), 0, 0); pre[q] = make_float4(w.x, w.y, w.z, w.w); }
          else pre[q] = *reinterpret_cast<const float4*>(src + g0 + l);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * F_NT) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const long g = g0 + l + e;
          v[e] = (l + e < SPAN && g >= 0 && g < nsamples) ? (float)src[g] : 0.0f;
        }
        pre[q] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };

  using P = PP<LOG2M, R, true>;
  float2 h[P::NCL][P::GP][F_MT];                                 // taps of the pair indices this thread owns
  pp_taps<LOG2M, R, true>(proto, tid, h);
  const int s = chan / N, nch = chan % N;
  const long kstride = (long)N * T_stride;
  const float hg = 0.5f * gain;

  // Round 3: the span of the next tile is NOT carried in registers through the FFT passes any more (6-15 float4 per thread; at
  // M = 2048 that spilled 56-63 registers at two workgroups per CU): it is fetched at the top of its own tile and the other
  // workgroups of the CU cover the latency.  profiles/fb_ab.py: M = 256 1.84 -> 1.65 ms, M = 1024 1.73 -> 1.56, M = 2048 2.13 -> 1.92
  constexpr bool PREFETCH = false;
  if (PREFETCH) fetch(tile_first);
  for (int tile = tile_first; tile < tile_end; tile++) {
    const long tt0 = (long)tile * TT;
    if (!PREFETCH) fetch(tile);
#pragma unroll
    for (int q = 0; q < NV4; q++) {
      const int l = (tid + q * F_NT) * 4;
      if (l < SPAN) *reinterpret_cast<float4*>(xs + l) = pre[q];
    }
    __syncthreads();
    if (PREFETCH && tile + 1 < tile_end) fetch(tile + 1);

    // ---- polyphase with register windows (all windows are pulled before the frames overwrite the span)
    {
      float2 win[P::NCL][P::NW];
      pp_pull<LOG2M, R, true>(xs, tid, win);
      __syncthreads();
      pp_emit<LOG2M, R, true>(fbuf, tid, win, h);
    }
    __syncthreads();

    wave_fft<LOG2M, false>(fbuf + wave * G::FPW * FRS, twj, lane);
    __syncthreads();

    // ---- Hermitian post-pass fused into the store
    {
      constexpr int KQ = G::KQ;
      const int f = tid % TT, kq = tid / TT;
      const bool live = tt0 + f < tcount;
      const float2* zf = fbuf + f * FRS;
      float2* xo = X + ((long)s * K * N + nch) * T_stride + tt0 + f;
#pragma unroll 4
      for (int it = 0; it < NF / KQ; it++) {
        const int k = kq + KQ * it;
        const int kp = (NF - k) & (NF - 1);
        const float2 zk = zf[zidx<LOG2M>(k)];
        const float2 zq = zf[zidx<LOG2M>(kp)];
        const float2 e = make_float2(hg * (zk.x + zq.x), hg * (zk.y - zq.y));
        const float2 o = make_float2(hg * (zk.y + zq.y), -hg * (zk.x - zq.x));
        const float2 w = twg[k];                                  // e^{+j 2 pi k / M}, L1-resident
        if (live && (!SHARD || (k >= k0 && k < k1)))
          btk_st<false>(xo + (long)(k - (SHARD ? k0 : 0)) * kstride, make_float2(e.x + (w.x * o.x - w.y * o.y), e.y + (w.x * o.y + w.y * o.x)));
      }
      if (kq == 0 && live && (!SHARD || (NF >= k0 && NF < k1))) {
        const float2 z0 = zf[0];
        xo[(long)(NF - (SHARD ? k0 : 0)) * kstride] = make_float2(gain * (z0.x - z0.y), 0.f);
      }
    }
    __syncthreads();
  }
}

template <int LOG2M, int R, typename PT = float>
int launch_fast_analysis(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, float2* X,
                         long T_stride, long t0, long tcount, hipStream_t st)
{
  using G = FG<LOG2M>;
  constexpr int D = G::M / R;
  constexpr int SPAN = (G::TT - 1) * D + F_MT * G::M;
  constexpr int FB_BYTES = G::TT * G::FRS * 8;
  constexpr int REG_U = ((SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES + 15) & ~15;
  const size_t lds = REG_U + sizeof(float2) * G::NF;
  if (lds > 160 * 1024) return 0;
  const int nchan = S * N;
  const int ntiles = (int)((tcount + G::TT - 1) / G::TT);
  const int nruns = (ntiles + F_RUN - 1) / F_RUN;
  const long nblocks = (long)((nchan + 7) / 8) * nruns * 8;
  const bool shard = !(fb->kx0 == 0 && fb->kx1 == fb->K);
  auto kern = shard ? fast_analysis_kernel<LOG2M, R, true, PT> : fast_analysis_kernel<LOG2M, R, false, PT>;
  // per launch: the attribute is per device, and one process may drive several GPUs (btk_set_device)
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float gain = fb->gain_factor > 0 ? (float)fb->gain_factor : 1.0f;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(F_NT), lds, st, pcm, nsamples, pcm_stride, fb->d_proto, fb->d_tw,
                     fb->laN, gain, N, fb->kx1 - fb->kx0, X, T_stride, t0, tcount, ntiles, nruns, nchan, fb->kx0, fb->kx1);
  BTK_HIP_CHECK(hipGetLastError());
  return 1;
}

template <int LOG2M, typename PT = float>
int fast_analysis_r(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, float2* X,
                    long T_stride, long t0, long tcount, hipStream_t st)
{
  switch (fb->R) {
    case 1: return launch_fast_analysis<LOG2M, 1, PT>(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, st);
    case 2: return launch_fast_analysis<LOG2M, 2, PT>(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, st);
    case 4: return launch_fast_analysis<LOG2M, 4, PT>(fb, pcm, nsamples, pcm_stride, S, N, X, T_stride, t0, tcount, st);
  }
  return 0;
}


// ------------------------------------------------------------------------------------------------ fused analysis -> apply
// OverSampledDFTAnalysisBank x N -> SubbandDS/GSC/MVDR::next for static weights (beamformer.cc:1267-1311 over
// modulated.cc:375-409) at M = 256, the reference's default geometry, m = 4 (M = 512 has its own kernel in
// fb_analysis512.hip; the template also instantiates for M = 1024 / 2048 but spills there and is not dispatched).
// One workgroup = one (stream, TT-frame tile); it walks over the N channels, transforms each exactly as
// fast_analysis_kernel does and adds conj(w_n[k]) X_n[k][t] to the bins it owns: thread (f = tid % TT, kq = tid / TT)
// keeps bins kq, kq + KQ, ... of frame f in registers.  The N x K snapshot block never reaches HBM.
// Wt [Sw][N][K] complex64: one contiguous column of K weights per channel.
__global__ void f_transpose_weights_kernel(const float2* __restrict__ W, float2* __restrict__ Wt, int K, int N, int Sw)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Sw * K * N) return;
  const int n = (int)(i % N);
  const int k = (int)((i / N) % K);
  const long s = i / ((long)N * K);
  Wt[(s * N + n) * K + k] = W[i];
}

template <int LOG2M, int R, typename PT = float>
__global__ __launch_bounds__(F_NT, 2)
void fast_analysis_bf_kernel(const PT* __restrict__ pcm, long nsamples, long pcm_stride,
                             const float* __restrict__ proto, const float2* __restrict__ twg,
                             int laN, float gain, int N, int K, const float2* __restrict__ Wt, long w_stream_stride,
                             float2* __restrict__ Y, long T_stride, long t0, long tcount, int ntiles, int tiles_per_xcd, int S)
{
  using G = FG<LOG2M>;
  constexpr int M = G::M, NF = G::NF, TT = G::TT, FRS = G::FRS, P2 = G::P2, LA = G::LA;
  constexpr int D = M / R;
  constexpr int SPAN = (TT - 1) * D + F_MT * M;
  constexpr int FB_BYTES = TT * FRS * 8;
  constexpr int REG_U = ((SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES + 15) & ~15;
  constexpr int NV4 = (SPAN / 4 + F_NT - 1) / F_NT;
  constexpr int NWP = (NF + 1 + F_NT - 1) / F_NT;                // weights of a column per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float2* fbuf = reinterpret_cast<float2*>(smem);
  float2* twj = reinterpret_cast<float2*>(smem + REG_U);        // [P1][P2] W_NF^{j k1}
  float2* wcol = twj + NF;                                      // [NF + 1] weights of the current channel

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware mapping: each XCD owns a contiguous range of tiles of every stream (neighbouring tiles share their halo in L2)
  const int b = blockIdx.x;
  const int xcd = b & 7, j0 = b >> 3;
  const int s = j0 / tiles_per_xcd;
  const int tile = xcd * tiles_per_xcd + j0 % tiles_per_xcd;
  if (s >= S || tile >= ntiles) return;
  const long tt0 = (long)tile * TT;

  for (int i = tid; i < NF; i += F_NT) twj[i] = twg[(2 * (i % P2) * (i / P2)) & (M - 1)];

  constexpr bool I16 = sizeof(PT) == 2;
  const bool vec_ok = ((pcm_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & (I16 ? 7 : 15)) == 0) && (!I16 || nsamples < (1L << 30));
  const long g0 = (t0 + tt0 + laN + 1) * (long)D - (long)F_MT * M;
  const bool inb = vec_ok && g0 >= 0 && g0 + SPAN <= nsamples;
  const float2* wts = Wt + (long)s * w_stream_stride;
  float4 pre[NV4];
  float2 wpre[NWP];
  auto fetch = [&](int n) {
    const PT* src = pcm + ((long)s * N + n) * pcm_stride;
    if (inb) {
      const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<PT*>(src), 0, 0x7fffffff, I16 ? BTK_RSRC_I16X4_SSCALED : 0);
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * F_NT) * 4;
        if (l < SPAN) {
          if constexpr (I16) { const btk_f4v w = btk_buffer_load_i16x4_f32(rs16, (int)((g0 + l) * 2), 0, 0); pre[q] = make_float4(w.x, w.y, w.z, w.w); }
          else pre[q] = *reinterpret_cast<const float4*>(src + g0 + l);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * F_NT) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const long g = g0 + l + e;
          v[e] = (l + e < SPAN && g >= 0 && g < nsamples) ? (float)src[g] : 0.0f;
        }
        pre[q] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int q = 0; q < NWP; q++) {
      const int k = tid + q * F_NT;
      wpre[q] = (k <= NF) ? wts[(long)n * K + k] : make_float2(0.f, 0.f);
    }
  };

  constexpr int NOWN = G::NOWN, FPT = G::FPT, KQ = G::KQ, NIT = NF / KQ;
  const int nbase = (NF >= F_NT) ? tid : (tid % NF);
  const int f0 = (NF >= F_NT) ? 0 : (tid / NF) * FPT;
  float2 h[NOWN][F_MT];
#pragma unroll
  for (int q = 0; q < NOWN; q++)
#pragma unroll
    for (int k = 0; k < F_MT; k++) h[q][k] = *reinterpret_cast<const float2*>(proto + 2 * (nbase + q * F_NT) + M * k);
  const float hg = 0.5f * gain;
  const int f = tid % TT, kq = tid / TT;
  float2 acc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; it++) acc[it] = make_float2(0.f, 0.f);
  float2 accN = make_float2(0.f, 0.f);                          // bin NF = M/2 (kq == 0 threads)

  // M >= 1024: the span prefetch (12-24 float4 per thread) does not fit next to the accumulators -> fetched at the top.  (Unlike the
  // staged analysis kernel above, this one keeps its prefetch at M = 256: without it 2.86 -> 3.08 ms, profiles/fused256_ab.py.)
  constexpr bool PREFETCH = LOG2M <= 9;
  if (PREFETCH) fetch(0);
  for (int n = 0; n < N; n++) {
    if (!PREFETCH) fetch(n);
#pragma unroll
    for (int q = 0; q < NV4; q++) {
      const int l = (tid + q * F_NT) * 4;
      if (l < SPAN) *reinterpret_cast<float4*>(xs + l) = pre[q];
    }
#pragma unroll
    for (int q = 0; q < NWP; q++) {
      const int k = tid + q * F_NT;
      if (k <= NF) wcol[k] = wpre[q];
    }
    __syncthreads();

    // ---- polyphase with register windows (all windows are pulled before the frames overwrite the span)
    {
      constexpr int NW = FPT + (F_MT - 1) * R;
      float2 win[NOWN][NW];
#pragma unroll
      for (int q = 0; q < NOWN; q++) {
        const float* wbase = xs + (M - 2 - 2 * (nbase + q * F_NT)) + f0 * D;
#pragma unroll
        for (int i = 0; i < NW; i++) win[q][i] = *reinterpret_cast<const float2*>(wbase + i * D);
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NOWN; q++) {
        const int nn = nbase + q * F_NT;
        const int zoff = (nn / P2) * LA + (nn % P2);
#pragma unroll
        for (int ff = 0; ff < FPT; ff++) {
          float p0 = 0.f, p1 = 0.f;
#pragma unroll
          for (int k = 0; k < F_MT; k++) {
            const float2 x = win[q][ff + R * (F_MT - 1 - k)];
            p0 = fmaf(h[q][k].x, x.y, p0);
            p1 = fmaf(h[q][k].y, x.x, p1);
          }
          fbuf[(f0 + ff) * FRS + zoff] = make_float2(p0, p1);
        }
      }
    }
    __syncthreads();
    if (PREFETCH && n + 1 < N) fetch(n + 1);                      // lands under the FFT and the accumulation

    wave_fft<LOG2M, false>(fbuf + wave * G::FPW * FRS, twj, lane);
    __syncthreads();

    // ---- Hermitian post-pass + beamformer sum: y_k += conj(w_k) X_k
    {
      const float2* zf = fbuf + f * FRS;
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int k = kq + KQ * it;
        const int kp = (NF - k) & (NF - 1);
        const float2 zk = zf[zidx<LOG2M>(k)];
        const float2 zq = zf[zidx<LOG2M>(kp)];
        const float2 e = make_float2(hg * (zk.x + zq.x), hg * (zk.y - zq.y));
        const float2 o = make_float2(hg * (zk.y + zq.y), -hg * (zk.x - zq.x));
        const float2 w = twg[k];                                  // e^{+j 2 pi k / M}, L1-resident
        const float xr = e.x + (w.x * o.x - w.y * o.y), xi = e.y + (w.x * o.y + w.y * o.x);
        const float2 wk = wcol[k];
        acc[it].x = fmaf(wk.x, xr, fmaf(wk.y, xi, acc[it].x));
        acc[it].y = fmaf(wk.x, xi, fmaf(-wk.y, xr, acc[it].y));
      }
      if (kq == 0) {
        const float2 z0 = zf[0];
        const float xr = gain * (z0.x - z0.y);
        const float2 wk = wcol[NF];
        accN.x = fmaf(wk.x, xr, accN.x);
        accN.y = fmaf(-wk.y, xr, accN.y);
      }
    }
    __syncthreads();
  }

  if (tt0 + f < tcount) {
    float2* yo = Y + (long)s * K * T_stride + tt0 + f;
#pragma unroll
    for (int it = 0; it < NIT; it++) yo[(long)(kq + KQ * it) * T_stride] = acc[it];
    if (kq == 0) yo[(long)NF * T_stride] = accN;
  }
}

template <int LOG2M, int R, typename PT = float>
int launch_fast_analysis_bf(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, const float2* W,
                            int per_stream, float2* Wt, float2* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  using G = FG<LOG2M>;
  constexpr int D = G::M / R;
  constexpr int SPAN = (G::TT - 1) * D + F_MT * G::M;
  constexpr int FB_BYTES = G::TT * G::FRS * 8;
  constexpr int REG_U = ((SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES + 15) & ~15;
  const size_t lds = REG_U + sizeof(float2) * (G::NF + G::NF + 1);
  if (lds > 160 * 1024) return 0;
  const int K = fb->K, Sw = per_stream ? S : 1;
  const long nw = (long)Sw * K * N;
  hipLaunchKernelGGL(f_transpose_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, W, Wt, K, N, Sw);
  const int ntiles = (int)((tcount + G::TT - 1) / G::TT);
  const int tiles_per_xcd = (ntiles + 7) / 8;
  const long nblocks = (long)8 * tiles_per_xcd * S;
  auto kern = fast_analysis_bf_kernel<LOG2M, R, PT>;
  // per launch: the attribute is per device, and one process may drive several GPUs (btk_set_device)
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float gain = fb->gain_factor > 0 ? (float)fb->gain_factor : 1.0f;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(F_NT), lds, st, pcm, nsamples, pcm_stride, fb->d_proto, fb->d_tw,
                     fb->laN, gain, N, K, Wt, per_stream ? (long)N * K : 0L, Y, T_stride, t0, tcount, ntiles, tiles_per_xcd, S);
  BTK_HIP_CHECK(hipGetLastError());
  return 1;
}

template <int LOG2M, typename PT = float>
int fast_analysis_bf_r(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, const float2* W,
                       int per_stream, float2* Wt, float2* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  switch (fb->R) {
    case 1: return launch_fast_analysis_bf<LOG2M, 1, PT>(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream, Wt, Y, T_stride, t0, tcount, st);
    case 2: return launch_fast_analysis_bf<LOG2M, 2, PT>(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream, Wt, Y, T_stride, t0, tcount, st);
    case 4: return launch_fast_analysis_bf<LOG2M, 4, PT>(fb, pcm, nsamples, pcm_stride, S, N, W, per_stream, Wt, Y, T_stride, t0, tcount, st);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ synthesis
// Same schedule as synthesis512_kernel (fb_analysis512.hip) for any M: a workgroup walks through F_SRUN output
// blocks of one stream; the real sequences v_f live in a ring of TT + m R frame buffers; every iteration adds
// TT frames (Hermitian pre-pass from Y, wave-private forward FFT through conjugation) and emits TT blocks
// (register-window polyphase + overlap-add, float32 running sum in the reference's order).
constexpr int F_SRUN = 256;

template <int LOG2M, int R>
__global__ __launch_bounds__(F_NT, 1)
void fast_synthesis_kernel(const float2* __restrict__ Y, long nframes, long T_stride, int K,
                           const float* __restrict__ proto, const float2* __restrict__ twg,
                           int pd, float gain, float* __restrict__ out, long out_stride, long b0, long bcount, int srun)
{
  using G = FG<LOG2M>;
  constexpr int M = G::M, NF = G::NF, TT = G::TT, FRS = G::FRS, P2 = G::P2, KQ = G::KQ;
  constexpr int D = M / R;
  constexpr int HALO = F_MT * R - 1;
  constexpr int NRING = TT + HALO + 1;
  constexpr int DSPL = D < F_NT ? F_NT / D : 1;               // block groups when D < 256
  constexpr int DPT = D > F_NT ? D / F_NT : 1;                // output samples per thread
  constexpr int BPT = TT / DSPL;                              // blocks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* ring = reinterpret_cast<float2*>(smem);             // [NRING][FRS]
  float2* twj = ring + NRING * FRS;                           // [NF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.y;
  const long bt0 = b0 + (long)blockIdx.x * srun;
  const long bend = (bt0 + srun < b0 + bcount) ? bt0 + srun : b0 + bcount;
  const float2* Ys = Y + (long)s * K * T_stride;
  float* os = out + (long)s * out_stride;

  for (int i = tid; i < NF; i += F_NT) twj[i] = twg[(2 * (i % P2) * (i / P2)) & (M - 1)];
  __syncthreads();

  const long f_lo = bt0 + pd - HALO;
  // A. Hermitian pre-pass: thread (fi = tid % TT, kq = tid / TT) loads the bin pairs (k, NF - k), k = kq + KQ it < NF/2,
  //    of frame fi once and forms both Zc[k] and Zc[NF - k] (bin NF/2 pairs with itself: the kq == 0 threads); the loads
  //    of the next chunk are issued before the FFT and the overlap-add of the current one (see synthesis512_kernel).
  constexpr int NPAIR = NF / KQ / 2;
  const int fi = tid % TT, kq = tid / TT;
  float2 pa[NPAIR], pb[NPAIR], pmid = make_float2(0.f, 0.f);
  auto prefetch = [&](long fc0) {
    const long f = fc0 + fi;
    const bool fok = f >= f_lo && f >= 0 && f < nframes;
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const int k = kq + KQ * it;
      pa[it] = fok ? Ys[(long)k * T_stride + f] : make_float2(0.f, 0.f);
      pb[it] = fok ? Ys[(long)(NF - k) * T_stride + f] : make_float2(0.f, 0.f);
    }
    if (kq == 0) pmid = fok ? Ys[(long)(NF / 2) * T_stride + f] : make_float2(0.f, 0.f);
  };
  auto zc = [&](float2 a, float2 bq, int k) {                 // Zc[k] from Y[k] = a, Y[NF-k] = bq
    const float2 sm = make_float2(a.x + bq.x, a.y - bq.y), df = make_float2(a.x - bq.x, a.y + bq.y);
    const float2 w = twg[k];
    const float2 t = make_float2(w.x * df.x + w.y * df.y, w.x * df.y - w.y * df.x);
    return make_float2(sm.x - t.y, sm.y + t.x);
  };
  prefetch(f_lo + HALO - TT);
  // ring slot of frame f is (f - f_lo) mod NRING; s0 = slot of the chunk's first frame, carried from chunk to chunk: the 64-bit
  // modulo per ring access (per lane in phases A and B) cost hundreds of instructions per chunk
  auto wrap = [](int v) { return v >= NRING ? v - NRING : (v < 0 ? v + NRING : v); };
  int s0 = wrap(HALO - TT + NRING);
  for (long fc0 = f_lo + HALO - TT; fc0 < bend + pd; fc0 += TT, s0 = wrap(s0 + TT)) {
    {
      const long f = fc0 + fi;
      if (f >= f_lo) {
        float2* zf = ring + wrap(s0 + fi) * FRS;
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          const int k = kq + KQ * it;
          float2 a = pa[it], bq = pb[it];
          if (k == 0) { a.y = 0.f; bq.y = 0.f; }                 // imaginary parts of bins 0 and M/2 are ignored
          zf[(k / P2) * G::LA + (k % P2)] = zc(a, bq, k);       // input layout of pass 1: n = P2 r + j
          if (k > 0) {
            const int kk = NF - k;
            zf[(kk / P2) * G::LA + (kk % P2)] = zc(bq, a, kk);
          }
        }
        if (kq == 0) zf[((NF / 2) / P2) * G::LA + ((NF / 2) % P2)] = zc(pmid, pmid, NF / 2);
      }
    }
    __syncthreads();
    if (fc0 + TT < bend + pd) prefetch(fc0 + TT);              // lands under B and C
    // ---- B. forward FFT of this wave's FPW new frames
    {
      // frames of a wave are consecutive ring slots only when they do not wrap; handle frame by frame groups
      const long fw0 = fc0 + (long)wave * G::FPW;
      if (fw0 + G::FPW - 1 >= f_lo) {
        // ring slots of consecutive frames are consecutive modulo NRING; wave_fft needs a contiguous block,
        // so the ring is laid out such that a wave's FPW frames never straddle the wrap (NRING % FPW == 0 is
        // not guaranteed) -> process through a per-frame slot table
        using GG = G;
        constexpr int P1 = GG::P1, LA = GG::LA, LB = GG::LB;
#pragma unroll
        for (int rd = 0; rd < GG::FPW / GG::FP1; rd++) {
          const int fl = lane / P2, j = lane % P2;
          const long f = fw0 + rd * GG::FP1 + fl;
          if (f >= f_lo) {
            float2* fb = ring + wrap(s0 + wave * GG::FPW + rd * GG::FP1 + fl) * FRS;
            float2 v[P1];
#pragma unroll
            for (int r = 0; r < P1; r++) v[r] = cconjf(fb[r * LA + j]);
            f_dftp<P1>(v);
#pragma unroll
            for (int k1 = 1; k1 < P1; k1++) v[k1] = cmulf(v[k1], twj[k1 * P2 + j]);
#pragma unroll
            for (int k1 = 0; k1 < P1; k1++) fb[j * LB + k1] = v[k1];
          }
        }
#pragma unroll
        for (int rd = 0; rd < GG::FPW / GG::FP2; rd++) {
          const int fl = lane / P1, k1 = lane % P1;
          const long f = fw0 + rd * GG::FP2 + fl;
          if (f >= f_lo) {
            float2* fb = ring + wrap(s0 + wave * GG::FPW + rd * GG::FP2 + fl) * FRS;
            float2 v[P2];
#pragma unroll
            for (int jp = 0; jp < P2; jp++) v[jp] = fb[jp * LB + k1];
            f_dftp<P2>(v);
#pragma unroll
            for (int k2 = 0; k2 < P2; k2++) fb[k2 * LB + k1] = cconjf(v[k2]);
          }
        }
      }
    }
    __syncthreads();
    // ---- C. polyphase + overlap-add: thread owns sample(s) d and BPT of the TT blocks
    if (fc0 >= f_lo + HALO) {
      const int dbase = (D >= F_NT) ? tid : tid % D;
      const int bb0 = (D >= F_NT) ? 0 : (tid / D) * BPT;
#pragma unroll
      for (int q = 0; q < DPT; q++) {
        const int d = dbase + q * F_NT;
        float gco[R][F_MT];
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
          for (int k = 0; k < F_MT; k++) gco[j][k] = proto[(M - 1 - (d + j * D)) + M * k];
        float win[R][BPT + HALO];
#pragma unroll
        for (int j = 0; j < R; j++) {
          const int i = d + j * D;
          const int zi = zidx<LOG2M>(i >> 1);
#pragma unroll
          for (int wdx = 0; wdx < BPT + HALO; wdx++) {
            const float2 zz = ring[wrap(s0 + bb0 - HALO + wdx) * FRS + zi];          // frame fc0 + bb0 - HALO + wdx
            win[j][wdx] = (i & 1) ? zz.y : zz.x;
          }
        }
#pragma unroll
        for (int bb = 0; bb < BPT; bb++) {
          const long bglob = fc0 + bb0 + bb - pd;
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < R; j++) {
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < F_MT; k++)
              sv = fmaf(gco[j][k], win[j][bb + HALO - (R - 1 - j) - R * k], sv);
            if (bglob - (R - 1 - j) >= 0) acc += sv;
          }
          if (gain > 0.f) acc *= gain;
          if (bglob >= bt0 && bglob < bend) os[(bglob - b0) * D + (D - 1 - d)] = acc;
        }
      }
    }
    __syncthreads();
  }
}


// ---- round 3: M = 1024 / 2048 (R = 1, 2; aligned launches).  The ring form above needs TT + m R frame buffers of M/2 complex words: 87 KB at
// M = 1024 and 101-135 KB at M = 2048 -> ONE four-wave workgroup per CU, 4-byte stores, every ds_read_b64 of the overlap-add half used.
// Here the overlap-add lane owns FOUR consecutive samples (a "quad") of all its blocks and carries the m R - 1 history frames of its quad
// in REGISTERS from chunk to chunk, so LDS holds only the TT frames of the current chunk (68-70 KB: two workgroups per CU); window reads
// are two adjacent complex words per frame, a block leaves as one 16-byte store per lane, the pre-pass lane owns two consecutive frames
// of a bin pair (16-byte loads), the FFT is the packed wave-private one of the analysis kernels (fft_packed.h).
template <int LOG2M, int R>
__global__ __launch_bounds__(F_NT, 2)
void fast_synthesis_w_kernel(const float2* __restrict__ Y, long nframes, long T_stride, int K,
                             const float* __restrict__ proto, const float2* __restrict__ twg,
                             int pd, float gain, float* __restrict__ out, long out_stride, long b0, long bcount, int srun)
{
  using G = FG<LOG2M>;
  constexpr int M = G::M, NF = G::NF, TT = G::TT, FRS = G::FRS, P2 = G::P2;
  constexpr int D = M / R;
  constexpr int HALO = F_MT * R - 1;
  constexpr int SPL = (D / 4 >= F_NT) ? 4 : 2;                 // consecutive samples per overlap-add lane (16- or 8-byte stores)
  constexpr int NQ = D / SPL;                                  // lanes' sample groups per block
  constexpr int QPT = NQ / F_NT;                               // groups per thread
  constexpr int BPG = TT;
  constexpr int NWF = BPG + HALO;
  static_assert(NQ % F_NT == 0 && HALO <= BPG && (TT & 1) == 0, "geometry");
  constexpr int FP = TT / 2, KQW = F_NT / FP, NPW = NF / 2 / KQW;      // frame pairs, bin lanes, bin pairs per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* frm = reinterpret_cast<float2*>(smem);               // [TT][FRS]
  float2* twj = frm + TT * FRS;                                // [NF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.y;
  const long bt0 = b0 + (long)blockIdx.x * srun;
  const long bend = (bt0 + srun < b0 + bcount) ? bt0 + srun : b0 + bcount;
  const float2* Ys = Y + (long)s * K * T_stride;
  float* os = out + (long)s * out_stride;

  for (int i = tid; i < NF; i += F_NT) twj[i] = twg[(2 * (i % P2) * (i / P2)) & (M - 1)];

  const int qd = tid;
  float hist[QPT][R][HALO][SPL];                                 // the quad's samples of the HALO frames before the chunk
#pragma unroll
  for (int q = 0; q < QPT; q++)
#pragma unroll
    for (int j = 0; j < R; j++)
#pragma unroll
      for (int w = 0; w < HALO; w++)
#pragma unroll
        for (int dd = 0; dd < SPL; dd++) hist[q][j][w][dd] = 0.f;

  const long f_lo = bt0 + pd - HALO;
  // A. pre-pass lane: frames fc0 + 2 fp, + 1; bin pairs (k, NF - k), k = kq + KQW it < NF / 2; the kq == 0 lanes also own bin NF / 2
  const int fp = tid % FP, kq = tid / FP;
  const float2 twm = twg[NF / 2];
  float4 pa[NPW], pb[NPW], pmid = make_float4(0.f, 0.f, 0.f, 0.f);
  auto prefetch = [&](long fc0) {
    const long fA = fc0 + 2 * fp;
    const bool okA = fA >= 0 && fA < nframes, okB = fA + 1 >= 0 && fA + 1 < nframes;
    if (okA && okB) {                                          // 16-byte loads: fA is even and the rows are 16-byte aligned (launch check)
#pragma unroll
      for (int it = 0; it < NPW; it++) {
        const int k = kq + KQW * it;
        pa[it] = *reinterpret_cast<const float4*>(Ys + (long)k * T_stride + fA);
        pb[it] = *reinterpret_cast<const float4*>(Ys + (long)(NF - k) * T_stride + fA);
      }
      if (kq == 0) pmid = *reinterpret_cast<const float4*>(Ys + (long)(NF / 2) * T_stride + fA);
    } else {
      auto ld = [&](int k) {
        const float2 a = okA ? Ys[(long)k * T_stride + fA] : make_float2(0.f, 0.f);
        const float2 b = okB ? Ys[(long)k * T_stride + fA + 1] : make_float2(0.f, 0.f);
        return make_float4(a.x, a.y, b.x, b.y);
      };
#pragma unroll
      for (int it = 0; it < NPW; it++) { const int k = kq + KQW * it; pa[it] = ld(k); pb[it] = ld(NF - k); }
      if (kq == 0) pmid = ld(NF / 2);
    }
  };
  auto zc = [&](float2 a, float2 bq, float2 w) {              // Zc[k] from Y[k] = a, Y[NF-k] = bq, w = W_M^k
    const float2 sm = make_float2(a.x + bq.x, a.y - bq.y), df = make_float2(a.x - bq.x, a.y + bq.y);
    const float2 t = make_float2(w.x * df.x + w.y * df.y, w.x * df.y - w.y * df.x);
    return make_float2(sm.x - t.y, sm.y + t.x);
  };
  __syncthreads();
  for (long fc0 = f_lo + HALO - TT; fc0 < bend + pd; fc0 += TT) {
    // the 32-point passes of B leave no room to carry the next chunk's 64 registers of bins through B and C (the form with the loads a
    // phase ahead spills 18-255 registers); the second workgroup of the CU covers this one's load latency instead.  Pulling the next
    // chunk into L2 with one discarded dword per 16-byte piece made it SLOWER (M = 2048: 0.57 -> 0.98 ms): the pre-pass is bound by
    // the number of load instructions and the 16 rows each of them touches, not by latency
    prefetch(fc0);
    // ---- A. Hermitian pre-pass of this lane's two frames
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float2* zf = frm + (2 * fp + h) * FRS;
#pragma unroll
      for (int it = 0; it < NPW; it++) {
        const int k = kq + KQW * it;
        float2 a = h ? make_float2(pa[it].z, pa[it].w) : make_float2(pa[it].x, pa[it].y);
        float2 bq = h ? make_float2(pb[it].z, pb[it].w) : make_float2(pb[it].x, pb[it].y);
        if (k == 0) { a.y = 0.f; bq.y = 0.f; }                 // imaginary parts of bins 0 and M/2 are ignored
        const float2 w = twg[k];                               // W_M^k (L1-resident); W_M^{NF-k} = (-x, y)
        zf[(k / P2) * G::LA + (k % P2)] = zc(a, bq, w);        // input layout of pass 1: n = P2 r + j
        if (k > 0) {
          const int kk = NF - k;
          zf[(kk / P2) * G::LA + (kk % P2)] = zc(bq, a, make_float2(-w.x, w.y));
        }
      }
      if (kq == 0) { const float2 c = h ? make_float2(pmid.z, pmid.w) : make_float2(pmid.x, pmid.y); zf[((NF / 2) / P2) * G::LA + ((NF / 2) % P2)] = zc(c, c, twm); }
    }
    __syncthreads();
    // ---- B. forward FFT of this wave's FPW frames: conj -> positive-exponent passes -> conj
    wave_fft<LOG2M, true>(frm + wave * G::FPW * FRS, twj, lane);
    __syncthreads();
    // ---- C. polyphase + overlap-add: sample groups qd (+ q 256), blocks fc0 + b - pd, b < TT, in sub-chunks of SB blocks so that only
    //         SB + HALO frames of the window are live at a time (the scheduler would otherwise hoist all TT + HALO frame reads to the top)
    const bool emit = fc0 >= f_lo + HALO;
    constexpr int SB = 4;
#pragma unroll
    for (int q = 0; q < QPT; q++) {
      const int d0 = SPL * (qd + q * F_NT);
      int zi[R];
#pragma unroll
      for (int j = 0; j < R; j++) zi[j] = zidx<LOG2M>((d0 + j * D) >> 1);
      auto rd = [&](int fr, int j, float (&o)[SPL]) {
        const float2* zr = frm + fr * FRS + zi[j];
        const float2 z0 = zr[0];
        o[0] = z0.x; o[1] = z0.y;
        if constexpr (SPL == 4) { const float2 z1 = zr[1]; o[2] = z1.x; o[3] = z1.y; }
      };
      float win[R][NWF][SPL];
#pragma unroll
      for (int j = 0; j < R; j++)
#pragma unroll
        for (int w = 0; w < HALO; w++)
#pragma unroll
          for (int dd = 0; dd < SPL; dd++) win[j][w][dd] = hist[q][j][w][dd];
      float gco[SPL][R][F_MT];                                 // g[M-1-(d+jD)+M k], d = d0 + dd: one reversed wide load per (j, k)
#pragma unroll
      for (int j = 0; j < R; j++)
#pragma unroll
        for (int k = 0; k < F_MT; k++) {
          if constexpr (SPL == 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(proto + (M - 4 - (d0 + j * D)) + M * k);
            gco[0][j][k] = g4.w; gco[1][j][k] = g4.z; gco[2][j][k] = g4.y; gco[3][j][k] = g4.x;
          } else {
            const float2 g2 = *reinterpret_cast<const float2*>(proto + (M - 2 - (d0 + j * D)) + M * k);
            gco[0][j][k] = g2.y; gco[1][j][k] = g2.x;
          }
        }
#pragma unroll
      for (int sb = 0; sb < BPG; sb += SB) {
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
          for (int i = 0; i < SB; i++) rd(sb + i, j, win[j][HALO + sb + i]);
        if (emit) {
#pragma unroll
          for (int b = sb; b < sb + SB; b++) {
            const long bglob = fc0 + b - pd;
            float acc[SPL];
#pragma unroll
            for (int dd = 0; dd < SPL; dd++) {
              float a = 0.f;
#pragma unroll
              for (int j = 0; j < R; j++) {
                float sv = 0.f;
#pragma unroll
                for (int k = 0; k < F_MT; k++) sv = fmaf(gco[dd][j][k], win[j][b + HALO - (R - 1 - j) - R * k][dd], sv);
                if (bglob - (R - 1 - j) >= 0) a += sv;         // gsi_ is still zero before block 0 (modulated.cc:574-578,600)
              }
              if (gain > 0.f) a *= gain;
              acc[dd] = a;
            }
            if (bglob >= bt0 && bglob < bend) {
              float* o = os + (bglob - b0) * D + (D - SPL - d0);
              if constexpr (SPL == 4) btk_st<false>(reinterpret_cast<float4*>(o), make_float4(acc[3], acc[2], acc[1], acc[0]));
              else *reinterpret_cast<float2*>(o) = make_float2(acc[1], acc[0]);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // history for the next chunk: the last HALO frames of this one
#pragma unroll
      for (int j = 0; j < R; j++)
#pragma unroll
        for (int w = 0; w < HALO; w++)
#pragma unroll
          for (int dd = 0; dd < SPL; dd++) hist[q][j][w][dd] = win[j][BPG + w][dd];
    }
    __syncthreads();
  }
}

template <int LOG2M, int R>
int launch_fast_synthesis(const btk_fb* fb, const float2* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                          long b0, long bcount, hipStream_t st)
{
  using G = FG<LOG2M>;
  constexpr int HALO = F_MT * R - 1;
  constexpr int NRING = G::TT + HALO + 1;
  size_t lds = sizeof(float2) * ((size_t)NRING * G::FRS + G::NF);
  if (lds > 160 * 1024) return 0;
  auto kern = fast_synthesis_kernel<LOG2M, R>;
  // the register-history form (M >= 1024): 16-byte rows on both sides -- even frame index of every chunk start (b0 + pd even; runs are
  // multiples of TT), Y rows and output blocks 16-byte aligned
  if constexpr (LOG2M >= 10 && R <= 2) {
    const bool wide = !btk_switches().syn_narrow && ((b0 + fb->pd) & 1) == 0 && (T_stride & 1) == 0 && (out_stride & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    if (wide) { kern = fast_synthesis_w_kernel<LOG2M, R>; lds = sizeof(float2) * ((size_t)G::TT * G::FRS + G::NF); }
  }
  // per launch: the attribute is per device, and one process may drive several GPUs (btk_set_device)
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // shorter runs when few streams would leave the chip empty (see synthesis512): multiples of TT, aiming at >= 512 runs
  long srun = ((bcount * S / 512 + G::TT - 1) / G::TT) * G::TT;
  srun = srun < 2 * G::TT ? 2 * G::TT : (srun > F_SRUN ? F_SRUN : srun);
  const unsigned gx = (unsigned)((bcount + srun - 1) / srun);
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)S), dim3(F_NT), lds, st, Y, nframes, T_stride, fb->K, fb->d_proto, fb->d_tw,
                     fb->pd, (float)fb->gain_factor, out, out_stride, b0, bcount, (int)srun);
  BTK_HIP_CHECK(hipGetLastError());
  return 1;
}

template <int LOG2M>
int fast_synthesis_r(const btk_fb* fb, const float2* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                     long b0, long bcount, hipStream_t st)
{
  switch (fb->R) {
    case 1: return launch_fast_synthesis<LOG2M, 1>(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 2: return launch_fast_synthesis<LOG2M, 2>(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 4: return launch_fast_synthesis<LOG2M, 4>(fb, Y, nframes, T_stride, S, out, out_stride, b0, bcount, st);
  }
  return 0;
}

}  // namespace

// returns 1 handled / 0 geometry not covered / <0 error
int btk_fast_analysis_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                          long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->m != F_MT) return 0;
  float2* Xp = static_cast<float2*>(X);
  switch (fb->M) {
    case 256:  return fast_analysis_r<8>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
    case 512:  return fast_analysis_r<9>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
    case 1024: return fast_analysis_r<10>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
    case 2048: return fast_analysis_r<11>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
  }
  return 0;
}

// the same from 16-bit PCM (btk_fb_analysis_i16; M = 512 has its own kernel in fb_analysis512.hip)
int btk_fast_analysis_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                              long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->m != F_MT) return 0;
  float2* Xp = static_cast<float2*>(X);
  switch (fb->M) {
    case 256:  return fast_analysis_r<8, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
    case 512:  return fast_analysis_r<9, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
    case 1024: return fast_analysis_r<10, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
    case 2048: return fast_analysis_r<11, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st);
  }
  return 0;
}

// fused analysis + fixed-weight beamformer for M = 256 from 16-bit PCM (btk_fb_analysis_bf_i16)
int btk_fast_analysis_bf_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                                 int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->m != F_MT || fb->M != 256) return 0;
  return fast_analysis_bf_r<8, short>(fb, pcm, nsamples, pcm_stride, S, N, static_cast<const float2*>(W), per_stream, static_cast<float2*>(Wt_scratch),
                                      static_cast<float2*>(Y), T_stride, t0, tcount, st);
}

// fused analysis + fixed-weight beamformer for M = 256 (the reference's default geometry), m = 4; scratch: Sw K N complex64
int btk_fast_analysis_bf_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                             int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->m != F_MT) return 0;
  const float2* Wp = static_cast<const float2*>(W);
  float2* Wt = static_cast<float2*>(Wt_scratch);
  float2* Yp = static_cast<float2*>(Y);
  switch (fb->M) {
    // M = 1024 / 2048: 32 accumulators + 32-point passes + two pair indices per thread spill ~1 KB of scratch per thread
    // in this form; those geometries stay with the staged pair until the accumulators move to the Z domain
    case 256:  return fast_analysis_bf_r<8>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt, Yp, T_stride, t0, tcount, st);
  }
  return 0;
}

int btk_fast_synthesis_try(const btk_fb* fb, const void* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                           long b0, long bcount, hipStream_t st)
{
  if (fb->m != F_MT) return 0;
  const float2* Yp = static_cast<const float2*>(Y);
  switch (fb->M) {
    case 256:  return fast_synthesis_r<8>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 512:  return fast_synthesis_r<9>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 1024: return fast_synthesis_r<10>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
    case 2048: return fast_synthesis_r<11>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st);
  }
  return 0;
}
