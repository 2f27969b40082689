// fb_analysis512.hip -- analysis bank specialised for the headline geometry M = 512, m = 4 (gfx950).
//
// Same arithmetic contract as analysis_kernel<9,4> in fb_kernels.hip (reference
// modulated/modulated.cc:375-409) with a schedule built around the LDS, which bounds the generic
// kernel:
//   * polyphase FIR with a register sliding window: a thread owns one sample pair index n for all
//     16 frames of the tile; the 16+3R distinct sample pairs it needs are read from LDS once
//     (22 ds_read_b64 instead of 64), the 8 prototype taps stay in VGPRs;
//   * the 256-point complex FFT is two radix-16 passes held in registers (16x16 Cooley-Tukey,
//     inter-pass twiddles W_256^{j k} precomputed per lane); each wavefront owns 4 frames, so the
//     FFT needs NO workgroup barrier -- LDS ops of one wave execute in order;
//   * every LDS exchange uses a 17-column padded 16x16 layout: reads and writes are conflict-free;
//   * the Hermitian post-pass is fused into the final store (X[k] from Z[k], Z[256-k]); bins 0..256
//     leave as 128-byte runs of 16 frames.
//   * a workgroup walks through 16 consecutive tiles of its channel; the PCM span of tile i+1 is
//     fetched into registers while tile i is computed, so HBM latency hides under LDS/VALU work.
// LDS: the PCM span and the 16 FFT frames share one 35 KB region (the polyphase window is pulled
// into registers before the frames overwrite the span) + 4 KB of twiddles -> four workgroups per CU.
#include "btk_internal.h"
#include "fft_lds.h"
#include "fft_packed.h"
#include <cstdlib>
#include <cstdio>
#include <type_traits>

namespace {

constexpr int A_M = 512, A_NF = 256, A_MT = 4, A_TT = 16, A_NT = 256;
constexpr int FRS = 273;               // float2 per FFT frame buffer: 16 rows x 17 (+1: frame stride = 34 banks mod 64)

constexpr int WSTR = 320;              // float4 per channel in the weight-pair table: 257 used, padded to 5 x 64 for 1 KiB LDS-DMA pieces

constexpr int A_RUN = 16;             // consecutive tiles (16 frames each) one workgroup walks through

// Round 3: FOUR workgroups per CU instead of three.  The PCM span of the next tile is no longer carried in registers through the tile
// (24 VGPRs; the kernel now fits 126) but fetched at the top of its own tile, where the other three workgroups of the CU cover its
// latency: 5.70 -> 5.06 ms per 32 streams x 64 channels x 4096 frames on row-padded snapshots (0.57 -> 0.64 of the HBM roofline),
// 1.72 -> 1.35 ms in profiles/fb_ab.py.
constexpr int ANA512_WGS = 4;
constexpr bool ANA512_PREFETCH = false;
// PT = float: the samples as SampleFeature hands them out; PT = short (round 6): the 16-bit PCM they were read from, widened on the
// way into the LDS span (exact: the same bits) -- the bank reads 2 D instead of 4 D bytes per frame and channel of the 4 D + 8 K it moves
template <int R, bool SHARD, typename PT = float>     // R = M / D in {1, 2, 4}; SHARD: only the bins [k0, k1) are stored
__global__ __launch_bounds__(A_NT, ANA512_WGS)
void analysis512_kernel(const PT* __restrict__ pcm, long nsamples, long pcm_stride,
                        const float* __restrict__ proto, const float2* __restrict__ twg,
                        int laN, float gain, int N, int K, float2* __restrict__ X,
                        long T_stride, long t0, long tcount, int ntiles, int nruns, int nchan, int k0, int k1)
{
  // X [S][K][N][T_stride] holds the bins [k0, k1) of the plan (K = k1 - k0; the whole range for an unsharded plan)
  constexpr int D = A_M / R;
  constexpr int SPAN = (A_TT - 1) * D + A_MT * A_M;
  constexpr int FB_BYTES = A_TT * FRS * 8;
  constexpr int REG_U = (SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES;          // PCM span and FFT frames share one region
  constexpr int NV4 = (SPAN / 4 + A_NT - 1) / A_NT;                          // float4 per thread covering the span
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float2* fbuf = reinterpret_cast<float2*>(smem);                             // [16 frames][FRS], aliases xs
  float2* tw = reinterpret_cast<float2*>(smem + REG_U);                       // [257] e^{+j 2 pi k / 512}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware mapping: the runs of one channel stay on one XCD (its L2 serves the PCM halo re-reads)
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int chan = (slot / nruns) * 8 + xcd;
  const int run = slot % nruns;
  if (chan >= nchan) return;
  const int tile_first = run * A_RUN;
  const int tile_end = (tile_first + A_RUN < ntiles) ? tile_first + A_RUN : ntiles;

  for (int j = tid; j <= A_NF; j += A_NT) tw[j] = twg[j];
  float2* twj = tw + (A_NF + 1);                                              // [16 k1][16 j] W_256^{j k1} as (cos, tan): fft_packed.h tw_tangent
  { const float2 t = twg[(2 * (tid & 15) * (tid >> 4)) & 511]; const f2 ct = tw_tangent(t.x, t.y); twj[tid] = make_float2(ct.x, ct.y); }   // tid = k1*16 + j

  const PT* src = pcm + (long)chan * pcm_stride;
  constexpr bool I16 = sizeof(PT) == 2;
  const bool vec_ok = ((pcm_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & (I16 ? 7 : 15)) == 0) &&
                      (!I16 || nsamples < (1L << 30));                 // (the typed loads' byte offset within the row is a 32-bit register)
  float4 pre[NV4];
  __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<PT*>(src), 0, 0x7fffffff, I16 ? BTK_RSRC_I16X4_SSCALED : 0);
  // PCM span of a tile -> registers (global loads stay in flight while the previous tile is computed)
  auto fetch = [&](int tile) {
    const long g0 = (t0 + (long)tile * A_TT + laN + 1) * (long)D - (long)A_MT * A_M;
    if (vec_ok && g0 >= 0 && g0 + SPAN <= nsamples) {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * A_NT) * 4;
        if (l < SPAN) {
          if constexpr (I16) {
            // four samples in 8 bytes, widened by the load unit (typed buffer load, 16_16_16_16 SSCALED): no vector instruction
            // (the form that loads the words raw and converts them -- 2 instructions per sample -- ran 5.43 ms against the
            //  float bank's 5.09: the bank is co-limited by the vector ALU)
            const btk_f4v w = btk_buffer_load_i16x4_f32(rs16, (int)((g0 + l) * 2), 0, 0);
            pre[q] = make_float4(w.x, w.y, w.z, w.w);
          } else {
            pre[q] = *reinterpret_cast<const float4*>(src + g0 + l);
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * A_NT) * 4;
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

  constexpr int G = (R >= 2) ? 2 : 1;
  constexpr int NPG = 256 / G, FPT = A_TT / G, CG = R / G;
  constexpr int NWG = FPT + (A_MT - 1) * R + (G - 1) * CG;
  const int n0 = tid % NPG, fg = tid / NPG;
  float2 h[G][A_MT];                                                          // prototype taps of the pair indices n0 + q NPG
#pragma unroll
  for (int q = 0; q < G; q++)
#pragma unroll
    for (int k = 0; k < A_MT; k++) h[q][k] = *reinterpret_cast<const float2*>(proto + 2 * (n0 + q * NPG) + A_M * k);
  const int s = chan / N, nch = chan % N;
  const long kstride = (long)N * T_stride;
  const float hg = 0.5f * gain;
  const f2 k_hc = f2{0.70710678118654752f, 0.92387953251128674f}, k_t1 = f2{0.41421356237309503f, 0.41421356237309503f};

  if (ANA512_PREFETCH) fetch(tile_first);
  for (int tile = tile_first; tile < tile_end; tile++) {
    const long tt0 = (long)tile * A_TT;
    if (!ANA512_PREFETCH) fetch(tile);
    // ---- phase 1: registers -> LDS span, then start fetching the next tile
#pragma unroll
    for (int q = 0; q < NV4; q++) {
      const int l = (tid + q * A_NT) * 4;
      if (l < SPAN) *reinterpret_cast<float4*>(xs + l) = pre[q];
    }
    __syncthreads();
    if (ANA512_PREFETCH && tile + 1 < tile_end) fetch(tile + 1);

    // ---- phase 2: polyphase with a sliding register window.  With D = M / R the windows of the pair indices n and
    //      n + D/2 are the same LDS words shifted by one frame, so a thread takes G = 2 such indices (n0, n0 + 128)
    //      for half of the tile's frames and reads every word once:
    //      V[i] = xs[(M - 2 - 2 n0 - (G-1) 512/G) + (f0 + i) D]; index n0 + q NPG, frame f0 + g, tap k uses
    //      V[g + R (m-1-k) + (G-1-q) CG]  (= xs[f D + m M - 2 - 2 n - M k], modulated.cc:380-392).
    //      The window is pulled into registers first: after the barrier the span is dead and the FFT frames may
    //      overwrite it.
    {
      float2 win[NWG];
      const float* wbase = xs + (A_M - 2 - 2 * n0 - (G - 1) * (A_M / G)) + fg * FPT * D;
#pragma unroll
      for (int i = 0; i < NWG; i++) win[i] = *reinterpret_cast<const float2*>(wbase + i * D);
      __syncthreads();
      {
#pragma unroll
        for (int q = 0; q < G; q++) {
          const int nn = n0 + q * NPG;
          const int zoff = (nn >> 4) * 17 + (nn & 15);
#pragma unroll
          for (int g = 0; g < FPT; g++) {
            // (h.x x.y, h.y x.x) summed over the taps: one packed multiply-add per tap, the halves of x crossed by op_sel
            f2 po;
#pragma unroll
            for (int k = 0; k < A_MT; k++) {
              const float2 xw = win[g + R * (A_MT - 1 - k) + (G - 1 - q) * CG];
              const f2 x = f2{xw.x, xw.y}, hk = f2{h[q][k].x, h[q][k].y};
              if (k == 0) po = pk_mul_xswap(hk, x);
              else pk_fma_xswap(po, hk, x);
            }
            fbuf[(fg * FPT + g) * FRS + zoff] = make_float2(po.x, po.y);  // frame f -> wave f/4, slot f%4
          }
        }
      }
    }
    __syncthreads();

    // ---- phase 3: wave-private 256-point FFT of 4 frames (no workgroup barrier inside)
    {
      const int fl = lane >> 4, j = lane & 15;
      f2* fb = reinterpret_cast<f2*>(fbuf) + (wave * 4 + fl) * FRS;
      f2 v[16];
#pragma unroll
      for (int r = 0; r < 16; r++) v[r] = fb[r * 17 + j];              // x[16 r + j]
      // round 4: the folded-constant passes of the fused kernel (fft_packed.h: a twiddle c (1 + i tan) is one rotation FMA, its cosine
      // rides in the consuming butterfly; the inter-pass twiddles W_256^{j k1} -- symmetric in lane and register index -- are applied
      // behind the exchange, inside the second pass): 296 instead of 316 packed instructions per wavefront and channel-tile
      dft16t(v, k_hc, k_t1);                                          // A[j][k1]
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) fb[j * 17 + k1] = v[k1];         // row j
      // lane now plays k1 = j: column k1 over rows j'
#pragma unroll
      for (int jp = 0; jp < 16; jp++) v[jp] = fb[jp * 17 + j];
      f2 twr[15];                                                     // (from LDS per tile: fifteen more live registers would cost the fourth workgroup per CU)
#pragma unroll
      for (int k1 = 1; k1 < 16; k1++) { const float2 t = twj[k1 * 16 + j]; twr[k1 - 1] = f2{t.x, t.y}; }
      dft16t_tw(v, twr, k_hc, k_t1);                                  // Z[k1 + 16 k2]
#pragma unroll
      for (int k2 = 0; k2 < 16; k2++) fb[k2 * 17 + j] = v[k2];         // natural order: idx(k) = (k>>4)*17 + (k&15)
    }
    __syncthreads();

    // ---- phase 4: Hermitian post-pass fused into the store: X[k] = E + W^k O
    //      thread (kq = tid>>4, f = tid&15) walks k = kq, kq+16, ..., kq+240; bin 256 is handled by tid < 16
    {
      const int f = tid & 15, kq = tid >> 4;
      const float2* zf = fbuf + f * FRS;
      float2* xo = X + ((long)s * K * N + nch) * T_stride + tt0 + f + (long)kq * kstride;
      auto post = [&](int it) {
        const int k = kq + 16 * it;
        const int kp = (A_NF - k) & 255;
        const float2 zk = zf[it * 17 + kq];
        const float2 zq = zf[(kp >> 4) * 17 + (kp & 15)];                  // Z[256-k], conjugated below
        const float2 e = make_float2(hg * (zk.x + zq.x), hg * (zk.y - zq.y));
        const float2 o = make_float2(hg * (zk.y + zq.y), -hg * (zk.x - zq.x));   // -j (Z[k] - conj Z[256-k]) / 2
        const float2 w = tw[k];
        return make_float2(e.x + (w.x * o.x - w.y * o.y), e.y + (w.x * o.y + w.y * o.x));
      };
      // whole tiles of an unsharded plan store unconditionally: a per-store test costs the store stream -- what bounds this
      // kernel -- two branches each
      if (!SHARD && tt0 + A_TT <= tcount) {
#pragma unroll 4
        for (int it = 0; it < 16; it++) btk_st<true>(xo + (long)(16 * it) * kstride, post(it));      // non-temporal: 5.24 -> 5.17 ms at C0 (profiles/r06_nt_hints.txt)
        if (tid < 16) {                                                    // k = 256: W^256 = -1, partner Z[0]
          const float2 z0 = zf[0];
          X[((long)s * K * N + nch) * T_stride + tt0 + f + (long)A_NF * kstride] = make_float2(gain * (z0.x - z0.y), 0.f);
        }
      } else {
        const bool live = tt0 + f < tcount;
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
          const int k = kq + 16 * it;
          const float2 xv = post(it);
          if (live && (!SHARD || (k >= k0 && k < k1))) xo[(long)(16 * it - (SHARD ? k0 : 0)) * kstride] = xv;
        }
        if (tid < 16 && live && (!SHARD || (A_NF >= k0 && A_NF < k1))) {
          const float2 z0 = zf[0];
          X[((long)s * K * N + nch) * T_stride + tt0 + f + (long)(A_NF - (SHARD ? k0 : 0)) * kstride] = make_float2(gain * (z0.x - z0.y), 0.f);
        }
      }
    }
    __syncthreads();                                         // frames consumed before the next span overwrites them
  }
}

template <int R, typename PT = float>
int launch512(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, float2* X,
              long T_stride, long t0, long tcount, hipStream_t st)
{
  constexpr int D = A_M / R;
  constexpr int SPAN = (A_TT - 1) * D + A_MT * A_M;
  constexpr int FB_BYTES = A_TT * FRS * 8;
  constexpr int REG_U = (SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES;
  const size_t lds = REG_U + sizeof(float2) * (A_NF + 1 + 256);
  const int nchan = S * N;
  const int ntiles = (int)((tcount + A_TT - 1) / A_TT);
  const int nruns = (ntiles + A_RUN - 1) / A_RUN;
  const long nblocks = (long)((nchan + 7) / 8) * nruns * 8;
  const bool shard = !(fb->kx0 == 0 && fb->kx1 == fb->K);
  auto kern = shard ? analysis512_kernel<R, true, PT> : analysis512_kernel<R, false, PT>;
  // per launch: the attribute is per device, and one process may drive several GPUs (btk_set_device)
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float gain = fb->gain_factor > 0 ? (float)fb->gain_factor : 1.0f;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(A_NT), lds, st, pcm, nsamples, pcm_stride, fb->d_proto, fb->d_tw,
                     fb->laN, gain, N, fb->kx1 - fb->kx0, X, T_stride, t0, tcount, ntiles, nruns, nchan, fb->kx0, fb->kx1);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}


// ------------------------------------------------------------------------------------------------
// Fused analysis -> fixed-weight beamformer (SubbandDS/GSC/MVDR::next over OverSampledDFTAnalysisBank
// channels, reference beamformer.cc:1267-1311 + modulated.cc:375-409) for static weights.
// One workgroup = one (stream, 16-frame tile); it walks over the N channels, computes each channel's
// 256-point complex FFT exactly as analysis512_kernel does and accumulates the beamformer sum in
// registers.  The N x K snapshot block never goes to HBM: traffic drops from N(4D+8K)+8K(N+1) to
// 4 D N + 8 K bytes per frame.
//
// The beamformer sum is taken in the Z domain (Z = the 256-point complex FFT of the
// packed real frame) and the Hermitian post-pass runs ONCE per tile instead of once per channel.  With
//   X_n[k] = hg [(1 - j W^k) Z_n[k] + (1 + j W^k) conj Z_n[256-k]],   hg = gain / 2,  W = e^{+j 2 pi / 512},
//   Y[k]   = sum_n conj(w_n[k]) X_n[k] = hg [(1 - j W^k) A[k] + (1 + j W^k) B[k]],
//   A[k]   = sum_n conj(w_n[k]) Z_n[k],        B[k] = sum_n conj(w_n[k]) conj Z_n[(256-k) & 255],
// the FFT lane that holds Z_n[q] (q = j + 16 k2, registers k2 = 0..15) accumulates A[q] and B'[q] = B[(256-q) & 255]
// straight from its registers: no FFT result goes back to LDS, no cross-lane partner is needed per channel.
// Wq [Sw][N][WSTR] float4: entry i < 256 = (w[i], w[(256-i) & 255]), entry 256 = (w[256], 0, 0), the rest 0.
// PT = float: the samples as SampleFeature hands them out (un-normalised floats); PT = short: the 16-bit PCM they were read from
// (feature/feature.cc:265-269), widened in registers -- half the bytes of the kernel's dominant stream, the same values, the same
// bits out (btk_fb_analysis_bf_i16).
template <int R, int VAR, int TT = A_TT, typename PT = float>       // TT frames per workgroup tile (16: four wavefronts; 8: two, four workgroups per CU)
__global__ __launch_bounds__(TT * 16, 2)
void analysis512_bfz_kernel(const PT* __restrict__ pcm, long nsamples, long pcm_stride,
                            const float* __restrict__ proto, const float2* __restrict__ twg,
                            int laN, float gain, int N, int K, const float4* __restrict__ Wq, long w_stream_stride,
                            float2* __restrict__ Y, long T_stride, long t0, long tcount, int ntiles, int tiles_per_xcd, int S,
                            unsigned long long* __restrict__ phase_cycles /* VAR & 512: diagnostics, else unused */)
{
  constexpr int NT = TT * 16, NWAVE = NT / 64;
  constexpr int D = A_M / R;
  constexpr int SPAN = (TT - 1) * D + A_MT * A_M;
  // frame stride 272 = 16 (mod 32) float2: the four frames of a wavefront land on disjoint bank halves in both FFT passes
  // (with 273 the transposing writes of frames fl and fl+1 collide two-way)
  constexpr int FRZ = 272;
  constexpr int FB_BYTES = TT * FRZ * 8;
  constexpr int REG_U = (SPAN * 4 > FB_BYTES) ? SPAN * 4 : FB_BYTES;
  constexpr int NV4 = (SPAN / 4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // PIPE: the PCM span has its own region and the weight pairs are double-buffered, so channel n+1 is staged into LDS
  // while channel n is transformed -- two workgroup barriers per channel instead of four.  Without it (R = 1: the span
  // alone is 38 KB) the span and the FFT frames share one region.
  constexpr bool PIPE = (VAR & 2) && R >= 2;
  // GW (VAR & 4, interior tiles): the polyphase window goes HBM -> registers directly (coalesced 8-byte loads, issued one
  // channel ahead), the PCM span never enters LDS.  The LDS pipe is what bounds this kernel (profiles/ubench/lds_rate:
  // ~64 B/clk per CU for writes, ~120 for 8-byte reads, ~230 for contiguous 16-byte reads; the bfz form moves 23 KiB of
  // writes and 39 KiB of reads per wave and channel = ~630 of the ~680 cycles a channel takes per wave): dropping the
  // span saves its 5.75 KiB of DMA writes and 7.5 KiB of window reads per wave and channel.
  constexpr bool GW = PIPE && (VAR & 4);
  constexpr int FB_OFF = (PIPE && !GW) ? SPAN * 4 : 0;
  constexpr int WQ_OFF = PIPE ? FB_OFF + FB_BYTES : REG_U;
  float* xs = reinterpret_cast<float*>(smem);
  float2* fbuf = reinterpret_cast<float2*>(smem + FB_OFF);
  float4* wq = reinterpret_cast<float4*>(smem + WQ_OFF);                      // [1 or 2][WSTR] weight pairs of a channel

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int xcd = b & 7, j0 = b >> 3;
  const int s = j0 / tiles_per_xcd;
  const int tile = xcd * tiles_per_xcd + j0 % tiles_per_xcd;
  if (s >= S || tile >= ntiles) return;
  const long tt0 = (long)tile * TT;
  const int fl = lane >> 4, j = lane & 15;

  constexpr bool I16 = sizeof(PT) == 2;
  const bool vec_ok = ((pcm_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & (4 * sizeof(PT) - 1)) == 0);
  const long g0 = (t0 + tt0 + laN + 1) * (long)D - (long)A_MT * A_M;
  const bool inb = vec_ok && g0 >= 0 && g0 + SPAN <= nsamples;
  const float4* wts = Wq + (long)s * w_stream_stride;
  float4 pre[NV4];
  float4 wpre[256 / NT];
  float2 w256pre;
  const PT* pcm_e = pcm;                     // the edge path's own copies of the two base pointers: laundered through an
  const float4* wts_e = wts;                 // asm at its entry so that hipcc cannot hoist its loads above the branch
  auto fetch = [&](int n) {
    const PT* src = pcm_e + ((long)s * N + n) * pcm_stride;
    if (inb) {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * NT) * 4;
        if (l < SPAN) {
          if constexpr (I16) {
            const uint2 u = *reinterpret_cast<const uint2*>(src + g0 + l);      // four samples
            pre[q] = make_float4((float)(short)(u.x & 0xffff), (float)((int)u.x >> 16), (float)(short)(u.y & 0xffff), (float)((int)u.y >> 16));
          } else {
            pre[q] = *reinterpret_cast<const float4*>(src + g0 + l);
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        const int l = (tid + q * NT) * 4;
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
    for (int q = 0; q < 256 / NT; q++) wpre[q] = wts_e[(long)n * WSTR + tid + q * NT];
    const float4 t = wts_e[(long)n * WSTR + 256];
    w256pre = make_float2(t.x, t.y);
  };

  // polyphase mapping: with D = M / R the windows of the pair indices n and n + D/2 are the same LDS words shifted by one
  // frame, so a thread takes G = 2 such indices (n0, n0 + 128) for half of the tile's frames and reads every word once
  constexpr int G = (R >= 2 && (VAR & 1)) ? 2 : 1;
  constexpr int NPG = 256 / G, FPT = 16 / G, CG = R / G;            // FPT frames per thread; NT / NPG thread groups cover the tile
  constexpr int NWG = FPT + (A_MT - 1) * R + (G - 1) * CG;
  const int n0 = tid % NPG, fg = tid / NPG;
  float2 h[G][A_MT];
#pragma unroll
  for (int q = 0; q < G; q++)
#pragma unroll
    for (int k = 0; k < A_MT; k++) h[q][k] = *reinterpret_cast<const float2*>(proto + 2 * (n0 + q * NPG) + A_M * k);
  f2 twr[15];                                                                 // W_256^{j k1}, k1 = 1..15
#pragma unroll
  for (int k1 = 1; k1 < 16; k1++) { const float2 t = twg[(2 * j * k1) & 511]; twr[k1 - 1] = (VAR & 128) ? tw_tangent(t.x, t.y) : f2{t.x, t.y}; }
  // TAN (VAR & 128, round 4): folded-constant radix-16 passes (fft_packed.h: a twiddle is one rotation FMA, its cosine rides in
  // the next butterfly): twr holds (cos, tan) and is applied behind the LDS exchange (W_256^{j k1} is symmetric in lane and
  // register index).  296 instead of 316 packed instructions per wavefront and channel; measured -1.0 % (profiles/r04_fused_ab.txt)
  constexpr bool TAN = (VAR & 128) != 0;
  const f2 k_hc = f2{0.70710678118654752f, 0.92387953251128674f}, k_t1 = f2{0.41421356237309503f, 0.41421356237309503f};
  // The taps and twiddles are first used inside the channel loop; without a use in front of it hipcc keeps their
  // s_waitcnt vmcnt(k) inside the loop, where every iteration it also waits for the LDS-DMA of the NEXT channel that
  // was issued a few instructions earlier (vector-memory counters retire in order): the latency the DMA is meant
  // to hide lands in the middle of the FFT.  An empty asm that reads them retires those loads here.
#pragma unroll
  for (int q = 0; q < G; q++)
#pragma unroll
    for (int k = 0; k < A_MT; k++) asm volatile("" : "+v"(h[q][k].x), "+v"(h[q][k].y));
#pragma unroll
  for (int k1 = 0; k1 < 15; k1++) asm volatile("" : "+v"(twr[k1]));
  f2 accA[16], accB[16];
#pragma unroll
  for (int k2 = 0; k2 < 16; k2++) { accA[k2] = f2{0.f, 0.f}; accB[k2] = f2{0.f, 0.f}; }
  float2 acc256 = make_float2(0.f, 0.f);

  // registers -> LDS: PCM span + weight pairs of one channel
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NV4; q++) {
      const int l = (tid + q * NT) * 4;
      if (l < SPAN) *reinterpret_cast<float4*>(xs + l) = pre[q];
    }
#pragma unroll
    for (int q = 0; q < 256 / NT; q++) wq[buf * WSTR + tid + q * NT] = wpre[q];
    if (tid == 0) wq[buf * WSTR + 256] = make_float4(w256pre.x, w256pre.y, 0.f, 0.f);
  };
  // PIPE: LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane-linear destination) of the span and the
  // weight pairs of channel n; issued by asm so that hipcc does not order the FFT's LDS traffic behind it -- the
  // matching s_waitcnt vmcnt(0) sits before the barrier that opens channel n.  Edge tiles (span not inside the
  // recording, or unaligned) go through registers synchronously.
  const unsigned xs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte offset of the dynamic region
  auto glds16s = [&](const void* gbase, unsigned voff, unsigned lds_dst) {       // uniform base (SGPR pair) + 32-bit lane offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
  };
  const int wv = __builtin_amdgcn_readfirstlane(wave);                          // wave index in an SGPR: piece bookkeeping stays scalar
  auto dma = [&](int n, auto fast) {
    if constexpr (decltype(fast)::value) {
      if constexpr (!GW) {
        static_assert(!I16 || !decltype(fast)::value || GW, "the LDS-DMA of the span copies float samples");
        const PT* src = pcm + ((long)s * N + n) * pcm_stride + g0;
        constexpr int NCH = (SPAN * 4 + 1023) / 1024;
        constexpr int NI = (NCH + NWAVE - 1) / NWAVE;
#pragma unroll
        for (int i = 0; i < NI; i++) {
          const int c = wv + NWAVE * i;
          if (c < NCH) {
            const int l = c * 256 + lane * 4;
            if ((SPAN % 256) == 0 || l < SPAN) glds16s(src, (unsigned)l * 4u, xs_lds + c * 1024);
          }
        }
      }
      const float4* wsrc = wts + (long)n * WSTR;
      const unsigned wq_lds = xs_lds + WQ_OFF + (n & 1) * (WSTR * 16);
#pragma unroll
      for (int i = 0; i < (WSTR / 64 + NWAVE - 1) / NWAVE; i++) {
        const int c = wv + NWAVE * i;
        if (c < WSTR / 64) glds16s(wsrc, (unsigned)(c * 64 + lane) * 16u, wq_lds + c * 1024);
      }
    } else {
      fetch(n);
      stage(n & 1);
    }
  };

  // The channel loop exists twice: interior tiles (FAST: the loads of the loop are the asm LDS-DMA and, with GW, the window
  // loads; no edge-path load can be pending in it) and edge tiles (register staging through the span region, guarded
  // loads, compiler-managed waits).  One loop with a runtime branch let hipcc hoist edge-path loads above the branch.
  float2 win[NWG];                            // polyphase window of the channel about to be transformed (GW: loaded a channel ahead)
  auto channels = [&](auto fast) {
  constexpr bool FAST = decltype(fast)::value;
  constexpr bool GWF = GW && FAST;            // edge tiles of the GW form stage the span through the frame region (see below)
  // (the int16 forms without the direct window loads have this one loop only: nothing to hoist above a branch, and the backend
  //  refuses the scalar constraint there -- "illegal VGPR to SGPR copy")
  if constexpr (!FAST && (!I16 || GW)) asm volatile("" : "+s"(pcm_e), "+s"(wts_e));
  // GW: V[i] of channel n straight from HBM / L2 (8-byte loads, 512 contiguous bytes per wave-instruction)
  auto wload = [&](float2 (&win)[NWG], int n) {
    const PT* wsrc = pcm + ((long)s * N + n) * pcm_stride + g0 + (A_M - 2 - 2 * n0 - (G - 1) * (A_M / G)) + fg * FPT * D;
    if constexpr (I16) {
      // 15 four-byte TYPED buffer loads (256 contiguous bytes per wave-instruction): the load unit delivers the two samples as
      // floats (btk_internal.h), so the loop has not one instruction more than the float kernel's -- and moves half its bytes.
      // (A first form loaded the words raw and widened them with 30 v_cvt_f32_i32 per lane and channel: 7 % SLOWER than the float
      // kernel although it drew 350 W less, profiles/r06_fused_i16_forms.txt: the vector ALU is the co-limiter.)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<PT*>(pcm + ((long)s * N + n) * pcm_stride + g0), 0, 0x7fffffff,
                                                                          BTK_RSRC_I16X2_SSCALED);
      const int vo = ((A_M - 2 - 2 * n0 - (G - 1) * (A_M / G)) + fg * FPT * D) * 2;
#pragma unroll
      for (int i = 0; i < NWG; i++) { const btk_f2v t = btk_buffer_load_i16x2_f32(rs, vo, i * D * 2, 0); win[i] = make_float2(t.x, t.y); }
      (void)wsrc;
    } else {
      // (VAR & 2048, measurement: the window loads as non-temporal loads -- every sample is read once, the halo by a neighbour)
#pragma unroll
      for (int i = 0; i < NWG; i++) {
        if constexpr ((VAR & 2048) != 0) { const f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(wsrc + i * D)); win[i] = make_float2(t.x, t.y); }
        else win[i] = *reinterpret_cast<const float2*>(wsrc + i * D);
      }
    }
  };
  // SHARED: the PCM span is staged through registers into the region the FFT frames overwrite (R = 1, and the edge
  // tiles of the GW form): four barriers per channel instead of two
  constexpr bool SHARED = !PIPE || (GW && !FAST);
  // VAR & 512 (diagnostics, profiles/scripts/r02_phase_timing.sh): wave 0 of every workgroup adds up the shader-clock time it
  // spends between the marks below (s_memtime; each mark drains lgkmcnt, which the surrounding code does anyway)
  constexpr bool TIMED = (VAR & 512) != 0;
  constexpr bool SPREAD = (VAR & 8) != 0;
  constexpr bool XSWAP = (VAR & 64) != 0;      // polyphase products with op_sel-crossed halves (no v_pk_mov swap per output)
  // ABL (VAR bits 12-14, only built with -DBTK_FUSED_ABLATE; results are WRONG by design): what the kernel costs without ...
  //   1 the LDS exchange between the FFT passes, 2 the weight-pair reads, 3 the frame writes and first-pass reads,
  //   4 the window loads, 5 the two barriers, 6 the beamformer sums; 7 = window loads always from channel 0 (cache hits)
  constexpr int ABL = (VAR >> 12) & 7;
  long long tm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = TIMED ? clock64() : 0;
  auto mark = [&](int i) { if constexpr (TIMED) { const long long c = clock64(); tm[i] += c - tlast; tlast = c; } };
  auto body = [&](int n, float2 (&win)[NWG]) {
    mark(0);                                                         // (loop overhead / previous accumulate tail)
    constexpr bool W256S = (VAR & 256) != 0 && FAST;
    f2 w256s = f2{0.f, 0.f};
    if constexpr (W256S) {
      // (a plain load would become a VECTOR load: the asm statements of this loop clobber memory, so hipcc cannot call the table
      //  invariant.)  The scalar load lands long before the sums; its s_waitcnt sits in front of the packed FMA that reads the pair
      const float4* wp = wts + (long)n * WSTR + 256;
      asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(w256s) : "s"(wp));
    }
    // ---- phase 1: registers -> LDS (PCM span + weight pairs)
    if (SHARED) stage(PIPE ? (n & 1) : 0);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the LDS-DMA of channel n (and, GW, its window) has landed
    mark(1);                                                         // wait for memory
    if constexpr (ABL != 5) __syncthreads();
    mark(2);                                                         // barrier A
    const int wbuf = PIPE ? (n & 1) : 0;

    // PRIO (VAR & 32768, round 4): the polyphase stage -- 64 packed instructions between the channel's two barriers -- runs at wave priority 1.
    // The two workgroups of a CU are out of phase; a wavefront that is between its barriers holds three others up, one that is in its
    // FFT holds nobody up: letting the short stage win the issue arbitration measured -1.0 ... -1.5 % in 15 of 16 alternating pairs on
    // three boxes (profiles/r04_fused_ab.txt).  Priority around the FFT instead: +3.8 %; from the loop top through the load issue: +1.9 %.
    if constexpr ((VAR & 32768) != 0) __builtin_amdgcn_s_setprio(1);
    // ---- phase 2: polyphase (sliding register window), frames overwrite the span after the barrier.
    //      V[i] = xs[(M - 2 - 2 n0 - (G-1) 512/G) + (f0 + i) D]; index n0 + q NPG, frame f0 + g, tap k uses
    //      V[g + R (m-1-k) + (G-1-q) CG]  (= xs[f D + m M - 2 - 2 n - M k], modulated.cc:380-392)
    {
      if constexpr (!GWF) {
        const float* wbase = xs + (A_M - 2 - 2 * n0 - (G - 1) * (A_M / G)) + fg * FPT * D;
#pragma unroll
        for (int i = 0; i < NWG; i++) win[i] = *reinterpret_cast<const float2*>(wbase + i * D);
        if (SHARED) __syncthreads();                     // the frames overwrite the span
        else __builtin_amdgcn_sched_barrier(0);          // keep the window reads back to back (one LDS latency, not NWG)
      }
      // tap-major order: consecutive FMAs belong to different outputs (no dependent back-to-back packed FMAs)
      // z = (h.x x.y, h.y x.x) summed over the taps: one packed multiply-add per tap with the halves of x crossed by op_sel
      f2 po[G][FPT];
#pragma unroll
      for (int k = 0; k < A_MT; k++)
#pragma unroll
        for (int q = 0; q < G; q++)
#pragma unroll
          for (int g = 0; g < FPT; g++) {
            const float2 xw = win[g + R * (A_MT - 1 - k) + (G - 1 - q) * CG];
            const f2 x = f2{xw.x, xw.y}, hk = f2{h[q][k].x, h[q][k].y};
            if constexpr (XSWAP) {
              if (k == 0) po[q][g] = pk_mul_xswap(hk, x);
              else pk_fma_xswap(po[q][g], hk, x);
            } else {
              po[q][g].x = (k == 0) ? hk.x * x.y : fmaf(hk.x, x.y, po[q][g].x);
              po[q][g].y = (k == 0) ? hk.y * x.x : fmaf(hk.y, x.x, po[q][g].y);
            }
          }
#pragma unroll
      for (int q = 0; q < G; q++) {
        const int nn = n0 + q * NPG;
        const int zoff = (nn >> 4) * 17 + (nn & 15);
#pragma unroll
        for (int g = 0; g < FPT; g++) {
          if constexpr (ABL != 3) fbuf[(fg * FPT + g) * FRZ + zoff] = make_float2(po[q][g].x, po[q][g].y);
          else asm volatile("" :: "v"(po[q][g]));
        }
      }
    }
    if constexpr ((VAR & 32768) != 0) __builtin_amdgcn_s_setprio(0);
    mark(3);                                                         // polyphase (+ LDS window reads when staged)
    if constexpr (ABL != 5) __syncthreads();
    mark(4);                                                         // barrier B
    if (n + 1 < N) {
      if (SHARED) fetch(n + 1);           // lands under phases 3-4
      else {
        dma(n + 1, fast);                 // every window read of channel n is behind the barrier; lands under phases 3-4
        if constexpr (GWF && !SPREAD) wload(win, n + 1);
      }
    }
    // SPREAD: the window loads are unconditional (the last channel re-reads its own window) so that they share a basic
    // block with the FFT, and the scheduling groups at the end of the body interleave them with its VALU work: four
    // wavefronts issuing 15 loads back to back queue up behind the CU's one address path (~80 cycles per load and wave)
    if constexpr (GWF && SPREAD && ABL != 4) wload(win, ABL == 7 ? 0 : (n + 1 < N ? n + 1 : n));     // ABL 7: always channel 0 (cache-resident)

    mark(5);                                                         // issue of the next channel's loads
    // ---- phase 3: wave-private 256-point FFT of 4 frames; the result stays in registers
    f2 v[16];
    const f4* wl = reinterpret_cast<const f4*>(wq) + wbuf * WSTR + j;
    f4 wg[2][4];                                                      // weight pairs, fetched one group of 4 bins ahead
    {
      f2* fb = reinterpret_cast<f2*>(fbuf) + (wave * 4 + fl) * FRZ;
#pragma unroll
      for (int r = 0; r < 16; r++) { if constexpr (ABL != 3) v[r] = fb[r * 17 + j]; else { v[r] = f2{win[r % NWG].x, (float)r}; asm volatile("" : "+v"(v[r])); } }
      if constexpr (TAN) dft16t(v, k_hc, k_t1); else dft16q(v);
      if constexpr (TIMED) { asm volatile("" : "+v"(v[15])); mark(6); }   // first pass (reads + radix-16)
      if constexpr (!TAN) {
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++) v[k1] = cmulv(v[k1], twr[k1 - 1]);
      }
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) { if constexpr (ABL != 1) fb[j * 17 + k1] = v[k1]; }
#pragma unroll
      for (int jp = 0; jp < 16; jp++) { if constexpr (ABL != 1) v[jp] = fb[jp * 17 + j]; else asm volatile("" : "+v"(v[jp])); }
#pragma unroll
      for (int q = 0; q < 4; q++) { if constexpr (ABL == 2 || ABL == 6) { wg[0][q] = f4{1.f, 0.5f, 0.25f, 2.f}; wg[1][q] = wg[0][q]; asm volatile("" : "+v"(wg[0][q]), "+v"(wg[1][q])); } else wg[0][q] = wl[q * 16]; }
      if constexpr (TAN) dft16t_tw(v, twr, k_hc, k_t1); else dft16q(v);   // v[k2] = Z[j + 16 k2]
      if constexpr (TIMED) { asm volatile("" : "+v"(v[15])); mark(7); }   // twiddles, exchange, second pass
    }
    // ---- phase 4: A[q] += conj(w[q]) Z[q],  B'[q] += conj(w[(256-q)&255]) conj(Z[q])
    {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (g < 3) {
#pragma unroll
          for (int q = 0; q < 4; q++) { if constexpr (ABL != 2 && ABL != 6) wg[(g + 1) & 1][q] = wl[((g + 1) * 4 + q) * 16]; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k2 = g * 4 + q;
          const f4 w4 = wg[g & 1][q];
          if constexpr (ABL != 6) {
            acc_conjw_z(accA[k2], w4.xy, v[k2]);
            acc_conjw_conjz(accB[k2], w4.zw, v[k2]);
          } else { accA[k2] += v[k2]; }
        }
      }
      const float r = v[0].x - v[0].y;                                // bin 256 (lanes j == 0): X = gain (Z0.re - Z0.im)
      if constexpr (W256S) {
        // VAR & 256: the channel's bin-256 weight is wave-uniform -- a scalar load issued at the top of the channel, one packed FMA
        // with the SGPR pair (acc += (w.x, -w.y) r) instead of an LDS read and two FMAs
        f2 a2 = f2{acc256.x, acc256.y};
        const f2 rr = f2{r, r};
        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(a2) : "s"(w256s), "v"(rr));
        acc256 = make_float2(a2.x, a2.y);
      } else {
      const float4 w256 = wq[wbuf * WSTR + 256];
      acc256.x = fmaf(w256.x, r, acc256.x);
      acc256.y = fmaf(-w256.y, r, acc256.y);
      }
    }
    if constexpr (GWF && SPREAD) {
      constexpr int PAT = (VAR >> 4) & 1;
#pragma unroll
      for (int i = 0; i < NWG; i++) {
        if constexpr (PAT == 0) {                               // measured 1.411 ms (VAR 15) against 1.463 unspread (VAR 7)
          __builtin_amdgcn_sched_group_barrier(0x080, 3, 0);    // three LDS instructions (anchors: the FFT's data flow fixes their order) ...
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // ... then one window load
        } else {                                                // evenly between the VALU instructions: 1.420 ms (VAR 31)
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
    if constexpr (TIMED) { asm volatile("" : "+v"(accA[15]), "+v"(accB[15])); mark(8); }   // beamformer sums
    if (SHARED) __syncthreads();                                      // frames and weight pairs consumed
  };
  if (SHARED) fetch(0);
  else dma(0, fast);
  if constexpr (GWF) wload(win, 0);
  for (int n = 0; n < N; n++) body(n, win);
  if constexpr (TIMED && FAST) {
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 9; i++) atomicAdd(phase_cycles + i, (unsigned long long)tm[i]);
      atomicAdd(phase_cycles + 9, 1ull);
    }
  }
  };
  if constexpr (PIPE && (GW || !I16)) {
    if (inb) channels(std::true_type{});
    else channels(std::false_type{});
  } else {
    channels(std::false_type{});                // (int16 without the direct window loads: widened while staged through registers)
  }

  if (PIPE) __syncthreads();
  // ---- once per tile: B[k] = B'[(256-k)&255] through the wave's own frame buffers, Hermitian post-pass,
  //      then a transposed store Y[s][k][tt0 .. tt0+15] (128-byte runs per bin)
  {
    const float hg = 0.5f * gain;
    float2* fb = fbuf + (wave * 4 + fl) * FRZ;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) fb[k2 * 17 + j] = make_float2(accB[k2].x, accB[k2].y);
    float2 yv[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
      const int k = j + 16 * k2;
      const int kp = (A_NF - k) & 255;
      const float2 Bk = fb[(kp >> 4) * 17 + (kp & 15)];
      const float2 w = twg[k];
      const float2 c1 = make_float2(1.f + w.y, -w.x), c2 = make_float2(1.f - w.y, w.x);
      const float2 a = make_float2(accA[k2].x, accA[k2].y);
      yv[k2] = make_float2(hg * ((c1.x * a.x - c1.y * a.y) + (c2.x * Bk.x - c2.y * Bk.y)),
                           hg * ((c1.x * a.y + c1.y * a.x) + (c2.x * Bk.y + c2.y * Bk.x)));
    }
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) fb[k2 * 17 + j] = yv[k2];
    if (j == 0) reinterpret_cast<float2*>(wq)[wave * 4 + fl] = make_float2(gain * acc256.x, gain * acc256.y);   // weights are dead
  }
  __syncthreads();
  {
    const int f = tid % TT, kq = tid / TT;                 // NT / TT = 16 bin columns
    if (tt0 + f < tcount) {
      float2* yo = Y + (long)s * K * T_stride + tt0 + f;
      const float2* zf = fbuf + f * FRZ;
#pragma unroll 4
      for (int it = 0; it < 16; it++) {
        // non-temporal stores (round 6): Y is written once and read by another kernel -- 2.2 % faster when the rows are a multiple of
        // 4 KiB apart (a C-ABI caller's contiguous [S][K][T] block with T a power of two), nothing when they are padded
        // (engine.padded_rows); VAR & 65536 = plain stores, for measurement (profiles/r06_fused_nt.txt)
        if constexpr ((VAR & 65536) == 0) __builtin_nontemporal_store(f2{zf[it * 17 + kq].x, zf[it * 17 + kq].y}, reinterpret_cast<f2*>(yo + (long)(kq + 16 * it) * T_stride));
        else yo[(long)(kq + 16 * it) * T_stride] = zf[it * 17 + kq];
      }
      if (kq == 0) yo[(long)A_NF * T_stride] = reinterpret_cast<const float2*>(wq)[f];
    }
  }
}

// W [Sw][K][N] -> Wq [Sw][N][WSTR] float4 (see analysis512_bfz_kernel)
__global__ void pair_weights_kernel(const float2* __restrict__ W, float4* __restrict__ Wq, int K, int N, int Sw)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Sw * N * WSTR) return;
  const int e = (int)(i % WSTR);
  const int n = (int)((i / WSTR) % N);
  const long s = i / ((long)N * WSTR);
  const float2* Ws = W + s * (long)K * N;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e <= 256) {
    const float2 a = Ws[(long)e * N + n];
    const float2 bq = (e < 256) ? Ws[(long)((256 - e) & 255) * N + n] : make_float2(0.f, 0.f);
    o = make_float4(a.x, a.y, bq.x, bq.y);
  }
  Wq[i] = o;
}

template <int R, typename PT = float>
int launch512_bf(const btk_fb* fb, const PT* pcm, long nsamples, long pcm_stride, int S, int N, const float2* W,
                 int per_stream, void* scratch, float2* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  constexpr bool I16 = sizeof(PT) == 2;
  constexpr int D = A_M / R;
  constexpr int SPAN = (A_TT - 1) * D + A_MT * A_M;
  const int K = fb->K;
  const int Sw = per_stream ? S : 1;
  const float gain = fb->gain_factor > 0 ? (float)fb->gain_factor : 1.0f;
  float4* Wq = static_cast<float4*>(scratch);
  // BTK_FUSED_VAR (diagnostics, read once): 1 = register staging with the span and the frames sharing one LDS region (the only
  // form for R = 1, whose 38 KB span leaves no room for a separate region), 3 = LDS-DMA staging of the span,
  // 7 = polyphase window straight from HBM, only frames and weights in LDS, 15 = 7 with the window loads interleaved with the FFT
  // (31: the other interleaving pattern), 79 = 15 with the polyphase products' halves crossed by op_sel, 207 =
  // 79 with the folded-constant radix-16 passes, 463 = 207 with the bin-256 weight of a channel taken from a scalar load,
  // 33231 (default for R = 2) = 463 with the polyphase stage at wave priority 1
  const int var = btk_switches().fused_var >= 0 ? btk_switches().fused_var : (R == 2 ? 33231 : 3);
  const bool pipe = (var & 2) && R >= 2;
  // (int16 samples: the production forms only -- the default kernel of R = 2 (33231) and the staged forms 1 / 3; the other
  //  diagnostic variants are float builds)
  const bool gw = pipe && (var & 4) && R == 2 && (!I16 || ((var & 8) && !(var & 1024)));
  // (TT = 8 -- two wavefronts per workgroup, four workgroups per CU, the same occupancy with less barrier coupling -- measured
  //  1.50 ms against 1.47 for the 16-frame tile: BTK_FUSED_VAR=1031 in profiles/scripts/r02_fused_ab.sh)
  const bool t8 = gw && (var & 1024);
  const int TTv = t8 ? 8 : A_TT;
  const int ntiles = (int)((tcount + TTv - 1) / TTv);
  const int tiles_per_xcd = (ntiles + 7) / 8;
  const long nblocks = (long)8 * tiles_per_xcd * S;
  const int fbz = TTv * 272 * 8;
  const int regz = SPAN * 4 > fbz ? SPAN * 4 : fbz;
  const size_t lds = pipe ? (size_t)(gw ? 0 : SPAN * 4) + fbz + sizeof(float4) * WSTR * 2 : (size_t)regz + sizeof(float4) * WSTR;
  const long nw = (long)Sw * N * WSTR;
  hipLaunchKernelGGL(pair_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, W, Wq, K, N, Sw);
  auto kern = pipe ? analysis512_bfz_kernel<R, 3, A_TT, PT> : analysis512_bfz_kernel<R, 1, A_TT, PT>;
  if constexpr (I16) {
    if (gw) kern = analysis512_bfz_kernel<2, 33231, A_TT, PT>;
  } else {
  if (gw) kern = analysis512_bfz_kernel<2, 7>;
  if (t8) kern = analysis512_bfz_kernel<2, 7, 8>;
  if (gw && !t8 && (var & 8)) kern = (var & 16) ? analysis512_bfz_kernel<2, 31> : ((var & 64) ? ((var & 128) ? ((var & 256) ? ((var & 32768) ? analysis512_bfz_kernel<2, 33231> : analysis512_bfz_kernel<2, 463>) : analysis512_bfz_kernel<2, 207>) : analysis512_bfz_kernel<2, 79>) : analysis512_bfz_kernel<2, 15>);
  }
  // (measurement forms of the default kernel: 2048 = non-temporal window loads (7 % slower: the halo a neighbouring tile re-reads
  //  no longer comes from L2), 65536 = plain instead of non-temporal stores of Y; profiles/r06_fused_nt.txt)
  if constexpr (!I16) if (gw && !t8 && (var & 33231) == 33231) {
    if ((var & 2048) && (var & 65536)) kern = analysis512_bfz_kernel<2, 33231 + 2048 + 65536>;
    else if (var & 2048) kern = analysis512_bfz_kernel<2, 33231 + 2048>;
    else if (var & 65536) kern = analysis512_bfz_kernel<2, 33231 + 65536>;
  }
#ifdef BTK_FUSED_ABLATE
  if constexpr (!I16) if (gw && !t8) switch ((var >> 12) & 7) {
    case 1: kern = analysis512_bfz_kernel<2, 15 + 4096 * 1>; break;
    case 2: kern = analysis512_bfz_kernel<2, 15 + 4096 * 2>; break;
    case 3: kern = analysis512_bfz_kernel<2, 15 + 4096 * 3>; break;
    case 4: kern = analysis512_bfz_kernel<2, 15 + 4096 * 4>; break;
    case 5: kern = analysis512_bfz_kernel<2, 15 + 4096 * 5>; break;
    case 6: kern = analysis512_bfz_kernel<2, 15 + 4096 * 6>; break;
    case 7: kern = analysis512_bfz_kernel<2, 15 + 4096 * 7>; break;
    default: break;
  }
#endif
  unsigned long long* phase = nullptr;
  if constexpr (!I16) if (R == 2 && (var & 512)) {   // diagnostics: per-phase shader cycles of wave 0, printed by every launch
    kern = gw ? analysis512_bfz_kernel<2, 33743> : analysis512_bfz_kernel<2, 515>;   // 33743 = the default form (33231) with the marks
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) BTK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbuf), 16 * sizeof(unsigned long long)));
    BTK_HIP_CHECK(hipMemsetAsync(dbuf, 0, 16 * sizeof(unsigned long long), st));
    phase = dbuf;
  }
  // per launch: the attribute is per device, and one process may drive several GPUs (btk_set_device)
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(TTv * 16), lds, st, pcm, nsamples, pcm_stride, fb->d_proto, fb->d_tw,
                     fb->laN, gain, N, K, Wq, per_stream ? (long)N * WSTR : 0L, Y, T_stride, t0, tcount, ntiles, tiles_per_xcd, S, phase);
  BTK_HIP_CHECK(hipGetLastError());
  if (phase) {
    unsigned long long h[10];
    BTK_HIP_CHECK(hipMemcpyAsync(h, phase, sizeof(h), hipMemcpyDeviceToHost, st));
    BTK_HIP_CHECK(hipStreamSynchronize(st));
    static const char* names[9] = {"loop", "wait_memory", "barrier_A", "polyphase", "barrier_B", "issue_loads", "fft_pass1", "fft_pass2", "beamformer_sums"};
    double tot = 0; for (int i = 0; i < 9; i++) tot += (double)h[i];
    fprintf(stderr, "fused kernel phases (wave 0 of %llu interior workgroups, %d channels): %.0f cycles per channel:", h[9], N, tot / (h[9] ? h[9] : 1) / N);
    for (int i = 0; i < 9; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * h[i] / (tot > 0 ? tot : 1));
    fprintf(stderr, "\n");
  }
  return BTK_OK;
}


// ------------------------------------------------------------------------------------------------
// Synthesis bank specialised for M = 512, m = 4 (reference modulated/modulated.cc:553-612).
// One workgroup walks through S_RUN consecutive output blocks of one stream.  The real sequences
// v_f = Re FFT_fwd(Y_f) live in a ring of 16 + m R - 1 frame buffers in LDS (the m R - 1 frames of
// history are computed once per run, not once per tile); every iteration adds 16 frames:
//   A. Hermitian pre-pass straight from Y[s][k][t] (128-byte runs of 16 frames per bin),
//   B. wave-private 256-point FFT (two in-register radix-16 passes; forward sign through conjugation),
//   C. polyphase + overlap-add with a register window: thread d reads each of the 16 + m R - 1 frames
//      once at i = d and i = d + jD and produces 16 output samples; float32 running sum in the reference's order.
constexpr int S_RUN = 256;

template <int R>
__global__ __launch_bounds__(A_NT, 2)
void synthesis512_kernel(const float2* __restrict__ Y, long nframes, long T_stride, int K,
                         const float* __restrict__ proto, const float2* __restrict__ twg,
                         int pd, float gain, float* __restrict__ out, long out_stride, long b0, long bcount, int srun)
{
  constexpr int D = A_M / R;
  constexpr int HALO = A_MT * R - 1;
  constexpr int NRING = 16 + HALO + 1;
  constexpr int DPT = (D + A_NT - 1) / A_NT;                  // output samples (d) per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* ring = reinterpret_cast<float2*>(smem);             // [NRING][FRS]
  float2* tw = ring + NRING * FRS;                            // [257]
  float2* twj = tw + (A_NF + 1);                              // [256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.y;
  const long bt0 = b0 + (long)blockIdx.x * srun;              // first block of this run
  const long bend = (bt0 + srun < b0 + bcount) ? bt0 + srun : b0 + bcount;
  const float2* Ys = Y + (long)s * K * T_stride;
  float* os = out + (long)s * out_stride;

  for (int j = tid; j <= A_NF; j += A_NT) tw[j] = twg[j];
  twj[tid] = twg[(2 * (tid & 15) * (tid >> 4)) & 511];

  // synthesis taps of this thread: g[M-1-(d+jD)+M k]
  float gco[DPT][R][A_MT];
#pragma unroll
  for (int q = 0; q < DPT; q++) {
    const int d = tid + q * A_NT;
#pragma unroll
    for (int j = 0; j < R; j++)
#pragma unroll
      for (int k = 0; k < A_MT; k++)
        gco[q][j][k] = (d < D) ? proto[(A_M - 1 - (d + j * D)) + A_M * k] : 0.f;
  }
  __syncthreads();

  const long f_lo = bt0 + pd - HALO;                          // oldest frame the run needs
  // A. Hermitian pre-pass: Zc[k] = (Y[k] + conj Y[256-k]) + j W^-k (Y[k] - conj Y[256-k]).  Thread (fi = tid & 15,
  //    kq = tid >> 4) loads the bin pairs (k, 256 - k), k = kq + 16 it, it < 8, of frame fi ONCE and forms both Zc[k]
  //    and Zc[256-k] (bin 128 pairs with itself: the kq == 0 threads).  The loads of chunk c+1 are issued before the
  //    FFT and the overlap-add of chunk c: their HBM latency was the longest stretch of a chunk.
  const int fi = tid & 15, kq = tid >> 4;
  float2 pa[8], pb[8], p128 = make_float2(0.f, 0.f);
  auto prefetch = [&](long fc0) {
    const long f = fc0 + fi;
    const bool fok = f >= f_lo && f >= 0 && f < nframes;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int k = kq + 16 * it;
      pa[it] = fok ? Ys[(long)k * T_stride + f] : make_float2(0.f, 0.f);
      pb[it] = fok ? Ys[(long)(A_NF - k) * T_stride + f] : make_float2(0.f, 0.f);
    }
    if (kq == 0) p128 = fok ? Ys[(long)128 * T_stride + f] : make_float2(0.f, 0.f);
  };
  auto zc = [&](float2 a, float2 bq, int k) {                 // Zc[k] from Y[k] = a, Y[256-k] = bq
    const float2 sm = make_float2(a.x + bq.x, a.y - bq.y), df = make_float2(a.x - bq.x, a.y + bq.y);
    const float2 w = tw[k];
    const float2 t = make_float2(w.x * df.x + w.y * df.y, w.x * df.y - w.y * df.x);     // conj(W^k) * df
    return make_float2(sm.x - t.y, sm.y + t.x);
  };
  prefetch(f_lo + HALO - 16);
  // ring slot of frame f is (f - f_lo) mod NRING; s0 = slot of the chunk's first frame, carried from chunk to chunk (a 64-bit
  // modulo by 24 per ring access was a thousand scalar instructions per chunk)
  auto wrap = [](int v) { return v >= NRING ? v - NRING : (v < 0 ? v + NRING : v); };
  int s0 = wrap(HALO - 16 + NRING);                              // (fc0 - f_lo) mod NRING for the history chunk, fc0 - f_lo = HALO - 16 < 0
  // chunk c covers frames fc0 .. fc0+15; chunk -1 is the history (only its last HALO frames matter)
  for (long fc0 = f_lo + HALO - 16; fc0 < bend + pd; fc0 += 16, s0 = wrap(s0 + 16)) {
    {
      const long f = fc0 + fi;
      const int slot = wrap(s0 + fi);
      float2* zf = ring + slot * FRS;
      if (f >= f_lo) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int k = kq + 16 * it;
          float2 a = pa[it], bq = pb[it];
          if (k == 0) { a.y = 0.f; bq.y = 0.f; }                 // imaginary parts of bins 0 and M/2 are ignored
          zf[it * 17 + kq] = zc(a, bq, k);
          if (k > 0) {
            const int kk = A_NF - k;                             // 144 .. 255
            zf[(kk >> 4) * 17 + (kk & 15)] = zc(bq, a, kk);
          }
        }
        if (kq == 0) zf[8 * 17] = zc(p128, p128, 128);
      }
    }
    __syncthreads();
    if (fc0 + 16 < bend + pd) prefetch(fc0 + 16);             // lands under B and C
    // ---- B. forward FFT of the 16 new frames (4 per wavefront): conj -> positive-exponent passes -> conj
    {
      const int fl = lane >> 4, j = lane & 15;
      const long f = fc0 + wave * 4 + fl;
      if (f >= f_lo) {
        const int slot = wrap(s0 + wave * 4 + fl);
        f2* fb = reinterpret_cast<f2*>(ring) + slot * FRS;
        const f2* twq = reinterpret_cast<const f2*>(twj);
        f2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { const f2 z = fb[r * 17 + j]; v[r] = f2{z.x, -z.y}; }
        dft16q(v);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++) v[k1] = cmulv(v[k1], twq[k1 * 16 + j]);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) fb[j * 17 + k1] = v[k1];
#pragma unroll
        for (int jp = 0; jp < 16; jp++) v[jp] = fb[jp * 17 + j];
        dft16q(v);
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) fb[k2 * 17 + j] = f2{v[k2].x, -v[k2].y};     // z[n]: v[2n] = Re, v[2n+1] = Im
      }
    }
    __syncthreads();
    // ---- C. polyphase + overlap-add for the 16 blocks whose newest frame is in this chunk
    if (fc0 >= f_lo + HALO) {
#pragma unroll
      for (int q = 0; q < DPT; q++) {
        const int d = tid + q * A_NT;
        if (d < D) {
          // window: frames fc0-HALO .. fc0+15 at i = d + jD
          float win[R][16 + HALO];
#pragma unroll
          for (int j = 0; j < R; j++) {
            const int i = d + j * D;
            const int zi = ((i >> 1) >> 4) * 17 + ((i >> 1) & 15);
#pragma unroll
            for (int wdx = 0; wdx < 16 + HALO; wdx++) {
              const int slot = wrap(s0 - HALO + wdx);          // frame fc0 - HALO + wdx
              const float2 zz = ring[slot * FRS + zi];
              win[j][wdx] = (i & 1) ? zz.y : zz.x;
            }
          }
#pragma unroll
          for (int bb = 0; bb < 16; bb++) {
            const long bglob = fc0 + bb - pd;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < R; j++) {
              // s_{f-(R-1-j)}[d + jD] = sum_k g[...] v_{f-(R-1-j)-Rk}[d + jD];  window index of frame f' is f' - fc0 + HALO
              float sv = 0.f;
#pragma unroll
              for (int k = 0; k < A_MT; k++)
                sv = fmaf(gco[q][j][k], win[j][bb + HALO - (R - 1 - j) - R * k], sv);
              if (bglob - (R - 1 - j) >= 0) acc += sv;       // gsi_ is still zero before block 0 (modulated.cc:574-578,600)
            }
            if (gain > 0.f) acc *= gain;
            if (bglob >= bt0 && bglob < bend) os[(bglob - b0) * D + (D - 1 - d)] = acc;
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---- round 3: the same schedule with wide memory instructions (R = 1, 2; aligned launches).  s_memtime marks on the kernel above
// (profiles/r03_synthesis_phases.txt) showed 18 800 cycles per 16-frame chunk: 9 000 in the overlap-add (46 ds_read_b64 of which
// each lane used half, sixteen 4-byte stores per lane), 3 300 issuing the 17 strided 8-byte loads of the next chunk, 3 000 in the
// FFT, 2 400 in the pre-pass.  Here
//   * the overlap-add lane owns FOUR consecutive output samples of BPG consecutive blocks: its window is read with ds_read2_b64
//     (every byte used), and a block leaves as one 16-byte store per lane -- 1 KiB per wave-instruction, a quarter of the stores;
//   * the pre-pass lane owns TWO consecutive frames of a bin pair: 16-byte loads, half the load instructions;
//   * the ring has 32 slots (a mask instead of compare-and-subtract chains), the inter-pass twiddles live in registers.
// Arithmetic and summation order per output sample are those of the kernel above (the two agree to 2 ulp: the compiler contracts
// the multiply-adds of the two bodies differently).  C0 bench launch: 0.156 -> 0.123 ms (0.32 -> 0.41 of the HBM roofline);
// placing the next chunk's loads one by one between the FFT's LDS operations (as the fused analysis kernel does) measured 0.128 ms
// and was not kept (profiles/r03_synthesis_ab.txt).
template <int R>
__global__ __launch_bounds__(A_NT, 2)
void synthesis512w_kernel(const float2* __restrict__ Y, long nframes, long T_stride, int K,
                          const float* __restrict__ proto, const float2* __restrict__ twg,
                          int pd, float gain, float* __restrict__ out, long out_stride, long b0, long bcount, int srun)
{
  constexpr int D = A_M / R;
  constexpr int HALO = A_MT * R - 1;
  constexpr int NRING = 32;
  constexpr int NQ = D / 4, NBG = A_NT / NQ, BPG = 16 / NBG;   // sample quads per block, block groups, blocks per group
  constexpr int NWF = BPG + HALO;                              // frames in a lane's window
  static_assert(NRING >= 16 + HALO + 1 && NBG * BPG == 16, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* ring = reinterpret_cast<float2*>(smem);             // [NRING][FRS]
  float2* tw = ring + NRING * FRS;                            // [257]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.y;
  const long bt0 = b0 + (long)blockIdx.x * srun;
  const long bend = (bt0 + srun < b0 + bcount) ? bt0 + srun : b0 + bcount;
  const float2* Ys = Y + (long)s * K * T_stride;
  float* os = out + (long)s * out_stride;

  for (int j = tid; j <= A_NF; j += A_NT) tw[j] = twg[j];
  const int fl = lane >> 4, jj = lane & 15;
  f2 twr[15];                                                 // W_256^{j k1}, k1 = 1..15
#pragma unroll
  for (int k1 = 1; k1 < 16; k1++) { const float2 t = twg[(2 * jj * k1) & 511]; twr[k1 - 1] = tw_tangent(t.x, t.y); }   // (cos, tan): folded-constant passes below
  const f2 k_hc = f2{0.70710678118654752f, 0.92387953251128674f}, k_t1 = f2{0.41421356237309503f, 0.41421356237309503f};

  const int dq = tid % NQ, bg = tid / NQ, d0 = 4 * dq;
  float gco[4][R][A_MT];                                      // g[M-1-(d+jD)+M k], d = d0 + dd
#pragma unroll
  for (int dd = 0; dd < 4; dd++)
#pragma unroll
    for (int j = 0; j < R; j++)
#pragma unroll
      for (int k = 0; k < A_MT; k++) gco[dd][j][k] = proto[(A_M - 1 - (d0 + dd + j * D)) + A_M * k];
  __syncthreads();

  const long f_lo = bt0 + pd - HALO;
  // A. pre-pass lane: frames fc0 + 2 fp, + 1; bin pairs (k, 256 - k), k = kq + 32 it
  const int fp = tid & 7, kq = tid >> 3;
  float4 pa[4], pb[4], p128 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto prefetch = [&](long fc0) {
    const long fA = fc0 + 2 * fp;
    const bool okA = fA >= f_lo && fA >= 0 && fA < nframes, okB = fA + 1 >= f_lo && fA + 1 >= 0 && fA + 1 < nframes;
    if (okA && okB) {                                          // 16-byte loads: fA is even and the rows are 16-byte aligned (launch check)
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int k = kq + 32 * it;
        pa[it] = btk_ld<false>(reinterpret_cast<const float4*>(Ys + (long)k * T_stride + fA));
        pb[it] = btk_ld<false>(reinterpret_cast<const float4*>(Ys + (long)(A_NF - k) * T_stride + fA));
      }
      if (kq == 0) p128 = *reinterpret_cast<const float4*>(Ys + (long)128 * T_stride + fA);
    } else {
      auto ld = [&](int k) {
        const float2 a = okA ? Ys[(long)k * T_stride + fA] : make_float2(0.f, 0.f);
        const float2 b = okB ? Ys[(long)k * T_stride + fA + 1] : make_float2(0.f, 0.f);
        return make_float4(a.x, a.y, b.x, b.y);
      };
#pragma unroll
      for (int it = 0; it < 4; it++) { const int k = kq + 32 * it; pa[it] = ld(k); pb[it] = ld(A_NF - k); }
      if (kq == 0) p128 = ld(128);
    }
  };
  auto zc = [&](float2 a, float2 bq, int k) {                 // Zc[k] from Y[k] = a, Y[256-k] = bq
    const float2 sm = make_float2(a.x + bq.x, a.y - bq.y), df = make_float2(a.x - bq.x, a.y + bq.y);
    const float2 w = tw[k];
    const float2 t = make_float2(w.x * df.x + w.y * df.y, w.x * df.y - w.y * df.x);     // conj(W^k) * df
    return make_float2(sm.x - t.y, sm.y + t.x);
  };
  prefetch(f_lo + HALO - 16);
  int s0 = (HALO - 16) & (NRING - 1);                         // ring slot of the chunk's first frame: (fc0 - f_lo) mod 32
  for (long fc0 = f_lo + HALO - 16; fc0 < bend + pd; fc0 += 16, s0 = (s0 + 16) & (NRING - 1)) {
    // ---- A. Hermitian pre-pass of the two frames of this lane
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const long f = fc0 + 2 * fp + h;
      if (f >= f_lo) {
        float2* zf = ring + ((s0 + 2 * fp + h) & (NRING - 1)) * FRS;
#pragma unroll
        for (int it = 0; it < 4; it++) {
          const int k = kq + 32 * it;
          float2 a = h ? make_float2(pa[it].z, pa[it].w) : make_float2(pa[it].x, pa[it].y);
          float2 bq = h ? make_float2(pb[it].z, pb[it].w) : make_float2(pb[it].x, pb[it].y);
          if (k == 0) { a.y = 0.f; bq.y = 0.f; }               // imaginary parts of bins 0 and M/2 are ignored
          zf[(k >> 4) * 17 + (k & 15)] = zc(a, bq, k);
          if (k > 0) {
            const int kk = A_NF - k;                           // 129 .. 255
            zf[(kk >> 4) * 17 + (kk & 15)] = zc(bq, a, kk);
          }
        }
        if (kq == 0) { const float2 c = h ? make_float2(p128.z, p128.w) : make_float2(p128.x, p128.y); zf[8 * 17] = zc(c, c, 128); }
      }
    }
    __syncthreads();
    if (fc0 + 16 < bend + pd) prefetch(fc0 + 16);             // lands under B and C
    // ---- B. forward FFT of the 16 new frames (4 per wavefront): conj -> positive-exponent passes -> conj
    {
      const long f = fc0 + wave * 4 + fl;
      if (f >= f_lo) {
        f2* fb = reinterpret_cast<f2*>(ring) + ((s0 + wave * 4 + fl) & (NRING - 1)) * FRS;
        f2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { const f2 z = fb[r * 17 + jj]; v[r] = f2{z.x, -z.y}; }
        dft16t(v, k_hc, k_t1);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) fb[jj * 17 + k1] = v[k1];
#pragma unroll
        for (int jp = 0; jp < 16; jp++) v[jp] = fb[jp * 17 + jj];
        dft16t_tw(v, twr, k_hc, k_t1);                                  // (inter-pass twiddles behind the exchange, fft_packed.h)
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) fb[k2 * 17 + jj] = f2{v[k2].x, -v[k2].y};     // z[n]: v[2n] = Re, v[2n+1] = Im
      }
    }
    __syncthreads();
    // ---- C. polyphase + overlap-add: samples d0 .. d0+3 of the blocks fc0 + bg BPG + b - pd, b < BPG
    if (fc0 >= f_lo + HALO) {
      // window: frames fc0 - HALO + bg BPG + w, w < NWF, at i = d0 + dd + j D  ->  two float2 at (d0 + j D) / 2
      float win[R][NWF][4];
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int h = (d0 + j * D) >> 1;
        const int zi = (h >> 4) * 17 + (h & 15);
#pragma unroll
        for (int w = 0; w < NWF; w++) {
          const float2* zr = ring + ((s0 - HALO + bg * BPG + w) & (NRING - 1)) * FRS + zi;
          const float2 z0 = zr[0], z1 = zr[1];
          win[j][w][0] = z0.x; win[j][w][1] = z0.y; win[j][w][2] = z1.x; win[j][w][3] = z1.y;
        }
      }
#pragma unroll
      for (int b = 0; b < BPG; b++) {
        const long bglob = fc0 + bg * BPG + b - pd;
        float acc[4];
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < R; j++) {
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < A_MT; k++) sv = fmaf(gco[dd][j][k], win[j][b + HALO - (R - 1 - j) - R * k][dd], sv);
            if (bglob - (R - 1 - j) >= 0) a += sv;             // gsi_ is still zero before block 0 (modulated.cc:574-578,600)
          }
          if (gain > 0.f) a *= gain;
          acc[dd] = a;
        }
        if (bglob >= bt0 && bglob < bend)
          btk_st<true>(reinterpret_cast<float4*>(os + (bglob - b0) * D + (D - 4 - d0)), make_float4(acc[3], acc[2], acc[1], acc[0]));
      }
    }
    __syncthreads();
  }
}

template <int R>
int launch_syn512(const btk_fb* fb, const float2* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                  long b0, long bcount, hipStream_t st)
{
  constexpr int HALO = A_MT * R - 1;
  constexpr int NRING = 16 + HALO + 1;
  size_t lds = sizeof(float2) * ((size_t)NRING * FRS + A_NF + 1 + 256);
  auto kern = synthesis512_kernel<R>;
  // the wide form needs 16-byte rows on both sides: even frame index of every chunk start (b0 + pd even; runs are multiples of 16),
  // Y rows and output blocks 16-byte aligned
  const bool wide = R <= 2 && !btk_switches().syn_narrow && ((b0 + fb->pd) & 1) == 0 && (T_stride & 1) == 0 && (out_stride & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if constexpr (R <= 2) {
    if (wide) { kern = synthesis512w_kernel<R>; lds = sizeof(float2) * ((size_t)32 * FRS + A_NF + 1); }
  }
  // per launch: the attribute is per device, and one process may drive several GPUs (btk_set_device)
  BTK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // run length: S_RUN blocks amortise the m R - 1 frame ring priming best, but few streams need shorter runs to fill
  // the chip (a single stream of 4096 blocks would be 32 workgroups); multiples of the 16-frame chunk, >= 512 runs
  long srun = ((bcount * S / 512 + 15) / 16) * 16;
  srun = srun < 16 ? 16 : (srun > S_RUN ? S_RUN : srun);
  const unsigned gx = (unsigned)((bcount + srun - 1) / srun);
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)S), dim3(A_NT), lds, st, Y, nframes, T_stride, fb->K, fb->d_proto, fb->d_tw,
                     fb->pd, (float)fb->gain_factor, out, out_stride, b0, bcount, (int)srun);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

}  // namespace

// returns 1 if handled, 0 if the geometry is not covered (caller falls back to the generic kernel), <0 on error
int btk_analysis512_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                        long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->M != A_M || fb->m != A_MT) return 0;
  float2* Xp = static_cast<float2*>(X);
  int rc;
  switch (fb->R) {
    case 1: rc = launch512<1>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st); break;
    case 2: rc = launch512<2>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st); break;
    case 4: rc = launch512<4>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st); break;
    default: return 0;
  }
  return rc == BTK_OK ? 1 : rc;
}

// the staged bank from 16-bit PCM (btk_fb_analysis_i16)
int btk_analysis512_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                            long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->M != A_M || fb->m != A_MT) return 0;
  float2* Xp = static_cast<float2*>(X);
  int rc;
  switch (fb->R) {
    case 1: rc = launch512<1, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st); break;
    case 2: rc = launch512<2, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st); break;
    case 4: rc = launch512<4, short>(fb, pcm, nsamples, pcm_stride, S, N, Xp, T_stride, t0, tcount, st); break;
    default: return 0;
  }
  return rc == BTK_OK ? 1 : rc;
}

// Fused analysis + fixed-weight beamformer; returns 1 if handled, 0 if the geometry is not covered, <0 on error
int btk_analysis512_bf_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                           int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->M != A_M || fb->m != A_MT) return 0;
  const float2* Wp = static_cast<const float2*>(W);
  void* Wt = Wt_scratch;
  float2* Yp = static_cast<float2*>(Y);
  int rc;
  switch (fb->R) {
    case 1: rc = launch512_bf<1>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt, Yp, T_stride, t0, tcount, st); break;
    case 2: rc = launch512_bf<2>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt, Yp, T_stride, t0, tcount, st); break;
    case 4: rc = launch512_bf<4>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt, Yp, T_stride, t0, tcount, st); break;
    default: return 0;
  }
  return rc == BTK_OK ? 1 : rc;
}

// the same from 16-bit PCM (btk_fb_analysis_bf_i16)
int btk_analysis512_bf_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                               int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st)
{
  if (fb->M != A_M || fb->m != A_MT) return 0;
  const float2* Wp = static_cast<const float2*>(W);
  float2* Yp = static_cast<float2*>(Y);
  int rc;
  switch (fb->R) {
    case 1: rc = launch512_bf<1, short>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt_scratch, Yp, T_stride, t0, tcount, st); break;
    case 2: rc = launch512_bf<2, short>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt_scratch, Yp, T_stride, t0, tcount, st); break;
    case 4: rc = launch512_bf<4, short>(fb, pcm, nsamples, pcm_stride, S, N, Wp, per_stream, Wt_scratch, Yp, T_stride, t0, tcount, st); break;
    default: return 0;
  }
  return rc == BTK_OK ? 1 : rc;
}

// Specialised synthesis (M = 512, m = 4); returns 1 if handled, 0 if the geometry is not covered, <0 on error
int btk_synthesis512_try(const btk_fb* fb, const void* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                         long b0, long bcount, hipStream_t st)
{
  if (fb->M != A_M || fb->m != A_MT) return 0;
  const float2* Yp = static_cast<const float2*>(Y);
  int rc;
  switch (fb->R) {
    case 1: rc = launch_syn512<1>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st); break;
    case 2: rc = launch_syn512<2>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st); break;
    case 4: rc = launch_syn512<4>(fb, Yp, nframes, T_stride, S, out, out_stride, b0, bcount, st); break;
    default: return 0;
  }
  return rc == BTK_OK ? 1 : rc;
}
