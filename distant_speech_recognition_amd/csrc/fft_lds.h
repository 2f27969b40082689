// fft_lds.h -- batched power-of-two complex FFT held in LDS, written for 64-wide wavefronts.
//
// A workgroup of NT threads transforms FRAMES independent sequences of NF complex points that
// live in one LDS array buf[FRAMES][STRIDE] (STRIDE >= NF+1).  Stockham autosort passes of
// radix 4 (plus one radix-2 pass when log2 NF is odd); each pass is "all threads read their
// butterflies into registers -> barrier -> all threads write" so it runs in place.
// Natural order in, natural order out, unnormalised (like gsl_fft_complex_radix2_{forward,backward},
// reference modulated.cc:396,559).  tw[j] = exp(+i 2 pi j / (2 NF)), j < 2 NF, in LDS.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cconjf(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 caddf(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csubf(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by +i (SIGN>0) or -i (SIGN<0)
template <int SIGN> __device__ __forceinline__ float2 cmul_i(float2 a)
{
  return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int LOG2NF, int FRAMES, int STRIDE, int NT, int SIGN>
__device__ __forceinline__ void fft_lds(float2* __restrict__ buf, const float2* __restrict__ tw, int tid)
{
  constexpr int NF = 1 << LOG2NF;
  constexpr int NB4 = FRAMES * (NF / 4);          // radix-4 butterflies per pass
  constexpr int U4 = NB4 / NT;
  static_assert(NB4 % NT == 0 && U4 >= 1, "FRAMES*NF/4 must be a multiple of the workgroup size");
  constexpr int NPASS4 = LOG2NF / 2;

  int Ns = 1;
#pragma unroll
  for (int pass = 0; pass < NPASS4; pass++) {
    float2 v[U4][4];
#pragma unroll
    for (int u = 0; u < U4; u++) {
      const int idx = tid + u * NT;
      const int f = idx / (NF / 4), j = idx % (NF / 4);
      const int k = j & (Ns - 1);
      // twiddle exp(SIGN i 2 pi k q / (4 Ns)) = tw[2 NF k q / (4 Ns)] (conjugated for SIGN<0)
      const int tstep = k * ((2 * NF) / (4 * Ns));
      const float2* p = buf + f * STRIDE + j;
      float2 a0 = p[0], a1 = p[NF / 4], a2 = p[NF / 2], a3 = p[3 * NF / 4];
      if (pass > 0) {
        float2 w1 = tw[tstep], w2 = tw[2 * tstep], w3 = tw[3 * tstep];
        if (SIGN < 0) { w1 = cconjf(w1); w2 = cconjf(w2); w3 = cconjf(w3); }
        a1 = cmulf(a1, w1); a2 = cmulf(a2, w2); a3 = cmulf(a3, w3);
      }
      const float2 s02 = caddf(a0, a2), d02 = csubf(a0, a2);
      const float2 s13 = caddf(a1, a3), d13 = cmul_i<SIGN>(csubf(a1, a3));
      v[u][0] = caddf(s02, s13);
      v[u][1] = caddf(d02, d13);
      v[u][2] = csubf(s02, s13);
      v[u][3] = csubf(d02, d13);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U4; u++) {
      const int idx = tid + u * NT;
      const int f = idx / (NF / 4), j = idx % (NF / 4);
      const int k = j & (Ns - 1);
      float2* q = buf + f * STRIDE + ((j - k) << 2) + k;
      q[0] = v[u][0]; q[Ns] = v[u][1]; q[2 * Ns] = v[u][2]; q[3 * Ns] = v[u][3];
    }
    __syncthreads();
    Ns <<= 2;
  }
  if (LOG2NF & 1) {                                // final radix-2 pass, Ns == NF/2
    constexpr int NB2 = FRAMES * (NF / 2);
    constexpr int U2 = NB2 / NT;
    static_assert(NB2 % NT == 0, "radix-2 pass must tile the workgroup");
    float2 v[U2][2];
#pragma unroll
    for (int u = 0; u < U2; u++) {
      const int idx = tid + u * NT;
      const int f = idx / (NF / 2), j = idx % (NF / 2);
      const float2* p = buf + f * STRIDE + j;      // k == j because Ns == NF/2
      float2 a0 = p[0], a1 = p[NF / 2];
      float2 w = tw[2 * j];                        // exp(i 2 pi j / NF)
      if (SIGN < 0) w = cconjf(w);
      a1 = cmulf(a1, w);
      v[u][0] = caddf(a0, a1);
      v[u][1] = csubf(a0, a1);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U2; u++) {
      const int idx = tid + u * NT;
      const int f = idx / (NF / 2), j = idx % (NF / 2);
      float2* q = buf + f * STRIDE + j;
      q[0] = v[u][0]; q[NF / 2] = v[u][1];
    }
    __syncthreads();
  }
}
