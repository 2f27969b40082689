// Micro-benchmark (round 3): the 16 x 16 complex transpose between the two radix-16 passes of the 256-point FFT,
//   LDS   : 16 ds_write_b64 [j*17+k] + 16 ds_read_b64 [k*17+j] per lane (what the fused kernel does), against
//   WAVE  : the same transpose without LDS -- a four-stage butterfly over (lane bit, register bit) pairs; with the lane layout
//           lane = (j>>2)<<4 | frame<<2 | (j&3) the two high bits of j are lane bits 4/5, exchanged by ONE v_permlane16_swap /
//           v_permlane32_swap per register pair and dword (gfx950), the two low bits are quad_perm DPP moves + selects.
// Both are checked (lane j, register k must end up with the value lane k, register j started with) and timed alone and with a
// radix-16 butterfly (the FFT's own vector work) between transposes, at 1 and 2 workgroups of 4 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../distant_speech_recognition_amd/csrc/fft_packed.h"

template <int D> __device__ __forceinline__ float quad_xor(float x)
{
  // quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), D == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true));
}

template <int B> __device__ __forceinline__ void stage_quad(f2 (&v)[16], bool hi)
{
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (r & (1 << B)) continue;
    const int r2 = r | (1 << B);
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const float keep0 = c ? v[r].y : v[r].x, keep1 = c ? v[r2].y : v[r2].x;
      const float send = hi ? keep0 : keep1;                     // lanes with the bit clear send register r2, the others register r
      const float recv = quad_xor<1 << B>(send);
      const float n0 = hi ? recv : keep0, n1 = hi ? keep1 : recv;
      if (c) { v[r].y = n0; v[r2].y = n1; } else { v[r].x = n0; v[r2].x = n1; }
    }
  }
}

template <int B> __device__ __forceinline__ void stage_swap(f2 (&v)[16])       // B = 2: lane bit 4 (permlane16), 3: lane bit 5 (permlane32)
{
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (r & (1 << B)) continue;
    const int r2 = r | (1 << B);
#pragma unroll
    for (int c = 0; c < 2; c++) {
      unsigned a = __builtin_bit_cast(unsigned, c ? v[r].y : v[r].x), b = __builtin_bit_cast(unsigned, c ? v[r2].y : v[r2].x);
      // odd rows / upper half of the first operand <-> even rows / lower half of the second.  Here the lanes with the bit SET must
      // give up register r and the lanes with it clear register r2: vdst = r, src = r2
      auto res = (B == 2) ? __builtin_amdgcn_permlane16_swap(a, b, false, false) : __builtin_amdgcn_permlane32_swap(a, b, false, false);
      const float n0 = __builtin_bit_cast(float, (unsigned)res[0]), n1 = __builtin_bit_cast(float, (unsigned)res[1]);
      if (c) { v[r].y = n0; v[r2].y = n1; } else { v[r].x = n0; v[r2].x = n1; }
    }
  }
}

template <int MODE>      // 0 LDS, 1 in-wave; +2: a radix-16 butterfly between transposes
__global__ __launch_bounds__(256) void k(float* out, int iters, int* bad)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr bool WAVE = (MODE & 1) != 0, WORK = (MODE & 2) != 0;
  const int j = WAVE ? (((lane >> 4) << 2) | (lane & 3)) : (lane & 15);
  const int fl = WAVE ? ((lane >> 2) & 3) : (lane >> 4);
  f2* fb = reinterpret_cast<f2*>(smem) + (wave * 4 + fl) * 272;
  f2 v[16];
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = f2{(float)(j * 16 + r + 1000 * fl), (float)(wave + 1)};
  // ---- one checked transpose
  if (WAVE) {
    stage_quad<0>(v, (lane & 1) != 0); stage_quad<1>(v, (lane & 2) != 0); stage_swap<2>(v); stage_swap<3>(v);
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) fb[j * 17 + r] = v[r];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = fb[r * 17 + j];
  }
  int nbad = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) nbad += (v[r].x != (float)(r * 16 + j + 1000 * fl)) || (v[r].y != (float)(wave + 1));
  if (nbad) atomicAdd(bad, nbad);
  __syncthreads();
  const long long c0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (WAVE) {
      stage_quad<0>(v, (lane & 1) != 0); stage_quad<1>(v, (lane & 2) != 0); stage_swap<2>(v); stage_swap<3>(v);
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) fb[j * 17 + r] = v[r];
#pragma unroll
      for (int r = 0; r < 16; r++) v[r] = fb[r * 17 + j];
    }
    if (WORK) {
      dft16q(v);
#pragma unroll
      for (int r = 0; r < 16; r++) v[r] *= 0.25f;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(v[r]));
  }
  const long long c1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[600 * 256] = (float)(c1 - c0) / (float)iters;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) s += v[r].x + v[r].y;
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> void run(const char* name, float* d, int* bad)
{
  const int iters = 2048;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int wg : {1, 2}) {
    hipMemset(bad, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) k<MODE><<<256 * wg, 256, 65536>>>(d, iters, bad);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256 * wg, 256, 65536>>>(d, iters, bad);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, d + 600 * 256, 4, hipMemcpyDeviceToHost);
    int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("%-66s %d WG/CU (%d waves per SIMD): %.3f ms, %.0f shader cycles per iteration and wave%s\n", name, wg, wg, ms, cyc, hb ? "  ** WRONG **" : "  (transpose checked)");
  }
}

int main()
{
  float* d; hipMalloc(&d, 1024 * 256 * sizeof(float));
  int* bad; hipMalloc(&bad, 4);
  run<0>("transpose through LDS (16 ds_write_b64 + 16 ds_read_b64)", d, bad);
  run<1>("transpose inside the wave (quad_perm DPP x2, permlane16/32_swap)", d, bad);
  run<2>("LDS transpose + radix-16 butterfly", d, bad);
  run<3>("in-wave transpose + radix-16 butterfly", d, bad);
  return 0;
}
