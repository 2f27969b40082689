// Micro-benchmark: issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950, alone and mixed with packed-f32 VALU work in the same wave
// (does the matrix pipe run under the VALU instructions of the same / another wave?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
  f4 acc[16]; f2 a[16]; f2 b = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
  float x = (float)threadIdx.x, y = 1.f;
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; a[i] = f2{(float)i, 1.f}; }
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {          // 16 independent MFMAs
#define X(i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i], 0, 0, 0);
      REP16(X)
#undef X
    } else if (MODE == 1) {   // 16 MFMA + 16 pk_fma interleaved (same wave)
#define X(i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i], 0, 0, 0); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 2) {   // 16 pk_fma only
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 3) {   // waves 0,1 of the SIMD pairing: even workgroups MFMA only, odd workgroups pk_fma only
      if (blockIdx.x & 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP16(X)
#undef X
      } else {
#define X(i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i], 0, 0, 0);
        REP16(X)
#undef X
      }
    } else if (MODE == 4) {   // 16 MFMA + 48 pk_fma interleaved (1:3, the ratio of the fused kernel)
#define X(i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i], 0, 0, 0); asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %3, %3, %1, %2\n\tv_pk_fma_f32 %4, %4, %1, %2" : "+v"(a[i]), "+v"(a[(i + 5) & 15]), "+v"(a[(i + 9) & 15]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 5) {   // 48 pk_fma only
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %3, %3, %1, %2\n\tv_pk_fma_f32 %4, %4, %1, %2" : "+v"(a[i]), "+v"(a[(i + 5) & 15]), "+v"(a[(i + 9) & 15]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w + a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* d, int wg_per_cu)
{
  const int iters = 8192, nb = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) k<MODE><<<nb, 256>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<nb, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s waves/SIMD %d: %.3f ms -> %.1f ns per loop iteration per SIMD-wave-slot set\n", name, wg_per_cu, ms, ms * 1e6 / iters);
}

int main()
{
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2}) {
    run<0>("16 x mfma_4x4x1", d, w);
    run<2>("16 x v_pk_fma_f32", d, w);
    run<1>("16 x (mfma + pk_fma) same wave", d, w);
    run<5>("48 x v_pk_fma_f32", d, w);
    run<4>("16 x (mfma + 3 pk_fma) same wave", d, w);
  }
  run<3>("even WGs: 16 mfma, odd WGs: 16 pk_fma (2 waves/SIMD)", d, 2);
  return 0;
}
