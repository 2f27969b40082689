// mfma16_chain.hip -- rate of v_mfma_f32_16x16x4_f32 as a function of the number of INDEPENDENT accumulator chains a wavefront interleaves
// (NC = 1, 2, 4, 8) and of the wavefronts per SIMD (1, 2): the register-resident Cholesky (csrc/chol_reg.h) updates one complex tile at a
// time = two chains (re, im) of eight dependent instructions each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NC>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed)
{
  f32x4 c[8];
  for (int i = 0; i < 8; i++) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = seed + threadIdx.x, b = seed * 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8 / NC; r++)
#pragma unroll
      for (int i = 0; i < NC; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    asm volatile("" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NC> void run(int wg)
{
  float* out; hipMalloc(&out, sizeof(float) * 64 * wg);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NC>, dim3(wg), dim3(64), 0, 0, out, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NC>, dim3(wg), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)wg / 1024.0;
  printf("%d chains, %.0f wavefront(s) per SIMD: %.2f ns per matrix instruction and SIMD (%.1f TFLOP/s)\n", NC, per_simd,
         ms * 1e6 / (iters * 8.0 * per_simd), (double)wg * iters * 8 * 2048 / ms / 1e9);
  hipFree(out);
}

int main()
{
  for (int wg : {1024, 2048}) { run<1>(wg); run<2>(wg); run<4>(wg); run<8>(wg); }
  return 0;
}
