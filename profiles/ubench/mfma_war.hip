// mfma_war.hip -- does a vector instruction that OVERWRITES the B operand of an in-flight 16-pass matrix instruction wait for it?
// v_mfma_f32_32x32x2_f32 (64 cycles on the matrix pipe) x 4 independent accumulators per iteration, one wavefront per SIMD (and two),
//   variant 0: matrix instructions only                                  (the pipe's own rate)
//   variant 1: each preceded by two vector instructions writing ITS B operand register, one register reused for all four
//   variant 2: the same vector work writing four different registers, consumed by the matrix instruction one group later
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed)
{
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a = seed + threadIdx.x, x = seed * 0.5f, y = 0.25f;
  float b0 = x, b1 = x + 1, b2 = x + 2, b3 = x + 3, n0 = b0, n1 = b1, n2 = b2, n3 = b3;
  for (int i = 0; i < iters; i++) {
    if (V == 0) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\tv_mfma_f32_32x32x2_f32 %1, %4, %6, %1\n\t"
                   "v_mfma_f32_32x32x2_f32 %2, %4, %7, %2\n\tv_mfma_f32_32x32x2_f32 %3, %4, %8, %3"
                   : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    } else if (V == 1) {
      asm volatile("v_mul_f32 %5, %6, %7\n\tv_fmac_f32 %5, %6, %4\n\ts_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\t"
                   "v_mul_f32 %5, %7, %7\n\tv_fmac_f32 %5, %6, %4\n\ts_nop 1\n\tv_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n\t"
                   "v_mul_f32 %5, %6, %6\n\tv_fmac_f32 %5, %7, %4\n\ts_nop 1\n\tv_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\t"
                   "v_mul_f32 %5, %7, %6\n\tv_fmac_f32 %5, %7, %4\n\ts_nop 1\n\tv_mfma_f32_32x32x2_f32 %3, %4, %5, %3"
                   : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+v"(a), "+v"(b0) : "v"(x), "v"(y));
    } else {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\tv_mul_f32 %9, %13, %14\n\tv_fmac_f32 %9, %13, %4\n\t"
                   "v_mfma_f32_32x32x2_f32 %1, %4, %6, %1\n\tv_mul_f32 %10, %14, %14\n\tv_fmac_f32 %10, %13, %4\n\t"
                   "v_mfma_f32_32x32x2_f32 %2, %4, %7, %2\n\tv_mul_f32 %11, %13, %13\n\tv_fmac_f32 %11, %14, %4\n\t"
                   "v_mfma_f32_32x32x2_f32 %3, %4, %8, %3\n\tv_mul_f32 %12, %14, %13\n\tv_fmac_f32 %12, %14, %4"
                   : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3)
                   : "v"(x), "v"(y));
      float t;
      t = b0; b0 = n0; n0 = t; t = b1; b1 = n1; n1 = t; t = b2; b2 = n2; n2 = t; t = b3; b3 = n3; n3 = t;
    }
  }
  float s = 0;
  for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 64 + threadIdx.x] = s + b0 + n0;
}

template <int V> void run(const char* name, int wg)
{
  float* out; hipMalloc(&out, sizeof(float) * 64 * wg);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(wg), dim3(64), 0, 0, out, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(wg), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)wg / 1024.0;               // wavefronts per SIMD
  printf("%-44s %5d wavefronts (%.0f per SIMD): %.3f ms -> %.1f ns per matrix instruction and SIMD slot, %.1f TFLOP/s\n", name, wg, per_simd, ms,
         ms * 1e6 / (iters * 4.0 * per_simd), (double)wg * iters * 4 * 4096 / ms / 1e9);
  hipFree(out);
}


// sixteen accumulator blocks (256 AGPRs: one wavefront per SIMD), four A registers x four B registers per step, matrix instructions only
__global__ __launch_bounds__(64) void k16(float* out, int iters, float seed)
{
  f32x16 c[16];
#pragma unroll
  for (int i = 0; i < 16; i++) c[i] = f32x16{0};
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int cb = 0; cb < 4; cb++) {
      const float b = cb == 0 ? b0 : cb == 1 ? b1 : cb == 2 ? b2 : b3;
      c[0 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, c[0 + cb], 0, 0, 0);
      c[4 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, c[4 + cb], 0, 0, 0);
      c[8 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, c[8 + cb], 0, 0, 0);
      c[12 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b, c[12 + cb], 0, 0, 0);
    }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
    for (int r = 0; r < 16; r++) s += c[i][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

void run16(int wg)
{
  float* out; hipMalloc(&out, sizeof(float) * 64 * wg);
  const int iters = 5000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k16, dim3(wg), dim3(64), 0, 0, out, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k16, dim3(wg), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %5d wavefronts: %.3f ms -> %.1f ns per matrix instruction and SIMD, %.1f TFLOP/s\n", "16 accumulator blocks, matrix instructions only", wg, ms,
         ms * 1e6 / (iters * 16.0 * (wg / 1024.0)), (double)wg * iters * 16 * 4096 / ms / 1e9);
  hipFree(out);
}

// the inner loop of wpe_lagprod_kernel in isolation: sixteen accumulator blocks, per two frames four products formed from LDS operands
// MODE bit 0: B operands come from vector instructions (mul + fma + select), bit 1: operands are re-read from LDS every step
template <int MODE>
__global__ __launch_bounds__(64) void k16m(float* out, int iters, float seed)
{
  __shared__ float4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = make_float4(seed + i, seed - i, 0.5f * i, 1.0f);
  __syncthreads();
  f32x16 c[16];
#pragma unroll
  for (int i = 0; i < 16; i++) c[i] = f32x16{0};
  const int lane = threadIdx.x;
  float4 x[4]; float4 y; float a[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = lds[(lane + 17 * q) & 1023]; a[q] = seed + q; }
  y = lds[(lane * 3) & 1023];
  const bool im = lane & 1;
  for (int i = 0; i < iters; i++) {
    if (MODE & 2) {
#pragma unroll
      for (int q = 0; q < 4; q++) x[q] = lds[(lane + 17 * q + 4 * i) & 1023];
      y = lds[(lane * 3 + i) & 1023];
      const float4 aa = lds[(lane + 7 * i) & 1023];
      a[0] = aa.x; a[1] = aa.y; a[2] = aa.z; a[3] = aa.w;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float yx = h ? y.z : y.x, yy = h ? y.w : y.y;
      const float p = im ? -yy : yx, q2 = im ? yx : yy;
#pragma unroll
      for (int cb = 0; cb < 4; cb++) {
        const float xx = h ? x[cb].z : x[cb].x, xy = h ? x[cb].w : x[cb].y;
        float b = (MODE & 1) ? fmaf(xx, p, xy * q2) : xx;
#pragma unroll
        for (int j = 0; j < 4; j++) c[4 * j + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, c[4 * j + cb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(MODE & 2)) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
    for (int r = 0; r < 16; r++) s += c[i][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE> void run16m(const char* name, int wg, int iters)
{
  float* out; hipMalloc(&out, sizeof(float) * 64 * wg);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k16m<MODE>, dim3(wg), dim3(64), 0, 0, out, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k16m<MODE>, dim3(wg), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %5d wavefronts x %5d steps: %8.3f ms -> %.1f TFLOP/s\n", name, wg, iters, ms, (double)wg * iters * 32 * 4096 / ms / 1e9);
  hipFree(out);
}

int main()
{
  for (int it : {2000, 20000}) {
    run16m<0>("lagprod loop: matrix instructions only", 1024, it);
    run16m<1>("lagprod loop: + vector products", 1024, it);
    run16m<2>("lagprod loop: + LDS operand reads", 1024, it);
    run16m<3>("lagprod loop: + both", 1024, it);
  }
  run16(1024); run16(2048); run16(4096);
  for (int wg : {1024, 2048, 4096}) {
    run<0>("matrix instructions only", wg);
    run<1>("B operand written just before, one register", wg);
    run<2>("B operands of the NEXT group written between", wg);
  }
  return 0;
}
