// valu_dep_chain.hip -- what one wavefront's serial float32 recurrence costs on gfx950 (the csvdc QR iteration, csrc/svd_linpack.hip,
// is such a chain): cycles per instruction of a DEPENDENT v_fma_f32 / v_mul_f32 chain with 1, 2 and 4 wavefronts on a SIMD, and of
// two independent chains in one wavefront.  hipcc --offload-arch=gfx950 -O3 valu_dep_chain.hip -o /tmp/valu_dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void chain_kernel(float* out, unsigned long long* cyc, int iters)
{
  float a = 1.0f + threadIdx.x * 1e-7f, b = 1.0f - threadIdx.x * 1e-7f;
  const float m = 0.999999f, c = 1e-9f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      a = __builtin_fmaf(a, m, c);
      if (CHAINS == 2) b = __builtin_fmaf(b, m, c);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
  float* d; unsigned long long* c;
  hipMalloc(&d, sizeof(float) * 1024 * 1024); hipMalloc(&c, sizeof(unsigned long long) * 4096);
  const int iters = 4096;
  for (int chains = 1; chains <= 2; chains++)
    for (int threads : {64, 256, 512, 1024}) {          // 1 wave on one SIMD; 1 per SIMD; 2 per SIMD; 4 per SIMD (one workgroup = one CU)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (chains == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(256), dim3(threads), 0, 0, d, c, iters);
        else hipLaunchKernelGGL(chain_kernel<2>, dim3(256), dim3(threads), 0, 0, d, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(256); hipMemcpy(h.data(), c, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost);
      const double per = (double)h[0] / ((double)iters * 16 * chains);
      printf("chains %d, %4d threads per workgroup (%d wavefronts per SIMD): %.3f ms, %.2f counter ticks (100 MHz) per fma of wave 0, %.1f ns per fma per wavefront\n",
             chains, threads, threads >= 256 ? threads / 256 : 1, ms, per, 1e6 * ms / ((double)iters * 16 * chains));
    }
  return 0;
}
