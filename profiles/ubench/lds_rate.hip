// Micro-benchmark: LDS throughput per CU on gfx950 for the access shapes of the fused kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f2* p2 = reinterpret_cast<f2*>(smem) + wave * 2048;
  f4* p4 = reinterpret_cast<f4*>(smem) + wave * 1024;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  f2 s2 = {0.f, 0.f}; f4 s4 = {0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {            // 16 x ds_read_b64, lanes contiguous (512 B per instruction)
#pragma unroll
      for (int r = 0; r < 16; r++) { f2 v = p2[r * 64 + lane]; s2 += v; }
    } else if (MODE == 1) {     // 16 x ds_read_b128, lanes contiguous (1 KiB per instruction)
#pragma unroll
      for (int r = 0; r < 16; r++) { f4 v = p4[r * 64 + lane]; s4 += v; }
    } else if (MODE == 2) {     // 16 x ds_read_b128, 16 distinct addresses broadcast to 4 lane groups (the weight-pair reads)
#pragma unroll
      for (int r = 0; r < 16; r++) { f4 v = p4[r * 16 + (lane & 15)]; s4 += v; }
    } else if (MODE == 3) {     // 16 x ds_write_b64 contiguous
#pragma unroll
      for (int r = 0; r < 16; r++) p2[r * 64 + lane] = s2 + f2{(float)r, (float)it};
    } else if (MODE == 4) {     // 16 x ds_read_b64 in the 17-column padded FFT layout: frame fl = lane >> 4 (stride 272), row r, column j
#pragma unroll
      for (int r = 0; r < 16; r++) { f2 v = reinterpret_cast<f2*>(smem)[(wave * 4 + (lane >> 4)) * 272 + r * 17 + (lane & 15)]; s2 += v; }
    } else if (MODE == 5) {     // transposed write of the FFT exchange: [j * 17 + k1]
#pragma unroll
      for (int r = 0; r < 16; r++) reinterpret_cast<f2*>(smem)[(wave * 4 + (lane >> 4)) * 272 + (lane & 15) * 17 + r] = s2 + f2{(float)r, (float)it};
    }
    else if (MODE == 6) {     // 16 x ds_write_b128 contiguous
#pragma unroll
      for (int r = 0; r < 16; r++) p4[r * 64 + lane] = s4 + f4{(float)r, (float)it, 0.f, 0.f};
    } else if (MODE == 7) {     // 8 x ds_write_b128, rows of 16 float2 at pitch 18 (lane j writes its row: the pass-1 output of the FFT)
#pragma unroll
      for (int r = 0; r < 8; r++) reinterpret_cast<f4*>(reinterpret_cast<f2*>(smem) + (wave * 4 + (lane >> 4)) * 296 + (lane & 15) * 18)[r] = s4 + f4{(float)r, (float)it, 0.f, 0.f};
    } else if (MODE == 8) {     // 16 x ds_read_b64 columns at pitch 18: [jp * 18 + j]
#pragma unroll
      for (int r = 0; r < 16; r++) { f2 v = reinterpret_cast<f2*>(smem)[(wave * 4 + (lane >> 4)) * 296 + r * 18 + (lane & 15)]; s2 += v; }
    } else if (MODE == 9) {     // 8 x ds_read_b128 rows at pitch 18
#pragma unroll
      for (int r = 0; r < 8; r++) { f4 v = reinterpret_cast<f4*>(reinterpret_cast<f2*>(smem) + (wave * 4 + (lane >> 4)) * 296 + (lane & 15) * 18)[r]; s4 += v; }
    } else if (MODE == 10) {    // 16 x ds_write_b64 contiguous issued as 8 x ds_write2_b64-style pairs (compiler's choice)
#pragma unroll
      for (int r = 0; r < 16; r++) p2[(r >> 1) * 128 + (r & 1) * 64 + lane] = s2 + f2{(float)r, (float)it};
    }
    asm volatile("" : "+v"(s2), "+v"(s4));
  }
  const long long c1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[600 * 256] = (float)(c1 - c0) / (float)iters;
  out[blockIdx.x * 256 + tid] = s2.x + s2.y + s4.x + s4.y + s4.z + s4.w;
}

template <int MODE> void run(const char* name, float* d, int bytes_per_inst)
{
  const int iters = 4096;
  for (int wg : {1, 2}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) k<MODE><<<256 * wg, 256, 65536>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256 * wg, 256, 65536>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)iters * 16 * bytes_per_inst * 4 * wg;      // 4 waves per workgroup
    float cyc; hipMemcpy(&cyc, d + 600 * 256, 4, hipMemcpyDeviceToHost);
    // one loop iteration = 16 instructions per wave; 4 * wg waves per CU run the loop concurrently
    printf("%-58s %d WG/CU: %.3f ms, %.0f shader cycles per 16-instruction iteration -> %.1f B/clk per CU (clock %.2f GHz)\n", name, wg, ms, cyc,
           16.0 * bytes_per_inst * 4 * wg / cyc, cyc * iters / (ms * 1e6));
  }
}

int main()
{
  float* d; hipMalloc(&d, 1024 * 256 * sizeof(float));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<0>("ds_read_b64 contiguous", d, 512);
  run<1>("ds_read_b128 contiguous", d, 1024);
  run<2>("ds_read_b128 16 addresses x 4 (weight pairs)", d, 1024);
  run<3>("ds_write_b64 contiguous", d, 512);
  run<4>("ds_read_b64 FFT layout [fl][r*17+j]", d, 512);
  run<5>("ds_write_b64 FFT transposed [fl][j*17+r]", d, 512);
  run<6>("ds_write_b128 contiguous", d, 1024);
  run<7>("ds_write_b128 rows at pitch 18 (8 per 16 float2)", d, 512);
  run<8>("ds_read_b64 columns at pitch 18", d, 512);
  run<9>("ds_read_b128 rows at pitch 18 (8 per 16 float2)", d, 512);
  run<10>("ds_write_b64 pairs", d, 512);
  return 0;
}
