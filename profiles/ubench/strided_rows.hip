// Micro-benchmark: HBM read bandwidth for the "many rows, a few frames of each" pattern of the per-bin recursive kernels
// (NLMS canceller: a wavefront stages TBF frames of 64*CPL snapshot rows X[s][k][n][t0 .. t0+TBF) per tile; rows are
// T_stride * 8 bytes apart).  RUN bytes of every row per visit (64 = 8 frames, 128 = 16, 256 = 32), rows of one (s, k)
// block consecutive.  One single-wavefront workgroup per 256-row block, tiles visited in time order, as the kernel does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int RUN>            // bytes of a row per visit
__global__ __launch_bounds__(64) void k(const char* __restrict__ X, long row_bytes, long T_bytes, float* out)
{
  constexpr int LPR = RUN / 16;               // lanes per row
  constexpr int RPP = 64 / LPR;               // rows per wave-load
  constexpr int NPASS = 256 / RPP;
  const int lane = threadIdx.x, lrow = lane / LPR, lc = lane % LPR;
  const char* base = X + (long)blockIdx.x * 256 * row_bytes;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long t = 0; t < T_bytes; t += RUN) {
    f4 v[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; q++) v[q] = *reinterpret_cast<const f4*>(base + (long)(q * RPP + lrow) * row_bytes + t + lc * 16);
#pragma unroll
    for (int q = 0; q < NPASS; q++) acc += v[q];
  }
  out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// 64 B of a row per load instruction, but the two halves of a 128-byte line are requested back to back (what a kernel with
// 8-frame LDS tiles can do: prefetch two tiles at a time into twice the registers)
__global__ __launch_bounds__(64) void k2x64(const char* __restrict__ X, long row_bytes, long T_bytes, float* out)
{
  const int lane = threadIdx.x, lrow = lane / 4, lc = lane % 4;
  const char* base = X + (long)blockIdx.x * 256 * row_bytes;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long t = 0; t < T_bytes; t += 128) {
    f4 v[32];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      v[2 * q] = *reinterpret_cast<const f4*>(base + (long)(q * 16 + lrow) * row_bytes + t + lc * 16);
      v[2 * q + 1] = *reinterpret_cast<const f4*>(base + (long)(q * 16 + lrow) * row_bytes + t + 64 + lc * 16);
    }
#pragma unroll
    for (int q = 0; q < 32; q++) acc += v[q];
  }
  out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

template <int RUN>
static void run(const char* X, long rows, long row_bytes, long T_bytes, float* out, const char* name)
{
  const int blocks = (int)(rows / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<RUN>, dim3(blocks), dim3(64), 0, 0, X, row_bytes, T_bytes, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<RUN>, dim3(blocks), dim3(64), 0, 0, X, row_bytes, T_bytes, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%-44s rows %7ld x %6ld B of %6ld: %8.3f ms  %7.0f GB/s\n", name, rows, T_bytes, row_bytes, ms, (double)rows * T_bytes / ms / 1e6);
}

int main()
{
  // the NLMS bench shapes: S streams x 257 bins x 64 channels rows of 4096 frames (T_stride 4096 -> 32 KiB rows), and padded rows
  for (long S : {16L, 32L}) {
    for (long row_bytes : {32768L, 33152L}) {
      const long rows = S * 257 * 64 / 256 * 256;
      char* X; float* out;
      if (hipMalloc(reinterpret_cast<void**>(&X), rows * row_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
      hipMalloc(reinterpret_cast<void**>(&out), rows / 256 * 64 * 4);
      hipMemset(X, 0, rows * row_bytes);
      printf("S = %ld streams, row pitch %ld B\n", S, row_bytes);
      run<64>(X, rows, row_bytes, 32768, out, "64 B of each row per visit (8 frames)");
      run<128>(X, rows, row_bytes, 32768, out, "128 B (16 frames)");
      run<256>(X, rows, row_bytes, 32768, out, "256 B (32 frames)");
      {
        const int blocks = (int)(rows / 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k2x64, dim3(blocks), dim3(64), 0, 0, X, row_bytes, 32768L, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k2x64, dim3(blocks), dim3(64), 0, 0, X, row_bytes, 32768L, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-44s rows %7ld x %6d B of %6ld: %8.3f ms  %7.0f GB/s\n", "2 x 64 B back to back (two 8-frame tiles)", rows, 32768, row_bytes, ms, (double)rows * 32768 / ms / 1e6);
      }
      hipFree(X); hipFree(out);
    }
  }
  return 0;
}
