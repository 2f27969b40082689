// Micro-benchmark: HBM write bandwidth for the store pattern of the staged analysis bank: a workgroup owns one (stream,
// channel) and a run of consecutive TT-frame tiles; per tile it writes TT * 8 bytes into each of the 257 bin rows
// X[s][k][n][t0 .. t0+TT), which are N * T_stride * 8 bytes apart (layout [S][K][N][T]).  TT = 16 is what
// analysis512_kernel does (128-byte runs, four rows per wave-instruction); 32 / 64 = what longer tiles would give.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TT>
__global__ __launch_bounds__(256) void k(float2* __restrict__ X, int N, int K, long T_stride, int tiles_per_run, int runs)
{
  const int b = blockIdx.x;
  const int run = b % runs, chan = b / runs;               // chan = s * N + n
  const int s = chan / N, n = chan % N;
  const int f = threadIdx.x % TT, kq = threadIdx.x / TT;
  constexpr int KQ = 256 / TT;
  float2* base = X + ((long)s * K * N + n) * T_stride;
  for (int tile = 0; tile < tiles_per_run; tile++) {
    const long t = ((long)run * tiles_per_run + tile) * TT + f;
    const float2 v = make_float2((float)tile, (float)f);
    for (int k = kq; k < K; k += KQ) base[(long)k * N * T_stride + t] = v;
  }
}

template <int TT>
static void run(float2* X, int S, int N, int K, long T, long T_stride)
{
  const int frames_per_run = 256;                            // 16 tiles of 16 frames, as analysis512_kernel's A_RUN
  const int tiles_per_run = frames_per_run / TT, runs = (int)(T / frames_per_run);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<TT>, dim3(S * N * runs), dim3(256), 0, 0, X, N, K, T_stride, tiles_per_run, runs);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<TT>, dim3(S * N * runs), dim3(256), 0, 0, X, N, K, T_stride, tiles_per_run, runs);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("  %3d frames (%4d B) per row and tile: %7.3f ms  %6.0f GB/s\n", TT, TT * 8, ms, (double)S * K * N * T * 8 / ms / 1e6);
}

int main()
{
  const int S = 16, N = 64, K = 257; const long T = 4096;
  for (long T_stride : {4096L, 4144L}) {
    float2* X;
    if (hipMalloc(reinterpret_cast<void**>(&X), (size_t)S * K * N * T_stride * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("snapshots [%d][%d][%d][%ld], row pitch %ld B (%.1f GB written per launch)\n", S, K, N, T, T_stride * 8, (double)S * K * N * T * 8 / 1e9);
    run<8>(X, S, N, K, T, T_stride);                         // 64-byte runs: what the M = 2048 bank stores (8-frame tiles)
    run<16>(X, S, N, K, T, T_stride);
    run<32>(X, S, N, K, T, T_stride);
    run<64>(X, S, N, K, T, T_stride);
    run<128>(X, S, N, K, T, T_stride);
    hipFree(X);
  }
  return 0;
}
