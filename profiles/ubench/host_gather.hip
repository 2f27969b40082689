// host_gather.hip -- many rows in SEPARATE pinned host allocations -> one device block: per-row hipMemcpyAsync (SDMA, one API call
// per row) against ONE gather kernel that reads the pinned host memory over PCIe through a table of row pointers.
// hipcc --offload-arch=gfx950 -O3 host_gather.hip -o /tmp/host_gather
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void gather_rows(const uint4* const* __restrict__ src, uint4* __restrict__ dst, long row16, long pitch16, int wg_per_row)
{
  const int row = blockIdx.x / wg_per_row, part = blockIdx.x % wg_per_row;
  const uint4* s = src[row];
  uint4* d = dst + (long)row * pitch16;
  const long per = (row16 + wg_per_row - 1) / wg_per_row;
  const long i0 = part * per, i1 = i0 + per < row16 ? i0 + per : row16;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i + u * (long)blockDim.x < i1) v[u] = s[i + u * (long)blockDim.x];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i + u * (long)blockDim.x < i1) d[i + u * (long)blockDim.x] = v[u];
  }
}

int main()
{
  for (long row_bytes : {262144L, 524288L, 1048576L, 4194304L}) {
    const int rows = (int)((1L << 30) / row_bytes);
    std::vector<void*> h(rows);
    for (int r = 0; r < rows; r++) { hipHostMalloc(&h[r], row_bytes, hipHostMallocDefault); memset(h[r], r & 255, row_bytes); }
    char* d; hipMalloc(&d, (size_t)rows * row_bytes);
    void** tab; hipMalloc(&tab, sizeof(void*) * rows);
    hipMemcpy(tab, h.data(), sizeof(void*) * rows, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double best_c = 1e9, best_k[3] = {1e9, 1e9, 1e9};
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < rows; r++) hipMemcpyAsync(d + (size_t)r * row_bytes, h[r], row_bytes, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
      double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt < best_c) best_c = dt;
      int wi = 0;
      for (int wpr : {1, 4, 16}) {
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(gather_rows, dim3(rows * wpr), dim3(256), 0, st, (const uint4* const*)tab, (uint4*)d, row_bytes / 16, row_bytes / 16, wpr);
        hipStreamSynchronize(st);
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best_k[wi]) best_k[wi] = dt;
        wi++;
      }
    }
    const double gb = (double)rows * row_bytes / 1e9;
    printf("%5d rows of %7ld B (%.2f GB): per-row hipMemcpyAsync %.1f GB/s | gather kernel, 1 / 4 / 16 workgroups per row: %.1f / %.1f / %.1f GB/s\n",
           rows, row_bytes, gb, gb / best_c, gb / best_k[0], gb / best_k[1], gb / best_k[2]);
    for (int r = 0; r < rows; r++) hipHostFree(h[r]);
    hipFree(d); hipFree(tab);
  }
  return 0;
}
