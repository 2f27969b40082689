#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out)
{
  const int l = threadIdx.x;
  // A[i][k] = (k == 0) ? i : 0  from lane i + 16 k ; B[k][j] = (k == 0) ? 100 + j : 0 from lane j + 16 k  -> D[i][j] = i * (100 + j)
  const double a = (l / 16 == 0) ? (double)(l % 16) : 0.0;
  const double b = (l / 16 == 0) ? (double)(100 + l % 16) : 0.0;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; v++) out[l * 4 + v] = c[v];
}
int main()
{
  double* d; hipMalloc(&d, 256 * 8); double h[256];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) { printf("lane %2d:", l); for (int v = 0; v < 4; v++) { double x = h[l * 4 + v]; int j = l % 16; int i = (int)(x / (100 + j) + 0.5); printf("  v%d: i=%d (j=%d)", v, i, j); } printf("\n"); }
  return 0;
}
