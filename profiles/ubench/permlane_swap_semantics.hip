// What v_permlane32_swap / v_permlane16_swap (gfx950) do to two registers, printed row by row (a row = 16 lanes).
// build: hipcc --offload-arch=gfx950 -O2 permlane_swap_semantics.hip -o permlane_swap_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
  const unsigned lane = threadIdx.x;
  unsigned a = 0x100 + lane, b = 0x200 + lane;            // a = first operand, b = second
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main()
{
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane32_swap r[0]", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]"};
  for (int i = 0; i < 4; i++) {
    printf("%s:", names[i]);
    for (int row = 0; row < 4; row++) printf("  row%d = %c.row%d", row, (h[64 * i + 16 * row] >> 8) == 1 ? 'a' : 'b', (h[64 * i + 16 * row] & 0xff) / 16);
    printf("\n");
  }
  return 0;
}
