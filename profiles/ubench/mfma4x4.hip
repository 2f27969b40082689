// Layout probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: which lane/register holds D[i][j] of block b, given A[i] and B[j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out)
{
  const int lane = threadIdx.x;
  // A value encodes (block, i) = lane: a = 1 + lane;  B value encodes (block, j): b = 1000 + lane  -> D = a * b identifies both
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + lane), (float)(1000 + lane), c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[lane * 4 + r] = c[r];
}
int main()
{
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 4; r++) {
      const int b = lane / 4, j = lane % 4;
      const float expect = (float)(1 + 4 * b + r) * (float)(1000 + 4 * b + j);      // hypothesis: D[i = reg][j = lane % 4] of block lane / 4
      if (h[lane * 4 + r] != expect) { ok = 0; if (lane < 8) printf("lane %d reg %d: got %g expect %g\n", lane, r, h[lane * 4 + r], expect); }
    }
  printf("hypothesis D[reg][lane%%4], A from lane 4b+i, B from lane 4b+j: %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
