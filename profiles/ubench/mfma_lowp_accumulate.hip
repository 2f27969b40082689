// How exactly do the 16-bit matrix instructions add up their 16 products before the float32 accumulator?  One
// v_mfma_f32_32x32x16_{bf16,f16}: A[i][0] = 1, A[i][k > 0] = e = 2^-n, B = 1, C = 0 or C = 2^20: exact result 1 + 15 e (+ C).
// hipcc --offload-arch=gfx950 -O2 mfma_lowp_accumulate.hip -o mfma_lowp_accumulate && ./mfma_lowp_accumulate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float e, float c0, float* out)
{
  const int lane = threadIdx.x, lk = lane >> 5;
  bf16x8 a, b; f16x8 ah, bh;
  for (int i = 0; i < 8; i++) {
    const float v = (lk == 0 && i == 0) ? 1.0f : e;
    a[i] = (__bf16)v; b[i] = (__bf16)1.0f; ah[i] = (_Float16)v; bh[i] = (_Float16)1.0f;
  }
  f32x16 c; for (int i = 0; i < 16; i++) c[i] = c0;
  f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  f32x16 dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  if (lane == 0) { out[0] = d[0]; out[1] = dh[0]; }
}
int main()
{
  float* d; hipMalloc(&d, 8);
  for (int n = 6; n <= 24; n += 2)
    for (float c0 : {0.0f, 1048576.0f}) {
      const float e = ldexpf(1.0f, -n);
      probe<<<1, 64>>>(e, c0, d);
      float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
      const double exact = 1.0 + 15.0 * e + c0;
      printf("e = 2^-%-2d C = %-8g exact %.10f  bf16 %.10f (err %.3g)  f16 %.10f (err %.3g)\n", n, c0, exact - c0, (double)h[0] - c0, h[0] - exact, (double)h[1] - c0, h[1] - exact);
    }
  return 0;
}
