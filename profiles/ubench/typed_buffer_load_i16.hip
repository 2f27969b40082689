#include <hip/hip_runtime.h>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
__device__ f2v llvm_raw_buffer_load_format_v2f32(i4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
__global__ void k(const short* __restrict__ in, float2* __restrict__ out, int n)
{
  // typed buffer: DATA_FORMAT 16_16 (5), NUM_FORMAT SSCALED (3), dst_sel x = R, y = G
  const unsigned long long p = (unsigned long long)in;
  i4v rs;
  rs.x = (int)(unsigned)p; rs.y = (int)(unsigned)(p >> 32); rs.z = 0x7fffffff; rs.w = 0x0002B02C;
  rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
  f2v acc = {0.f, 0.f};
  for (int i = 0; i < 4; i++) {
    f2v r = llvm_raw_buffer_load_format_v2f32(rs, (int)(threadIdx.x * 4u), i * 1024, 0);
    acc += r;
  }
  out[threadIdx.x] = make_float2(acc.x, acc.y);
}
int main()
{
  short* d; float2* o; const int n = 64;
  hipMalloc(&d, 8192 * 2); hipMalloc(&o, n * 8);
  short h[4096]; for (int i = 0; i < 4096; i++) h[i] = (short)(i * 37 - 20000);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, n>>>(d, o, n);
  float2 r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < n; t++) {
    float ex = 0, ey = 0; for (int i = 0; i < 4; i++) { ex += h[2 * t + 512 * i]; ey += h[2 * t + 1 + 512 * i]; }
    if (r[t].x != ex || r[t].y != ey) bad++;
  }
  printf("typed buffer load 16_16 SSCALED: %d mismatches of %d (lane 3: %g %g)\n", bad, n, r[3].x, r[3].y);
  return 0;
}
