// Micro-benchmark: issue rate of packed-f32 VALU forms on gfx950 (how many cycles does a wave64 v_pk_*_f32 occupy a SIMD?).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
  f2 a[16], b = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { a[i] = f2{(float)threadIdx.x + i, 1.f}; s[i] = (float)i; }
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
      REP16(X)
#undef X
    } else if (MODE == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 4) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (MODE == 5) {   // dependent chain of 4 (4 independent chains)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 6) {   // fully dependent
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (MODE == 7) {   // fully dependent plain fma
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[0]) : "v"(b.x), "v"(c.x));
      REP16(X)
#undef X
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* d, int wg_per_cu)
{
  const int iters = 4096, nb = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<nb, 256>>>(d, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<nb, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // each SIMD runs wg_per_cu waves (256 threads = 4 waves, one per SIMD), each issuing iters*16 instructions
  const double inst_per_simd = (double)iters * 16 * wg_per_cu;
  printf("%-44s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wg_per_cu, ms,
         ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main()
{
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4}) {
    run<0>("v_pk_fma_f32 (16 independent)", d, w);
    run<1>("v_pk_fma_f32 op_sel/neg (16 independent)", d, w);
    run<2>("v_fma_f32 (16 independent)", d, w);
    run<3>("v_pk_add_f32 op_sel/neg (16 independent)", d, w);
    run<4>("v_pk_mul_f32 (16 independent)", d, w);
    run<5>("v_pk_fma_f32 (4 chains)", d, w);
    run<6>("v_pk_fma_f32 (1 chain)", d, w);
    run<7>("v_fma_f32 (1 chain)", d, w);
  }
  return 0;
}
