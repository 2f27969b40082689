// Micro-benchmark: does the cost of packed-f32 VALU instructions on gfx950 depend on which VGPR banks (register index mod 4)
// their 64-bit operands live in?  Explicit registers v[100..171]; 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
             "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
             "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
             "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171"
// 8 independent instructions per block, destinations v[100+4i : 101+4i] (bank pair {0,1}) -- D = A op B [+ C]
#define ADD(d, a, b) "v_pk_add_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1]\n\t"
#define FMA(d, a, b, c) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #c ":" #c "+1]\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
  asm volatile("v_mov_b32 v140, 1.0\n\tv_mov_b32 v141, 1.0\n\tv_mov_b32 v142, 1.0\n\tv_mov_b32 v143, 1.0\n\t"
               "v_mov_b32 v144, 0\n\tv_mov_b32 v145, 0\n\tv_mov_b32 v146, 0\n\tv_mov_b32 v147, 0\n\t"
               "v_mov_b32 v148, 0\n\tv_mov_b32 v149, 0\n\tv_mov_b32 v150, 0\n\tv_mov_b32 v151, 0" ::: CLOB);
  const long long c0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0)        // pk_add, sources in the SAME bank pair: a = {0,1}, b = {0,1}
      asm volatile(ADD(100,100,144) ADD(104,104,148) ADD(108,108,144) ADD(112,112,148) ADD(116,116,144) ADD(120,120,148) ADD(124,124,144) ADD(128,128,148) ::: CLOB);
    else if (MODE == 1)   // pk_add, sources in DIFFERENT bank pairs: a = {0,1}, b = {2,3}
      asm volatile(ADD(100,100,146) ADD(104,104,150) ADD(108,108,146) ADD(112,112,150) ADD(116,116,146) ADD(120,120,150) ADD(124,124,146) ADD(128,128,150) ::: CLOB);
    else if (MODE == 2)   // pk_fma chain form d = a * b + d: a {0,1}, b {0,1}, d {0,1}
      asm volatile(FMA(100,140,144,100) FMA(104,140,148,104) FMA(108,140,144,108) FMA(112,140,148,112) FMA(116,140,144,116) FMA(120,140,148,120) FMA(124,140,144,124) FMA(128,140,148,128) ::: CLOB);
    else if (MODE == 3)   // pk_fma: a {0,1}, b {2,3}, d {0,1}
      asm volatile(FMA(100,140,146,100) FMA(104,140,150,104) FMA(108,140,146,108) FMA(112,140,150,112) FMA(116,140,146,116) FMA(120,140,150,120) FMA(124,140,146,124) FMA(128,140,150,128) ::: CLOB);
    else if (MODE == 4)   // pk_fma: a {2,3}, b {2,3}, d {0,1}
      asm volatile(FMA(100,142,146,100) FMA(104,142,150,104) FMA(108,142,146,108) FMA(112,142,150,112) FMA(116,142,146,116) FMA(120,142,150,120) FMA(124,142,146,124) FMA(128,142,150,128) ::: CLOB);
    else if (MODE == 5)   // pk_fma: a {0,1}, b {2,3}, d {2,3} destinations in {2,3}
      asm volatile(FMA(102,140,146,102) FMA(106,140,150,106) FMA(110,140,146,110) FMA(114,140,150,114) FMA(118,140,146,118) FMA(122,140,150,122) FMA(126,140,146,126) FMA(130,140,150,130) ::: CLOB);
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[600 * 256] = (float)(c1 - c0) / ((float)iters * 8);
  float r;
  asm volatile("v_add_f32 %0, v100, v104\n\tv_add_f32 %0, %0, v108\n\tv_add_f32 %0, %0, v102" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* d)
{
  const int iters = 16384, nb = 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) k<MODE><<<nb, 256>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<nb, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d + 600 * 256, 4, hipMemcpyDeviceToHost);
  printf("%-56s %.3f ms -> %.2f ns per instruction per SIMD (2 waves/SIMD); s_memtime: %.2f cycles per instruction of one wave -> clock %.2f GHz if the two waves alternate\n", name, ms, ms * 1e6 / ((double)iters * 8 * 2), cyc, cyc / 2 / (ms * 1e6 / ((double)iters * 8 * 2)));
}

int main()
{
  float* d; hipMalloc(&d, 1024 * 256 * sizeof(float));
  run<0>("v_pk_add_f32  a{0,1} b{0,1}", d);
  run<1>("v_pk_add_f32  a{0,1} b{2,3}", d);
  run<2>("v_pk_fma_f32  a{0,1} b{0,1} c=d{0,1}", d);
  run<3>("v_pk_fma_f32  a{0,1} b{2,3} c=d{0,1}", d);
  run<4>("v_pk_fma_f32  a{2,3} b{2,3} c=d{0,1}", d);
  run<5>("v_pk_fma_f32  a{0,1} b{2,3} c=d{2,3}", d);
  return 0;
}
