// fft_packed.h -- complex arithmetic and radix-8/16/32 DFTs in packed float32 (v_pk_*_f32) for gfx950.
#pragma once
#include <hip/hip_runtime.h>

// ---- packed-float32 complex arithmetic (v_pk_*_f32 with op_sel / neg modifiers).  The compiler forms packed adds and
// FMAs from float2 code on its own, but materialises every "multiply by +-i" and conjugation with v_xor + v_mov pairs
// (a quarter of the FFT's instructions); the modifiers do those swaps and sign flips for free, so the few shapes the FFT
// and the beamformer sum need are spelled out.  op_sel[i] / op_sel_hi[i] pick the low or high dword of source i for the
// low / high result, neg_lo / neg_hi negate that source for the low / high result.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 add_ib(f2 a, f2 b)      // a + i b = (a.x - b.y, a.y + b.x)
{
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 sub_ib(f2 a, f2 b)      // a - i b = (a.x + b.y, a.y - b.x)
{
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 cmulv(f2 a, f2 w)       // a w
{
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                    // (a.x w.x, a.x w.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // + (-a.y w.y, a.y w.x)
  return r;
}
// polyphase product of a tap pair with a sample pair: (lo, hi) = (h.lo x.hi, h.hi x.lo) -- the low result takes the HIGH half of x and
// vice versa through op_sel, so no v_pk_mov swap is needed afterwards (hipcc emits one per output otherwise)
__device__ __forceinline__ f2 pk_mul_xswap(f2 h, f2 x)
{
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ void pk_fma_xswap(f2& acc, f2 h, f2 x)
{
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(h), "v"(x));
}
__device__ __forceinline__ f2 cmulc(f2 a, f2 w) { return __builtin_elementwise_fma(a.yy, f2{-w.y, w.x}, a.xx * w); }   // constant w
__device__ __forceinline__ void acc_conjw_z(f2& A, f2 w, f2 z)          // A += conj(w) z
{
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(A) : "v"(w), "v"(z));                       // (w.x z.x, w.x z.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(A) : "v"(w), "v"(z));        // (w.y z.y, -w.y z.x)
}
__device__ __forceinline__ void acc_conjw_conjz(f2& B, f2 w, f2 z)      // B += conj(w) conj(z)
{
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(B) : "v"(w), "v"(z));        // (w.x z.x, -w.x z.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(B) : "v"(w), "v"(z));   // (-w.y z.y, -w.y z.x)
}
__device__ __forceinline__ void dft4q(f2& a0, f2& a1, f2& a2, f2& a3)
{
  const f2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, t = a1 - a3;
  a0 = s02 + s13; a1 = add_ib(d02, t); a2 = s02 - s13; a3 = sub_ib(d02, t);
}
__device__ __forceinline__ void dft4q_i2(f2& a0, f2& a1, f2& a2, f2& a3)   // a2 stands for i a2
{
  const f2 s02 = add_ib(a0, a2), d02 = sub_ib(a0, a2), s13 = a1 + a3, t = a1 - a3;
  a0 = s02 + s13; a1 = add_ib(d02, t); a2 = s02 - s13; a3 = sub_ib(d02, t);
}
// dft16p on packed registers (same factorisation and twiddles)
__device__ __forceinline__ void dft16q(f2 (&v)[16])
{
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  f2 t[4][4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    f2 x0 = v[b], x1 = v[4 + b], x2 = v[8 + b], x3 = v[12 + b];
    dft4q(x0, x1, x2, x3);
    t[b][0] = x0; t[b][1] = x1; t[b][2] = x2; t[b][3] = x3;
  }
  t[1][1] = cmulc(t[1][1], f2{C1, S1});
  t[1][2] = cmulc(t[1][2], f2{H, H});
  t[1][3] = cmulc(t[1][3], f2{S1, C1});
  t[2][1] = cmulc(t[2][1], f2{H, H});
  t[2][3] = cmulc(t[2][3], f2{-H, H});                  // t[2][2] *= i is folded into dft4q_i2
  t[3][1] = cmulc(t[3][1], f2{S1, C1});
  t[3][2] = cmulc(t[3][2], f2{-H, H});
  t[3][3] = cmulc(t[3][3], f2{-C1, -S1});
#pragma unroll
  for (int c = 0; c < 4; c++) {
    f2 y0 = t[0][c], y1 = t[1][c], y2 = t[2][c], y3 = t[3][c];
    if (c == 2) dft4q_i2(y0, y1, y2, y3); else dft4q(y0, y1, y2, y3);
    v[c] = y0; v[c + 4] = y1; v[c + 8] = y2; v[c + 12] = y3;
  }
}


// in-register 8-point DFT, positive exponent (two radix-4 halves + radix-2 combine)
__device__ __forceinline__ void dft8q(f2 (&v)[8])
{
  constexpr float H = 0.70710678118654752f;
  f2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  f2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4q(e0, e1, e2, e3);
  dft4q(o0, o1, o2, o3);
  o1 = cmulc(o1, f2{H, H});
  o3 = cmulc(o3, f2{-H, H});
  v[0] = e0 + o0; v[4] = e0 - o0;
  v[1] = e1 + o1; v[5] = e1 - o1;
  v[2] = add_ib(e2, o2); v[6] = sub_ib(e2, o2);          // o2 stands for i o2
  v[3] = e3 + o3; v[7] = e3 - o3;
}

// in-register 32-point DFT: two 16-point DFTs (even / odd inputs), odd half times W32^c, radix-2 combine
__device__ __forceinline__ void dft32q(f2 (&v)[32])
{
  constexpr float CW[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f,
                            0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                            -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
  constexpr float SW[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f,
                            0.70710678118654752f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f,
                            1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
  f2 e[16], o[16];
#pragma unroll
  for (int a = 0; a < 16; a++) { e[a] = v[2 * a]; o[a] = v[2 * a + 1]; }
  dft16q(e);
  dft16q(o);
#pragma unroll
  for (int c = 0; c < 16; c++) {
    if (c == 8) { v[c] = add_ib(e[c], o[c]); v[c + 16] = sub_ib(e[c], o[c]); continue; }     // W32^8 = i
    const f2 t = (c == 0) ? o[0] : cmulc(o[c], f2{CW[c], SW[c]});
    v[c] = e[c] + t;
    v[c + 16] = e[c] - t;
  }
}

template <int P> __device__ __forceinline__ void dftq(f2 (&v)[P]);
template <> __device__ __forceinline__ void dftq<8>(f2 (&v)[8]) { dft8q(v); }
template <> __device__ __forceinline__ void dftq<16>(f2 (&v)[16]) { dft16q(v); }
template <> __device__ __forceinline__ void dftq<32>(f2 (&v)[32]) { dft32q(v); }
