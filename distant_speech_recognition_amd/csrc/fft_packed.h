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


// ---- folded-constant forms (round 4).  A twiddle w = c (1 + i t), t = tan(arg w), costs ONE packed FMA for the rotation
// a (1 + i t) = (a.x - t a.y, a.y + t a.x); the real scale c is carried into the butterfly that consumes the value, whose
// add / subtract becomes an FMA with c (no extra instruction).  The W8-type twiddles H (1 + i) are one modifier add
// a + i a with the scale H folded the same way.  16-point DFT: 52 v_pk_add + 20 v_pk_fma instead of 64 + 8 v_pk_mul + 8;
// with the inter-pass twiddles of a 256-point transform folded into the first butterflies of the second pass the pass
// costs 15 + 7 + 16 + 16 + 40 = 94 packed instructions instead of 30 + 80.
// k operand: the low (KH = 0) or high (KH = 1) dword of an SGPR or VGPR pair, broadcast to both halves.
#define BTK_PK3(NAME, CONS, MODS0, MODS1)                                                                     \
  template <int KH> __device__ __forceinline__ f2 NAME(f2 k, f2 x, f2 a)                                      \
  {                                                                                                           \
    f2 r;                                                                                                     \
    if constexpr (KH == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS0 : "=v"(r) : CONS(k), "v"(x), "v"(a));    \
    else asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS1 : "=v"(r) : CONS(k), "v"(x), "v"(a));                      \
    return r;                                                                                                 \
  }
// a + k x, a - k x
BTK_PK3(fma_ks, "s", "op_sel:[0,0,0] op_sel_hi:[0,1,1]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
BTK_PK3(fms_ks, "s", "op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]", "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
BTK_PK3(fma_kv, "v", "op_sel:[0,0,0] op_sel_hi:[0,1,1]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
BTK_PK3(fms_kv, "v", "op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]", "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
// a + i k x = (a.x - k x.y, a.y + k x.x),  a - i k x = (a.x + k x.y, a.y - k x.x)
BTK_PK3(fma_ib_ks, "s", "op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]")
BTK_PK3(fms_ib_ks, "s", "op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]")
BTK_PK3(fma_ib_kv, "v", "op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]")
#undef BTK_PK3
template <int KH> __device__ __forceinline__ f2 mul_kv(f2 k, f2 x)          // k x
{
  f2 r;
  if constexpr (KH == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(k), "v"(x));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(k), "v"(x));
  return r;
}

// second radix-4 stage of the 16-point DFT on the untwiddled first-stage results t[b][c] (b = input residue, c = output
// residue): the twiddles W16^{b c} are folded as above.  hc = (H, C1) = (cos pi/4, cos pi/8), t1 = (tan pi/8, *).
__device__ __forceinline__ void dft16t_stage2(f2 (&t)[4][4], f2 (&v)[16], f2 hc, f2 t1)
{
  {
    f2 y0 = t[0][0], y1 = t[1][0], y2 = t[2][0], y3 = t[3][0];
    dft4q(y0, y1, y2, y3);
    v[0] = y0; v[4] = y1; v[8] = y2; v[12] = y3;
  }
  {   // c = 1: W^1 = C1 (1 + i T1), W^2 = H (1 + i), W^3 = i C1 (1 - i T1)
    const f2 u0 = t[0][1], u1 = t[1][1], u2 = t[2][1], u3 = t[3][1];
    const f2 p1 = fma_ib_ks<0>(t1, u1, u1), p3 = fms_ib_ks<0>(t1, u3, u3), g2 = add_ib(u2, u2);
    const f2 s02 = fma_ks<0>(hc, g2, u0), d02 = fms_ks<0>(hc, g2, u0);
    const f2 s13 = add_ib(p1, p3), td = sub_ib(p1, p3);
    v[1] = fma_ks<1>(hc, s13, s02); v[9] = fms_ks<1>(hc, s13, s02);
    v[5] = fma_ib_ks<1>(hc, td, d02); v[13] = fms_ib_ks<1>(hc, td, d02);
  }
  {   // c = 2: W^2 = H (1 + i), W^4 = i, W^6 = i H (1 + i)
    const f2 u0 = t[0][2], u1 = t[1][2], u2 = t[2][2], u3 = t[3][2];
    const f2 g1 = add_ib(u1, u1), g3 = add_ib(u3, u3);
    const f2 s02 = add_ib(u0, u2), d02 = sub_ib(u0, u2);
    const f2 s13 = add_ib(g1, g3), td = sub_ib(g1, g3);
    v[2] = fma_ks<0>(hc, s13, s02); v[10] = fms_ks<0>(hc, s13, s02);
    v[6] = fma_ib_ks<0>(hc, td, d02); v[14] = fms_ib_ks<0>(hc, td, d02);
  }
  {   // c = 3: W^3 = i C1 (1 - i T1), W^6 = i H (1 + i), W^9 = -C1 (1 + i T1)
    const f2 u0 = t[0][3], u1 = t[1][3], u2 = t[2][3], u3 = t[3][3];
    const f2 q1 = fms_ib_ks<0>(t1, u1, u1), q3 = fma_ib_ks<0>(t1, u3, u3), g2 = add_ib(u2, u2);
    const f2 s02 = fma_ib_ks<0>(hc, g2, u0), d02 = fms_ib_ks<0>(hc, g2, u0);
    const f2 e = sub_ib(q3, q1), f = add_ib(q3, q1);          // y1 + y3 = -C1 e,  y1 - y3 = C1 f
    v[3] = fms_ks<1>(hc, e, s02); v[11] = fma_ks<1>(hc, e, s02);
    v[7] = fma_ib_ks<1>(hc, f, d02); v[15] = fms_ib_ks<1>(hc, f, d02);
  }
}
__device__ __forceinline__ void dft16t(f2 (&v)[16], f2 hc, f2 t1)
{
  f2 t[4][4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    f2 x0 = v[b], x1 = v[4 + b], x2 = v[8 + b], x3 = v[12 + b];
    dft4q(x0, x1, x2, x3);
    t[b][0] = x0; t[b][1] = x1; t[b][2] = x2; t[b][3] = x3;
  }
  dft16t_stage2(t, v, hc, t1);
}
// 16-point DFT of w_n v[n], n = 0..15, w_0 = 1, w_n = c_n (1 + i t_n) given as tw[n - 1] = (c_n, t_n)
__device__ __forceinline__ void dft16t_tw(f2 (&v)[16], const f2 (&tw)[15], f2 hc, f2 t1)
{
  f2 t[4][4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const f2 r2 = fma_ib_kv<1>(tw[8 + b - 1], v[8 + b], v[8 + b]);
    const f2 r1 = fma_ib_kv<1>(tw[4 + b - 1], v[4 + b], v[4 + b]);
    const f2 r3 = fma_ib_kv<1>(tw[12 + b - 1], v[12 + b], v[12 + b]);
    f2 m0;
    if (b == 0) m0 = v[0];
    else m0 = mul_kv<0>(tw[b > 0 ? b - 1 : 0], fma_ib_kv<1>(tw[b > 0 ? b - 1 : 0], v[b], v[b]));
    const f2 s02 = fma_kv<0>(tw[8 + b - 1], r2, m0), d02 = fms_kv<0>(tw[8 + b - 1], r2, m0);
    const f2 m1 = mul_kv<0>(tw[4 + b - 1], r1);
    const f2 s13 = fma_kv<0>(tw[12 + b - 1], r3, m1), td = fms_kv<0>(tw[12 + b - 1], r3, m1);
    t[b][0] = s02 + s13; t[b][1] = add_ib(d02, td); t[b][2] = s02 - s13; t[b][3] = sub_ib(d02, td);
  }
  dft16t_stage2(t, v, hc, t1);
}
// (cos, sin) -> (cos, tan); a vanishing cosine is replaced by 1e-20 (the product c t stays sin to rounding)
__device__ __forceinline__ f2 tw_tangent(float c, float s)
{
  const float cc = (fabsf(c) < 1e-30f) ? 1e-20f : c;
  return f2{cc, s / cc};
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
