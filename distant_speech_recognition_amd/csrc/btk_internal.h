// btk_internal.h -- shared declarations of the HIP engine (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "btkhip.h"

struct btk_fb {
  int M, m, r, R, D, K;
  int dct, synthesis;
  int pd;            // processing_delay_  (modulated.cc:246-264)
  int laN;           // laN_               (modulated.cc:246-264)
  int gain_factor;   // gain_factor_ (int, default 1)
  float* d_proto;    // [m*M] float32 prototype on device
  float2* d_tw;      // [M] e^{+j 2 pi j / M}
  int kx0, kx1;      // bin range the analysis kernels store (a bin shard writes only its bins; 0, K for a whole plan)
};

int btk_set_error(int code, const char* fmt, ...);

// Diagnostic A/B switches of the profiling scripts (profiles/): read from the environment ONCE per process, in one place, so
// that sizing and dispatch can never disagree; production runs leave them unset.
struct btk_switches_t {
  bool disable_analysis512, disable_synthesis512, disable_fast, disable_fused, nlms_v1, wpe_noskip, wpe_timing, syn_narrow, rls_packed, wpe_solve_panel, wpe_solve_reg, wpe_herk_blocks /* BTK_WPE_HERK_BLOCKS: the round-2 block HERK instead of the lag-product form */,
       wpe_predict_valu /* BTK_WPE_PREDICT_VALU: the vector prediction kernel instead of the matrix-core one */,
       wpe_lagprod_f32 /* BTK_WPE_LAGPROD_F32: the float32 matrix instruction in the lag-product kernel (round 4) instead of the float16-split form */;
  int wpe_lagprod_waves /* BTK_WPE_LAGPROD_WAVES: 4 = four wavefronts x one column block per task (A/B: 8.6 ms per stream) instead of two x two (7.3, the default) */;
  int mvdr_reg_min /* BTK_MVDR_REG_MIN: channel count from which the MVDR design runs on the register-resident solver (default 64) */, nlms_alt, fused_var /* -1: default */, pf_jb, pf_tpw /* 0: default */, pf_mfma_min /* channels from which the matrix-core statistics kernel runs */;
};
const btk_switches_t& btk_switches();

#define BTK_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess)                                                               \
      return btk_set_error(BTK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Two consecutive 16-bit PCM samples as two floats, converted by the LOAD: a typed buffer load (buffer_load_format_xy) through a
// resource whose descriptor says DATA_FORMAT 16_16, NUM_FORMAT SSCALED -- signed integers delivered as their float values, exact.
// hipcc has no builtin for the format loads; the LLVM intrinsic is reached by its name (the compiler keeps track of the load like
// of any other: no hand-written waits).  Descriptor word 3 for __builtin_amdgcn_make_buffer_rsrc: dst_sel x = R, y = G, SSCALED, 16_16.
typedef float btk_f2v __attribute__((ext_vector_type(2)));
typedef float btk_f4v __attribute__((ext_vector_type(4)));
// Cache-hint experiments (profiles/r06_nt_hints.txt): -DBTK_EXP=<bits> builds a library in which single global streams of the
// bandwidth-bound kernels are non-temporal.  btk_ld<NT> / btk_st<NT>: a plain or a non-temporal 8- / 16-byte access.
#ifndef BTK_EXP
#define BTK_EXP 0
#endif
#if defined(__HIPCC__)
template <bool NT> __device__ __forceinline__ float4 btk_ld(const float4* p)
{
  if constexpr (NT) { const btk_f4v t = __builtin_nontemporal_load(reinterpret_cast<const btk_f4v*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ float2 btk_ld(const float2* p)
{
  if constexpr (NT) { const btk_f2v t = __builtin_nontemporal_load(reinterpret_cast<const btk_f2v*>(p)); return make_float2(t.x, t.y); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ void btk_st(float2* p, float2 v)
{
  if constexpr (NT) __builtin_nontemporal_store(btk_f2v{v.x, v.y}, reinterpret_cast<btk_f2v*>(p));
  else *p = v;
}
template <bool NT> __device__ __forceinline__ void btk_st(float4* p, float4 v)
{
  if constexpr (NT) __builtin_nontemporal_store(btk_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<btk_f4v*>(p));
  else *p = v;
}
#endif
__device__ btk_f2v btk_buffer_load_i16x2_f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v2f32");
constexpr int BTK_RSRC_I16X2_SSCALED = 0x0002B02C;
// four samples: DATA_FORMAT 16_16_16_16 (12), NUM_FORMAT SSCALED (3), dst_sel x y z w
__device__ btk_f4v btk_buffer_load_i16x4_f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
constexpr int BTK_RSRC_I16X4_SSCALED = 0x00063FAC;

// fb_analysis512.hip: specialised analysis kernel (M=512, m=4); returns 1 handled / 0 not covered / <0 error
int btk_analysis512_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                        long T_stride, long t0, long tcount, hipStream_t st);
int btk_analysis512_bf_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                           int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st);
int btk_analysis512_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                            long T_stride, long t0, long tcount, hipStream_t st);
int btk_analysis512_bf_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                               int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st);
int btk_synthesis512_try(const btk_fb* fb, const void* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                         long b0, long bcount, hipStream_t st);
// fb_fast.hip: register-FFT kernels for M in {256,512,1024,2048}, m = 4
int btk_fast_analysis_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                          long T_stride, long t0, long tcount, hipStream_t st);
int btk_fast_analysis_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, void* X,
                              long T_stride, long t0, long tcount, hipStream_t st);
int btk_fast_analysis_bf_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                                 int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st);
int btk_fast_analysis_bf_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                             int per_stream, void* Wt_scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st);
// fb_fused_big.hip: fused analysis -> fixed-weight beamformer for M = 1024 / 2048 (m = 4, r = 1); scratch bytes 0 = geometry not covered
long btk_big_analysis_bf_scratch_bytes(const btk_fb* fb, int S, int N, int per_stream, long tcount);
int btk_big_analysis_bf_try(const btk_fb* fb, const float* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                            int per_stream, void* scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st);
int btk_big_analysis_bf_i16_try(const btk_fb* fb, const short* pcm, long nsamples, long pcm_stride, int S, int N, const void* W,
                                int per_stream, void* scratch, void* Y, long T_stride, long t0, long tcount, hipStream_t st);
int btk_fast_synthesis_try(const btk_fb* fb, const void* Y, long nframes, long T_stride, int S, float* out, long out_stride,
                           long b0, long bcount, hipStream_t st);
