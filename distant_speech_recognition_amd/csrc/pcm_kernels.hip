// pcm_kernels.hip -- the sample formats on either side of the path.
// The reference's SampleFeature reads 16-bit PCM and hands UN-NORMALISED floats to the analysis bank (feature/feature.cc:265-269),
// and its application scripts write the synthesis bank's float blocks back as int16 (unit_test/test_online_beamforming.py:209,
// `numpy.array(buf, numpy.int16)`, i.e. truncation toward zero).  Utterances therefore cross PCIe as int16 -- half the bytes of
// the float samples the kernels compute on -- and are widened / narrowed on the device.  Both kernels are pure HBM streams.
#include "btk_internal.h"

namespace {

// 8 samples per thread: one 16-byte load, two 16-byte stores
__global__ __launch_bounds__(256)
void pcm_i16_to_f32_kernel(const short* __restrict__ in, float* __restrict__ out, long n)
{
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i0 + 8 <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const int4 v = *reinterpret_cast<const int4*>(in + i0);
    const int w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int q = 0; q < 4; q++) { f[2 * q] = (float)(short)(w[q] & 0xffff); f[2 * q + 1] = (float)(short)(w[q] >> 16); }
    *reinterpret_cast<float4*>(out + i0) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(out + i0 + 4) = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    for (long i = i0; i < n && i < i0 + 8; i++) out[i] = (float)in[i];
  }
}

__global__ __launch_bounds__(256)
void pcm_f32_to_i16_kernel(const float* __restrict__ in, short* __restrict__ out, long n)
{
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  for (long i = i0; i < n && i < i0 + 8; i++) out[i] = (short)(int)in[i];       // numpy's float -> int16 cast: toward zero
}

// A multi-channel recording is stored frame by frame, in[l][c] (what libsndfile hands SampleFeature::read, which then copies out
// channel chX with stride chN: feature/feature.cc:333-334, once per channel node).  One pass de-interleaves ALL channels:
// 64 x 64 tiles through LDS, reads coalesced along c, writes coalesced along l.  out[c][l] float32, rows out_stride apart.
__global__ __launch_bounds__(256)
void pcm_i16_deinterleave_kernel(const short* __restrict__ in, float* __restrict__ out, long L, int N, long out_stride)
{
  __shared__ float tile[64][65];
  const long l0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const long l = l0 + ty + 4 * r;
    const int c = c0 + tx;
    tile[ty + 4 * r][tx] = (l < L && c < N) ? (float)in[l * N + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int c = c0 + ty + 4 * r;
    const long l = l0 + tx;
    if (c < N && l < L) out[(long)c * out_stride + l] = tile[tx][ty + 4 * r];
  }
}

}  // namespace

extern "C" int btk_pcm_i16_deinterleave(const short* in, float* out, long L, int N, long out_stride, void* stream)
{
  if (L < 0 || N <= 0 || out_stride < L || (L > 0 && (!in || !out))) return btk_set_error(BTK_ERR_PARAMETER, "btk_pcm_i16_deinterleave: bad argument");
  if (L == 0) return BTK_OK;
  hipLaunchKernelGGL(pcm_i16_deinterleave_kernel, dim3((unsigned)((L + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, as_stream(stream),
                     in, out, L, N, out_stride);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

extern "C" int btk_pcm_i16_to_f32(const short* in, float* out, long n, void* stream)
{
  if (n < 0 || (n > 0 && (!in || !out))) return btk_set_error(BTK_ERR_PARAMETER, "btk_pcm_i16_to_f32: bad argument");
  if (n == 0) return BTK_OK;
  hipLaunchKernelGGL(pcm_i16_to_f32_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, as_stream(stream), in, out, n);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

extern "C" int btk_pcm_f32_to_i16(const float* in, short* out, long n, void* stream)
{
  if (n < 0 || (n > 0 && (!in || !out))) return btk_set_error(BTK_ERR_PARAMETER, "btk_pcm_f32_to_i16: bad argument");
  if (n == 0) return BTK_OK;
  hipLaunchKernelGGL(pcm_f32_to_i16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, as_stream(stream), in, out, n);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}
