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

// Rows that lie in SEPARATE host allocations (one per channel node: every SampleFeature owns its utterance) -> one device block.
// A hipMemcpyAsync per row costs the API call and the DMA engine's set-up per row: 2 048 rows of 0.5 MB (32 graphs x 64 channels
// of a round) arrive at 27 GB/s, rows of 4 MB at 49 (profiles/r06_ubench_host_gather.txt).  One kernel that reads the pinned host
// memory itself, through a table of row pointers that also lies in pinned memory, runs at the link's 57 GB/s whatever the rows'
// length: 16-byte loads, four in flight per thread, the bytes behind a shorter row zeroed.
// The launch is a FIXED, small number of workgroups that walk the (row, 16 KB piece) items in contiguous runs: the kernel's
// wavefronts wait on the link for milliseconds, and a grid that fills every wavefront slot of the chip (the first form: 2 048
// workgroups) keeps the kernels of the block BEFORE from starting -- the upload is there to run under them.
__global__ __launch_bounds__(256)
void gather_rows_kernel(const btk_row_t* __restrict__ table, char* __restrict__ dst, long dst_pitch, long pieces_per_row, long items)
{
  const long per_wg = (items + gridDim.x - 1) / gridDim.x;
  const long it0 = blockIdx.x * per_wg, it1 = it0 + per_wg < items ? it0 + per_wg : items;
  const long p16 = dst_pitch >> 4;
  long row = -1, n16 = 0, tail = 0, skip = -1;
  const uint4* s = nullptr;
  uint4* d = nullptr;
  for (long it = it0; it < it1; it++) {
    const long r = it / pieces_per_row, piece = it - r * pieces_per_row;
    if (r != row) {                                            // (a table entry crosses the link once per run of a row's pieces)
      row = r;
      const btk_row_t e = table[row];
      n16 = e.bytes >> 4; tail = e.bytes & 15;
      skip = tail ? n16 : -1;                                  // the partial word of the row is written below, by one place only
      s = static_cast<const uint4*>(e.src);
      d = reinterpret_cast<uint4*>(dst + row * dst_pitch);
    }
    const long i = piece * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const long j = i + 256 * u; v[u] = j < n16 ? s[j] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
    for (int u = 0; u < 4; u++) { const long j = i + 256 * u; if (j < p16 && j != skip) d[j] = v[u]; }
    // the last, partial 16-byte word of a row whose length is no multiple of 16 (its 2-byte samples one by one; zeros behind them)
    if (tail && piece == n16 / 1024 && threadIdx.x < 8) {
      const unsigned short* s2 = reinterpret_cast<const unsigned short*>(s + n16);
      unsigned short* d2 = reinterpret_cast<unsigned short*>(d + n16);
      d2[threadIdx.x] = (long)threadIdx.x * 2 < tail ? s2[threadIdx.x] : (unsigned short)0;
    }
  }
}

}  // namespace

extern "C" int btk_gather_rows(const btk_row_t* table, void* dst, int nrows, long dst_pitch_bytes, void* stream)
{
  if (nrows < 0 || dst_pitch_bytes < 0 || (nrows > 0 && (!table || !dst))) return btk_set_error(BTK_ERR_PARAMETER, "btk_gather_rows: bad argument");
  if ((dst_pitch_bytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15))
    return btk_set_error(BTK_ERR_PARAMETER, "btk_gather_rows: dst and dst_pitch_bytes must be multiples of 16");
  if (nrows == 0 || dst_pitch_bytes == 0) return BTK_OK;
  // 16 KB pieces, walked by a fixed number of workgroups (BTK_GATHER_WGS: measurement knob)
  static const int wgs_env = getenv("BTK_GATHER_WGS") ? atoi(getenv("BTK_GATHER_WGS")) : 0;
  const long ppr = (dst_pitch_bytes / 16 + 1023) / 1024;
  const long items = (long)nrows * ppr;
  long wgs = wgs_env > 0 ? wgs_env : 64;     // (32 ... 128 measure alike and best, 256 and more get in the kernels' way: profiles/r06_node_api_gather.txt)
  if (wgs > items) wgs = items;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)wgs), dim3(256), 0, as_stream(stream), table, static_cast<char*>(dst),
                     dst_pitch_bytes, ppr, items);
  BTK_HIP_CHECK(hipGetLastError());
  return BTK_OK;
}

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
