// comm_rccl.hip -- the one collective of the frequency-bin-sharded path: after every rank has beamformed its
// bin range, ONE all-gather of the beamformed block precedes synthesis.  The reference has no counterpart (it is a
// single-threaded CPU library); the shape of the exchange follows from its per-bin loops (beamformer.cc:1298,
// postfilter.cc:184, pybeamformer.py:674): everything between the analysis FFT and the synthesis FFT is independent per bin.
//
// RCCL is bound at run time (dlsym in the process, then dlopen("librccl.so")): libbtkhip.so itself has no RCCL dependency,
// single-GPU users never load it, and a host program that already initialised RCCL gets ITS instance (a communicator must
// not cross library instances -- which is also why the Python layer, where torch brings its own RCCL, keeps using
// torch.distributed for this step: sharding.allgather_bins).
#include "btk_internal.h"
#include <dlfcn.h>
#include <cstddef>
#include <mutex>

namespace {
typedef int (*nccl_group_fn)();
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
struct Rccl { nccl_group_fn start = nullptr, end = nullptr; nccl_bcast_fn bcast = nullptr; nccl_allgather_fn allgather = nullptr; nccl_errstr_fn errstr = nullptr; };
Rccl g_rccl;
std::once_flag g_rccl_once;

// Bound once per process (std::call_once: a second thread waits for the first instead of seeing a half-filled table).  Order:
// symbols already visible in the process; an RCCL the host program loaded RTLD_LOCAL (torch does) through RTLD_NOLOAD -- the SAME
// instance, a communicator must not cross library instances --; only then a fresh dlopen.
bool bind_rccl()
{
  std::call_once(g_rccl_once, [] {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllGather")) {
      h = dlopen("librccl.so", RTLD_NOLOAD | RTLD_NOW);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOLOAD | RTLD_NOW);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return;
    }
    Rccl r;
    r.start = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupStart"));
    r.end = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupEnd"));
    r.bcast = reinterpret_cast<nccl_bcast_fn>(dlsym(h, "ncclBroadcast"));
    r.allgather = reinterpret_cast<nccl_allgather_fn>(dlsym(h, "ncclAllGather"));
    r.errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(h, "ncclGetErrorString"));
    if (r.start && r.end && r.bcast && r.allgather) g_rccl = r;
  });
  return g_rccl.bcast != nullptr;
}
constexpr int kNcclFloat = 7;           // ncclFloat32 (rccl.h)
}  // namespace

extern "C" {

// bin range of a rank: ceil(K / world) bins per rank, trailing ranks short or empty (the partition of sharding.py)
void btk_bin_range(int K, int rank, int world, int* k0, int* k1)
{
  const int per = (K + world - 1) / world;
  const int a = rank * per < K ? rank * per : K;
  const int b = a + per < K ? a + per : K;
  if (k0) *k0 = a;
  if (k1) *k1 = b;
}

int btk_allgather_bins(void* nccl_comm, const void* Y_local, void* Y, int S, int K, long T_stride, int rank, int world, void* stream)
{
  if (!nccl_comm || !Y) return btk_set_error(BTK_ERR_PARAMETER, "btk_allgather_bins: null argument");
  if (S < 1 || K < 1 || T_stride < 1 || world < 1 || rank < 0 || rank >= world)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_allgather_bins: bad sizes S=%d K=%d T_stride=%ld rank=%d world=%d", S, K, T_stride, rank, world);
  if (!bind_rccl()) return btk_set_error(BTK_ERR_PARAMETER, "btk_allgather_bins: RCCL (librccl.so) is not available in this process");
  int myk0, myk1;
  btk_bin_range(K, rank, world, &myk0, &myk1);
  if (myk1 > myk0 && !Y_local) return btk_set_error(BTK_ERR_PARAMETER, "btk_allgather_bins: null local block");
  const float* src = static_cast<const float*>(Y_local);
  float* dst = static_cast<float*>(Y);
  // One broadcast per (owner rank, stream) inside one group: uneven and empty shards need no padding or staging buffer,
  // and every block lands at its place in Y [S][K][T_stride].
  int rc = g_rccl.start();
  for (int r = 0; r < world && rc == 0; r++) {
    int a, b;
    btk_bin_range(K, r, world, &a, &b);
    if (b == a) continue;
    const size_t count = (size_t)(b - a) * T_stride * 2;
    for (int s = 0; s < S && rc == 0; s++) {
      float* recv = dst + ((size_t)s * K + a) * T_stride * 2;
      const float* send = (r == rank) ? src + (size_t)s * (b - a) * T_stride * 2 : recv;
      rc = g_rccl.bcast(send, recv, count, kNcclFloat, r, nccl_comm, as_stream(stream));
    }
  }
  const int rc2 = g_rccl.end();
  if (rc == 0) rc = rc2;
  if (rc != 0) return btk_set_error(BTK_ERR_HIP, "btk_allgather_bins: RCCL error %d (%s)", rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
  return BTK_OK;
}

// rows per stream of a buffer the in-place all-gather can run in: world x ceil(K / world) >= K (the last shard padded)
int btk_bin_rows_padded(int K, int world) { return (K < 1 || world < 1) ? 0 : world * ((K + world - 1) / world); }

// The even form: Y [S][Kp][T_stride], Kp = btk_bin_rows_padded(K, world); every rank has ALREADY written its beamformed bins at
// rows [rank per, ...) of each stream (per = ceil(K / world); the apply kernel writes there directly), so the exchange is ONE
// in-place ncclAllGather per stream -- for the single-stream large-array case of SURVEY 8(e) literally a single all-gather, no
// staging copy on either side.  Rows >= K are padding.
int btk_allgather_bins_inplace(void* nccl_comm, void* Y, int S, int K, long T_stride, int rank, int world, void* stream)
{
  if (!nccl_comm || !Y) return btk_set_error(BTK_ERR_PARAMETER, "btk_allgather_bins_inplace: null argument");
  if (S < 1 || K < 1 || T_stride < 1 || world < 1 || rank < 0 || rank >= world)
    return btk_set_error(BTK_ERR_DIMENSION, "btk_allgather_bins_inplace: bad sizes S=%d K=%d T_stride=%ld rank=%d world=%d", S, K, T_stride, rank, world);
  if (!bind_rccl()) return btk_set_error(BTK_ERR_PARAMETER, "btk_allgather_bins_inplace: RCCL (librccl.so) is not available in this process");
  const int per = (K + world - 1) / world;
  const size_t count = (size_t)per * T_stride * 2, rows = (size_t)per * world;
  float* base = static_cast<float*>(Y);
  int rc = 0;
  if (S > 1) rc = g_rccl.start();
  for (int s = 0; s < S && rc == 0; s++) {
    float* recv = base + (size_t)s * rows * T_stride * 2;
    rc = g_rccl.allgather(recv + (size_t)rank * count, recv, count, kNcclFloat, nccl_comm, as_stream(stream));
  }
  if (S > 1) { const int rc2 = g_rccl.end(); if (rc == 0) rc = rc2; }
  if (rc != 0) return btk_set_error(BTK_ERR_HIP, "btk_allgather_bins_inplace: RCCL error %d (%s)", rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
  return BTK_OK;
}

}  // extern "C"
