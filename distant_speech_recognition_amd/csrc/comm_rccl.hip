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

namespace {
typedef int (*nccl_group_fn)();
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
struct Rccl { nccl_group_fn start = nullptr, end = nullptr; nccl_bcast_fn bcast = nullptr; nccl_errstr_fn errstr = nullptr; bool tried = false; };
Rccl g_rccl;

bool bind_rccl()
{
  if (g_rccl.tried) return g_rccl.bcast != nullptr;
  g_rccl.tried = true;
  void* h = RTLD_DEFAULT;
  if (!dlsym(h, "ncclBroadcast")) {
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
  }
  g_rccl.start = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupStart"));
  g_rccl.end = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupEnd"));
  g_rccl.bcast = reinterpret_cast<nccl_bcast_fn>(dlsym(h, "ncclBroadcast"));
  g_rccl.errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.start || !g_rccl.end || !g_rccl.bcast) { g_rccl.bcast = nullptr; return false; }
  return true;
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

}  // extern "C"
