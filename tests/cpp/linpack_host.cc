// linpack_host.cc -- TEST harness: the kernel body of csrc/svd_linpack.hip (csrc/linpack_f32.h, csvdc_values<Ctx>) compiled
// by g++ with a one-thread context, so the CPU suite can compare it bit for bit with the reference's own compiled csvdc
// (oracle/_ref) -- tests/test_linpack_f32.py.  Build: g++ -O2 -ffp-contract=off -shared -fPIC.
#include <vector>
#include <cstring>
#include "linpack_f32.h"

extern "C" int lpk_host_csvdc_values(const float* a /* [n][p] interleaved re, im, row-major */, int n, int p, float* s, float* e)
{
  using namespace lpk;
  std::vector<cf> x((size_t)n * p), col(n + 1), ev(p + 1), work(n + 1), sc(n + p + 2), ec(n + p + 2), t(2);
  std::memcpy(x.data(), a, sizeof(cf) * (size_t)n * p);
  int flag[2] = {0, 0};
  Work w{col.data(), ev.data(), work.data(), sc.data(), ec.data(), t.data(), flag};
  SerialCtx cx;
  return csvdc_values(cx, x.data(), p, n, p, w, s, e);
}
