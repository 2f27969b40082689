// ref_driver.cc -- ORACLE / test infrastructure only.
// extern "C" entry points around the REFERENCE's own csvdc (J. Burkardt's C++ LINPACK,
// /root/reference/btk20_src/matrix/linpack_c.cc:9516), compiled from the sources where they
// lie by oracle/Makefile into oracle/_ref/libbtkref_linpack.so.  Nothing from the reference is
// copied into this repository; this file only forwards plain-C arrays to the reference symbol.
#include <complex>
#include <cstring>
using namespace std;
#include "matrix/blas1_c.h"
#include "matrix/linpack_c.h"

extern "C" {

// x: column-major n x p complex<float> (interleaved re,im), destroyed like the reference.
// s,e: n+p entries; u: ldu*n; v: ldv*p.  Returns csvdc's INFO.
int ref_csvdc(float* x, int ldx, int n, int p, float* s, float* e,
              float* u, int ldu, float* v, int ldv, int job)
{
  return csvdc(reinterpret_cast<complex<float>*>(x), ldx, n, p,
               reinterpret_cast<complex<float>*>(s), reinterpret_cast<complex<float>*>(e),
               reinterpret_cast<complex<float>*>(u), ldu,
               reinterpret_cast<complex<float>*>(v), ldv, job);
}

}
