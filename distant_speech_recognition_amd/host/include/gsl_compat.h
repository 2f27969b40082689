// gsl_compat.h -- layout-compatible minimal GSL vector/matrix PODs.
// The reference's node API traffics in gsl_vector_float / gsl_vector_complex / gsl_matrix_complex
// (stream/stream.h:76-86).  GSL is not a dependency of this engine; these PODs keep the
// {size, stride, data, block, owner} layout and the accessor names existing callers use
// (e.g. gsl_vector_float_get(data, i), src/beamformerDS.cc:193) so such code compiles unchanged.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>

typedef struct { double dat[2]; } gsl_complex;
#define GSL_REAL(z) ((z).dat[0])
#define GSL_IMAG(z) ((z).dat[1])
#define GSL_SET_COMPLEX(zp, x, y) do { (zp)->dat[0] = (x); (zp)->dat[1] = (y); } while (0)
static inline gsl_complex gsl_complex_rect(double x, double y) { gsl_complex z; z.dat[0] = x; z.dat[1] = y; return z; }
static inline gsl_complex gsl_complex_conjugate(gsl_complex a) { return gsl_complex_rect(a.dat[0], -a.dat[1]); }

#define BTK_GSL_VECTOR(NAME, T, MULT)                                                          \
  typedef struct { size_t size; T* data; } NAME##_block;                                       \
  typedef struct { size_t size; size_t stride; T* data; NAME##_block* block; int owner; } NAME; \
  static inline NAME* NAME##_calloc(size_t n) {                                                \
    NAME* v = (NAME*)malloc(sizeof(NAME));                                                     \
    v->block = (NAME##_block*)malloc(sizeof(NAME##_block));                                    \
    v->block->size = n; v->block->data = (T*)calloc(n ? n * MULT : 1, sizeof(T));               \
    v->size = n; v->stride = 1; v->data = v->block->data; v->owner = 1; return v; }            \
  static inline NAME* NAME##_alloc(size_t n) { return NAME##_calloc(n); }                      \
  static inline void NAME##_free(NAME* v) { if (!v) return; if (v->owner) { free(v->block->data); free(v->block); } free(v); } \
  static inline void NAME##_set_zero(NAME* v) { memset(v->data, 0, sizeof(T) * v->size * MULT); }

BTK_GSL_VECTOR(gsl_vector, double, 1)
BTK_GSL_VECTOR(gsl_vector_float, float, 1)
BTK_GSL_VECTOR(gsl_vector_short, short, 1)
BTK_GSL_VECTOR(gsl_vector_char, char, 1)
BTK_GSL_VECTOR(gsl_vector_complex, double, 2)

static inline double gsl_vector_get(const gsl_vector* v, size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_set(gsl_vector* v, size_t i, double x) { v->data[i * v->stride] = x; }
static inline float gsl_vector_float_get(const gsl_vector_float* v, size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_float_set(gsl_vector_float* v, size_t i, float x) { v->data[i * v->stride] = x; }
static inline gsl_complex gsl_vector_complex_get(const gsl_vector_complex* v, size_t i) {
  return gsl_complex_rect(v->data[2 * i * v->stride], v->data[2 * i * v->stride + 1]); }
static inline void gsl_vector_complex_set(gsl_vector_complex* v, size_t i, gsl_complex z) {
  v->data[2 * i * v->stride] = z.dat[0]; v->data[2 * i * v->stride + 1] = z.dat[1]; }

typedef struct { size_t size1, size2, tda; double* data; void* block; int owner; } gsl_matrix;
typedef struct { size_t size1, size2, tda; double* data; void* block; int owner; } gsl_matrix_complex;
static inline gsl_matrix* gsl_matrix_alloc(size_t n1, size_t n2) {
  gsl_matrix* m = (gsl_matrix*)malloc(sizeof(gsl_matrix)); m->size1 = n1; m->size2 = n2; m->tda = n2;
  m->data = (double*)calloc((n1 * n2) != 0 ? n1 * n2 : 1, sizeof(double)); m->block = NULL; m->owner = 1; return m; }
static inline void gsl_matrix_free(gsl_matrix* m) { if (m) { free(m->data); free(m); } }
static inline double gsl_matrix_get(const gsl_matrix* m, size_t i, size_t j) { return m->data[i * m->tda + j]; }
static inline void gsl_matrix_set(gsl_matrix* m, size_t i, size_t j, double x) { m->data[i * m->tda + j] = x; }
static inline gsl_matrix_complex* gsl_matrix_complex_alloc(size_t n1, size_t n2) {
  gsl_matrix_complex* m = (gsl_matrix_complex*)malloc(sizeof(gsl_matrix_complex)); m->size1 = n1; m->size2 = n2; m->tda = n2;
  m->data = (double*)calloc((n1 * n2) != 0 ? 2 * n1 * n2 : 1, sizeof(double)); m->block = NULL; m->owner = 1; return m; }
static inline void gsl_matrix_complex_free(gsl_matrix_complex* m) { if (m) { free(m->data); free(m); } }
static inline gsl_complex gsl_matrix_complex_get(const gsl_matrix_complex* m, size_t i, size_t j) {
  return gsl_complex_rect(m->data[2 * (i * m->tda + j)], m->data[2 * (i * m->tda + j) + 1]); }
static inline void gsl_matrix_complex_set(gsl_matrix_complex* m, size_t i, size_t j, gsl_complex z) {
  m->data[2 * (i * m->tda + j)] = z.dat[0]; m->data[2 * (i * m->tda + j) + 1] = z.dat[1]; }
