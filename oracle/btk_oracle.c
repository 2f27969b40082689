/*
 * btk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A loop-faithful float64 restatement, in plain C, of the reference's subband
 * beamforming hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path
 * (distant_speech_recognition_amd/) never links, imports or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * btk20_src/ of kkumatani/distant_speech_recognition).
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - filter banks: pinned by the reference's own prototype fixtures
 *     (unit_test/prototype.ny/{h,g}-M256-m4-r1.pickle) through the
 *     analysis->synthesis reconstruction identity the reference's
 *     tools/filterbank/test_oversampled_dft_filter.py measures, and by the
 *     frame-count bookkeeping of modulated.cc.  The reference C++ itself needs
 *     GSL (absent) so it cannot be compiled here: output-level parity is
 *     otherwise UNPINNED for modulated.cc / beamformer.cc / postfilter.cc /
 *     dereverberation.cc.
 *   - blocking matrix, array manifold, NLMS canceller, covariance accumulation:
 *     pinned against the reference's own lib/pybeamformer.py executed in the
 *     dev container (tests/golden/gen_golden_pybeamformer.py).
 *   - pseudo-inverse: pinned against the reference's csvdc compiled from
 *     matrix/linpack_c.cc (oracle/_ref, see oracle/Makefile).
 *
 * Third-party arithmetic restated here because the dependency is absent from
 * /root/reference: GSL (unpinned version, btk20_src/CMakeLists.txt:54) --
 * radix-2 complex FFT, BLAS level-1/2 loops, sinc, complex Cholesky.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { double re, im; } cplx;

static inline cplx c_make(double r, double i) { cplx z = { r, i }; return z; }
static inline cplx c_add(cplx a, cplx b) { return c_make(a.re + b.re, a.im + b.im); }
static inline cplx c_sub(cplx a, cplx b) { return c_make(a.re - b.re, a.im - b.im); }
static inline cplx c_mul(cplx a, cplx b) { return c_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline cplx c_conj(cplx a) { return c_make(a.re, -a.im); }
static inline cplx c_scale(cplx a, double s) { return c_make(a.re * s, a.im * s); }
static inline double c_abs2(cplx a) { return a.re * a.re + a.im * a.im; }
static inline double c_abs(cplx a) { return hypot(a.re, a.im); }
static inline cplx c_div(cplx a, cplx b)
{
  double d = b.re * b.re + b.im * b.im;
  return c_make((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}
static inline cplx c_polar(double r, double th) { return c_make(r * cos(th), r * sin(th)); }

/* ------------------------------------------------------------------------
 * GSL stand-in arithmetic: in-place radix-2 complex FFT, unnormalised.
 * sign=+1 == gsl_fft_complex_radix2_backward (e^{+j2pi kn/N}),
 * sign=-1 == gsl_fft_complex_radix2_forward  (modulated.cc:396, :559).
 * data is interleaved re,im; n must be a power of two.
 * ---------------------------------------------------------------------- */
static void fft_radix2(double* data, unsigned n, int sign)
{
  unsigned j = 0;
  for (unsigned i = 0; i + 1 < n; i++) {          /* bit reversal permutation */
    if (i < j) {
      double tr = data[2*i], ti = data[2*i+1];
      data[2*i] = data[2*j]; data[2*i+1] = data[2*j+1];
      data[2*j] = tr; data[2*j+1] = ti;
    }
    unsigned k = n >> 1;
    while (k <= j && k > 0) { j -= k; k >>= 1; }
    j += k;
  }
  for (unsigned len = 2; len <= n; len <<= 1) {   /* decimation in time */
    unsigned half = len >> 1;
    for (unsigned b = 0; b < half; b++) {
      double ang = sign * 2.0 * M_PI * (double)b / (double)len;
      double wr = cos(ang), wi = sin(ang);
      for (unsigned s = b; s < n; s += len) {
        unsigned t = s + half;
        double xr = wr * data[2*t] - wi * data[2*t+1];
        double xi = wr * data[2*t+1] + wi * data[2*t];
        data[2*t]   = data[2*s]   - xr;
        data[2*t+1] = data[2*s+1] - xi;
        data[2*s]   += xr;
        data[2*s+1] += xi;
      }
    }
  }
}

/* ------------------------------------------------------------------------
 * RealBuffer_ : modulated/modulated.h:56-140.  nsamp rows of len doubles;
 * zero_ is the index of the most recent row, sample(t,i) is the row written
 * t calls ago.  The integer index arithmetic here is the bit-exact contract.
 * ---------------------------------------------------------------------- */
typedef struct {
  unsigned len, nsamp, zero;
  double* rows;                    /* [nsamp][len] */
} realbuf;

static void rb_init(realbuf* b, unsigned len, unsigned nsamp)
{
  b->len = len; b->nsamp = nsamp; b->zero = nsamp - 1;          /* modulated.h:66 */
  b->rows = (double*)calloc((size_t)len * nsamp, sizeof(double));
}
static void rb_free(realbuf* b) { free(b->rows); b->rows = NULL; }
static void rb_zero(realbuf* b)
{                                                              /* modulated.h:123-127 */
  memset(b->rows, 0, sizeof(double) * (size_t)b->len * b->nsamp);
  b->zero = b->nsamp - 1;
}
static unsigned rb_index(const realbuf* b, unsigned t)
{                                                              /* modulated.h:130-134 */
  return (b->zero + b->nsamp - t) % b->nsamp;
}
static double rb_sample(const realbuf* b, unsigned t, unsigned i)
{                                                              /* modulated.h:80-84 */
  return b->rows[(size_t)rb_index(b, t) * b->len + i];
}
/* nextSample(const gsl_vector*, reverse): modulated.h:86-103; s==NULL -> zeros */
static void rb_next(realbuf* b, const double* s, int reverse)
{
  b->zero = (b->zero + 1) % b->nsamp;
  double* row = b->rows + (size_t)b->zero * b->len;
  if (!s) { memset(row, 0, sizeof(double) * b->len); return; }
  if (reverse) for (unsigned i = 0; i < b->len; i++) row[i] = s[b->len - i - 1];
  else memcpy(row, s, sizeof(double) * b->len);
}
/* nextSample(const gsl_vector_float*): modulated.h:105-113 */
static void rb_next_f(realbuf* b, const float* s)
{
  b->zero = (b->zero + 1) % b->nsamp;
  double* row = b->rows + (size_t)b->zero * b->len;
  for (unsigned i = 0; i < b->len; i++) row[i] = s[i];
}

/* exported so tests can pin the ring index arithmetic bit-exactly */
unsigned orc_ring_index(unsigned zero, unsigned nsamp, unsigned t)
{
  return (zero + nsamp - t) % nsamp;
}

/* ------------------------------------------------------------------------
 * PCM block source == SampleFeature(block_len=D, shift_len=D, pad_zeros=true)
 * feature/feature.cc:605-649.  returns 1 (jiterator_error) at end.
 * ---------------------------------------------------------------------- */
typedef struct {
  const float* samples; long total; long cur; unsigned D; int is_end; float* vec;
} pcm_source;

static int pcm_next(pcm_source* s)
{
  if (s->is_end) return 1;                                      /* :608-610 */
  if (s->cur >= s->total) { s->is_end = 1; return 1; }          /* :618-625 */
  if (s->cur + (long)s->D >= s->total) {                        /* :627-632, pad_zeros */
    memset(s->vec, 0, sizeof(float) * s->D);
    long rem = s->total - s->cur;
    for (long i = 0; i < rem; i++) s->vec[i] = s->samples[s->cur + i];
  } else {
    for (unsigned i = 0; i < s->D; i++) s->vec[i] = s->samples[s->cur + i];
  }
  s->cur += s->D;
  return 0;
}

/* ------------------------------------------------------------------------
 * OverSampledDFTFilterBank ctor delay logic: modulated/modulated.cc:232-268
 * ---------------------------------------------------------------------- */
void orc_fb_delays(unsigned m, unsigned r, int synthesis, unsigned dct,
                   unsigned* processing_delay, unsigned* laN)
{
  unsigned R = 1u << r;
  *laN = 0;
  switch (dct) {
  case 1:  *processing_delay = m * R - 1; break;                /* :248-250 */
  case 2:                                                       /* :251-258 */
    if (synthesis) *processing_delay = m * R / 2;
    else { *processing_delay = m * R - 1; *laN = m * R / 2 - 1; }
    break;
  default: *processing_delay = 2 * m - 1; break;                /* :260-262 */
  }
}

/* ------------------------------------------------------------------------
 * OverSampledDFTAnalysisBank : modulated/modulated.cc:311-469
 * ---------------------------------------------------------------------- */
typedef struct {
  unsigned M, m, r, R, D, pd, laN;
  int gain_factor;
  double* proto;                    /* [m*M] copy, modulated.cc:243-244 */
  realbuf buffer;                   /* (M, m*R) */
  realbuf gsi;                      /* (D, R)   */
  double* convert;                  /* [M] */
  double* ppout;                    /* [2M] polyphase_output_ */
  pcm_source src;
  int frame_no, is_end;
  unsigned frames_padded;
} orc_analysis;

orc_analysis* orc_analysis_new(const double* proto, unsigned M, unsigned m, unsigned r,
                               unsigned dct, const float* pcm, long len)
{
  orc_analysis* a = (orc_analysis*)calloc(1, sizeof(*a));
  a->M = M; a->m = m; a->r = r; a->R = 1u << r; a->D = M / a->R;
  orc_fb_delays(m, r, 0, dct, &a->pd, &a->laN);
  a->gain_factor = 1;
  a->proto = (double*)malloc(sizeof(double) * m * M);
  memcpy(a->proto, proto, sizeof(double) * m * M);
  rb_init(&a->buffer, M, m * a->R);
  rb_init(&a->gsi, a->D, a->R);
  a->convert = (double*)calloc(M, sizeof(double));
  a->ppout = (double*)calloc(2 * M, sizeof(double));
  a->src.samples = pcm; a->src.total = len; a->src.cur = 0; a->src.D = a->D;
  a->src.is_end = 0; a->src.vec = (float*)calloc(a->D, sizeof(float));
  a->frame_no = -1; a->is_end = 0; a->frames_padded = 0;
  return a;
}

void orc_analysis_free(orc_analysis* a)
{
  if (!a) return;
  free(a->proto); rb_free(&a->buffer); rb_free(&a->gsi);
  free(a->convert); free(a->ppout); free(a->src.vec); free(a);
}

/* update_buf_: modulated.cc:363-373 */
static void analysis_update_buf(orc_analysis* a)
{
  for (unsigned s = 0; s < a->R; s++)
    for (unsigned d = 0; d < a->D; d++)
      a->convert[d + s * a->D] = rb_sample(&a->gsi, a->R - s - 1, d);
  rb_next(&a->buffer, a->convert, /*reverse=*/1);
}

/* update_buffer_: modulated.cc:419-469 (frame_no < 0 path: "take next frame") */
static int analysis_update_buffer(orc_analysis* a)
{
  if (a->is_end) return 1;
  if (a->laN > 0 && a->frame_no == -1) {                        /* :425-439 look-ahead skip */
    for (unsigned it = 0; it < a->laN; it++) {
      if (pcm_next(&a->src)) a->is_end = 1;
      if (!a->is_end) { rb_next_f(&a->gsi, a->src.vec); analysis_update_buf(a); }
    }
  }
  if (a->frames_padded == 0) {                                  /* :440-457 */
    if (pcm_next(&a->src)) a->frames_padded++;
    if (a->frames_padded == 0) rb_next_f(&a->gsi, a->src.vec);
    else rb_next(&a->gsi, NULL, 0);
    analysis_update_buf(a);
  } else if (a->frames_padded < a->pd) {                        /* :458-462 */
    rb_next(&a->gsi, NULL, 0);
    analysis_update_buf(a);
    a->frames_padded++;
  } else {
    a->is_end = 1;                                              /* :463-465 */
  }
  return a->is_end;
}

/* next(): modulated.cc:375-409. out = M complex (interleaved). polyphase (optional)
   receives the M real polyphase sums before the FFT (used to pin indexing). */
int orc_analysis_next(orc_analysis* a, double* out, double* polyphase)
{
  if (analysis_update_buffer(a)) return 1;                      /* jiterator_error */
  for (unsigned i = 0; i < a->M; i++) {                         /* :384-391 */
    double sum = 0.0;
    for (unsigned k = 0; k < a->m; k++)
      sum += a->proto[i + a->M * k] * rb_sample(&a->buffer, a->R * k, i);
    a->ppout[2*i] = sum; a->ppout[2*i+1] = 0.0;
    if (polyphase) polyphase[i] = sum;
  }
  fft_radix2(a->ppout, a->M, +1);                               /* :396 backward */
  for (unsigned i = 0; i < 2 * a->M; i++) out[i] = a->ppout[i];
  if (a->gain_factor > 0)                                       /* :400-404 */
    for (unsigned i = 0; i < 2 * a->M; i++) out[i] *= a->gain_factor;
  a->frame_no++;
  return 0;
}

/* Whole-utterance driver: returns the number of frames written (<= max_frames). */
long orc_analysis_run(const double* proto, unsigned M, unsigned m, unsigned r, unsigned dct,
                      const float* pcm, long len, double* out, double* polyphase, long max_frames)
{
  orc_analysis* a = orc_analysis_new(proto, M, m, r, dct, pcm, len);
  long t = 0;
  while (t < max_frames) {
    if (orc_analysis_next(a, out + (size_t)2 * M * t, polyphase ? polyphase + (size_t)M * t : NULL)) break;
    t++;
  }
  orc_analysis_free(a);
  return t;
}

/* ------------------------------------------------------------------------
 * OverSampledDFTSynthesisBank : modulated/modulated.cc:474-621
 * Input is an array source of T complex frames (the upstream node).
 * ---------------------------------------------------------------------- */
typedef struct {
  unsigned M, m, r, R, D, pd;
  int gain_factor;
  double* proto;
  realbuf buffer;                  /* (M, m*R) */
  realbuf gsi;                     /* (M, R)   */
  double* convert;                 /* [M] */
  double* ppin;                    /* [2M] */
  const double* in; long T; long in_pos;   /* upstream frames [T][2M] */
  int frame_no, is_end;
  float* vec;                      /* [D] output block (float32 like vector_) */
} orc_synthesis;

orc_synthesis* orc_synthesis_new(const double* proto, unsigned M, unsigned m, unsigned r,
                                 unsigned dct, int gain_factor, const double* frames, long T)
{
  orc_synthesis* s = (orc_synthesis*)calloc(1, sizeof(*s));
  unsigned la;
  s->M = M; s->m = m; s->r = r; s->R = 1u << r; s->D = M / s->R;
  orc_fb_delays(m, r, 1, dct, &s->pd, &la);
  s->gain_factor = gain_factor;
  s->proto = (double*)malloc(sizeof(double) * m * M);
  memcpy(s->proto, proto, sizeof(double) * m * M);
  rb_init(&s->buffer, M, m * s->R);
  rb_init(&s->gsi, M, s->R);
  s->convert = (double*)calloc(M, sizeof(double));
  s->ppin = (double*)calloc(2 * M, sizeof(double));
  s->in = frames; s->T = T; s->in_pos = 0;
  s->frame_no = -1; s->is_end = 0;
  s->vec = (float*)calloc(s->D, sizeof(float));
  return s;
}

void orc_synthesis_free(orc_synthesis* s)
{
  if (!s) return;
  free(s->proto); rb_free(&s->buffer); rb_free(&s->gsi);
  free(s->convert); free(s->ppin); free(s->vec); free(s);
}

/* update_buf_(block): modulated.cc:553-567 */
static void synthesis_update_buf(orc_synthesis* s, const double* block)
{
  memcpy(s->ppin, block, sizeof(double) * 2 * s->M);
  fft_radix2(s->ppin, s->M, -1);                                /* :559 forward */
  for (unsigned i = 0; i < s->M; i++) s->convert[i] = s->ppin[2*i];   /* real part only */
  rb_next(&s->buffer, s->convert, 0);
}

/* update_buffer_: modulated.cc:533-551 */
static int synthesis_update_buffer(orc_synthesis* s)
{
  if (s->in_pos >= s->T) { s->is_end = 1; return 1; }
  synthesis_update_buf(s, s->in + (size_t)2 * s->M * s->in_pos);
  s->in_pos++;
  return 0;
}

/* next(): modulated.cc:569-612.  out = D floats. returns 1 at end of samples. */
int orc_synthesis_next(orc_synthesis* s, float* out)
{
  if (s->frame_no == -1)                                        /* :574-578 prime */
    for (unsigned it = 0; it < s->pd; it++)
      if (synthesis_update_buffer(s)) return 1;
  if (synthesis_update_buffer(s)) return 1;                     /* :583-590 */
  s->frame_no++;
  for (unsigned i = 0; i < s->M; i++) {                         /* :594-599 */
    double sum = 0.0;
    for (unsigned k = 0; k < s->m; k++)
      sum += s->proto[(s->M - i - 1) + s->M * k] * rb_sample(&s->buffer, s->R * k, i);
    s->convert[i] = sum;
  }
  rb_next(&s->gsi, s->convert, 0);                              /* :600 */
  memset(s->vec, 0, sizeof(float) * s->D);                      /* :603 */
  for (unsigned j = 0; j < s->R; j++)                           /* :604-606 float32 running sum */
    for (unsigned d = 0; d < s->D; d++)
      s->vec[s->D - d - 1] = (float)(s->vec[s->D - d - 1] + rb_sample(&s->gsi, s->R - j - 1, d + j * s->D));
  if (s->gain_factor > 0)                                       /* :608-609 */
    for (unsigned d = 0; d < s->D; d++) s->vec[d] *= (float)s->gain_factor;
  memcpy(out, s->vec, sizeof(float) * s->D);
  return 0;
}

long orc_synthesis_run(const double* proto, unsigned M, unsigned m, unsigned r, unsigned dct,
                       int gain_factor, const double* frames, long T, float* out, long max_blocks)
{
  orc_synthesis* s = orc_synthesis_new(proto, M, m, r, dct, gain_factor, frames, T);
  long b = 0;
  while (b < max_blocks) {
    if (orc_synthesis_next(s, out + (size_t)s->D * b)) break;
    b++;
  }
  orc_synthesis_free(s);
  return b;
}

/* ------------------------------------------------------------------------
 * BeamformerWeights::calcMainlobe (halfBandShift == false)
 * beamformer/beamformer.cc:502-565.  wq is [M][N] complex; samplerate is a
 * float in the reference signature (beamformer.h), delays are doubles.
 * ---------------------------------------------------------------------- */
void orc_calc_mainlobe(unsigned M, unsigned N, float samplerate, const double* delays, cplx* wq)
{
  unsigned M2 = M / 2;
  for (unsigned c = 0; c < N; c++)                               /* :533-535 */
    wq[c] = c_scale(c_polar(1.0, 0.0), 1.0 / N);
  for (unsigned k = 1; k < M2; k++) {                            /* :537-545 */
    for (unsigned c = 0; c < N; c++) {
      double val = -2.0 * M_PI * k * delays[c] * samplerate / M;
      cplx p = c_polar(1.0, val), q = c_polar(1.0, -val);
      wq[(size_t)k * N + c]       = c_make(p.re / N, p.im / N);
      wq[(size_t)(M - k) * N + c] = c_make(q.re / N, q.im / N);
    }
  }
  for (unsigned c = 0; c < N; c++) {                             /* :547-551 */
    double val = -M_PI * samplerate * delays[c];
    cplx p = c_polar(1.0, val);
    wq[(size_t)M2 * N + c] = c_make(p.re / N, p.im / N);
  }
}


/* ------------------------------------------------------------------------
 * LCMV quiescent weights: calc_inverse_22mat_ (beamformer.cc:181-221),
 * calc_null_beamformer_ (:299-363) and BeamformerWeights::calcMainlobeN (:600-721),
 * halfBandShift == false, NC == 2 (the closed-form 2x2 inverse; NC > 2 takes the
 * float32 SVD pseudo-inverse in the reference and is handled by the Python side of the oracle).
 * ---------------------------------------------------------------------- */
static void inverse_22(cplx m[4], double beta)
{                                                                   /* beamformer.cc:181-221 */
  cplx m00 = m[0], m01 = m[1], m10 = m[2], m11 = m[3];
  cplx det = c_sub(c_mul(m00, m11), c_mul(m01, m10));
  if (c_abs(det) < 1.0E-07) {                                       /* MINDET_THRESHOLD */
    m00.re += beta; m11.re += beta;
    det = c_sub(c_mul(m00, m11), c_mul(m01, m10));
  }
  m[0] = c_div(m11, det);
  m[3] = c_div(m00, det);
  m[1] = c_scale(c_div(m01, det), -1.0);
  m[2] = c_scale(c_div(m10, det), -1.0);
}

/* wt (in/out, target manifold), wj interference manifold; wt <- C (C^H C)^-1 g, g = (1,0) */
static void null_beamformer2(cplx* wt, const cplx* wj, unsigned N)
{                                                                   /* beamformer.cc:299-363 with NC = 2 */
  cplx G[4] = { {0,0}, {0,0}, {0,0}, {0,0} };
  for (unsigned i = 0; i < N; i++) {                                /* C^H C */
    G[0] = c_add(G[0], c_mul(c_conj(wt[i]), wt[i]));
    G[1] = c_add(G[1], c_mul(c_conj(wt[i]), wj[i]));
    G[2] = c_add(G[2], c_mul(c_conj(wj[i]), wt[i]));
    G[3] = c_add(G[3], c_mul(c_conj(wj[i]), wj[i]));
  }
  inverse_22(G, 0.01);
  cplx v0 = G[0], v1 = G[2];                                        /* inv * (1,0)^T */
  for (unsigned i = 0; i < N; i++) wt[i] = c_add(c_mul(wt[i], v0), c_mul(wj[i], v1));
}

void orc_calc_mainlobe_2(unsigned M, unsigned N, float samplerate, const double* delaysT, const double* delaysI, cplx* wq)
{
  unsigned M2 = M / 2;
  cplx* pWj = (cplx*)calloc(N, sizeof(cplx));
  orc_calc_mainlobe(M, N, samplerate, delaysT, wq);                 /* :638 */
  for (unsigned c = 0; c < N; c++) wq[c] = c_make(1.0 / N, 0.0);    /* :665-667 */
  for (unsigned k = 1; k < M2; k++) {                               /* :670-689 */
    cplx* vec = wq + (size_t)k * N;
    for (unsigned c = 0; c < N; c++) {
      vec[c] = c_scale(vec[c], (double)N);
      double valJ = -2.0 * M_PI * k * samplerate * delaysI[c] / M;
      pWj[c] = c_polar(1.0, valJ);
    }
    null_beamformer2(vec, pWj, N);
  }
  {                                                                 /* :692-703, bin M/2, reproduced literally: */
    cplx* vec = wq + (size_t)M2 * N;                                /* the interferer's manifold is written into */
    for (unsigned c = 0; c < N; c++) {                              /* vec and the solve runs inside the channel */
      vec[c] = c_scale(vec[c], (double)N);                          /* loop with pWj left over from bin M/2-1    */
      double val = -M_PI * samplerate * delaysI[c];
      cplx p = c_polar(1.0, val);
      vec[c] = c_make(p.re / N, p.im / N);
      null_beamformer2(vec, pWj, N);
    }
  }
  free(pWj);
}

/* BLAS level-1 helpers written as the loops GSL's CBLAS performs. */
static double dznrm2(const cplx* x, unsigned n)
{
  /* scaled 2-norm as in reference BLAS dznrm2 */
  double scale = 0.0, ssq = 1.0;
  for (unsigned i = 0; i < n; i++) {
    double v[2] = { x[i].re, x[i].im };
    for (int p = 0; p < 2; p++) {
      if (v[p] != 0.0) {
        double a = fabs(v[p]);
        if (scale < a) { ssq = 1.0 + ssq * (scale / a) * (scale / a); scale = a; }
        else ssq += (a / scale) * (a / scale);
      }
    }
  }
  return scale * sqrt(ssq);
}
static cplx zdotc(const cplx* x, const cplx* y, unsigned n)     /* sum conj(x_i) y_i */
{
  cplx s = c_make(0, 0);
  for (unsigned i = 0; i < n; i++) s = c_add(s, c_mul(c_conj(x[i]), y[i]));
  return s;
}

/* calc_blocking_matrix_: beamformer/beamformer.cc:373-454
 * (== calc_blocking_matrix, lib/pybeamformer.py:309-341).
 * a = arrayManifold [N]; B is row-major [N][N-NC]. returns 0 on success. */
int orc_blocking_matrix(const cplx* a, unsigned N, unsigned NC, cplx* B)
{
  int bsize = (int)N - (int)NC;
  if (bsize <= 0) return -1;                                     /* :380-383 */
  memset(B, 0, sizeof(cplx) * N * bsize);
  cplx* P = (cplx*)calloc((size_t)N * N, sizeof(cplx));
  cplx* vec = (cplx*)calloc(N, sizeof(cplx));
  cplx* rvec = (cplx*)calloc(N, sizeof(cplx));
  double norm_vs = dznrm2(a, N);
  norm_vs = norm_vs * norm_vs;                                   /* :401-402 */
  double alpha = -1.0 / norm_vs;
  for (unsigned i = 0; i < N; i++) P[(size_t)i * N + i] = c_make(1.0, 0.0);
  for (unsigned i = 0; i < N; i++)                               /* zgeru: P += alpha*conj(a) a^T, :406-409 */
    for (unsigned j = 0; j < N; j++)
      P[(size_t)i * N + j] = c_add(P[(size_t)i * N + j], c_scale(c_mul(c_conj(a[i]), a[j]), alpha));
  for (int idim = 0; idim < bsize; idim++) {                     /* :411-432 classical Gram-Schmidt */
    for (unsigned i = 0; i < N; i++) vec[i] = P[(size_t)i * N + idim];
    for (int jdim = 0; jdim < idim; jdim++) {
      for (unsigned i = 0; i < N; i++) rvec[i] = B[(size_t)i * bsize + jdim];
      cplx ip = zdotc(rvec, vec, N);
      ip = c_scale(ip, -1.0);
      for (unsigned i = 0; i < N; i++) vec[i] = c_add(vec[i], c_mul(ip, rvec[i]));   /* zaxpy */
    }
    double nv = dznrm2(vec, N);
    for (unsigned i = 0; i < N; i++) B[(size_t)i * bsize + idim] = c_scale(vec[i], 1.0 / nv);
  }
  free(P); free(vec); free(rvec);
  return 0;
}

/* calcSidelobeCancellerU_f: beamformer.cc:752-767.  wl = B * wa (zgemv NoTrans) */
void orc_sidelobe_canceller(const cplx* B, const cplx* wa, unsigned N, unsigned NC, cplx* wl)
{
  unsigned bs = N - NC;
  for (unsigned i = 0; i < N; i++) {
    cplx s = c_make(0, 0);
    for (unsigned j = 0; j < bs; j++) s = c_add(s, c_mul(B[(size_t)i * bs + j], wa[j]));
    wl[i] = s;
  }
}

/* SnapShotArray::update: beamformer.cc:62-70.  samples [N][M] -> snapshots [M][N] */
void orc_snapshot_update(const cplx* samples, unsigned M, unsigned N, cplx* snapshots)
{
  for (unsigned k = 0; k < M; k++)
    for (unsigned c = 0; c < N; c++)
      snapshots[(size_t)k * N + c] = samples[(size_t)c * M + k];
}

/* calc_gsc_output: beamformer.cc:1208-1243 */
static cplx gsc_output(const cplx* x, const cplx* wl, const cplx* wq, unsigned N, int normalize, cplx* tmp)
{
  for (unsigned i = 0; i < N; i++) tmp[i] = c_sub(wq[i], wl[i]);
  if (normalize) {                                               /* :1228-1237 */
    double norm = dznrm2(tmp, N);
    for (unsigned i = 0; i < N; i++) tmp[i] = c_make(tmp[i].re / (norm * N), tmp[i].im / (norm * N));
  }
  return zdotc(tmp, x, N);
}

/* SubbandGSC::next body for one frame (halfBandShift == false): beamformer.cc:1286-1311.
 * snapshots [M][N], wq [M][N], wl [M][N] (wl may be NULL == SubbandDS::next, :1132-1151),
 * out [M]. */
void orc_gsc_frame(const cplx* snapshots, const cplx* wq, const cplx* wl, unsigned M, unsigned N,
                   int normalize, cplx* out)
{
  unsigned M2 = M / 2;
  cplx* tmp = (cplx*)malloc(sizeof(cplx) * N);                   /* the per-bin temp of :1214 */
  out[0] = zdotc(wq, snapshots, N);                              /* :1288-1291 */
  for (unsigned k = 1; k <= M2; k++) {
    cplx val;
    if (wl) val = gsc_output(snapshots + (size_t)k * N, wl + (size_t)k * N, wq + (size_t)k * N, N, normalize, tmp);
    else val = zdotc(wq + (size_t)k * N, snapshots + (size_t)k * N, N);
    if (k < M2) { out[k] = val; out[M - k] = c_conj(val); }
    else out[M2] = val;
  }
  free(tmp);
}

/* ------------------------------------------------------------------------
 * Zelinski post-filter: postfilter/postfilter.cc:8-219, 424-491
 * CSDs is the beamformer's BeamformerWeights::CSDs_ state [M][N*N]
 * (beamformer.cc:874-887), wp1 [M] the post-filter weights.
 * ---------------------------------------------------------------------- */
#define ORC_SPECTRAL_FLOOR 0.0001
#define ORC_TYPE_ZELINSKI1_REAL 0x01

/* ZelinskiFilter_f: postfilter.cc:57-140 */
static double zelinski_f(const cplx* d, const cplx* x, unsigned N, cplx* csd, double alpha, int pftype, cplx* ta)
{
  double num, den = 0.0;
  for (unsigned i = 0; i < N; i++) ta[i] = c_mul(c_conj(d[i]), x[i]);      /* time_alignment_: :30-43 */
  cplx sum = c_make(0, 0);
  for (unsigned i = 0; i + 1 < N; i++)                                     /* :77-87 */
    for (unsigned j = i + 1; j < N; j++) {
      unsigned idx = i * N + j;
      cplx xx = c_mul(ta[i], c_conj(ta[j]));
      cplx est = (alpha > 0.0) ? c_add(c_scale(csd[idx], alpha), c_scale(xx, 1.0 - alpha)) : xx;   /* calc_CSD_: :8-21 */
      sum = c_add(sum, est);
      csd[idx] = est;
    }
  if (pftype & ORC_TYPE_ZELINSKI1_REAL) { num = sum.re; if (num < 0.0) num = 0.0; }   /* :90-98 */
  else num = c_abs(sum);
  for (unsigned i = 0; i < N; i++) {                                       /* :100-116 */
    unsigned idx = i * N + i;
    double est = (alpha > 0.0) ? alpha * csd[idx].re + (1.0 - alpha) * c_abs2(ta[i]) : c_abs2(ta[i]);
    den += est;
    csd[idx] = c_make(est, 0.0);
  }
  double W = (num / den) * (2.0 / (N - 1.0));                              /* :118-121 */
  if (W >= 1.0) W = 1.0;
  if (W < ORC_SPECTRAL_FLOOR) W = ORC_SPECTRAL_FLOOR;
  return W;
}

/* ZelinskiPostFilter::next for one frame: postfilter.cc:424-491 + ZelinskiFilter :157-219.
 * frame_no_pre is frame_no_ BEFORE increment_ (starts at -1).  y [M] is copied from the
 * beamformer output and filtered in place.  d == wq if (type & 8) else ta_ (:452-457);
 * callers pass the right array. */
void orc_zelinski_frame(const cplx* d, const cplx* snapshots, unsigned M, unsigned N,
                        cplx* csds, cplx* wp1, double alpha_cfg, int type, int min_frames,
                        int frame_no_pre, cplx* y)
{
  unsigned M2 = M / 2;
  double alpha = (frame_no_pre > 0) ? alpha_cfg : 0.0;                     /* :460-463 */
  int pftype = (frame_no_pre < min_frames) ? 0 : type;                     /* :468-473 */
  cplx* ta = (cplx*)malloc(sizeof(cplx) * N);
  for (unsigned k = 0; k <= M2; k++) {                                     /* :184-194 */
    double r = zelinski_f(d + (size_t)k * N, snapshots + (size_t)k * N, N, csds + (size_t)k * N * N, alpha, pftype, ta);
    wp1[k] = c_make(r, 0.0);
    if (k > 0 && k < M2) wp1[M - k] = c_make(r, -0.0);
  }
  free(ta);
  if (pftype == 0) return;                                                 /* :197-199 */
  for (unsigned k = 0; k <= M2; k++) {                                     /* :209-216 */
    cplx o = c_mul(wp1[k], y[k]);
    y[k] = o;
    if (k > 0 && k < M2) y[M - k] = c_conj(o);
  }
}

/* ------------------------------------------------------------------------
 * SubbandGSCLMSBeamformer.__iter__ : lib/pybeamformer.py:659-734
 * State mirrors reset_stats (:745-758).  One call == one frame.
 * BmH [K][N-Nc][N] (= transpose(B), :742), wqH [K][N] (= conj(vs), :743).
 * ---------------------------------------------------------------------- */
typedef struct {
  unsigned M, N, Nc;
  double beta, init_gamma, init_diagonal_load, reg, energy_floor, sil_thresh, max_wa_l2norm;
  int min_frames, slowdown_after;
  int isamp, ttl_updates;
  double gamma, energy;
  double* subband_energy;          /* [K] */
  cplx* waH;                       /* [K][N-Nc] */
} orc_nlms;

orc_nlms* orc_nlms_new(unsigned M, unsigned N, unsigned Nc, double beta, double gamma,
                       double init_diagonal_load, double reg, double energy_floor,
                       double sil_thresh, double max_wa_l2norm, int min_frames, int slowdown_after)
{
  orc_nlms* s = (orc_nlms*)calloc(1, sizeof(*s));
  unsigned K = M / 2 + 1;
  s->M = M; s->N = N; s->Nc = Nc; s->beta = beta; s->init_gamma = gamma;
  s->init_diagonal_load = init_diagonal_load; s->reg = reg; s->energy_floor = energy_floor;
  s->sil_thresh = sil_thresh; s->max_wa_l2norm = max_wa_l2norm;
  s->min_frames = min_frames; s->slowdown_after = slowdown_after;
  s->subband_energy = (double*)malloc(sizeof(double) * K);
  s->waH = (cplx*)calloc((size_t)K * (N - Nc), sizeof(cplx));
  /* reset_stats: pybeamformer.py:745-758 */
  s->isamp = 0; s->ttl_updates = 0; s->gamma = gamma; s->energy = init_diagonal_load;
  for (unsigned k = 0; k < K; k++) s->subband_energy[k] = init_diagonal_load;
  return s;
}
void orc_nlms_free(orc_nlms* s) { if (s) { free(s->subband_energy); free(s->waH); free(s); } }
cplx* orc_nlms_wa(orc_nlms* s) { return s->waH; }
double* orc_nlms_subband_energy(orc_nlms* s) { return s->subband_energy; }
double orc_nlms_energy(const orc_nlms* s) { return s->energy; }

/* samples [N][M] are the analysis outputs of this frame; snapshots [M][N] their transpose. */
void orc_nlms_frame(orc_nlms* s, const cplx* samples, const cplx* snapshots,
                    const cplx* BmH, const cplx* wqH, cplx* out)
{
  unsigned M = s->M, N = s->N, bs = s->N - s->Nc, M2 = M / 2;
  /* update_snapshot_array(chan_no=0): pybeamformer.py:263-277 ; :665 */
  cplx e0 = zdotc(samples, samples, M);
  double energy = c_abs(e0) / M;
  memset(out, 0, sizeof(cplx) * M);
  if (s->isamp > 0 && (s->isamp % s->slowdown_after) == 0) s->gamma /= 2.0;       /* :668-670 */
  int adapt = energy > (s->energy / s->sil_thresh);
  if (adapt) s->ttl_updates++;                                                     /* :672-673 */
  cplx* ZK = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* wat = (cplx*)malloc(sizeof(cplx) * bs);
  for (unsigned k = 0; k <= M2; k++) {
    const cplx* XK = snapshots + (size_t)k * N;
    const cplx* Bk = BmH + (size_t)k * bs * N;
    cplx* wa = s->waH + (size_t)k * bs;
    for (unsigned i = 0; i < bs; i++) {                                            /* :677 ZK = BmH . XK */
      cplx z = c_make(0, 0);
      for (unsigned c = 0; c < N; c++) z = c_add(z, c_mul(Bk[(size_t)i * N + c], XK[c]));
      ZK[i] = z;
    }
    cplx YcK = c_make(0, 0);                                                       /* :679 */
    for (unsigned c = 0; c < N; c++) YcK = c_add(YcK, c_mul(wqH[(size_t)k * N + c], XK[c]));
    double xx = c_abs(zdotc(XK, XK, N));
    double se = (s->isamp > 0) ? s->subband_energy[k] * s->beta + (1.0 - s->beta) * xx : xx;   /* :682-685 */
    if (se < s->energy_floor) se = s->energy_floor;                                /* :687-688 */
    if (adapt) {                                                                   /* :690-720 */
      cplx dot = c_make(0, 0);
      for (unsigned i = 0; i < bs; i++) dot = c_add(dot, c_mul(wa[i], ZK[i]));
      cplx epa = c_sub(YcK, dot);
      double alphaK = s->gamma / se;
      for (unsigned i = 0; i < bs; i++)
        wat[i] = c_add(wa[i], c_scale(c_mul(epa, c_conj(ZK[i])), alphaK));
      if (s->reg > 0)
        for (unsigned i = 0; i < bs; i++)
          wat[i] = c_sub(wat[i], c_scale(wa[i], alphaK * s->reg));
      cplx nn = c_make(0, 0);
      for (unsigned i = 0; i < bs; i++) nn = c_add(nn, c_mul(wat[i], c_conj(wat[i])));
      double norm = c_abs(nn);
      if (norm > s->max_wa_l2norm) {
        double cK = sqrt(s->max_wa_l2norm / norm);
        for (unsigned i = 0; i < bs; i++) wa[i] = c_scale(wat[i], cK);
      } else {
        for (unsigned i = 0; i < bs; i++) wa[i] = wat[i];
      }
      s->subband_energy[k] = se;
    }
    if (s->isamp >= s->min_frames) {                                               /* :723-726 */
      cplx dot = c_make(0, 0);
      for (unsigned i = 0; i < bs; i++) dot = c_add(dot, c_mul(wa[i], ZK[i]));
      out[k] = c_sub(YcK, dot);
    } else out[k] = YcK;
    if (k > 0 && k < M2) out[M - k] = c_conj(out[k]);                              /* :727-728 */
  }
  s->energy = s->energy * s->beta + (1.0 - s->beta) * energy;                      /* :731 */
  s->isamp++;
  free(ZK); free(wat);
}

/* ------------------------------------------------------------------------
 * Covariance accumulation: lib/pybeamformer.py:967-985 (label version) and
 * :1136-1147 (TF-mask version).  R [K][N][N] += w * x x^H.
 * ---------------------------------------------------------------------- */
void orc_cov_accumulate_frame(const cplx* snapshots, unsigned M, unsigned N,
                              const double* mask_k /* NULL -> weight 1 */, cplx* R)
{
  unsigned K = M / 2 + 1;
  for (unsigned k = 0; k < K; k++) {
    double w = mask_k ? mask_k[k] : 1.0;
    if (mask_k && !(w > 0)) continue;
    const cplx* x = snapshots + (size_t)k * N;
    cplx* Rk = R + (size_t)k * N * N;
    for (unsigned i = 0; i < N; i++)
      for (unsigned j = 0; j < N; j++)
        Rk[(size_t)i * N + j] = c_add(Rk[(size_t)i * N + j], c_scale(c_mul(x[i], c_conj(x[j])), w));
  }
}

/* energy of channel 0 as used for gating: pybeamformer.py:263-277 */
double orc_frame_energy(const cplx* samples_ch0, unsigned M)
{
  return c_abs(zdotc(samples_ch0, samples_ch0, M)) / M;
}

/* ------------------------------------------------------------------------
 * SubbandMVDR: diffuse model, diagonal loading, MVDR weights.
 * ---------------------------------------------------------------------- */
static double gsl_sinc(double x) { return (x == 0.0) ? 1.0 : sin(M_PI * x) / (M_PI * x); }

/* set_diffuse_noise_model: beamformer.cc:2442-2509.  mpos [N][3]; R [K][N][N] */
void orc_diffuse_noise_model(const double* mpos, unsigned N, unsigned M, float samplerate, float sspeed, cplx* R)
{
  unsigned K = M / 2 + 1;
  double* dm = (double*)calloc((size_t)N * N, sizeof(double));
  for (unsigned a = 0; a < N; a++)
    for (unsigned b = 0; b < a; b++) {
      double dx = mpos[a*3] - mpos[b*3], dy = mpos[a*3+1] - mpos[b*3+1], dz = mpos[a*3+2] - mpos[b*3+2];
      dm[(size_t)a * N + b] = sqrt(dx * dx + dy * dy + dz * dz);
    }
  for (unsigned k = 0; k < K; k++) {
    double omega_d_c = 2.0 * samplerate * k / (M * sspeed);                 /* :2485 */
    cplx* Rk = R + (size_t)k * N * N;
    for (unsigned a = 0; a < N; a++)
      for (unsigned b = 0; b < a; b++)
        Rk[(size_t)a * N + b] = c_make(gsl_sinc(omega_d_c * dm[(size_t)a * N + b]), 0.0);
    for (unsigned a = 0; a < N; a++) Rk[(size_t)a * N + a] = c_make(1.0, 0.0);
    for (unsigned a = 0; a < N; a++)
      for (unsigned b = a + 1; b < N; b++) Rk[(size_t)a * N + b] = Rk[(size_t)b * N + a];
  }
  free(dm);
}

/* set_all_diagonal_loading: beamformer.cc:2511-2523 (diagonalWeight is a float) */
void orc_diagonal_loading(cplx* R, unsigned M, unsigned N, float w)
{
  unsigned K = M / 2 + 1;
  for (unsigned k = 0; k < K; k++)
    for (unsigned c = 0; c < N; c++) R[((size_t)k * N + c) * N + c].re += w;
}

/* calc_mvdr_weights given the (pseudo-)inverse: beamformer.cc:2368-2397.
 * invR [K][N][N] (bin 0 unused), d = wq [M][N], w [K][N]; w_0 = all ones (:2369-2371). */
void orc_mvdr_weights_from_inverse(const cplx* invR, const cplx* wq, unsigned M, unsigned N, cplx* w)
{
  unsigned K = M / 2 + 1;
  cplx* tmpH = (cplx*)malloc(sizeof(cplx) * N);
  for (unsigned c = 0; c < N; c++) w[c] = c_make(1.0, 0.0);
  for (unsigned k = 1; k < K; k++) {
    const cplx* d = wq + (size_t)k * N;
    const cplx* iR = invR + (size_t)k * N * N;
    for (unsigned i = 0; i < N; i++) {                                      /* tmpH = invR^H d, :2386 */
      cplx s = c_make(0, 0);
      for (unsigned j = 0; j < N; j++) s = c_add(s, c_mul(c_conj(iR[(size_t)j * N + i]), d[j]));
      tmpH[i] = s;
    }
    cplx lambda = zdotc(tmpH, d, N);                                        /* :2387 */
    cplx norm = c_scale(lambda, (double)N);                                 /* :2388 */
    for (unsigned c = 0; c < N; c++) w[(size_t)k * N + c] = c_div(tmpH[c], norm);
  }
  free(tmpH);
}

/* ------------------------------------------------------------------------
 * GSL complex Cholesky (lower) + solve, as used by dereverberation.cc:677-681.
 * A is [n][n] row-major; only the lower triangle + diagonal are read.
 * ---------------------------------------------------------------------- */
int orc_cholesky_solve(cplx* A, const cplx* b, unsigned n, cplx* x)
{
  for (unsigned j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j].re;
    for (unsigned k = 0; k < j; k++) d -= c_abs2(A[(size_t)j * n + k]);
    if (d <= 0.0) return -1;
    d = sqrt(d);
    A[(size_t)j * n + j] = c_make(d, 0.0);
    for (unsigned i = j + 1; i < n; i++) {
      cplx s = A[(size_t)i * n + j];
      for (unsigned k = 0; k < j; k++) s = c_sub(s, c_mul(A[(size_t)i * n + k], c_conj(A[(size_t)j * n + k])));
      A[(size_t)i * n + j] = c_scale(s, 1.0 / d);
    }
  }
  for (unsigned i = 0; i < n; i++) {            /* L y = b */
    cplx s = b[i];
    for (unsigned k = 0; k < i; k++) s = c_sub(s, c_mul(A[(size_t)i * n + k], x[k]));
    x[i] = c_scale(s, 1.0 / A[(size_t)i * n + i].re);
  }
  for (int i = (int)n - 1; i >= 0; i--) {       /* L^H x = y */
    cplx s = x[i];
    for (unsigned k = i + 1; k < n; k++) s = c_sub(s, c_mul(c_conj(A[(size_t)k * n + i]), x[k]));
    x[i] = c_scale(s, 1.0 / A[(size_t)i * n + i].re);
  }
  return 0;
}

/* ------------------------------------------------------------------------
 * MultiChannelWPEDereverberation: dereverberation/dereverberation.cc:312-698.
 * frames: Y [T][C][M] complex (all buffered frames, as fill_buffer_ :506-538).
 * G [C][M][C*L] filters (in/out, start at zero), L = upper-lower+1.
 * Estimation for bins 0..M-1 restricted to the reference's band test (:672).
 * ---------------------------------------------------------------------- */
static void wpe_lags(const cplx* Y, long T, unsigned C, unsigned M, unsigned L,
                     unsigned bin, long sampleX, cplx* lags)
{                                                                           /* get_lags_: :540-555 */
  unsigned tot = 0;
  for (unsigned c = 0; c < C; c++)
    for (unsigned l = 0; l < L; l++) {
      long idx = sampleX - (long)l;
      lags[tot++] = (idx < 0) ? c_make(0, 0) : Y[((size_t)idx * C + c) * M + bin];
    }
  (void)T;
}

int orc_wpe_estimate(const cplx* Y, long T, unsigned C, unsigned M, unsigned lowerN, unsigned upperN,
                     unsigned iterations, double load_db, unsigned lower_bw, unsigned upper_bw,
                     double diagonal_bias, cplx* G)
{
  unsigned L = upperN - lowerN + 1, P = L * C;
  double load_factor = pow(10.0, load_db / 10.0);
  double* theta = (double*)malloc(sizeof(double) * (size_t)C * T * M);
  cplx* lags = (cplx*)malloc(sizeof(cplx) * P);
  cplx* R = (cplx*)malloc(sizeof(cplx) * (size_t)P * P);
  cplx* rr = (cplx*)malloc(sizeof(cplx) * P);
  int rc = 0;
  for (unsigned it = 0; it < iterations && rc == 0; it++) {
    /* calc_Thetan_: :619-646 */
    for (long t = 0; t < T; t++)
      for (unsigned c = 0; c < C; c++)
        for (unsigned k = 0; k < M; k++) {
          cplx cur = Y[((size_t)t * C + c) * M + k];
          if (t >= (long)lowerN) {
            wpe_lags(Y, T, C, M, L, k, t - lowerN, lags);
            cur = c_sub(cur, zdotc(G + ((size_t)c * M + k) * P, lags, P));
          }
          double th = c_abs(cur);
          if (th < 1.0e-3) th = 1.0e-3;                                     /* subband_floor_ :617 */
          theta[((size_t)c * T + t) * M + k] = th * th;
        }
    for (unsigned k = 0; k < M && rc == 0; k++) {
      if (k > lower_bw && k < upper_bw) continue;                           /* :672 */
      for (unsigned c = 0; c < C && rc == 0; c++) {
        /* calc_Rr_: :557-615 */
        memset(R, 0, sizeof(cplx) * (size_t)P * P);
        memset(rr, 0, sizeof(cplx) * P);
        for (long t = lowerN; t < T; t++) {
          double th = theta[((size_t)c * T + t) * M + k];
          wpe_lags(Y, T, C, M, L, k, t - lowerN, lags);
          for (unsigned row = 0; row < P; row++)
            for (unsigned col = 0; col <= row; col++) {
              cplx v = c_mul(lags[row], c_conj(lags[col]));
              R[(size_t)row * P + col] = c_add(R[(size_t)row * P + col], c_make(v.re / th, v.im / th));
            }
          cplx cur = Y[((size_t)t * C + c) * M + k];
          for (unsigned l = 0; l < P; l++) {
            cplx v = c_mul(c_conj(cur), lags[l]);
            rr[l] = c_add(rr[l], c_make(v.re / th, v.im / th));
          }
        }
        for (unsigned row = 0; row < P; row++) R[(size_t)row * P + row].re += diagonal_bias;
        /* load_R_: :648-663 */
        double maxd = 0.0;
        for (unsigned i = 0; i < P; i++) { double d = c_abs(R[(size_t)i * P + i]); if (d > maxd) maxd = d; }
        for (unsigned i = 0; i < P; i++)
          R[(size_t)i * P + i] = c_make(c_abs(R[(size_t)i * P + i]) + maxd * load_factor, 0.0);
        /* cholesky decomp + solve: :676-681 */
        if (orc_cholesky_solve(R, rr, P, G + ((size_t)c * M + k) * P)) rc = -1;
      }
    }
  }
  free(theta); free(lags); free(R); free(rr);
  return rc;
}

/* calc_every_channel_output for all frames: dereverberation.cc:444-501.
 * out [T][C][M]; bins 0..M/2 computed, mirror conj. */
void orc_wpe_apply(const cplx* Y, long T, unsigned C, unsigned M, unsigned lowerN, unsigned upperN,
                   unsigned lower_bw, unsigned upper_bw, const cplx* G, cplx* out)
{
  unsigned L = upperN - lowerN + 1, P = L * C;
  cplx* lags = (cplx*)malloc(sizeof(cplx) * P);
  for (long t = 0; t < T; t++)
    for (unsigned c = 0; c < C; c++)
      for (unsigned k = 0; k <= M / 2; k++) {
        cplx cur = Y[((size_t)t * C + c) * M + k];
        if (t >= (long)lowerN && (k <= lower_bw || k >= upper_bw)) {
          /* the apply-time ring holds at most L frames (:471-480) so lags reach back L-1 frames
             from frame t-lowerN; older entries read as zero exactly like (index < 0). */
          long newest = t - lowerN;
          long window_first = (t + 1 > (long)L) ? t + 1 - (long)L : 0;
          unsigned tot = 0;
          for (unsigned cc = 0; cc < C; cc++)
            for (unsigned l = 0; l < L; l++) {
              long idx = newest - (long)l;
              lags[tot++] = (idx < window_first) ? c_make(0, 0) : Y[((size_t)idx * C + cc) * M + k];
            }
          cur = c_sub(cur, zdotc(G + ((size_t)c * M + k) * P, lags, P));
        }
        out[((size_t)t * C + c) * M + k] = cur;
        if (k > 0 && k < M / 2) out[((size_t)t * C + c) * M + (M - k)] = c_conj(cur);
      }
  free(lags);
}

/* ------------------------------------------------------------------------
 * Whole-graph CPU baseline: N analysis banks -> SnapShotArray -> SubbandGSC ->
 * synthesis bank, pulled frame by frame exactly like src/beamformerDS.cc:184-191
 * (without the post-filter).  pcm [N][len].  Returns output blocks written.
 * ---------------------------------------------------------------------- */
long orc_pipeline_gsc(const double* h, const double* g, unsigned M, unsigned m, unsigned r, unsigned dct,
                      const float* pcm, unsigned N, long len, const cplx* wq, const cplx* wl,
                      float* out, long max_blocks, long* frames_beamformed)
{
  orc_analysis** banks = (orc_analysis**)malloc(sizeof(void*) * N);
  for (unsigned c = 0; c < N; c++) banks[c] = orc_analysis_new(h, M, m, r, dct, pcm + (size_t)c * len, len);
  unsigned pd_s, la, R = 1u << r, D = M / R;
  orc_fb_delays(m, r, 1, dct, &pd_s, &la);
  cplx* samples = (cplx*)malloc(sizeof(cplx) * (size_t)N * M);
  cplx* snaps = (cplx*)malloc(sizeof(cplx) * (size_t)N * M);
  cplx* y = (cplx*)malloc(sizeof(cplx) * M);
  /* synthesis node fed one frame at a time */
  orc_synthesis* syn = orc_synthesis_new(g, M, m, r, dct, 1, (const double*)y, 0);
  long blocks = 0, nbf = 0; int primed = 0, end = 0;
  while (blocks < max_blocks && !end) {
    unsigned need = primed ? 1 : pd_s + 1;
    for (unsigned q = 0; q < need && !end; q++) {
      for (unsigned c = 0; c < N; c++)
        if (orc_analysis_next(banks[c], (double*)(samples + (size_t)c * M), NULL)) { end = 1; break; }
      if (end) break;
      orc_snapshot_update(samples, M, N, snaps);
      orc_gsc_frame(snaps, wq, wl, M, N, 0, y);
      nbf++;
      synthesis_update_buf(syn, (const double*)y);
    }
    if (end) break;
    primed = 1;
    syn->frame_no++;
    for (unsigned i = 0; i < M; i++) {
      double sum = 0.0;
      for (unsigned k = 0; k < m; k++)
        sum += syn->proto[(M - i - 1) + M * k] * rb_sample(&syn->buffer, R * k, i);
      syn->convert[i] = sum;
    }
    rb_next(&syn->gsi, syn->convert, 0);
    float* o = out + (size_t)D * blocks;
    memset(o, 0, sizeof(float) * D);
    for (unsigned j = 0; j < R; j++)
      for (unsigned d = 0; d < D; d++)
        o[D - d - 1] = (float)(o[D - d - 1] + rb_sample(&syn->gsi, R - j - 1, d + j * D));
    blocks++;
  }
  if (frames_beamformed) *frames_beamformed = nbf;
  for (unsigned c = 0; c < N; c++) orc_analysis_free(banks[c]);
  free(banks); free(samples); free(snaps); free(y); orc_synthesis_free(syn);
  return blocks;
}

/* ------------------------------------------------------------------------
 * RLS sidelobe cancellers ("next" row of the survey).  Two reference variants.
 *
 * (1) lib/pybeamformer.py:817-898  SubbandGSCRLSBeamformer.__iter__ (pure numpy, pinned by
 *     tests/golden/gen_golden_pybeamformer_rls.py).
 *     scal = { _energy, _isamp, _ttl_updates } in/out;
 *     par  = { beta, gamma, mu, init_diagonal_load, regularization_param, sil_thresh,
 *              constraint_option, alpha2, max_wa_l2norm, min_frames }.
 *     BmH [K][bs][N], wqH [K][N], Pz [K][bs][bs], waH [K][bs]; samples_ch0 [M]; snapshots [M][N].
 * ---------------------------------------------------------------------- */
void orc_rls_py_frame(unsigned M, unsigned N, unsigned Nc, const double* par, double* scal,
                      const cplx* samples_ch0, const cplx* snapshots,
                      const cplx* BmH, const cplx* wqH, cplx* Pz, cplx* waH, cplx* out)
{
  const double beta = par[0], gamma = par[1], mu = par[2], init_load = par[3], reg = par[4], sil = par[5];
  const int copt = (int)par[6];
  const double alpha2 = par[7], max_norm = par[8];
  const long min_frames = (long)par[9];
  unsigned bs = N - Nc, M2 = M / 2;
  long isamp = (long)scal[1];
  double energy = c_abs(zdotc(samples_ch0, samples_ch0, M)) / M;                   /* :824 */
  memset(out, 0, sizeof(cplx) * M);
  int adapt = energy > (scal[0] / sil);                                            /* :827 */
  if (adapt) scal[2] += 1.0;
  cplx* ZK = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* PzZ = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* gz = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* temp = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* PzK = (cplx*)malloc(sizeof(cplx) * bs * bs);
  cplx* waHK = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* va = (cplx*)malloc(sizeof(cplx) * bs);
  for (unsigned k = 0; k <= M2; k++) {
    const cplx* XK = snapshots + (size_t)k * N;
    const cplx* Bk = BmH + (size_t)k * bs * N;
    cplx* P = Pz + (size_t)k * bs * bs;
    cplx* wa = waH + (size_t)k * bs;
    for (unsigned i = 0; i < bs; i++) {                                            /* :832 */
      cplx z = c_make(0, 0);
      for (unsigned c = 0; c < N; c++) z = c_add(z, c_mul(Bk[(size_t)i * N + c], XK[c]));
      ZK[i] = z;
    }
    cplx YcK = c_make(0, 0);                                                       /* :834 */
    for (unsigned c = 0; c < N; c++) YcK = c_add(YcK, c_mul(wqH[(size_t)k * N + c], XK[c]));
    if (adapt) {
      cplx ip = c_make(0, 0);
      for (unsigned i = 0; i < bs; i++) {                                          /* :838 PzZ = Pz . ZK */
        cplx a = c_make(0, 0);
        for (unsigned j = 0; j < bs; j++) a = c_add(a, c_mul(P[(size_t)i * bs + j], ZK[j]));
        PzZ[i] = a;
      }
      for (unsigned i = 0; i < bs; i++) ip = c_add(ip, c_mul(c_conj(ZK[i]), PzZ[i]));   /* :839 */
      cplx den = c_make(mu + ip.re, ip.im);
      for (unsigned i = 0; i < bs; i++) gz[i] = c_div(PzZ[i], den);                /* :840 */
      for (unsigned j = 0; j < bs; j++) {                                          /* :841 temp = conj(ZK) . Pz */
        cplx a = c_make(0, 0);
        for (unsigned i = 0; i < bs; i++) a = c_add(a, c_mul(c_conj(ZK[i]), P[(size_t)i * bs + j]));
        temp[j] = a;
      }
      for (unsigned i = 0; i < bs; i++)                                            /* :842 */
        for (unsigned j = 0; j < bs; j++)
          PzK[(size_t)i * bs + j] = c_scale(c_sub(P[(size_t)i * bs + j], c_mul(gz[i], temp[j])), 1.0 / mu);
      cplx dot = c_make(0, 0);                                                     /* :845 */
      for (unsigned i = 0; i < bs; i++) dot = c_add(dot, c_mul(wa[i], ZK[i]));
      cplx ep = c_sub(YcK, dot);
      for (unsigned i = 0; i < bs; i++)                                            /* :846 */
        waHK[i] = c_add(wa[i], c_mul(c_scale(c_conj(gz[i]), gamma), ep));
      if (reg > 0) {                                                               /* :848-849 */
        for (unsigned i = 0; i < bs; i++) {
          cplx a = c_make(0, 0);
          for (unsigned j = 0; j < bs; j++) a = c_add(a, c_mul(c_conj(PzK[(size_t)i * bs + j]), wa[j]));
          waHK[i] = c_sub(waHK[i], c_scale(a, reg));
        }
      }
      if (copt > 0) {                                                              /* :852-871 */
        cplx nn = c_make(0, 0);
        for (unsigned i = 0; i < bs; i++) nn = c_add(nn, c_mul(waHK[i], c_conj(waHK[i])));
        double waK2 = c_abs(nn);
        if ((copt == 1 || copt == 3) && waK2 > alpha2) {
          for (unsigned i = 0; i < bs; i++) {                                      /* va = PzK . waK */
            cplx a = c_make(0, 0);
            for (unsigned j = 0; j < bs; j++) a = c_add(a, c_mul(PzK[(size_t)i * bs + j], c_conj(waHK[j])));
            va[i] = a;
          }
          cplx aa = c_make(0, 0), bb = c_make(0, 0);
          for (unsigned i = 0; i < bs; i++) {
            aa = c_add(aa, c_mul(va[i], c_conj(va[i])));
            bb = c_add(bb, c_mul(c_conj(va[i]), c_conj(waHK[i])));
          }
          double a = c_abs(aa), b = -2.0 * bb.re, c = waK2 - alpha2;
          double arg = b * b - 4.0 * a * c;
          double betaK = (arg > 0) ? -(b + sqrt(arg)) / (2.0 * a) : -b / (2.0 * a);
          for (unsigned i = 0; i < bs; i++) waHK[i] = c_sub(waHK[i], c_scale(c_conj(va[i]), betaK));
        }
        if (copt >= 2 && waK2 > max_norm) {
          double sc = sqrt(max_norm / waK2);
          for (unsigned i = 0; i < bs; i++) waHK[i] = c_scale(waHK[i], sc);
          for (unsigned i = 0; i < bs; i++)
            for (unsigned j = 0; j < bs; j++) PzK[(size_t)i * bs + j] = c_make(i == j ? 1.0 / init_load : 0.0, 0.0);
        }
      }
      memcpy(P, PzK, sizeof(cplx) * bs * bs);                                      /* :890-891 */
      memcpy(wa, waHK, sizeof(cplx) * bs);
    }
    if (isamp >= min_frames) {                                                     /* :894-897 */
      cplx dot = c_make(0, 0);
      for (unsigned i = 0; i < bs; i++) dot = c_add(dot, c_mul(wa[i], ZK[i]));
      out[k] = c_sub(YcK, dot);
    } else out[k] = YcK;
    if (k > 0 && k < M2) out[M - k] = c_conj(out[k]);
  }
  scal[0] = scal[0] * beta + (1.0 - beta) * energy;                                /* :902 */
  scal[1] = (double)(isamp + 1);
  free(ZK); free(PzZ); free(gz); free(temp); free(PzK); free(waHK); free(va);
}

/* (2) beamformer/beamformer.cc:1514-1645  SubbandGSCRLS::next + update_active_weight_vector2_
 *     (C++; needs GSL, cannot be compiled here -> restated, output-level parity UNPINNED).
 *     B [M][N][bs] (rows = channels), wq [M][N], wl [M][N] in/out, Pz [M][bs][bs] in/out,
 *     wa [M][bs] in/out (only bins 1..M/2 are touched); snapshots [M][N]; out [M].
 *     qctype: 0 none, 1 CONSTANT_NORM, 2 THRESHOLD_LIMITATION (beamformer.h:215-219).          */
void orc_rls_cc_frame(unsigned M, unsigned N, unsigned Nc, double mu, double diag_w, int qctype, double alpha,
                      int normalize, int update, const cplx* snapshots, const cplx* B, const cplx* wq,
                      cplx* wl, cplx* Pz, cplx* wa_all, cplx* out)
{
  unsigned bs = N - Nc, M2 = M / 2;
  cplx* Zf = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* PzHZ = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* gz = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* wa = (cplx*)malloc(sizeof(cplx) * bs);
  cplx* w = (cplx*)malloc(sizeof(cplx) * N);
  /* :1540-1543 direct component */
  out[0] = zdotc(wq, snapshots, N);
  for (unsigned k = 1; k <= M2; k++) {                                             /* :1546-1558 */
    const cplx* x = snapshots + (size_t)k * N;
    for (unsigned c = 0; c < N; c++) w[c] = c_sub(wq[(size_t)k * N + c], wl[(size_t)k * N + c]);
    if (normalize) {                                                               /* :1229-1237 */
      double nrm = 0;
      for (unsigned c = 0; c < N; c++) nrm += c_abs2(w[c]);
      nrm = sqrt(nrm);
      for (unsigned c = 0; c < N; c++) w[c] = c_scale(w[c], 1.0 / (nrm * N));
    }
    cplx val = zdotc(w, x, N);
    out[k] = val;
    if (k < M2) out[M - k] = c_conj(val);
  }
  if (update) {
    for (unsigned k = 1; k <= M2; k++) {                                           /* :1589-1644 */
      const cplx* x = snapshots + (size_t)k * N;
      const cplx* Bk = B + (size_t)k * N * bs;
      cplx* P = Pz + (size_t)k * bs * bs;
      cplx* old_wa = wa_all + (size_t)k * bs;
      for (unsigned i = 0; i < bs; i++) {                                          /* :1594 Z = B^H X */
        cplx z = c_make(0, 0);
        for (unsigned c = 0; c < N; c++) z = c_add(z, c_mul(c_conj(Bk[(size_t)c * bs + i]), x[c]));
        Zf[i] = z;
      }
      for (unsigned i = 0; i < bs; i++) {                                          /* :1597 PzH_Z = Pz^H Z */
        cplx a = c_make(0, 0);
        for (unsigned j = 0; j < bs; j++) a = c_add(a, c_mul(c_conj(P[(size_t)j * bs + i]), Zf[j]));
        PzHZ[i] = a;
      }
      for (unsigned i = 0; i < bs; i++) {                                          /* :1598 gz = Pz Z / mu */
        cplx a = c_make(0, 0);
        for (unsigned j = 0; j < bs; j++) a = c_add(a, c_mul(P[(size_t)i * bs + j], Zf[j]));
        gz[i] = c_scale(a, 1.0 / mu);
      }
      cplx de = zdotc(PzHZ, Zf, bs);                                               /* :1599-1600 */
      de = c_make(de.re * (1.0 / mu) + 1.0, de.im * (1.0 / mu));
      for (unsigned i = 0; i < bs; i++) gz[i] = c_div(gz[i], de);                  /* :1601-1606 */
      for (unsigned i = 0; i < bs; i++)                                            /* :1609-1617 */
        for (unsigned j = 0; j < bs; j++)
          P[(size_t)i * bs + j] = c_scale(c_sub(P[(size_t)i * bs + j], c_mul(gz[i], c_conj(PzHZ[j]))), 1.0 / mu);
      cplx epA = c_conj(out[k]);                                                   /* :1620 */
      for (unsigned i = 0; i < bs; i++) {                                          /* :1622-1625 (I - s Pz) wa */
        cplx a = c_make(0, 0);
        for (unsigned j = 0; j < bs; j++) {
          cplx m1 = c_scale(P[(size_t)i * bs + j], -diag_w);
          if (i == j) m1.re += 1.0;
          a = c_add(a, c_mul(m1, old_wa[j]));
        }
        wa[i] = a;
      }
      for (unsigned i = 0; i < bs; i++) wa[i] = c_add(wa[i], c_mul(gz[i], epA));   /* :1626-1630 */
      if (qctype == 1 || qctype == 2) {                                            /* :1631-1641 */
        double nrm = 0;
        for (unsigned i = 0; i < bs; i++) nrm += c_abs2(wa[i]);
        nrm = sqrt(nrm);
        if (qctype == 1 || nrm * nrm >= alpha)
          for (unsigned i = 0; i < bs; i++) wa[i] = c_scale(wa[i], alpha / nrm);
      }
      for (unsigned i = 0; i < bs; i++) old_wa[i] = wa[i];                         /* :1643 U_f: wa_ <- wa ; wl = B wa */
      for (unsigned c = 0; c < N; c++) {
        cplx a = c_make(0, 0);
        for (unsigned i = 0; i < bs; i++) a = c_add(a, c_mul(Bk[(size_t)c * bs + i], wa[i]));
        wl[(size_t)k * N + c] = a;
      }
    }
  }
  free(Zf); free(PzHZ); free(gz); free(wa); free(w);
}

/* ------------------------------------------------------------------------
 * McCowan / Lefkimmiatis post-filters ("next" row of the survey), postfilter/postfilter.cc.
 * C++ only (needs GSL) -> restated, output-level parity UNPINNED.
 * ---------------------------------------------------------------------- */
/* calculateSpectralDensities_f: postfilter.cc:692-736.  csd [N*N] in/out; returns sumOfPSD / N */
static double spectral_densities_f(const cplx* ta, unsigned N, cplx* csd, double alpha)
{
  double sum_psd = 0.0;
  for (unsigned i = 0; i + 1 < N; i++)
    for (unsigned j = i + 1; j < N; j++) {
      unsigned idx = i * N + j;
      cplx xx = c_mul(ta[i], c_conj(ta[j]));                                       /* calc_CSD_ :8-21 */
      csd[idx] = (alpha > 0.0) ? c_add(c_scale(csd[idx], alpha), c_scale(xx, 1.0 - alpha)) : xx;
    }
  for (unsigned i = 0; i < N; i++) {
    unsigned idx = i * N + i;
    double est = (alpha > 0.0) ? alpha * csd[idx].re + (1.0 - alpha) * c_abs2(ta[i]) : c_abs2(ta[i]);
    sum_psd += est;
    csd[idx] = c_make(est, 0.0);
  }
  return sum_psd / N;
}

/* McCowanPostFilter::estimate_average_clean_PSD_ (non-ORIGINAL_IAIN_PAPER branch): postfilter.cc:798-829 */
static double mccowan_clean_psd(const cplx* R, const cplx* csd, unsigned N, double thr, int type)
{
  cplx sum = c_make(0, 0);
  for (unsigned i = 0; i + 1 < N; i++) {
    double phi_ii = csd[i * N + i].re;
    for (unsigned j = i + 1; j < N; j++) {
      cplx phi_ij = csd[i * N + j];
      double phi_jj = csd[j * N + j].re;
      cplx R_ij = R[i * N + j];
      if (R_ij.re > thr && R_ij.im <= 0.0) R_ij = c_make(thr, 0);
      cplx nu = c_sub(phi_ij, c_scale(R_ij, 0.5 * (phi_ii + phi_jj)));
      cplx de = c_make(1.0 - R_ij.re, -R_ij.im);
      sum = c_add(sum, c_div(nu, de));
    }
  }
  double avg = (type & 1) ? sum.re : c_abs(sum);
  return 2.0 * avg / (N * (N - 1.0));
}

/* LefkimmiatisPostFilter::estimate_average_noise_PSD_ (non-ORIGINAL branch): postfilter.cc:1041-1077 */
static double lefkimmiatis_noise_psd(const cplx* R, const cplx* csd, unsigned N, double thr, int type)
{
  cplx sum = c_make(0, 0);
  for (unsigned i = 0; i + 1 < N; i++) {
    cplx phi_ii = csd[i * N + i];
    for (unsigned j = i + 1; j < N; j++) {
      cplx phi_ij = csd[i * N + j];
      cplx phi_jj = csd[j * N + j];
      cplx R_ij = R[i * N + j];
      if (R_ij.re > thr) R_ij = c_make(thr, 0);
      else if (R_ij.re == 1) R_ij = c_make(0.99, 0);
      cplx nu = c_sub(c_scale(c_add(phi_ii, phi_jj), 0.5), phi_ij);
      cplx de = c_make(1.0 - R_ij.re, -R_ij.im);
      sum = c_add(sum, c_div(nu, de));
    }
  }
  double avg = (type & 1) ? sum.re : c_abs(sum);
  return 2.0 * avg / (N * (N - 1.0));
}

/* McCowanPostFilter::post_filtering_ for one frame: postfilter.cc:843-898.
 * d [K..][N] = wq if (type & 8) else the array manifold (callers pass the right array); R [K][N][N];
 * csds [K][N*N] in/out; wp [M] out; y [M] in/out; frame_no_pre = frame_no_ before increment_. */
void orc_mccowan_frame(const cplx* d, const cplx* snapshots, const cplx* R, unsigned M, unsigned N,
                       cplx* csds, cplx* wp, double alpha_cfg, int type, int min_frames, double thr,
                       int frame_no_pre, cplx* y)
{
  unsigned M2 = M / 2;
  double alpha = (frame_no_pre > 0) ? alpha_cfg : 0.0;                             /* :866-869 */
  cplx* ta = (cplx*)malloc(sizeof(cplx) * N);
  for (unsigned k = 0; k <= M2; k++) {
    const cplx* dk = d + (size_t)k * N;
    const cplx* x = snapshots + (size_t)k * N;
    cplx* csd = csds + (size_t)k * N * N;
    for (unsigned i = 0; i < N; i++) ta[i] = c_mul(c_conj(dk[i]), x[i]);            /* time_alignment_ :30-43 */
    double de = spectral_densities_f(ta, N, csd, alpha);
    double nu = mccowan_clean_psd(R + (size_t)k * N * N, csd, N, thr, type);
    double weight = nu / de;
    if (weight > 1.0) weight = 1.0;
    if (weight < ORC_SPECTRAL_FLOOR) weight = ORC_SPECTRAL_FLOOR;
    wp[k] = c_make(weight, 0);
    if (k > 0 && k < M2) wp[M - k] = c_make(weight, 0);
    if (frame_no_pre >= min_frames) {                                              /* :889-894 */
      cplx o = c_scale(y[k], weight);
      y[k] = o;
      if (k > 0 && k < M2) y[M - k] = c_conj(o);
    }
  }
  free(ta);
}

/* LefkimmiatisPostFilter::post_filtering_ for one frame: postfilter.cc:1081-1157 with calcLambda :982-995.
 * d [K][N] = array manifold; invR [K][N][N] (pseudoinverse or identity, :967-980); fbinX1. */
void orc_lefkimmiatis_frame(const cplx* d, const cplx* snapshots, const cplx* R, const cplx* invR, unsigned M,
                            unsigned N, cplx* csds, cplx* wp, double alpha_cfg, int type, int min_frames,
                            double thr, unsigned fbinX1, int frame_no_pre, cplx* y)
{
  unsigned M2 = M / 2;
  double alpha = (frame_no_pre > 0) ? alpha_cfg : 0.0;                             /* :1104-1107 */
  cplx* ta = (cplx*)malloc(sizeof(cplx) * N);
  cplx* tmpH = (cplx*)malloc(sizeof(cplx) * N);
  for (unsigned k = 0; k <= M2; k++) {
    const cplx* dk = d + (size_t)k * N;
    const cplx* x = snapshots + (size_t)k * N;
    cplx* csd = csds + (size_t)k * N * N;
    for (unsigned i = 0; i < N; i++) ta[i] = c_mul(c_conj(dk[i]), x[i]);
    spectral_densities_f(ta, N, csd, alpha);
    double phi_ss = mccowan_clean_psd(R + (size_t)k * N * N, csd, N, thr, type);
    double phi_vv = lefkimmiatis_noise_psd(R + (size_t)k * N * N, csd, N, thr, type);
    double weight;
    if (k < fbinX1) weight = phi_ss / (phi_ss + phi_vv);
    else {
      const cplx* iR = invR + (size_t)k * N * N;
      for (unsigned i = 0; i < N; i++) {                                           /* tmpH = invR^H d */
        cplx a = c_make(0, 0);
        for (unsigned j = 0; j < N; j++) a = c_add(a, c_mul(c_conj(iR[(size_t)j * N + i]), dk[j]));
        tmpH[i] = a;
      }
      cplx Lambda = zdotc(tmpH, dk, N);
      double phi_nn = phi_vv / ((type & 1) ? Lambda.re : c_abs(Lambda));
      weight = phi_ss / (phi_ss + phi_nn);
    }
    if (weight > 1.0) weight = 1.0;
    if (weight < ORC_SPECTRAL_FLOOR) weight = ORC_SPECTRAL_FLOOR;
    wp[k] = c_make(weight, 0);
    if (k > 0 && k < M2) wp[M - k] = c_make(weight, 0);
    if (frame_no_pre >= min_frames) {
      cplx o = c_scale(y[k], weight);
      y[k] = o;
      if (k > 0 && k < M2) y[M - k] = c_conj(o);
    }
  }
  free(ta); free(tmpH);
}
