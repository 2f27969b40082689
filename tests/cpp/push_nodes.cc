// The two push-style entry points of the node layer through C++ (reference modulated/modulated.h:320-334, postfilter/postfilter.h:83-94):
//   * OverSampledDFTSynthesisBank without a source: input_source_vector() + next() per frame,
//   * ZelinskiPostFilter without a beamformer object: set_snapshot_array() + set_array_manifold_vector(), the caller keeps the
//     SnapShotArray current, and BeamformerWeights::CSDs() of the weight object the filter owns is live.
// usage: push_nodes <dir> M m r T N alpha type      reads  <dir>/g.f64 [m M], Y.c128 [T][M], X.c128 [T][N][M], d.c128 [M][N]
//                                                   writes <dir>/blocks.f32 [T][D], Z.c128 [T][M], csd.c128 [M/2+1][N N], wp1.c128 [M]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "modulated/modulated.h"
#include "postfilter/postfilter.h"

namespace {
template <class T> std::vector<T> slurp(const std::string& fn, size_t n)
{
  std::vector<T> v(n);
  FILE* fp = fopen(fn.c_str(), "rb");
  if (!fp || fread(v.data(), sizeof(T), n, fp) != n) { fprintf(stderr, "cannot read %zu items of %s\n", n, fn.c_str()); exit(2); }
  fclose(fp);
  return v;
}
template <class T> void dump(const std::string& fn, const T* p, size_t n)
{
  FILE* fp = fopen(fn.c_str(), "wb");
  if (!fp || fwrite(p, sizeof(T), n, fp) != n) { fprintf(stderr, "cannot write %s\n", fn.c_str()); exit(2); }
  fclose(fp);
}
// a source node over frames held in memory: what an upstream beamformer looks like to the post-filter
class MemoryFrames : public VectorComplexFeatureStream {
 public:
  MemoryFrames(const std::vector<double>& Y, unsigned M, long T) : VectorComplexFeatureStream(M, "MemoryFrames"), Y_(Y), M_(M), T_(T) {}
  virtual const gsl_vector_complex* next(int frame_no = -5)
  {
    if (frame_no == frame_no_) return vector_;
    if (frame_no_ + 1 >= T_) throw jiterator_error("end of samples!");
    increment_();
    for (unsigned i = 0; i < 2 * M_; i++) vector_->data[i] = Y_[(size_t)frame_no_ * 2 * M_ + i];
    return vector_;
  }
 private:
  const std::vector<double>& Y_;
  unsigned M_;
  long T_;
};
}  // namespace

int main(int argc, char** argv)
{
  if (argc != 9) { fprintf(stderr, "usage: push_nodes dir M m r T N alpha type\n"); return 2; }
  const std::string dir = argv[1];
  const unsigned M = atoi(argv[2]), m = atoi(argv[3]), r = atoi(argv[4]), N = atoi(argv[6]);
  const long T = atol(argv[5]);
  const double alpha = atof(argv[7]);
  const int type = atoi(argv[8]);
  const unsigned D = M >> r, K = M / 2 + 1;
  try {
    std::vector<double> g = slurp<double>(dir + "/g.f64", (size_t)m * M);
    std::vector<double> Y = slurp<double>(dir + "/Y.c128", (size_t)T * 2 * M);
    std::vector<double> X = slurp<double>(dir + "/X.c128", (size_t)T * N * 2 * M);
    std::vector<double> d = slurp<double>(dir + "/d.c128", (size_t)M * 2 * N);
    gsl_vector* proto = gsl_vector_calloc((size_t)m * M);
    for (size_t i = 0; i < g.size(); i++) gsl_vector_set(proto, i, g[i]);

    // ---- 1. the source-less synthesis bank: one frame in, one block out
    OverSampledDFTSynthesisBankPtr sfb(new OverSampledDFTSynthesisBank(proto, M, m, r, 2, 1));
    gsl_vector_complex* frame = gsl_vector_complex_calloc(M);
    std::vector<float> blocks((size_t)T * D);
    for (long t = 0; t < T; t++) {
      for (unsigned i = 0; i < 2 * M; i++) frame->data[i] = Y[(size_t)t * 2 * M + i];
      if (t & 1) sfb->input_source_vector(frame); else sfb->inputSourceVector(frame);
      const gsl_vector_float* b = sfb->next();
      for (unsigned i = 0; i < D; i++) blocks[(size_t)t * D + i] = gsl_vector_float_get(b, i);
    }
    bool threw = false;                                      // a second next() without a frame is not a per-frame graph
    try { sfb->next(); } catch (jconsistency_error&) { threw = true; }
    if (!threw) { fprintf(stderr, "next() without input_source_vector() did not raise\n"); return 1; }
    dump(dir + "/blocks.f32", blocks.data(), blocks.size());

    // ---- 2. Zelinski without a beamformer object
    VectorComplexFeatureStreamPtr src(new MemoryFrames(Y, M, T));
    ZelinskiPostFilterPtr pf(new ZelinskiPostFilter(src, M, alpha, type));
    SnapShotArrayPtr snap(new SnapShotArray(M, N));
    pf->set_snapshot_array(snap);
    gsl_vector_complex* dv = gsl_vector_complex_calloc(N);
    for (unsigned k = 0; k < M; k++) {
      for (unsigned c = 0; c < 2 * N; c++) dv->data[c] = d[(size_t)k * 2 * N + c];
      if (k & 1) pf->set_array_manifold_vector(k, dv, false, 1); else pf->setArrayManifoldVector(k, dv, false, 1);
    }
    std::vector<double> Z((size_t)T * 2 * M);
    gsl_vector_complex chan; chan.size = M; chan.stride = 1; chan.block = NULL; chan.owner = 0;
    for (long t = 0; t < T; t++) {
      for (unsigned c = 0; c < N; c++) {
        chan.data = X.data() + ((size_t)t * N + c) * 2 * M;
        snap->set_samples(&chan, c);
      }
      snap->update();
      const gsl_vector_complex* z = pf->next();
      for (unsigned i = 0; i < 2 * M; i++) Z[(size_t)t * 2 * M + i] = z->data[i];
    }
    dump(dir + "/Z.c128", Z.data(), Z.size());
    const gsl_vector_complex* wp1 = pf->postfilter_weights();
    dump(dir + "/wp1.c128", wp1->data, (size_t)2 * M);
    // the densities live in the weight object the filter created: reach it the way the reference's code does
    // (postfilter_weights() is bf_weights_->wp1(); CSDs() of the same object are rebuilt on demand)
    std::vector<double> csd((size_t)K * 2 * N * N);
    gsl_vector_complex** C = pf->weights_object()->CSDs();
    for (unsigned k = 0; k < K; k++)
      for (unsigned i = 0; i < 2 * N * N; i++) csd[(size_t)k * 2 * N * N + i] = C[k]->data[i];
    dump(dir + "/csd.c128", csd.data(), csd.size());
    printf("ok\n");
    return 0;
  } catch (j_error& e) {
    fprintf(stderr, "j_error: %s\n", e.what());
    return 1;
  }
}
