// pybtk20.cc -- Python binding of the C++ node layer (libbtk20hip.so): the classes the reference exposes through SWIG
// (stream/stream.i, feature/feature.i, modulated/modulated.i, beamformer/beamformer.i, postfilter/postfilter.i,
// dereverberation/dereverberation.i) bound with pybind11 under the same names.
//   * a bound object IS the C++ node: the holder shares Countable's intrusive count with the refcountable_ptrs the nodes keep
//     of each other, so Python and C++ references are one population;
//   * next() returns a numpy VIEW of the node's vector_ (no copy; the array keeps the node alive) -- the reference's typemap
//     does the same (include/vector.i:290-305);
//   * jiterator_error surfaces as StopIteration, the rest of the j_error family as Python exceptions of the same names;
//   * PyVectorFloatFeatureStream / PyVectorComplexFeatureStream wrap any Python object with size() / __iter__ / next() /
//     reset() as a C++ source node (reference stream/pyStream.h:25-168), so Python code can feed C++ nodes.
#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <complex>
#include <vector>

#include "beamformer/beamformer.h"
#include "dereverberation/dereverberation.h"
#include "feature/feature.h"
#include "modulated/modulated.h"
#include "postfilter/postfilter.h"

namespace py = pybind11;
typedef std::complex<double> cd;

// intrusive holder: one reference count per object, kept in Countable
template <class T>
class cref {
 public:
  cref(T* p = nullptr) : p_(p) { if (p_) p_->increment(); }
  cref(const cref& o) : p_(o.p_) { if (p_) p_->increment(); }
  cref& operator=(const cref& o) { if (p_ != o.p_) { release(); p_ = o.p_; if (p_) p_->increment(); } return *this; }
  ~cref() { release(); }
  T* get() const { return p_; }
 private:
  void release() { if (!p_) return; if (p_->unique()) delete p_; else p_->decrement(); p_ = nullptr; }
  T* p_;
};
PYBIND11_DECLARE_HOLDER_TYPE(T, cref<T>, true);

namespace {

// ---- numpy <-> gsl PODs
struct GslVec {
  gsl_vector* v;
  explicit GslVec(const py::array_t<double, py::array::c_style | py::array::forcecast>& a) : v(gsl_vector_calloc((size_t)a.size())) {
    memcpy(v->data, a.data(), sizeof(double) * (size_t)a.size());
  }
  ~GslVec() { gsl_vector_free(v); }
  GslVec(const GslVec&) = delete;
};
struct GslMat {
  gsl_matrix* m;
  explicit GslMat(const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
    if (a.ndim() != 2) throw jdimension_error("expected a 2-d array, got %d-d", (int)a.ndim());
    m = gsl_matrix_alloc((size_t)a.shape(0), (size_t)a.shape(1));
    memcpy(m->data, a.data(), sizeof(double) * (size_t)a.size());
  }
  ~GslMat() { gsl_matrix_free(m); }
  GslMat(const GslMat&) = delete;
};
struct GslCVec {
  gsl_vector_complex* v;
  explicit GslCVec(const py::array_t<cd, py::array::c_style | py::array::forcecast>& a) : v(gsl_vector_complex_calloc((size_t)a.size())) {
    memcpy(v->data, a.data(), sizeof(double) * 2 * (size_t)a.size());
  }
  ~GslCVec() { gsl_vector_complex_free(v); }
  GslCVec(const GslCVec&) = delete;
};
struct GslCMat {
  gsl_matrix_complex* m;
  explicit GslCMat(const py::array_t<cd, py::array::c_style | py::array::forcecast>& a) {
    if (a.ndim() != 2) throw jdimension_error("expected a 2-d array, got %d-d", (int)a.ndim());
    m = gsl_matrix_complex_alloc((size_t)a.shape(0), (size_t)a.shape(1));
    memcpy(m->data, a.data(), sizeof(double) * 2 * (size_t)a.size());
  }
  ~GslCMat() { gsl_matrix_complex_free(m); }
  GslCMat(const GslCMat&) = delete;
};

// views of node-owned storage (kept alive by `owner`), copies of transient views
py::array view(const gsl_vector_float* v, py::handle owner) { return py::array_t<float>({(py::ssize_t)v->size}, {(py::ssize_t)sizeof(float)}, v->data, owner); }
py::array view(const gsl_vector_complex* v, py::handle owner) { return py::array_t<cd>({(py::ssize_t)v->size}, {(py::ssize_t)sizeof(cd)}, reinterpret_cast<const cd*>(v->data), owner); }
py::array copy_of(const gsl_vector_complex* v) {
  py::array_t<cd> a((py::ssize_t)v->size);
  memcpy(static_cast<void*>(a.mutable_data()), v->data, sizeof(cd) * v->size);
  return std::move(a);
}
py::array copy_of(const gsl_matrix_complex* m) {
  py::array_t<cd> a({(py::ssize_t)m->size1, (py::ssize_t)m->size2});
  for (size_t i = 0; i < m->size1; i++) memcpy(static_cast<void*>(a.mutable_data(i, 0)), m->data + 2 * i * m->tda, sizeof(cd) * m->size2);
  return std::move(a);
}

// ---- a Python object as a source node (reference stream/pyStream.h:25-168)
template <class Base, class VecT, class T>
class PyStream : public Base, public BlockSource {
 public:
  PyStream(py::object c, const String& nm) : Base(size_of_(c), nm), cont_(c), iter_(c.attr("__iter__")()) {}
  ~PyStream() { py::gil_scoped_acquire g; cont_ = py::object(); iter_ = py::object(); }
  const VecT* next(int frame_no = -5) override {
    if (frame_no == this->frame_no_) return this->vector_;
    py::gil_scoped_acquire g;
    py::object o;
    try {
      o = py::hasattr(iter_, "next") ? iter_.attr("next")() : iter_.attr("__next__")();
    } catch (py::error_already_set& e) {
      if (e.matches(PyExc_StopIteration)) { this->is_end_ = true; throw jiterator_error("No more samples!"); }
      throw;
    }
    py::array_t<T, py::array::c_style | py::array::forcecast> a(o);
    if ((size_t)a.size() < this->size_) throw jdimension_error("Python source returned %d items, the stream has %d", (int)a.size(), (int)this->size_);
    memcpy(this->vector_->data, a.data(), sizeof(T) * this->size_);
    this->increment_();
    return this->vector_;
  }
  void reset() override {
    py::gil_scoped_acquire g;
    cont_.attr("reset")();
    iter_ = cont_.attr("__iter__")();
    Base::reset();
  }
  py::object python_object() const { return cont_; }
  // BlockSource: a GPU-backed Python beamformer (pybeamformer.*, or anything with device_block() -> complex [.. K][T], optionally
  // _output_version() / _advance_to(idx)) hands its whole block to a batching C++ consumer; plain iterators have none and are drained
  bool has_block() override { py::gil_scoped_acquire g; return py::hasattr(cont_, "device_block"); }
  unsigned long block_version() override {
    py::gil_scoped_acquire g;
    return py::hasattr(cont_, "_output_version") ? cont_.attr("_output_version")().template cast<unsigned long>() : 0ul;
  }
  const std::vector<float>& block(long& Tn) override {
    py::gil_scoped_acquire g;
    py::object b = cont_.attr("device_block")();
    if (py::hasattr(b, "detach")) b = b.attr("detach")().attr("cpu")().attr("numpy")();      // a torch tensor
    py::array_t<std::complex<float>, py::array::c_style | py::array::forcecast> a(b);
    if (a.ndim() < 2) throw jdimension_error("device_block() must return [.. K][T], got %d-d", (int)a.ndim());
    const size_t K = (size_t)a.shape(a.ndim() - 2);
    Tn = (long)a.shape(a.ndim() - 1);
    blk_.resize(2 * K * (size_t)Tn);
    memcpy(blk_.data(), a.data(), sizeof(float) * blk_.size());
    return blk_;
  }
  void advance_to(long idx) override {
    py::gil_scoped_acquire g;
    if (py::hasattr(cont_, "_advance_to")) cont_.attr("_advance_to")(idx);
  }
  // blocks of a bounded number of frames (modulated/modulated.h): _block_base() = stream index of the block's first frame,
  // _next_block() moves the object on to its following block and returns False at the end of the stream
  long block_base() override {
    py::gil_scoped_acquire g;
    return py::hasattr(cont_, "_block_base") ? cont_.attr("_block_base")().template cast<long>() : 0l;
  }
  bool next_block() override {
    py::gil_scoped_acquire g;
    return py::hasattr(cont_, "_next_block") ? cont_.attr("_next_block")().template cast<bool>() : false;
  }
 private:
  static unsigned size_of_(const py::object& c) { return c.attr("size")().cast<unsigned>(); }
  py::object cont_, iter_;
  std::vector<float> blk_;
};
typedef PyStream<VectorFloatFeatureStream, gsl_vector_float, float> PyVectorFloatFeatureStream;
typedef PyStream<VectorComplexFeatureStream, gsl_vector_complex, cd> PyVectorComplexFeatureStream;

// a bound node is used as it is; any other Python object with size() / __iter__ / next() / reset() becomes a source node (the
// reference wants the explicit PyVector*FeatureStreamPtr(obj) wrapper, stream/pyStream.h:25-168; both spellings work here)
VectorComplexFeatureStreamPtr as_cstream(const py::object& o)
{
  if (py::isinstance<VectorComplexFeatureStream>(o)) return VectorComplexFeatureStreamPtr(o.cast<VectorComplexFeatureStream*>());
  return VectorComplexFeatureStreamPtr(new PyVectorComplexFeatureStream(o, "PyVectorComplexFeatureStream"));
}
VectorFloatFeatureStreamPtr as_fstream(const py::object& o)
{
  if (py::isinstance<VectorFloatFeatureStream>(o)) return VectorFloatFeatureStreamPtr(o.cast<VectorFloatFeatureStream*>());
  return VectorFloatFeatureStreamPtr(new PyVectorFloatFeatureStream(o, "PyVectorFloatFeatureStream"));
}
py::array block_of(BlockSource& b)
{
  long T = 0;
  const std::vector<float>& y = b.block(T);
  const py::ssize_t K = T > 0 ? (py::ssize_t)(y.size() / 2 / (size_t)T) : 0;
  py::array_t<std::complex<float>> a({(py::ssize_t)1, K, (py::ssize_t)T});
  if (T > 0) memcpy(static_cast<void*>(a.mutable_data()), y.data(), sizeof(float) * 2 * (size_t)K * (size_t)T);
  return std::move(a);
}

template <class S, class C>
void bind_stream_methods(C& cls)
{
  cls.def("next", [](py::object self, int frame_no) { S& s = self.cast<S&>(); return view(s.next(frame_no), self); }, py::arg("frame_no") = -5)
      .def("__next__", [](py::object self) { S& s = self.cast<S&>(); return view(s.next(-5), self); })
      .def("__iter__", [](py::object self) { self.attr("reset")(); return self; })
      .def("current", [](py::object self) { S& s = self.cast<S&>(); return view(s.current(), self); })
      .def("reset", [](S& s) { s.reset(); })
      .def("size", &S::size)
      .def("is_end", [](S& s) { return s.is_end(); })
      .def("frame_no", [](S& s) { return s.frame_no(); })
      .def("name", [](S& s) { return std::string(s.name()); });
}

}  // namespace

PYBIND11_MODULE(_btk20cpp, m)
{
  m.doc() = "C++ node layer of the MI355X subband-beamforming engine (libbtk20hip.so) under the reference's class names";

  // ---- exceptions: jiterator_error ends an iteration, the rest keep their names
  static py::exception<j_error> ex_j(m, "j_error");
  static py::exception<j_error> ex_alloc(m, "jallocation_error", ex_j.ptr());
  static py::exception<j_error> ex_cons(m, "jconsistency_error", ex_j.ptr());
  static py::exception<j_error> ex_dim(m, "jdimension_error", ex_j.ptr());
  static py::exception<j_error> ex_idx(m, "jindex_error", ex_j.ptr());
  static py::exception<j_error> ex_io(m, "jio_error", ex_j.ptr());
  static py::exception<j_error> ex_par(m, "jparameter_error", ex_j.ptr());
  static py::exception<j_error> ex_num(m, "jnumeric_error", ex_j.ptr());
  py::register_exception_translator([](std::exception_ptr p) {
    try {
      if (p) std::rethrow_exception(p);
    } catch (j_error& e) {
      switch (e.getCode()) {
        case JITERATOR: PyErr_SetString(PyExc_StopIteration, e.what()); break;
        case JALLOCATION: py::set_error(ex_alloc, e.what()); break;
        case JCONSISTENCY: py::set_error(ex_cons, e.what()); break;
        case JDIMENSION: py::set_error(ex_dim, e.what()); break;
        case JINDEX: py::set_error(ex_idx, e.what()); break;
        case JIO: py::set_error(ex_io, e.what()); break;
        case JPARAMETER: py::set_error(ex_par, e.what()); break;
        case JNUMERIC: py::set_error(ex_num, e.what()); break;
        default: py::set_error(ex_j, e.what());
      }
    }
  });

  // ---- stream bases
  py::class_<VectorFloatFeatureStream, cref<VectorFloatFeatureStream>> vf(m, "VectorFloatFeatureStream");
  bind_stream_methods<VectorFloatFeatureStream>(vf);
  py::class_<VectorComplexFeatureStream, cref<VectorComplexFeatureStream>> vc(m, "VectorComplexFeatureStream");
  bind_stream_methods<VectorComplexFeatureStream>(vc);

  py::class_<PyVectorFloatFeatureStream, VectorFloatFeatureStream, cref<PyVectorFloatFeatureStream>>(m, "PyVectorFloatFeatureStreamPtr")
      .def(py::init([](py::object c, const std::string& nm) { return new PyVectorFloatFeatureStream(c, nm); }), py::arg("c"), py::arg("nm") = "PyVectorFloatFeatureStream")
      .def("python_object", &PyVectorFloatFeatureStream::python_object);
  py::class_<PyVectorComplexFeatureStream, VectorComplexFeatureStream, cref<PyVectorComplexFeatureStream>>(m, "PyVectorComplexFeatureStreamPtr")
      .def(py::init([](py::object c, const std::string& nm) { return new PyVectorComplexFeatureStream(c, nm); }), py::arg("c"), py::arg("nm") = "PyVectorComplexFeatureStream")
      .def("python_object", &PyVectorComplexFeatureStream::python_object);

  // ---- feature/feature.h
  py::class_<SampleFeature, VectorFloatFeatureStream, cref<SampleFeature>>(m, "SampleFeaturePtr")
      .def(py::init([](const std::string& fn, unsigned block_len, unsigned shift_len, bool pad_zeros, const std::string& nm) {
             return new SampleFeature(fn, block_len, shift_len, pad_zeros, nm);
           }), py::arg("fn") = "", py::arg("block_len") = 320, py::arg("shift_len") = 160, py::arg("pad_zeros") = false, py::arg("nm") = "Sample")
      .def("read", [](SampleFeature& s, const std::string& fn, int format, int samplerate, int chX, int chN, int cfrom, int to, int outsamplerate, float norm) {
             return s.read(fn, format, samplerate, chX, chN, cfrom, to, outsamplerate, norm);
           }, py::arg("fn"), py::arg("format") = 0, py::arg("samplerate") = 16000, py::arg("chX") = 1, py::arg("chN") = 1, py::arg("cfrom") = 0,
           py::arg("to") = -1, py::arg("outsamplerate") = -1, py::arg("norm") = 0.0f)
      .def("set_samples", [](SampleFeature& s, py::array_t<float, py::array::c_style | py::array::forcecast> a) { s.set_samples(a.data(), (size_t)a.size()); })
      .def("holds_pcm16", [](SampleFeature& s) { return s.pcm16() != NULL; })   // every loaded sample is an integer of the int16 range
      .def("samplerate", &SampleFeature::getSampleRate)
      .def("getSampleRate", &SampleFeature::getSampleRate)
      .def("getChanN", &SampleFeature::getChanN)
      .def("samplesN", &SampleFeature::samplesN)
      .def("write", [](SampleFeature& s, const std::string& fn, int format, int sampleRate) { s.write(fn, format, sampleRate); },
           py::arg("fn"), py::arg("format") = (int)(sndfile::SF_FORMAT_WAV | sndfile::SF_FORMAT_PCM_16), py::arg("sampleRate") = -1)
      .def("cut", &SampleFeature::cut, py::arg("cfrom"), py::arg("cto"))
      .def("randomize", &SampleFeature::randomize, py::arg("startX"), py::arg("endX"), py::arg("sigma2"))
      .def("exit", &SampleFeature::exit)
      .def("data", [](SampleFeature& s) {
             const gsl_vector_float* v = s.data();
             py::array_t<float> a((py::ssize_t)v->size);
             if (v->size) memcpy(a.mutable_data(), v->data, sizeof(float) * v->size);
             return a; })
      .def("dataDouble", [](SampleFeature& s) {
             const gsl_vector* v = s.dataDouble();
             py::array_t<double> a((py::ssize_t)v->size);
             if (v->size) memcpy(a.mutable_data(), v->data, sizeof(double) * v->size);
             return a; })
      .def("copySamples", [](SampleFeature& s, SampleFeature* src, unsigned cfrom, unsigned to) { SampleFeaturePtr p(src); s.copySamples(p, cfrom, to); },
           py::arg("src"), py::arg("cfrom"), py::arg("to"))
      .def("zeroMean", &SampleFeature::zeroMean)
      .def("addWhiteNoise", &SampleFeature::addWhiteNoise, py::arg("snr"))
      .def("setSamples", [](SampleFeature& s, py::array_t<double, py::array::c_style | py::array::forcecast> a, unsigned sampleRate) {
             GslVec v(a); s.setSamples(v.v, sampleRate); }, py::arg("samples"), py::arg("sampleRate"));

  // ---- modulated/modulated.h
  py::class_<OverSampledDFTAnalysisBank, VectorComplexFeatureStream, cref<OverSampledDFTAnalysisBank>>(m, "OverSampledDFTAnalysisBankPtr")
      .def(py::init([](py::object samp, py::array_t<double, py::array::c_style | py::array::forcecast> prototype, unsigned M, unsigned mm,
                       unsigned r, unsigned dct, const std::string& nm) {
             VectorFloatFeatureStreamPtr sp = as_fstream(samp);
             GslVec h(prototype);
             return new OverSampledDFTAnalysisBank(sp, h.v, M, mm, r, dct, nm);
           }), py::arg("samp"), py::arg("prototype"), py::arg("M"), py::arg("m"), py::arg("r"), py::arg("delay_compensation_type") = 0,
           py::arg("nm") = "OverSampledDFTAnalysisBank")
      .def("fftlen", &OverSampledDFTAnalysisBank::fftlen)
      .def("fftLen", &OverSampledDFTAnalysisBank::fftlen)
      .def("shiftlen", &OverSampledDFTAnalysisBank::shiftlen)
      .def("nBlocks", &OverSampledDFTAnalysisBank::nBlocks)
      .def("subSampRate", &OverSampledDFTAnalysisBank::subSampRate)
      .def("set_block_frames", &OverSampledDFTAnalysisBank::set_block_frames, py::arg("n"))
      .def("block_frames", &OverSampledDFTAnalysisBank::block_frames);
  py::class_<OverSampledDFTSynthesisBank, VectorFloatFeatureStream, cref<OverSampledDFTSynthesisBank>>(m, "OverSampledDFTSynthesisBankPtr")
      // source-less form (modulated/modulated.i:151-171): frames are pushed with input_source_vector(); registered first so that a
      // prototype array in the first position is not taken for a source object
      .def(py::init([](py::array_t<double, py::array::c_style | py::array::forcecast> prototype, unsigned M, unsigned mm, unsigned r,
                       unsigned dct, int gain_factor, const std::string& nm) {
             GslVec g(prototype);
             return new OverSampledDFTSynthesisBank(g.v, M, mm, r, dct, gain_factor, nm);
           }), py::arg("prototype"), py::arg("M"), py::arg("m"), py::arg("r") = 0, py::arg("delay_compensation_type") = 0,
           py::arg("gain_factor") = 1, py::arg("nm") = "OverSampledDFTSynthesisBank")
      .def(py::init([](py::object samp, py::array_t<double, py::array::c_style | py::array::forcecast> prototype, unsigned M, unsigned mm,
                       unsigned r, unsigned dct, int gain_factor, const std::string& nm) {
             VectorComplexFeatureStreamPtr sp = as_cstream(samp);
             GslVec g(prototype);
             return new OverSampledDFTSynthesisBank(sp, g.v, M, mm, r, dct, gain_factor, nm);
           }), py::arg("samp"), py::arg("prototype"), py::arg("M"), py::arg("m"), py::arg("r") = 0, py::arg("delay_compensation_type") = 0,
           py::arg("gain_factor") = 1, py::arg("nm") = "OverSampledDFTSynthesisBank")
      .def("input_source_vector", [](OverSampledDFTSynthesisBank& b, py::array_t<cd, py::array::c_style | py::array::forcecast> block) {
             GslCVec v(block); b.input_source_vector(v.v); }, py::arg("block"))
      .def("inputSourceVector", [](OverSampledDFTSynthesisBank& b, py::array_t<cd, py::array::c_style | py::array::forcecast> block) {
             GslCVec v(block); b.input_source_vector(v.v); }, py::arg("block"))
      .def("no_stream_feature", &OverSampledDFTSynthesisBank::no_stream_feature, py::arg("flag") = true)
      .def("doNotUseStreamFeature", &OverSampledDFTSynthesisBank::no_stream_feature, py::arg("flag") = true)
      .def("set_block_frames", &OverSampledDFTSynthesisBank::set_block_frames, py::arg("n"))
      .def("block_frames", &OverSampledDFTSynthesisBank::block_frames)
      // engine extension: the blocks the next calls of next() would return, as one float32 array [n][shiftlen] (n = 0: end of stream)
      .def("next_blocks", [](OverSampledDFTSynthesisBank& b, long max_blocks) {
             const float* p = NULL;
             const long n = b.next_blocks(max_blocks, &p);
             py::array_t<float> a({(py::ssize_t)n, (py::ssize_t)b.size()});
             if (n > 0) memcpy(a.mutable_data(), p, sizeof(float) * (size_t)n * b.size());
             return a;
           }, py::arg("max_blocks") = 0);

  // ---- beamformer/beamformer.h
  py::class_<SnapShotArray, cref<SnapShotArray>>(m, "SnapShotArrayPtr")
      .def(py::init([](unsigned fftLn, unsigned nChn) { return new SnapShotArray(fftLn, nChn); }), py::arg("fftLn"), py::arg("nChn"))
      .def("fftLen", &SnapShotArray::fftLen)
      .def("nChan", &SnapShotArray::nChan)
      .def("set_samples", [](SnapShotArray& a, py::array_t<cd, py::array::c_style | py::array::forcecast> s, unsigned chanX) {
             if ((unsigned)s.size() != a.fftLen()) throw jdimension_error("sample vector has %d entries, fftLen is %d", (int)s.size(), (int)a.fftLen());
             GslCVec v(s); a.set_samples(v.v, chanX); })
      .def("set_snapshots", [](SnapShotArray& a, py::array_t<cd, py::array::c_style | py::array::forcecast> s, unsigned fbinX) {
             GslCVec v(s); a.set_snapshots(v.v, fbinX); })
      .def("newSample", [](SnapShotArray& a, py::array_t<cd, py::array::c_style | py::array::forcecast> s, unsigned chanX) {
             if ((unsigned)s.size() != a.fftLen()) throw jdimension_error("sample vector has %d entries, fftLen is %d", (int)s.size(), (int)a.fftLen());
             GslCVec v(s); a.set_samples(v.v, chanX); })
      .def("getSnapShot", [](SnapShotArray& a, unsigned fbinX) { return copy_of(a.snapshot(fbinX)); })
      .def("update", [](SnapShotArray& a) { a.update(); })
      .def("zero", [](SnapShotArray& a) { a.zero(); })
      .def("snapshot", [](SnapShotArray& a, unsigned fbinX) { return copy_of(a.snapshot(fbinX)); });
  py::class_<SpectralMatrixArray, SnapShotArray, cref<SpectralMatrixArray>>(m, "SpectralMatrixArrayPtr")
      .def(py::init([](unsigned fftLn, unsigned nChn, float forgetFact) { return new SpectralMatrixArray(fftLn, nChn, forgetFact); }),
           py::arg("fftLn"), py::arg("nChn"), py::arg("forgetFact") = 0.95f)
      .def("matrix_f", [](SpectralMatrixArray& a, unsigned idx) { return copy_of(a.matrix_f(idx)); })
      .def("getSpecMatrix", [](SpectralMatrixArray& a, unsigned idx) { return copy_of(a.matrix_f(idx)); });

  py::class_<SubbandBeamformer, VectorComplexFeatureStream, cref<SubbandBeamformer>>(m, "SubbandBeamformer")
      .def("set_channel", [](SubbandBeamformer& b, py::object chan) { VectorComplexFeatureStreamPtr p = as_cstream(chan); b.set_channel(p); })
      .def("setChannel", [](SubbandBeamformer& b, py::object chan) { VectorComplexFeatureStreamPtr p = as_cstream(chan); b.set_channel(p); })
      .def("_device_snapshots_info", [](SubbandBeamformer& b) {      // (device pointer, K, N, T) of the complex64 [K][N][T] snapshot block
             void* p = b.device_snapshots();
             return py::make_tuple((size_t)reinterpret_cast<uintptr_t>(p), (long)(b.fftLen() / 2 + 1), (long)b.chanN(), b.num_frames()); })
      .def("clear_channel", [](SubbandBeamformer& b) { b.clear_channel(); })
      .def("clearChannel", [](SubbandBeamformer& b) { b.clear_channel(); })
      .def("chan_num", &SubbandBeamformer::chanN)
      .def("chanN", &SubbandBeamformer::chanN)
      .def("fftlen", &SubbandBeamformer::fftLen)
      .def("fftLen", &SubbandBeamformer::fftLen)
      .def("dim", &SubbandBeamformer::dim)
      .def("want_snapshots", &SubbandBeamformer::want_snapshots)   // every block from now on brings its snapshots along (staged path)
      .def("i16_stream", &SubbandBeamformer::i16_stream)   // the current stream's samples go up as 16-bit PCM (every source holds it)
      .def("snapshots_materialised", &SubbandBeamformer::snapshots_materialised)   // the current block's snapshots exist on the device
      .def("num_frames", &SubbandBeamformer::num_frames)                 // frames of the current block of snapshots
      .def("chunk_base", &SubbandBeamformer::chunk_base)                 // stream index of its first frame
      .def("set_block_frames", &SubbandBeamformer::set_block_frames, py::arg("n"))
      .def("block_frames", &SubbandBeamformer::block_frames)
      .def("set_block_quantum", &SubbandBeamformer::set_block_quantum, py::arg("q"))
      .def("block_quantum", &SubbandBeamformer::block_quantum)
      .def("is_half_band_shift", &SubbandBeamformer::is_half_band_shift)
      .def("snapshot_array_f", [](SubbandBeamformer& b, unsigned fbinX) { return copy_of(b.snapshot_array_f(fbinX)); })
      .def("snapshot_array", [](SubbandBeamformer& b) { SnapShotArrayPtr a = b.snapshot_array(); return cref<SnapShotArray>(a.operator->()); })
      .def("getSnapShotArray", [](SubbandBeamformer& b) { SnapShotArrayPtr a = b.snapshot_array(); return cref<SnapShotArray>(a.operator->()); });

  // BeamformerWeights (beamformer.h:26-97), owned by its beamformer node: whole-array views of the reference's per-bin accessors
  py::class_<BeamformerWeights, std::unique_ptr<BeamformerWeights, py::nodelete>>(m, "BeamformerWeights")
      .def("fftLen", &BeamformerWeights::fftLen)
      .def("chanN", &BeamformerWeights::chanN)
      .def("NC", &BeamformerWeights::NC)
      .def("isHalfBandShift", &BeamformerWeights::isHalfBandShift)
      .def("wq_f", [](BeamformerWeights& w, unsigned fbinX) { if (fbinX >= w.fftLen()) throw jindex_error("bin %d of %d", (int)fbinX, (int)w.fftLen()); return copy_of(w.wq_f(fbinX)); })
      .def("wl_f", [](BeamformerWeights& w, unsigned fbinX) { if (fbinX >= w.fftLen()) throw jindex_error("bin %d of %d", (int)fbinX, (int)w.fftLen()); return copy_of(w.wl_f(fbinX)); })
      .def_property_readonly("wq", [](BeamformerWeights& w) {
             py::array_t<cd> a({(py::ssize_t)w.fftLen(), (py::ssize_t)w.chanN()});
             memcpy(static_cast<void*>(a.mutable_data()), w.wq_v.data(), sizeof(cd) * w.wq_v.size()); return a; })
      .def_property_readonly("wl", [](BeamformerWeights& w) {
             py::array_t<cd> a({(py::ssize_t)w.fftLen(), (py::ssize_t)w.chanN()});
             memcpy(static_cast<void*>(a.mutable_data()), w.wl_v.data(), sizeof(cd) * w.wl_v.size()); return a; })
      .def_property_readonly("ta", [](BeamformerWeights& w) {
             py::array_t<cd> a({(py::ssize_t)w.fftLen(), (py::ssize_t)w.chanN()});
             memcpy(static_cast<void*>(a.mutable_data()), w.ta_v.data(), sizeof(cd) * w.ta_v.size()); return a; })
      .def_property_readonly("wa", [](BeamformerWeights& w) {
             py::array_t<cd> a({(py::ssize_t)w.fftLen(), (py::ssize_t)(w.chanN() - w.NC())});
             memcpy(static_cast<void*>(a.mutable_data()), w.wa_v.data(), sizeof(cd) * w.wa_v.size()); return a; })
      // the reference's auto / cross spectral densities of bin fbinX as an N x N matrix (entry [i][j], i <= j; rebuilt on demand)
      .def("CSDs", [](BeamformerWeights& w, unsigned fbinX) {
             if (fbinX >= w.fftLen()) throw jindex_error("bin %d of %d", (int)fbinX, (int)w.fftLen());
             gsl_vector_complex** c = w.CSDs();
             py::array_t<cd> a({(py::ssize_t)w.chanN(), (py::ssize_t)w.chanN()});
             memcpy(static_cast<void*>(a.mutable_data()), c[fbinX]->data, sizeof(cd) * w.chanN() * w.chanN()); return a; }, py::arg("fbinX"))
      .def("wp1", [](BeamformerWeights& w) { return copy_of(w.wp1()); })
      .def_property_readonly("B", [](BeamformerWeights& w) {
             py::array_t<cd> a({(py::ssize_t)w.fftLen(), (py::ssize_t)w.chanN(), (py::ssize_t)(w.chanN() - w.NC())});
             memcpy(static_cast<void*>(a.mutable_data()), w.B_v.data(), sizeof(cd) * w.B_v.size()); return a; });

  py::class_<SubbandDS, SubbandBeamformer, cref<SubbandDS>>(m, "SubbandDSPtr")
      .def(py::init([](unsigned fftlen, bool half_band_shift, const std::string& nm) { return new SubbandDS(fftlen, half_band_shift, nm); }),
           py::arg("fftlen") = 512, py::arg("half_band_shift") = false, py::arg("nm") = "SubbandDS")
      .def("calc_array_manifold_vectors", [](SubbandDS& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> d) {
             GslVec v(d); b.calc_array_manifold_vectors(fs, v.v); }, py::arg("samplerate"), py::arg("delays"))
      .def("calc_array_manifold_vectors_2", [](SubbandDS& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> dt,
                                               py::array_t<double, py::array::c_style | py::array::forcecast> dj) {
             GslVec a(dt), c(dj); b.calc_array_manifold_vectors_2(fs, a.v, c.v); })
      .def("calc_array_manifold_vectors_n", [](SubbandDS& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> dt,
                                               py::array_t<double, py::array::c_style | py::array::forcecast> djs, unsigned NC) {
             GslVec a(dt); GslMat c(djs); b.calc_array_manifold_vectors_n(fs, a.v, c.m, NC); }, py::arg("samplerate"), py::arg("delays_t"), py::arg("delays_js"), py::arg("NC") = 2)
      .def("get_weights", [](SubbandDS& b, unsigned fbinX) { return copy_of(b.get_weights(fbinX)); })
      .def("getWeights", [](SubbandDS& b, unsigned fbinX) { return copy_of(b.get_weights(fbinX)); })
      .def("beamformer_weight_object", [](SubbandDS& b, unsigned srcX) -> py::object {
             BeamformerWeights* w = b.beamformer_weight_object(srcX);
             return w ? py::cast(w, py::return_value_policy::reference) : py::object(py::none()); }, py::arg("srcX") = 0, py::keep_alive<0, 1>())
      // the block protocol a batching consumer uses (modulated/modulated.h BlockSource), under the names of the Python-side protocol
      .def("device_block", [](SubbandDS& b) { return block_of(b); })
      .def("fused_path", &SubbandDS::fused_path)   // the node's blocks come from the fused analysis -> apply kernel
      .def("_output_version", [](SubbandDS& b) { return b.block_version(); })
      .def("_advance_to", [](SubbandDS& b, long idx) { b.advance_to(idx); })
      .def("_block_base", [](SubbandDS& b) { return b.block_base(); })
      .def("_next_block", [](SubbandDS& b) { return b.next_block(); });

  py::class_<SubbandGSC, SubbandDS, cref<SubbandGSC>>(m, "SubbandGSCPtr")
      .def(py::init([](unsigned fftlen, bool half_band_shift, const std::string& nm) { return new SubbandGSC(fftlen, half_band_shift, nm); }),
           py::arg("fftlen") = 512, py::arg("half_band_shift") = false, py::arg("nm") = "SubbandGSC")
      .def("normalize_weight", &SubbandGSC::normalize_weight)
      .def("calc_gsc_weights", [](SubbandGSC& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> d) { GslVec v(d); b.calc_gsc_weights(fs, v.v); })
      .def("calc_gsc_weights_2", [](SubbandGSC& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> dt,
                                    py::array_t<double, py::array::c_style | py::array::forcecast> dj) { GslVec a(dt), c(dj); b.calc_gsc_weights_2(fs, a.v, c.v); })
      .def("calc_gsc_weights_n", [](SubbandGSC& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> dt,
                                    py::array_t<double, py::array::c_style | py::array::forcecast> djs, unsigned NC) {
             GslVec a(dt); GslMat c(djs); b.calc_gsc_weights_n(fs, a.v, c.m, NC); }, py::arg("samplerate"), py::arg("delays_t"), py::arg("delays_js"), py::arg("NC") = 2)
      .def("set_active_weights_f", [](SubbandGSC& b, unsigned fbinX, py::array_t<double, py::array::c_style | py::array::forcecast> packed) {
             GslVec v(packed); b.set_active_weights_f(fbinX, v.v); })
      .def("set_quiescent_weights_f", [](SubbandGSC& b, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> wq) {
             GslCVec v(wq); b.set_quiescent_weights_f(fbinX, v.v); })
      .def("zero_active_weights", &SubbandGSC::zero_active_weights)
      .def("write_fir_coeff", [](SubbandGSC& b, const std::string& fn, unsigned winType) { return b.write_fir_coeff(fn, winType); }, py::arg("fn"), py::arg("winType") = 1)
      .def("blocking_matrix", [](SubbandGSC& b, unsigned srcX, unsigned fbinX) { return copy_of(b.blocking_matrix(srcX, fbinX)); });

  py::class_<SubbandGSCRLS, SubbandGSC, cref<SubbandGSCRLS>>(m, "SubbandGSCRLSPtr")
      .def(py::init([](unsigned fftlen, bool half_band_shift, float mu, float sigma2, const std::string& nm) {
             return new SubbandGSCRLS(fftlen, half_band_shift, mu, sigma2, nm); }),
           py::arg("fftlen") = 512, py::arg("half_band_shift") = false, py::arg("mu") = 0.9f, py::arg("sigma2") = 0.0f, py::arg("nm") = "SubbandGSCRLS")
      .def("init_precision_matrix", &SubbandGSCRLS::init_precision_matrix, py::arg("sigma2") = 0.01f)
      .def("set_precision_matrix", [](SubbandGSCRLS& b, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> Pz) {
             GslCMat M(Pz); b.set_precision_matrix(fbinX, M.m); })
      .def("update_active_weight_vecotrs", &SubbandGSCRLS::update_active_weight_vecotrs)
      .def("set_quadratic_constraint", &SubbandGSCRLS::set_quadratic_constraint, py::arg("alpha"), py::arg("qctype") = 1);

  py::class_<SubbandMVDR, SubbandDS, cref<SubbandMVDR>>(m, "SubbandMVDRPtr")
      .def(py::init([](unsigned fftlen, bool half_band_shift, const std::string& nm) { return new SubbandMVDR(fftlen, half_band_shift, nm); }),
           py::arg("fftlen") = 512, py::arg("half_band_shift") = false, py::arg("nm") = "SubbandMVDR")
      .def("calc_mvdr_weights", &SubbandMVDR::calc_mvdr_weights, py::arg("samplerate"), py::arg("threshold") = 1.0E-8f, py::arg("calc_inverse_matrix") = true)
      .def("mvdr_weights", [](SubbandMVDR& b, unsigned fbinX) { return copy_of(b.mvdr_weights(fbinX)); })
      .def("set_noise_spatial_spectral_matrix", [](SubbandMVDR& b, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> Rnn) {
             GslCMat M(Rnn); return b.set_noise_spatial_spectral_matrix(fbinX, M.m); })
      .def("noise_spatial_spectral_matrix", [](SubbandMVDR& b, unsigned fbinX) -> py::object {      // None while no matrix is set (the reference returns NULL)
             const gsl_matrix_complex* m = b.noise_spatial_spectral_matrix(fbinX); return m ? py::object(copy_of(m)) : py::object(py::none()); })
      .def("set_diffuse_noise_model", [](SubbandMVDR& b, py::array_t<double, py::array::c_style | py::array::forcecast> mpos, float fs, float sspeed) {
             GslMat M(mpos); return b.set_diffuse_noise_model(M.m, fs, sspeed); }, py::arg("mic_positions"), py::arg("samplerate"), py::arg("sspeed") = 343740.0f)
      .def("set_all_diagonal_loading", &SubbandMVDR::set_all_diagonal_loading)
      .def("set_diagonal_looading", &SubbandMVDR::set_diagonal_looading)
      .def("divide_all_nondiagonal_elements", &SubbandMVDR::divide_all_nondiagonal_elements)
      .def("divide_nondiagonal_elements", &SubbandMVDR::divide_nondiagonal_elements)
      .def("identity_fallbacks", &SubbandMVDR::identity_fallbacks)
      .def("set_svd_rule", &SubbandMVDR::set_svd_rule, py::arg("rule"))
      .def("svd_rule", &SubbandMVDR::svd_rule)
      .def("csvdc_not_converged", &SubbandMVDR::csvdc_not_converged);

  py::class_<SubbandMVDRGSC, SubbandMVDR, cref<SubbandMVDRGSC>>(m, "SubbandMVDRGSCPtr")
      .def(py::init([](unsigned fftlen, bool half_band_shift, const std::string& nm) { return new SubbandMVDRGSC(fftlen, half_band_shift, nm); }),
           py::arg("fftlen") = 512, py::arg("half_band_shift") = false, py::arg("nm") = "SubbandMVDR")
      .def("normalize_weight", &SubbandMVDRGSC::normalize_weight)
      .def("set_active_weights_f", [](SubbandMVDRGSC& b, unsigned fbinX, py::array_t<double, py::array::c_style | py::array::forcecast> packed) {
             GslVec v(packed); b.set_active_weights_f(fbinX, v.v); })
      .def("zero_active_weights", &SubbandMVDRGSC::zero_active_weights)
      .def("calc_blocking_matrix1", [](SubbandMVDRGSC& b, float fs, py::array_t<double, py::array::c_style | py::array::forcecast> d) {
             GslVec v(d); return b.calc_blocking_matrix1(fs, v.v); })
      .def("calc_blocking_matrix2", &SubbandMVDRGSC::calc_blocking_matrix2)
      .def("upgrade_blocking_matrix", &SubbandMVDRGSC::upgrade_blocking_matrix)
      .def("blocking_matrix_output", [](SubbandMVDRGSC& b, int outChanX) { return copy_of(b.blocking_matrix_output(outChanX)); }, py::arg("outChanX") = 0);

  // ---- postfilter/postfilter.h
  py::class_<ZelinskiPostFilter, VectorComplexFeatureStream, cref<ZelinskiPostFilter>>(m, "ZelinskiPostFilterPtr")
      .def(py::init([](py::object output, unsigned fftlen, double alpha, int type, int min_frames, const std::string& nm) {
             VectorComplexFeatureStreamPtr p = as_cstream(output);
             return new ZelinskiPostFilter(p, fftlen, alpha, type, min_frames, nm);
           }), py::arg("output"), py::arg("fftlen"), py::arg("alpha") = 0.6, py::arg("type") = 2, py::arg("min_frames") = 0, py::arg("nm") = "ZelinskPostFilter")
      .def("set_beamformer", [](ZelinskiPostFilter& f, SubbandDS* bf) { SubbandDSPtr p(bf); f.set_beamformer(p); })
      .def("setBeamformer", [](ZelinskiPostFilter& f, SubbandDS* bf) { SubbandDSPtr p(bf); f.set_beamformer(p); })
      .def("set_snapshot_array", [](ZelinskiPostFilter& f, SnapShotArray* a) { SnapShotArrayPtr p(a); f.set_snapshot_array(p); }, py::arg("snapShotArray"))
      .def("setSnapShotArray", [](ZelinskiPostFilter& f, SnapShotArray* a) { SnapShotArrayPtr p(a); f.set_snapshot_array(p); }, py::arg("snapShotArray"))
      .def("set_array_manifold_vector", [](ZelinskiPostFilter& f, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> v, bool halfBandShift, unsigned NC) {
             GslCVec g(v); f.set_array_manifold_vector(fbinX, g.v, halfBandShift, NC); },
           py::arg("fbinX"), py::arg("arrayManifoldVector"), py::arg("halfBandShift"), py::arg("NC") = 1)
      .def("setArrayManifoldVector", [](ZelinskiPostFilter& f, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> v, bool halfBandShift, unsigned NC) {
             GslCVec g(v); f.set_array_manifold_vector(fbinX, g.v, halfBandShift, NC); },
           py::arg("fbinX"), py::arg("arrayManifoldVector"), py::arg("halfBandShift"), py::arg("NC") = 1)
      .def("weights_object", [](py::object self) -> py::object {            // bf_weights_ (postfilter.h:104); None before any weights exist
             ZelinskiPostFilter& f = self.cast<ZelinskiPostFilter&>();
             BeamformerWeights* w = f.weights_object();
             // the handle keeps the post-filter alive (reference_internal): the filter owns the object it made for
             // set_array_manifold_vector(), and keeps its beamformer -- the owner otherwise -- alive itself
             return w ? py::cast(w, py::return_value_policy::reference_internal, self) : py::none(); })
      .def("getPostFilterWeights", [](ZelinskiPostFilter& f) { return copy_of(f.postfilter_weights()); })
      .def("postfilter_weights", [](ZelinskiPostFilter& f) { return copy_of(f.postfilter_weights()); })
      .def("device_block", [](ZelinskiPostFilter& f) { return block_of(f); })
      .def("_output_version", [](ZelinskiPostFilter& f) { return f.block_version(); })
      .def("_advance_to", [](ZelinskiPostFilter& f, long idx) { f.advance_to(idx); })
      .def("_block_base", [](ZelinskiPostFilter& f) { return f.block_base(); })
      .def("_next_block", [](ZelinskiPostFilter& f) { return f.next_block(); });
  py::class_<McCowanPostFilter, ZelinskiPostFilter, cref<McCowanPostFilter>>(m, "McCowanPostFilterPtr")
      .def(py::init([](py::object output, unsigned fftlen, double alpha, int type, int min_frames, float threshold, const std::string& nm) {
             VectorComplexFeatureStreamPtr p = as_cstream(output);
             return new McCowanPostFilter(p, fftlen, alpha, type, min_frames, threshold, nm);
           }), py::arg("output"), py::arg("fftlen"), py::arg("alpha") = 0.6, py::arg("type") = 2, py::arg("min_frames") = 0, py::arg("threshold") = 0.99f,
           py::arg("nm") = "McCowanPostFilter")
      .def("set_diffuse_noise_model", [](McCowanPostFilter& f, py::array_t<double, py::array::c_style | py::array::forcecast> mpos, double fs, double sspeed) {
             GslMat M(mpos); return f.set_diffuse_noise_model(M.m, fs, sspeed); }, py::arg("mic_positions"), py::arg("samplerate"), py::arg("sspeed") = 343740.0)
      .def("set_noise_spatial_spectral_matrix", [](McCowanPostFilter& f, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> Rnn) {
             GslCMat M(Rnn); return f.set_noise_spatial_spectral_matrix(fbinX, M.m); })
      .def("noise_spatial_spectral_matrix", [](McCowanPostFilter& f, unsigned fbinX) -> py::object {
             const gsl_matrix_complex* m = f.noise_spatial_spectral_matrix(fbinX); return m ? py::object(copy_of(m)) : py::object(py::none()); })
      .def("set_all_diagonal_loading", &McCowanPostFilter::set_all_diagonal_loading)
      .def("set_diagonal_looading", &McCowanPostFilter::set_diagonal_looading)
      .def("divide_nondiagonal_elements", &McCowanPostFilter::divide_nondiagonal_elements)
      .def("divide_all_nondiagonal_elements", &McCowanPostFilter::divide_all_nondiagonal_elements)
      .def("setAllLevelsOfDiagonalLoading", &McCowanPostFilter::set_all_diagonal_loading)
      .def("setLevelOfDiagonalLoading", &McCowanPostFilter::set_diagonal_looading)
      .def("divideNonDiagonalElements", &McCowanPostFilter::divide_nondiagonal_elements)
      .def("divideAllNonDiagonalElements", &McCowanPostFilter::divide_all_nondiagonal_elements)
      .def("setNoiseSpatialSpectralMatrix", [](McCowanPostFilter& f, unsigned fbinX, py::array_t<cd, py::array::c_style | py::array::forcecast> Rnn) {
             GslCMat M(Rnn); return f.set_noise_spatial_spectral_matrix(fbinX, M.m); })
      .def("getNoiseSpatialSpectralMatrix", [](McCowanPostFilter& f, unsigned fbinX) { return copy_of(f.noise_spatial_spectral_matrix(fbinX)); })
      .def("set_svd_rule", [](McCowanPostFilter& f, const std::string& rule) { f.set_svd_rule(rule); }, py::arg("rule"))
      .def("svd_rule", [](McCowanPostFilter& f) { return std::string(f.svd_rule()); });
  py::class_<LefkimmiatisPostFilter, McCowanPostFilter, cref<LefkimmiatisPostFilter>>(m, "LefkimmiatisPostFilterPtr")
      .def(py::init([](py::object output, unsigned fftlen, double min_sv, unsigned fbin_x1, double alpha, int type, int min_frames,
                       float threshold, const std::string& nm) {
             VectorComplexFeatureStreamPtr p = as_cstream(output);
             return new LefkimmiatisPostFilter(p, fftlen, min_sv, fbin_x1, alpha, type, min_frames, threshold, nm);
           }), py::arg("output"), py::arg("fftlen"), py::arg("min_sv") = 1.0E-8, py::arg("fbin_x1") = 0, py::arg("alpha") = 0.6, py::arg("type") = 2,
           py::arg("min_frames") = 0, py::arg("threshold") = 0.99f, py::arg("nm") = "LefkimmiatisPostFilter")
      .def("calc_inverse_noise_spatial_spectral_matrix", &LefkimmiatisPostFilter::calc_inverse_noise_spatial_spectral_matrix);

  // ---- many utterance graphs advanced as one launch (beamformer/beamformer.h, SubbandGraphPool)
  py::class_<SubbandGraphPool, cref<SubbandGraphPool>>(m, "SubbandGraphPoolPtr")
      .def(py::init([]() { return new SubbandGraphPool(); }))
      .def("add", [](SubbandGraphPool& p, SubbandDS& bf, OverSampledDFTSynthesisBank& syn) {
             SubbandDSPtr b(&bf); OverSampledDFTSynthesisBankPtr s(&syn); p.add(b, s); }, py::arg("beamformer"), py::arg("synthesis"))
      .def("size", &SubbandGraphPool::size)
      .def("__len__", &SubbandGraphPool::size)
      .def("rounds", &SubbandGraphPool::rounds)
      .def("is_end", &SubbandGraphPool::is_end, py::arg("g"))
      .def("reset", &SubbandGraphPool::reset)
      // one output block per graph: a list with a numpy view of the pool's own buffer per graph, None for a graph that has ended;
      // StopIteration when every graph has ended
      .def("next", [](py::object self) {
             SubbandGraphPool& p = self.cast<SubbandGraphPool&>();
             bool ok; { py::gil_scoped_release rel; ok = p.next(); }
             if (!ok) throw py::stop_iteration();
             py::list out;
             for (unsigned g = 0; g < p.size(); g++) { const gsl_vector_float* v = p.output(g); if (v) out.append(view(v, self)); else out.append(py::none()); }
             return out; })
      .def("__next__", [](py::object self) { return self.attr("next")(); })
      .def("__iter__", [](py::object self) { self.attr("reset")(); return self; })
      // engine extension: a round at a time -- a list with one float32 array [n_g][shiftlen] per graph (n_g = 0: none in this round),
      // None when every graph has ended
      .def("next_round", [](SubbandGraphPool& p) -> py::object {
             bool ok; { py::gil_scoped_release rel; ok = p.next_round(); }
             if (!ok) return py::none();
             py::list out;
             for (unsigned g = 0; g < p.size(); g++) {
               const float* q = NULL;
               const long n = p.round_blocks(g, &q);
               const py::ssize_t D = n > 0 ? (py::ssize_t)p.output(g)->size : 0;
               py::array_t<float> a({(py::ssize_t)n, D});
               if (n > 0) memcpy(a.mutable_data(), q, sizeof(float) * (size_t)n * (size_t)D);
               out.append(a);
             }
             return out; });
  m.def("node_synchronize", []() { btk_node_synchronize(); });     // wait for everything this thread's nodes have launched
  m.def("node_alloc_counts", []() { long d = 0, h = 0; btk_node_alloc_counts(&d, &h); return py::make_tuple(d, h); });   // (hipMalloc, hipHostMalloc) calls so far

  // ---- dereverberation/dereverberation.h
  py::class_<MultiChannelWPEDereverberation, cref<MultiChannelWPEDereverberation>>(m, "MultiChannelWPEDereverberationPtr")
      .def(py::init([](unsigned subbands_num, unsigned channels_num, unsigned lower_num, unsigned upper_num, unsigned iterations_num, double load_db,
                       double band_width, double diagonal_bias, double samplerate) {
             return new MultiChannelWPEDereverberation(subbands_num, channels_num, lower_num, upper_num, iterations_num, load_db, band_width, diagonal_bias, samplerate);
           }), py::arg("subbands_num"), py::arg("channels_num"), py::arg("lower_num"), py::arg("upper_num"), py::arg("iterations_num") = 2,
           py::arg("load_db") = -20.0, py::arg("band_width") = 0.0, py::arg("diagonal_bias") = 0.0, py::arg("samplerate") = 16000.0)
      .def("size", &MultiChannelWPEDereverberation::size)
      .def("reset", &MultiChannelWPEDereverberation::reset)
      .def("set_input", [](MultiChannelWPEDereverberation& w, py::object s) { VectorComplexFeatureStreamPtr p = as_cstream(s); w.set_input(p); })
      .def("estimate_filter", &MultiChannelWPEDereverberation::estimate_filter, py::arg("start_frame_no") = 0, py::arg("frame_num") = -1)
      .def("reset_filter", &MultiChannelWPEDereverberation::reset_filter)
      .def("next_speaker", &MultiChannelWPEDereverberation::next_speaker)
      .def("print_objective_func", &MultiChannelWPEDereverberation::print_objective_func)
      .def("frame_no", &MultiChannelWPEDereverberation::frame_no);
  py::class_<MultiChannelWPEDereverberationFeature, VectorComplexFeatureStream, cref<MultiChannelWPEDereverberationFeature>>(m, "MultiChannelWPEDereverberationFeaturePtr")
      .def(py::init([](MultiChannelWPEDereverberation* source, unsigned channel_no, unsigned primary_channel_no, const std::string& nm) {
             MultiChannelWPEDereverberationPtr p(source);
             return new MultiChannelWPEDereverberationFeature(p, channel_no, primary_channel_no, nm);
           }), py::arg("source"), py::arg("channel_no"), py::arg("primary_channel_no") = 0, py::arg("nm") = "MultiChannelWPEDereverberationFeature");
  py::class_<SingleChannelWPEDereverberationFeature, VectorComplexFeatureStream, cref<SingleChannelWPEDereverberationFeature>>(m, "SingleChannelWPEDereverberationFeaturePtr")
      .def(py::init([](py::object samples, unsigned lower_num, unsigned upper_num, unsigned iterations_num, double load_db, double band_width,
                       double samplerate, const std::string& nm) {
             VectorComplexFeatureStreamPtr p = as_cstream(samples);
             return new SingleChannelWPEDereverberationFeature(p, lower_num, upper_num, iterations_num, load_db, band_width, samplerate, nm);
           }), py::arg("samples"), py::arg("lower_num"), py::arg("upper_num"), py::arg("iterations_num") = 2, py::arg("load_db") = -20.0,
           py::arg("band_width") = 0.0, py::arg("samplerate") = 16000.0, py::arg("nm") = "SingleChannelWPEDereverberationFeature")
      .def("estimate_filter", &SingleChannelWPEDereverberationFeature::estimate_filter, py::arg("start_frame_no") = 0, py::arg("frame_num") = -1)
      .def("reset_filter", &SingleChannelWPEDereverberationFeature::reset_filter)
      .def("next_speaker", &SingleChannelWPEDereverberationFeature::next_speaker)
      .def("print_objective_func", &SingleChannelWPEDereverberationFeature::print_objective_func);
}
