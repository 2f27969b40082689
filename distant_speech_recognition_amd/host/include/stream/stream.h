// stream/stream.h -- FeatureStream<Type,item_type> with the reference's interface, unchanged
// (reference stream/stream.h:16-86): the drop-in boundary of the node layer.
#pragma once
#include "common/refcount.h"
#include "gsl_compat.h"

template <typename Type> struct btk_vec_ops;
template <> struct btk_vec_ops<gsl_vector_char>    { static gsl_vector_char* make(size_t n) { return gsl_vector_char_calloc(n); } static void drop(gsl_vector_char* v) { gsl_vector_char_free(v); } };
template <> struct btk_vec_ops<gsl_vector_short>   { static gsl_vector_short* make(size_t n) { return gsl_vector_short_calloc(n); } static void drop(gsl_vector_short* v) { gsl_vector_short_free(v); } };
template <> struct btk_vec_ops<gsl_vector_float>   { static gsl_vector_float* make(size_t n) { return gsl_vector_float_calloc(n); } static void drop(gsl_vector_float* v) { gsl_vector_float_free(v); } };
template <> struct btk_vec_ops<gsl_vector>         { static gsl_vector* make(size_t n) { return gsl_vector_calloc(n); } static void drop(gsl_vector* v) { gsl_vector_free(v); } };
template <> struct btk_vec_ops<gsl_vector_complex> { static gsl_vector_complex* make(size_t n) { return gsl_vector_complex_calloc(n); } static void drop(gsl_vector_complex* v) { gsl_vector_complex_free(v); } };

template <typename Type, typename item_type>
class FeatureStream : public Countable {
 public:
  virtual ~FeatureStream() { btk_vec_ops<Type>::drop(vector_); }
  const String& name() const { return name_; }
  unsigned size() const { return size_; }
  virtual const Type* next(int frame_no = -5) = 0;
  const Type* current() {
    if (frame_no_ < 0) throw jconsistency_error("Frame index (%d) < 0.", frame_no_);
    return next(frame_no_);
  }
  bool is_end() { return is_end_; }
  virtual void reset() { frame_no_ = frame_reset_no_; is_end_ = false; }
  virtual int frame_no() const { return frame_no_; }
  size_t itemsize() { return sizeof(item_type); }
 protected:
  FeatureStream(unsigned sz, const String& nm)
      : frame_reset_no_(-1), size_(sz), frame_no_(-1), vector_(btk_vec_ops<Type>::make(sz)), is_end_(false), name_(nm) {}
  void increment_() { frame_no_++; }
  const int frame_reset_no_;
  const unsigned size_;
  int frame_no_;
  Type* vector_;
  bool is_end_;
 private:
  const String name_;
};

typedef FeatureStream<gsl_vector_char, char>            VectorCharFeatureStream;
typedef FeatureStream<gsl_vector_short, short>          VectorShortFeatureStream;
typedef FeatureStream<gsl_vector_float, float>          VectorFloatFeatureStream;
typedef FeatureStream<gsl_vector, double>               VectorFeatureStream;
typedef FeatureStream<gsl_vector_complex, gsl_complex>  VectorComplexFeatureStream;
typedef refcountable_ptr<VectorCharFeatureStream>       VectorCharFeatureStreamPtr;
typedef refcountable_ptr<VectorShortFeatureStream>      VectorShortFeatureStreamPtr;
typedef refcountable_ptr<VectorFloatFeatureStream>      VectorFloatFeatureStreamPtr;
typedef refcountable_ptr<VectorFeatureStream>           VectorFeatureStreamPtr;
typedef refcountable_ptr<VectorComplexFeatureStream>    VectorComplexFeatureStreamPtr;
