// postfilter/postfilter.h -- ZelinskiPostFilter (reference postfilter/postfilter.h:74-108).
#pragma once
#include "beamformer/beamformer.h"

typedef enum { TYPE_ZELINSKI1_REAL = 0x01, TYPE_ZELINSKI1_ABS = 0x02, TYPE_APAB = 0x04, TYPE_ZELINSKI2 = 0x08,
               NO_USE_POST_FILTER = 0x00 } PostfilterType;

class ZelinskiPostFilter : public VectorComplexFeatureStream {
 public:
  ZelinskiPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha = 0.6, int type = 2,
                     int minFrames = 0, const String& nm = "ZelinskPostFilter");
  ~ZelinskiPostFilter();
  virtual const gsl_vector_complex* next(int frame_no = -5);
  virtual void reset();
  void set_beamformer(SubbandDSPtr& beamformer);
  void setBeamformer(SubbandDSPtr& beamformer) { set_beamformer(beamformer); }
  const gsl_vector_complex* postfilter_weights();
 private:
  void compute_(long from_frame);
  unsigned fftLen_;
  VectorComplexFeatureStreamPtr samp_;
  PostfilterType type_;
  double alpha_;
  int min_frames_;
  SubbandDSPtr bf_ptr_;
  bool has_bf_ptr_;
  std::vector<float> Yhost_, wlast_;
  long T_;
  bool prepared_;
  unsigned long bf_version_;
  void *dPhi_, *dPsi_, *dWl_;
  gsl_vector_complex* wp1_;
};
typedef Inherit<ZelinskiPostFilter, VectorComplexFeatureStreamPtr> ZelinskiPostFilterPtr;
