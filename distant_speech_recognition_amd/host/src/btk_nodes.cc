// btk_nodes.cc -- C++ node layer: the reference's FeatureStream nodes for the beamforming hot path,
// implemented over the C-ABI of libbtkhip (include/btkhip.h).
//
// A node computes BLOCKS of frames on the device (modulated/modulated.h, BlockSource) and serves frames from a host mirror,
// so next() keeps the reference's per-frame contract: node-owned buffer, same-frame caching, consecutive frame numbers,
// jiterator_error("end of samples!") at the end.  Weight changes between frames recompute only the frames not yet served.
// Steady state (common/devmem.h): one non-NULL HIP stream per host thread, node-owned grow-only device / pinned buffers,
// asynchronous copies from / to pinned memory; a beamformer over analysis banks whose snapshots nobody asks for runs the
// FUSED analysis -> apply kernel (btk_fb_analysis_bf) and hands its block to the synthesis bank on the device.
#include <cstdlib>
#include <sched.h>
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>

#include "feature/feature.h"
#include "modulated/modulated.h"
#include "beamformer/beamformer.h"
#include "postfilter/postfilter.h"
#include "dereverberation/dereverberation.h"

namespace {

typedef std::complex<double> cd;

void check_abi(int rc)
{
  if (rc == BTK_OK) return;
  const char* msg = btk_last_error();
  switch (rc) {
    case BTK_ERR_DIMENSION: throw jdimension_error("%s", msg);
    case BTK_ERR_CONSISTENCY: throw jconsistency_error("%s", msg);
    case BTK_ERR_ALLOCATION: throw jallocation_error("%s", msg);
    case BTK_ERR_PARAMETER: throw jparameter_error("%s", msg);
    case BTK_ERR_NUMERIC: throw jnumeric_error("%s", msg);
    default: throw j_error("%s", msg);
  }
}

void check_hip(hipError_t e, const char* what)
{
  if (e == hipSuccess) return;
  if (e == hipErrorOutOfMemory) throw jallocation_error("%s: %s", what, hipGetErrorString(e));
  throw j_error("%s: %s", what, hipGetErrorString(e));
}

// Helper threads for the one bulk job of the node layer that is pure host memory traffic: analysis banks drawing a round of input
// blocks out of SampleFeature sources into their pinned windows (64 channels x 8192 blocks x 1 KB = 537 MB per block at C0: 30 ms
// on one core, which is what a drop-in caller's frame rate was made of).  The reference is single-threaded and so is every node's
// interface; the helpers only ever run SampleFeature::next_blocks of DIFFERENT source objects side by side, never a Python source,
// and the caller waits for them.  BTK_NODE_THREADS sets the number (default: the cores this process may use, at most 8; 1 = none).
class PullPool {
 public:
  static PullPool& get() { static PullPool* p = new PullPool(); return *p; }     // (never destroyed: no join at process exit)
  int threads() const { return nthreads_; }
  // fn(i) for i in [0, n): the calling thread takes part; the first exception of any of them is rethrown here
  void parallel_for(int n, const std::function<void(int)>& fn)
  {
    if (n <= 0) return;
    if (nthreads_ <= 1 || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; n_ = n; next_ = 0; done_ = 0; err_ = nullptr; gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return done_ == n_; });
    fn_ = NULL;
    if (err_) { std::exception_ptr e = err_; err_ = nullptr; lk.unlock(); std::rethrow_exception(e); }
  }
 private:
  PullPool() : nthreads_(1), fn_(NULL), n_(0), next_(0), done_(0), gen_(0)
  {
    int n = (int)std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1, CPU_COUNT(&set));
#endif
    n = n < 1 ? 1 : (n > 8 ? 8 : n);
    const char* e = getenv("BTK_NODE_THREADS");
    if (e && *e) n = std::max(1, atoi(e));
    nthreads_ = n;
    for (int i = 1; i < nthreads_; i++) std::thread([this] { loop(); }).detach();
  }
  void work()
  {
    for (;;) {
      int i;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (!fn_ || next_ >= n_) return;
        i = next_++;
      }
      try { (*fn_)(i); } catch (...) { std::lock_guard<std::mutex> lk(mu_); if (!err_) err_ = std::current_exception(); }
      std::lock_guard<std::mutex> lk(mu_);
      if (++done_ == n_) cv_done_.notify_all();
    }
  }
  void loop()
  {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
      }
      work();
    }
  }
  int nthreads_;
  std::mutex mu_;
  std::condition_variable cv_, cv_done_;
  const std::function<void(int)>* fn_;
  int n_, next_, done_;
  unsigned long gen_;
  std::exception_ptr err_;
};

std::atomic<long> g_device_allocs(0), g_pinned_allocs(0);
std::atomic<long long> g_pull_ns(0), g_upload_ns(0), g_device_ns(0);
struct ScopedNs {                                   // adds the scope's wall time to one of the counters of btk_node_timers()
  std::atomic<long long>& acc;
  std::chrono::steady_clock::time_point t0;
  explicit ScopedNs(std::atomic<long long>& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
  ~ScopedNs() { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

hipStream_t nstream() { return static_cast<hipStream_t>(btk_node_stream()); }
// a host thread's second stream: uploads of the block AFTER the current one (16-bit streams, SubbandBeamformer::prefetch_next_)
hipStream_t cstream()
{
  static thread_local hipStream_t st = NULL;
  if (!st) check_hip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  return st;
}
void nsync() { check_hip(hipStreamSynchronize(nstream()), "hipStreamSynchronize"); }

// one-off device blocks of the design-time paths (weight design, WPE estimate, CSD rebuild): not the per-block steady state
void* dev_alloc(size_t bytes)
{
  void* p = NULL;
  check_hip(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc");
  g_device_allocs++;
  return p;
}
void dev_free(void* p) { if (p) (void)hipFree(p); }
// copies between pageable host memory and the device, complete on return, ordered on the node stream like every launch
void h2d(void* d, const void* h, size_t n)
{
  if (!n) return;
  check_hip(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, nstream()), "hipMemcpyAsync H2D");
  nsync();
}
void d2h(void* h, const void* d, size_t n)
{
  if (!n) return;
  check_hip(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, nstream()), "hipMemcpyAsync D2H");
  nsync();
}
// the asynchronous forms: `h` is pinned memory that stays untouched until the stream has passed the copy
void h2d_async(void* d, const void* h, size_t n)
{
  if (n) check_hip(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, nstream()), "hipMemcpyAsync H2D");
}
void d2h_async(void* h, const void* d, size_t n)
{
  if (n) check_hip(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, nstream()), "hipMemcpyAsync D2H");
}
void d2d_async(void* dst, const void* src, size_t n)
{
  if (n) check_hip(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, nstream()), "hipMemcpyAsync D2D");
}
// n rows of `bytes` bytes that lie in SEPARATE pinned allocations (every channel's source owns its utterance) -> rows `pitch` bytes
// apart of one device block, on `stream`; the bytes of a row behind `bytes` are zeroed.  `tab` [pinned, n entries, src filled in by
// the caller] and the rows stay untouched until the stream has passed the call.  ONE kernel that reads the host memory through the
// table (btk_gather_rows: the link's 57 GB/s whatever the rows' length) where a copy per row reaches 27 GB/s at 0.5 MB per row and
// 49 at 4 MB (profiles/r06_ubench_host_gather.txt); rows that are not 16-byte aligned -- a shift length that is no multiple of 8
// samples -- take the copies.  BTK_NODE_GATHER=0: always the copies.
void upload_rows(btk_row_t* tab, unsigned n, size_t bytes, char* dst, size_t pitch, hipStream_t stream)
{
  const char* env = getenv("BTK_NODE_GATHER");                      // (read per upload: a test switches it inside one process)
  const bool off = env && atoi(env) == 0;
  if (!n || !pitch) return;
  bool aligned = !off && ((reinterpret_cast<uintptr_t>(dst) | pitch) & 15) == 0;
  for (unsigned c = 0; c < n && aligned; c++) aligned = (reinterpret_cast<uintptr_t>(tab[c].src) & 15) == 0;
  if (aligned) {
    for (unsigned c = 0; c < n; c++) tab[c].bytes = (long)bytes;
    check_abi(btk_gather_rows(tab, dst, (int)n, (long)pitch, stream));
    return;
  }
  if (bytes < pitch) check_hip(hipMemset2DAsync(dst + bytes, pitch, 0, pitch - bytes, n, stream), "hipMemset2DAsync");
  if (bytes)
    for (unsigned c = 0; c < n; c++)
      check_hip(hipMemcpyAsync(dst + (size_t)c * pitch, tab[c].src, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
}
// rows x width bytes between two pitched device arrays
void d2d_2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows)
{
  if (!width || !rows) return;
  check_hip(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToDevice, nstream()), "hipMemcpy2DAsync D2D");
}
void dev_zero_async(void* d, size_t n)
{
  if (n) check_hip(hipMemsetAsync(d, 0, n, nstream()), "hipMemsetAsync");
}

// [K][T] complex64 -> frame t as gsl_vector_complex of M bins with conjugate mirror
void serve_frame(const float* Y, long T, unsigned M, long t, gsl_vector_complex* out)
{
  const unsigned K = M / 2 + 1;
  for (unsigned k = 0; k < K; k++) {
    const double re = Y[2 * ((size_t)k * T + t)], im = Y[2 * ((size_t)k * T + t) + 1];
    out->data[2 * k] = re; out->data[2 * k + 1] = im;
    if (k > 0 && k < M / 2) { out->data[2 * (M - k)] = re; out->data[2 * (M - k) + 1] = -im; }
  }
}

// halfBandShift: every bin has its own output, nothing is mirrored.  Y [M][T] complex64
void serve_frame_all_bins(const float* Y, long T, unsigned M, long t, gsl_vector_complex* out)
{
  for (unsigned k = 0; k < M; k++) { out->data[2 * k] = Y[2 * ((size_t)k * T + t)]; out->data[2 * k + 1] = Y[2 * ((size_t)k * T + t) + 1]; }
}

// drain a complex node: frames [T][K] complex64 (bins 0..M/2)
long drain_complex(VectorComplexFeatureStreamPtr& src, unsigned M, std::vector<float>& frames)
{
  const unsigned K = M / 2 + 1;
  long T = 0;
  for (;;) {
    const gsl_vector_complex* v;
    try { v = src->next(); } catch (jiterator_error&) { break; }
    frames.resize((size_t)(T + 1) * K * 2);
    for (unsigned k = 0; k < K; k++) {
      frames[2 * ((size_t)T * K + k)] = (float)v->data[2 * k];
      frames[2 * ((size_t)T * K + k) + 1] = (float)v->data[2 * k + 1];
    }
    T++;
  }
  return T;
}

}  // namespace

// ================================================================================ node stream, device / pinned buffers
void* btk_node_stream()
{
  // one stream per host thread, never destroyed (a thread_local destructor would run after the HIP runtime's own teardown)
  static thread_local hipStream_t st = NULL;
  if (!st) check_hip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  return st;
}
void btk_node_synchronize() { nsync(); }
void btk_node_alloc_counts(long* device_allocs, long* pinned_allocs)
{
  if (device_allocs) *device_allocs = g_device_allocs.load();
  if (pinned_allocs) *pinned_allocs = g_pinned_allocs.load();
}

void btk_node_timers(double* pull_s, double* upload_s, double* device_s)
{
  if (pull_s) *pull_s = 1e-9 * (double)g_pull_ns.load();
  if (upload_s) *upload_s = 1e-9 * (double)g_upload_ns.load();
  if (device_s) *device_s = 1e-9 * (double)g_device_ns.load();
}
void btk_node_timers_reset() { g_pull_ns = 0; g_upload_ns = 0; g_device_ns = 0; }

void* DeviceBuffer::ensure(size_t bytes)
{
  if (bytes <= cap_ && p_) return p_;
  const size_t want = std::max(bytes ? bytes : (size_t)16, cap_ + cap_ / 2);
  if (p_) { nsync(); (void)hipFree(p_); p_ = NULL; cap_ = 0; }          // launches that still read the old block
  check_hip(hipMalloc(&p_, want), "hipMalloc");
  g_device_allocs++;
  cap_ = want;
  return p_;
}
void DeviceBuffer::release() { if (p_) (void)hipFree(p_); p_ = NULL; cap_ = 0; }

void* PinnedBuffer::ensure(size_t bytes) { return ensure_keep(bytes, 0); }
void* PinnedBuffer::ensure_keep(size_t bytes, size_t keep_bytes)
{
  if (bytes <= cap_ && p_) return p_;
  const size_t want = std::max(bytes ? bytes : (size_t)16, cap_ + cap_ / 2);
  void* q = NULL;
  // (mapped: the gather kernel of upload_rows reads this memory itself; portable: whichever device of the process runs it)
  check_hip(hipHostMalloc(&q, want, hipHostMallocPortable | hipHostMallocMapped), "hipHostMalloc");
  g_pinned_allocs++;
  if (p_) {
    nsync();                                                                // copies that still read the old block
    if (keep_bytes) memcpy(q, p_, std::min(keep_bytes, cap_));
    (void)hipHostFree(p_);
  }
  p_ = q; cap_ = want;
  return p_;
}
void PinnedBuffer::release() { if (p_) (void)hipHostFree(p_); p_ = NULL; cap_ = 0; }

// ================================================================================ SampleFeature
SampleFeature::SampleFeature(const String& fn, unsigned blockLen, unsigned shiftLen, bool padZeros, const String& nm)
    : VectorFloatFeatureStream(blockLen, nm), have_samples_(false), norm_(0.0f), shiftLen_(shiftLen), cur_(0),
      pad_zeros_(padZeros), samplerate_(0), nChan_(1), format_(sndfile::SF_FORMAT_WAV | sndfile::SF_FORMAT_PCM_16),
      copy_fsamples_(NULL), copy_dsamples_(NULL), pcm16_state_(0), samples_gen_(0)
{
  if (fn != "") read(fn);
  is_end_ = false;
}

SampleFeature::~SampleFeature()
{
  if (copy_fsamples_) gsl_vector_float_free(copy_fsamples_);
  if (copy_dsamples_) gsl_vector_free(copy_dsamples_);
}

unsigned SampleFeature::read(const String& fn, int, int, int chX, int, int cfrom, int to, int, float norm)
{
  norm_ = norm;
  samples_.clear(); have_samples_ = false;
  FILE* fp = fopen(fn.c_str(), "rb");
  if (!fp) throw jio_error("Could not open file %s.", fn.c_str());
  unsigned char hdr[12];
  if (fread(hdr, 1, 12, fp) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) {
    fclose(fp);
    throw jio_error("%s is not a RIFF/WAVE file", fn.c_str());
  }
  int channels = 1, bits = 16, fmt = 1;
  std::vector<short> raw;
  for (;;) {
    unsigned char ck[8];
    if (fread(ck, 1, 8, fp) != 8) break;
    const unsigned len = ck[4] | (ck[5] << 8) | (ck[6] << 16) | ((unsigned)ck[7] << 24);
    if (!memcmp(ck, "fmt ", 4)) {
      std::vector<unsigned char> f(len);
      if (fread(f.data(), 1, len, fp) != len) break;
      fmt = f[0] | (f[1] << 8); channels = f[2] | (f[3] << 8);
      samplerate_ = f[4] | (f[5] << 8) | (f[6] << 16) | (f[7] << 24);
      bits = f[14] | (f[15] << 8);
      if (len & 1) fseek(fp, 1, SEEK_CUR);
    } else if (!memcmp(ck, "data", 4)) {
      raw.resize(len / 2);
      size_t got = fread(raw.data(), 2, raw.size(), fp);
      raw.resize(got);
      break;
    } else {
      fseek(fp, len + (len & 1), SEEK_CUR);
    }
  }
  fclose(fp);
  if (fmt != 1 || bits != 16) throw jio_error("Only 16-bit PCM WAV is supported (%s)", fn.c_str());
  nChan_ = channels;
  const long nfr = (long)(raw.size() / channels);
  if (to < 0 || to >= nfr) to = (int)nfr - 1;                       // feature.cc:286-293
  if (cfrom < 0) cfrom = 0;
  if (cfrom > to || cfrom > nfr) throw jio_error("Cannot load samples from %d to %d.", cfrom, to);
  if (chX > channels || chX < 1) {
    if (chX == 0) throw jconsistency_error("Multi-channel read is not yet supported.");
    throw jconsistency_error("Selected channel out of range of available channels.");
  }
  // norm == 0: un-normalised (int16-scale) floats; otherwise libsndfile's normalised floats (x / 32768) times norm
  const float scale = (norm == 0.0f) ? 1.0f : (1.0f / 32768.0f) * (norm != 1.0f ? norm : 1.0f);
  std::vector<float> s((size_t)(to - cfrom + 1));
  for (long i = cfrom; i <= to; i++) s[(size_t)(i - cfrom)] = (float)raw[(size_t)i * channels + (chX - 1)] * scale;
  format_ = sndfile::SF_FORMAT_WAV | sndfile::SF_FORMAT_PCM_16;
  set_samples(s.data(), s.size());
  return (unsigned)samples_.size();
}

void SampleFeature::write(const String& fn, int format, int sampleRate)
{
  if (format != (sndfile::SF_FORMAT_WAV | sndfile::SF_FORMAT_PCM_16))
    throw jio_error("Error opening file %s: only SF_FORMAT_WAV|SF_FORMAT_PCM_16 is written (format 0x%x asked).", fn.c_str(), format);
  (void)sampleRate;                                                  // no SRCONV: the file keeps samplerate_ (feature.cc:439-443)
  FILE* fp = fopen(fn.c_str(), "wb");
  if (!fp) throw jio_error("Error opening file %s.", fn.c_str());
  const size_t n = samples_.size();
  std::vector<short> pcm(n);
  // norm_ == 0: un-normalised floats are int16 values; else samples / norm_ are [-1, 1) values scaled by 0x7FFF (libsndfile's
  // float -> short conversion with SFC_SET_NORM_FLOAT on); rounding lrintf as libsndfile does, clipped instead of wrapped
  const float k = (norm_ == 0.0f) ? 1.0f : 32767.0f / ((norm_ != 1.0f) ? norm_ : 1.0f);
  for (size_t i = 0; i < n; i++) {
    long v = lrintf(samples_[i] * k);
    pcm[i] = (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
  const unsigned rate = (unsigned)(samplerate_ > 0 ? samplerate_ : 16000), bytes = (unsigned)(2 * n);
  unsigned char h[44];
  auto le32 = [&](int o, unsigned v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; h[o + 2] = (v >> 16) & 255; h[o + 3] = (v >> 24) & 255; };
  auto le16 = [&](int o, unsigned v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; };
  memcpy(h, "RIFF", 4); le32(4, 36 + bytes); memcpy(h + 8, "WAVEfmt ", 8); le32(16, 16); le16(20, 1); le16(22, 1);
  le32(24, rate); le32(28, rate * 2); le16(32, 2); le16(34, 16); memcpy(h + 36, "data", 4); le32(40, bytes);
  const bool ok = fwrite(h, 1, 44, fp) == 44 && fwrite(pcm.data(), 2, n, fp) == n;
  fclose(fp);
  if (!ok) fprintf(stderr, "unable to write all samples to %s\n", fn.c_str());
}

void SampleFeature::set_samples(const float* samples, size_t n)
{
  samples_.assign(samples, samples + n);
  have_samples_ = true;
  cur_ = 0;
  reset();
  is_end_ = false;
  samples_changed_();
  (void)pcm16();                       // load time, like read(): the 16-bit view of the utterance (a WAV is 16-bit PCM to begin with)
}

static bool node_i16_enabled()                        // BTK_NODE_I16=0: no 16-bit views, no 16-bit streams (read whenever it matters)
{
  const char* e = getenv("BTK_NODE_I16");
  return !(e && *e == '0');
}

// The loaded samples as 16-bit PCM (feature/feature.h): every sample must be an integer of the int16 range for the filter bank
// to see the same values either way.  (-0.0f counts as 0: a WAV read never produces it, numpy.rint does.  The only trace it could
// leave is the SIGN of an exactly-zero polyphase sum -- all m samples of a tap column zero and at least one of them -0.0f.)
const short* SampleFeature::pcm16()
{
  const size_t sz = size();
  if (!have_samples_ || samples_.empty() || shiftLen_ != sz || sz == 0 || !node_i16_enabled()) return NULL;
  if (pcm16_state_ == 0) {
    const size_t n = samples_.size(), padded = (n / sz + 3) * sz;
    short* q = NULL;
    try { q = static_cast<short*>(pcm16_.ensure(sizeof(short) * padded)); } catch (j_error&) { q = NULL; }
    if (!q) { pcm16_state_ = -1; return NULL; }              // no pinned memory (a host without a device): the float path remains
    const float* x = samples_.data();
    unsigned bad = 0;
    for (size_t i = 0; i < n; i++) {
      const float v = x[i], c = fminf(fmaxf(v, -32768.0f), 32767.0f);
      const int iv = (int)c;
      bad |= (unsigned)((float)iv != v);
      q[i] = (short)iv;
    }
    memset(q + n, 0, sizeof(short) * (padded - n));
    pcm16_state_ = bad ? -1 : 1;
  }
  return pcm16_state_ == 1 ? static_cast<const short*>(pcm16_.get()) : NULL;
}

void SampleFeature::setSamples(const gsl_vector* samples, unsigned sampleRate)      // feature.cc:669-679
{
  samplerate_ = (int)sampleRate;
  samples_.resize(samples->size);
  for (size_t i = 0; i < samples->size; i++) samples_[i] = (float)gsl_vector_get(samples, i);
  have_samples_ = true;
  samples_changed_();
  reset();
}

void SampleFeature::copySamples(SampleFeaturePtr& src, unsigned cfrom, unsigned to)  // feature.cc:651-667
{
  size_t n;
  if (to == 0) {
    n = src->samples_.size();
  } else {
    if (to <= cfrom) throw jindex_error("cfrom = %d and to = %d are inconsistent.", cfrom, to);
    if (to >= src->samples_.size()) to = (unsigned)src->samples_.size() - 1;
    n = to - cfrom;
  }
  if ((size_t)cfrom + n > src->samples_.size()) throw jindex_error("cfrom = %d and to = %d are inconsistent.", cfrom, to);
  std::vector<float> tmp(src->samples_.begin() + cfrom, src->samples_.begin() + cfrom + n);
  samples_.swap(tmp);
  have_samples_ = true;
  samples_changed_();
}

const gsl_vector_float* SampleFeature::data()
{
  if (copy_fsamples_) gsl_vector_float_free(copy_fsamples_);
  copy_fsamples_ = gsl_vector_float_calloc(samples_.size());
  for (size_t i = 0; i < samples_.size(); i++) gsl_vector_float_set(copy_fsamples_, i, samples_[i]);
  return copy_fsamples_;
}

const gsl_vector* SampleFeature::dataDouble()
{
  if (copy_dsamples_) gsl_vector_free(copy_dsamples_);
  copy_dsamples_ = gsl_vector_calloc(samples_.size());
  for (size_t i = 0; i < samples_.size(); i++) gsl_vector_set(copy_dsamples_, i, samples_[i]);
  return copy_dsamples_;
}

void SampleFeature::zeroMean()                                       // feature.cc:556-570: truncation toward zero after the int16 clamp
{
  if (!have_samples_) throw jconsistency_error("Must first load data before setting mean to zero.");
  double mean = 0.0;
  for (size_t i = 0; i < samples_.size(); i++) mean += samples_[i];
  mean /= (double)samples_.size();
  for (size_t i = 0; i < samples_.size(); i++) {
    const double x = samples_[i] - mean;
    samples_[i] = (float)(int)(x < -32768.0 ? -32768.0 : (x < 32767.0 ? x : 32767.0));
  }
  samples_changed_();
}

void SampleFeature::cut(unsigned cfrom, unsigned cto)                // feature.cc:572-586 (both bounds inclusive)
{
  if (cfrom >= cto) throw j_error("Cut bounds (%d,%d) do not match.", cfrom, cto);
  if (cto >= samples_.size()) throw j_error("Do not have enough samples (%d,%d).", cto, (int)samples_.size());
  std::vector<float> tmp(samples_.begin() + cfrom, samples_.begin() + cto + 1);
  samples_.swap(tmp);
  samples_changed_();
}

// feature.cc:589-603: gsl_rng_default (mt19937, default seed: GSL turns seed 0 into 4357) and gsl_ran_gaussian (polar
// Box-Muller over gsl_rng_uniform_pos-style uniforms: x, y = -1 + 2 u), restated from GSL's published sources; the reference
// hands sigma2 to it as the standard deviation
void SampleFeature::randomize(int startX, int endX, double sigma2)
{
  printf("Randomizing from %6.2f to %6.2f\n", startX / 16000.0, endX / 16000.0);
  unsigned long mt[624];
  int mti = 624;
  mt[0] = 4357UL;
  for (int i = 1; i < 624; i++) mt[i] = (1812433253UL * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned long)i) & 0xffffffffUL;
  auto next_u32 = [&]() -> unsigned long {
    if (mti >= 624) {
      for (int kk = 0; kk < 624; kk++) {
        const unsigned long y = (mt[kk] & 0x80000000UL) | (mt[(kk + 1) % 624] & 0x7fffffffUL);
        mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
      }
      mti = 0;
    }
    unsigned long k = mt[mti++];
    k ^= (k >> 11); k ^= (k << 7) & 0x9d2c5680UL; k ^= (k << 15) & 0xefc60000UL; k ^= (k >> 18);
    return k & 0xffffffffUL;
  };
  auto uniform = [&]() { return (double)next_u32() / 4294967296.0; };
  for (int n = startX; n <= endX; n++) {
    if (n < 0 || (size_t)n >= samples_.size()) throw jindex_error("randomize: sample %d outside of the %d loaded", n, (int)samples_.size());
    double x, y, r2;
    do { x = -1 + 2 * uniform(); y = -1 + 2 * uniform(); r2 = x * x + y * y; } while (r2 > 1.0 || r2 == 0);
    samples_[(size_t)n] = (float)(sigma2 * y * sqrt(-2.0 * log(r2) / r2));
  }
  samples_changed_();
}

// feature.cc:391-427, literally: the noise is drawn into SHORT integers (rand() truncated, then (x / max - 0.5) truncated
// again), so every noise sample is 0 or -(desired level / mean |noise|) -- kept as the reference computes it, time-seeded
void SampleFeature::addWhiteNoise(float snr)
{
  const size_t n = samples_.size();
  if (n == 0) return;
  std::vector<short> noise(n);
  double avgSig = 0.0, avgNoi = 0.0;
  int max = -2147483647 - 1;
  for (size_t i = 0; i < n; i++) avgSig += fabsf(samples_[i]);
  avgSig /= (double)n;
  srand((unsigned)time(NULL));
  for (size_t i = 0; i < n; i++) { noise[i] = (short)rand(); if (noise[i] > max) max = noise[i]; }
  for (size_t i = 0; i < n; i++) { noise[i] = (short)((noise[i] / (float)max) - 0.5); avgNoi += std::abs((int)noise[i]); }
  avgNoi /= (double)n;
  const double desiredNoA = avgSig / pow(10.0, snr / 20.0);
  for (size_t i = 0; i < n; i++) noise[i] = (short)(desiredNoA * noise[i] / avgNoi);
  for (size_t i = 0; i < n; i++) samples_[i] += noise[i];
  samples_changed_();
}

const gsl_vector_float* SampleFeature::next(int frame_no)
{
  if (is_end_) throw jiterator_error("end of samples!");
  if (frame_no == frame_no_) return vector_;
  if (frame_no >= 0 && frame_no - 1 != frame_no_)
    throw jindex_error("Problem in Feature %s: %d != %d\n", name().c_str(), frame_no - 1, frame_no_);
  const size_t ttl = samples_.size();
  if (!have_samples_ || cur_ >= ttl) {
    is_end_ = true; have_samples_ = false; samples_.clear();
    throw jiterator_error("end of samples!");
  }
  if (cur_ + size() >= ttl) {
    if (pad_zeros_) {
      gsl_vector_float_set_zero(vector_);
      for (size_t i = 0; i < ttl - cur_; i++) vector_->data[i] = samples_[cur_ + i];
    } else {
      is_end_ = true; have_samples_ = false; samples_.clear();
      throw jiterator_error("end of samples!");
    }
  } else {
    for (unsigned i = 0; i < size(); i++) vector_->data[i] = samples_[cur_ + i];
  }
  cur_ += shiftLen_;
  increment_();
  return vector_;
}

// next() nmax times in a row, the blocks written one after the other to dst (an analysis bank filling its sample window): the
// same state transitions as the loop of next() calls it replaces -- frame counter, read position, zero padding of the last
// block, and at the end of the samples exactly what the throwing call does (is_end_, samples dropped) with the count returned
// instead of the exception.  vector_ holds the last block handed out, as after next().
long SampleFeature::next_blocks(float* dst, long nmax)
{
  long n = 0;
  const unsigned sz = size();
  while (n < nmax) {
    if (is_end_) break;
    const size_t ttl = samples_.size();
    if (!have_samples_ || cur_ >= ttl) { is_end_ = true; have_samples_ = false; samples_.clear(); break; }
    float* o = dst + (size_t)n * sz;
    if (cur_ + sz >= ttl) {
      if (!pad_zeros_) { is_end_ = true; have_samples_ = false; samples_.clear(); break; }
      memset(o, 0, sizeof(float) * sz);
      memcpy(o, samples_.data() + cur_, sizeof(float) * (ttl - cur_));
    } else {
      memcpy(o, samples_.data() + cur_, sizeof(float) * sz);
    }
    cur_ += shiftLen_;
    increment_();
    n++;
  }
  if (n > 0) memcpy(vector_->data, dst + (size_t)(n - 1) * sz, sizeof(float) * sz);
  return n;
}

// next() nmax times in a row without handing the blocks out: the caller (an analysis bank that uploads the utterance as 16-bit
// PCM) reads them from pcm16() -- block j of this call is the samples first_sample + j shiftLen ... of it, the zero padding of
// the last block included.  State transitions as in next_blocks(); vector_ holds the last block, as after next().
long SampleFeature::advance_blocks(long nmax, size_t* first_sample)
{
  long n = 0;
  const unsigned sz = size();
  const short* q = pcm16();
  if (!q && !is_end_ && have_samples_ && !samples_.empty())
    throw jconsistency_error("SampleFeature %s: advance_blocks() without a 16-bit view of the samples\n", name().c_str());
  if (first_sample) *first_sample = cur_;
  if (nmax <= 0 || is_end_) return 0;
  // closed form of the loop of next() calls: block j starts at cur_ + j shiftLen_ and is handed out while it starts inside the
  // samples (zero-padded tail) or, without padding, while it ENDS before the last sample (next(): `cur_ + size() >= ttl` ends it)
  const size_t ttl = samples_.size();
  long avail = 0;
  if (have_samples_ && cur_ < ttl) {
    if (pad_zeros_) avail = (long)((ttl - cur_ + shiftLen_ - 1) / shiftLen_);
    else if (cur_ + sz < ttl) avail = (long)((ttl - sz - cur_ + shiftLen_ - 1) / shiftLen_);
  }
  n = avail < nmax ? avail : nmax;
  const size_t last = cur_ + (size_t)(n > 0 ? n - 1 : 0) * shiftLen_;
  cur_ += (size_t)n * shiftLen_;
  frame_no_ += (int)n;
  if (n > 0 && q) for (unsigned i = 0; i < sz; i++) vector_->data[i] = (float)q[last + i];
  if (n < nmax) { is_end_ = true; have_samples_ = false; samples_.clear(); }       // the call after the last block: next() ends the stream
  return n;
}

// ================================================================================ analysis bank
long btk_default_block_frames()
{
  const char* e = getenv("BTK_BLOCK_FRAMES");
  if (!e || !*e) return 8192;
  const long v = atol(e);
  return v < 0 ? 0 : v;
}

OverSampledDFTAnalysisBank::OverSampledDFTAnalysisBank(VectorFloatFeatureStreamPtr& samp, gsl_vector* prototype, unsigned M,
                                                       unsigned m, unsigned r, unsigned delayCompensationType, const String& nm)
    : VectorComplexFeatureStream(M, nm), samp_(samp), M_(M), m_(m), r_(r), D_(M >> r), dct_(delayCompensationType),
      plan_(NULL), src16_(NULL), src16_pos0_(0), src16_gen_(0), block_frames_(btk_default_block_frames()), win_b0_(0), nblk_(0), eos_(false),
      chunk_base_(0), chunk_len_(0)
{
  if (prototype->size != (size_t)M * m)
    throw jconsistency_error("Prototype sizes do not match (%d vs. %d).", (int)prototype->size, (int)(M * m));
  if (samp_->size() != D_) throw jdimension_error("Input block length (%d) != D_ (%d)\n", samp_->size(), D_);
  std::vector<double> h(prototype->size);
  for (size_t i = 0; i < h.size(); i++) h[i] = gsl_vector_get(prototype, i);
  check_abi(btk_fb_create(&plan_, (int)M, (int)m, (int)r, (int)delayCompensationType, 0, h.data()));
}

OverSampledDFTAnalysisBank::~OverSampledDFTAnalysisBank() { btk_fb_destroy(plan_); }

void OverSampledDFTAnalysisBank::append_(const float* blocks, long n)
{
  const size_t have = (size_t)(nblk_ - win_b0_) * D_;
  float* w = static_cast<float*>(win_.ensure_keep(sizeof(float) * (have + (size_t)n * D_), sizeof(float) * have));
  if (blocks) memcpy(w + have, blocks, sizeof(float) * (size_t)n * D_);
  nblk_ += n;
}

// the window sized for one more round (+ the m R blocks of history + a beamformer's block quantum): a steady-state round then
// allocates nothing; an unbounded round (block_frames() == 0) grows geometrically as it pulls
void OverSampledDFTAnalysisBank::reserve_round()
{
  const size_t have = (size_t)(nblk_ - win_b0_) * D_;
  if (block_frames_ > 0)
    win_.ensure_keep(sizeof(float) * (have + (size_t)(block_frames_ + (long)m_ * (1L << r_) + 64) * D_), sizeof(float) * have);
}

// the source is a SampleFeature and rounds are bounded: pull_more() is then plain host memory traffic on objects no other bank
// shares, and a beamformer node may run the pull_more() of several of its banks side by side (PullPool)
bool OverSampledDFTAnalysisBank::parallel_pull_ok() const
{
  return block_frames_ > 0 && !src16_ && dynamic_cast<SampleFeature*>(samp_.operator->()) != NULL;
}

bool OverSampledDFTAnalysisBank::i16_source_ok() const
{
  SampleFeature* sf = dynamic_cast<SampleFeature*>(samp_.operator->());
  return sf && nblk_ == 0 && !eos_ && sf->size() == D_ && sf->shiftlen() == D_ && sf->pcm16() != NULL;
}

void OverSampledDFTAnalysisBank::begin_i16()
{
  SampleFeature* sf = dynamic_cast<SampleFeature*>(samp_.operator->());
  if (!i16_source_ok()) throw jconsistency_error("OverSampledDFTAnalysisBank %s: the source holds no 16-bit PCM\n", name().c_str());
  src16_ = sf->pcm16();
  size_t pos = 0;
  (void)sf->advance_blocks(0, &pos);                    // where the source stands: block 0 of this stream
  src16_pos0_ = pos;
  src16_gen_ = sf->samples_generation();
}

// one round of input: at most block_frames() blocks of D samples from the upstream node (modulated.cc:419-438 pulls one per frame)
bool OverSampledDFTAnalysisBank::pull_more()
{
  if (eos_) return false;
  long got = 0;
  SampleFeature* sf = dynamic_cast<SampleFeature*>(samp_.operator->());
  if (src16_) {
    // 16-bit mode: the blocks stay where they are (SampleFeature::pcm16); the source only moves on
    for (;;) {
      const long ask = block_frames_ == 0 ? (1L << 20) : block_frames_ - got;
      if (ask <= 0) break;
      size_t pos = 0;
      const long n = sf->advance_blocks(ask, &pos);
      if (n > 0 && (pos != src16_pos0_ + (size_t)nblk_ * D_ || sf->samples_generation() != src16_gen_))
        throw jconsistency_error("OverSampledDFTAnalysisBank %s: the source was moved or its samples changed while the bank streamed "
                                 "them as 16-bit PCM; reset() the graph after changing an utterance\n", name().c_str());
      nblk_ += n; got += n;
      if (n < ask) { eos_ = true; break; }
    }
    return got > 0;
  }
  reserve_round();
  if (sf && sf->size() == D_) {
    // a SampleFeature hands its blocks over in bulk, straight into the window (SampleFeature::next_blocks == the loop below)
    for (;;) {
      const long ask = block_frames_ == 0 ? 4096 : block_frames_ - got;
      if (ask <= 0) break;
      const size_t h = (size_t)(nblk_ - win_b0_) * D_;
      float* w = static_cast<float*>(win_.ensure_keep(sizeof(float) * (h + (size_t)ask * D_), sizeof(float) * h));
      const long n = sf->next_blocks(w + h, ask);
      nblk_ += n; got += n;
      if (n < ask) { eos_ = true; break; }
    }
    return got > 0;
  }
  while (block_frames_ == 0 || got < block_frames_) {
    const gsl_vector_float* b;
    try { b = samp_->next(); } catch (jiterator_error&) { eos_ = true; break; }
    append_(b->data, 1);
    got++;
  }
  return got > 0;
}

// frame t reads the samples (t + laN + 1 - m R) D .. (t + laN + 1) D - 1 (csrc/fb_kernels.hip: closed form of the ring of
// modulated.cc:375-409); once the source has ended the remaining pd frames read zeros beyond it (btk_fb_analysis_num_frames)
long OverSampledDFTAnalysisBank::frames_ready() const
{
  const long laN = btk_fb_lookahead(plan_), pd = btk_fb_processing_delay(plan_);
  if (eos_) return nblk_ < laN ? 0 : nblk_ - laN + pd;
  return nblk_ > laN ? nblk_ - laN : 0;
}

long OverSampledDFTAnalysisBank::first_block_of_frame(long t) const
{
  const long b = t + btk_fb_lookahead(plan_) + 1 - (long)m_ * (1L << r_);
  return b < 0 ? 0 : b;
}

void OverSampledDFTAnalysisBank::release_before(long t)
{
  if (src16_) return;                                   // the utterance stays with its source
  const long keep = std::min(first_block_of_frame(t), nblk_);
  if (keep > win_b0_) {
    float* w = static_cast<float*>(win_.get());
    const size_t drop = (size_t)(keep - win_b0_) * D_, left = (size_t)(nblk_ - keep) * D_;
    if (left) memmove(w, w + drop, sizeof(float) * left);
    win_b0_ = keep;
  }
}

// the next block of frames of a bank that is pulled frame by frame (a beamformer node batches its banks itself)
bool OverSampledDFTAnalysisBank::load_chunk_()
{
  const long f0 = chunk_base_ + chunk_len_;
  while (!eos_ && frames_ready() <= f0) pull_more();
  const long f1 = frames_ready();
  if (f1 <= f0) return false;
  if (src16_)
    throw jconsistency_error("OverSampledDFTAnalysisBank %s is a channel of a beamformer node that streams it as 16-bit PCM; pull the "
                             "bank through the beamformer or on its own, not both\n", name().c_str());
  // a bank is either a channel of a beamformer node (which releases the samples its blocks are done with) or pulled frame by
  // frame; both at once would find the window already cut -- frames computed as if the stream began later: refuse instead
  if (win_b0_ > first_block_of_frame(f0))
    throw jconsistency_error("OverSampledDFTAnalysisBank %s: frame %ld needs samples from input block %ld on, but a beamformer node "
                             "this bank is a channel of has released everything before block %ld; pull the bank through the "
                             "beamformer or on its own, not both\n", name().c_str(), f0, first_block_of_frame(f0), win_b0_);
  const unsigned K = M_ / 2 + 1;
  const long b0 = first_block_of_frame(f0), L = (nblk_ - b0) * (long)D_, Tn = f1 - f0;
  float* dp = static_cast<float*>(dPcm_.ensure(sizeof(float) * (L ? L : 1)));
  void* dX = dX_.ensure(sizeof(float) * 2 * K * Tn);
  float* Xh = static_cast<float*>(hX_.ensure(sizeof(float) * 2 * K * Tn));
  h2d_async(dp, window(b0), sizeof(float) * L);
  // the window starts at block b0: stream frame t is frame t - b0 of the window, and nothing it reads lies before the window
  check_abi(btk_fb_analysis(plan_, dp, L, L ? L : 1, 1, 1, dX, Tn, f0 - b0, Tn, nstream()));
  d2h_async(Xh, dX, sizeof(float) * 2 * K * Tn);
  nsync();
  frames_.assign((size_t)Tn * 2 * M_, 0.0);
  gsl_vector_complex tmp; tmp.size = M_; tmp.stride = 1;
  for (long t = 0; t < Tn; t++) {
    tmp.data = frames_.data() + (size_t)t * 2 * M_;
    serve_frame(Xh, Tn, M_, t, &tmp);
  }
  chunk_base_ = f0; chunk_len_ = Tn;
  release_before(f1);
  return true;
}

const gsl_vector_complex* OverSampledDFTAnalysisBank::next(int frame_no)
{
  if (frame_no == frame_no_) return vector_;
  const long idx = frame_no_ + 1;
  while (idx >= chunk_base_ + chunk_len_)
    if (!load_chunk_()) { is_end_ = true; throw jiterator_error("end of samples!"); }
  memcpy(vector_->data, frames_.data() + (size_t)(idx - chunk_base_) * 2 * M_, sizeof(double) * 2 * M_);
  increment_();
  return vector_;
}

void OverSampledDFTAnalysisBank::reset()
{
  samp_->reset();
  VectorComplexFeatureStream::reset();
  win_b0_ = 0; nblk_ = 0; eos_ = false; frames_.clear(); chunk_base_ = 0; chunk_len_ = 0;
  src16_ = NULL; src16_pos0_ = 0;
}

// ================================================================================ synthesis bank
OverSampledDFTSynthesisBank::OverSampledDFTSynthesisBank(VectorComplexFeatureStreamPtr& samp, gsl_vector* prototype, unsigned M,
                                                         unsigned m, unsigned r, unsigned delayCompensationType, int gainFactor,
                                                         const String& nm)
    : VectorFloatFeatureStream(M >> r, nm), samp_(samp), M_(M), m_(m), r_(r), D_(M >> r), gain_(gainFactor), plan_(NULL),
      nblocks_(0), blk_base_(0), prepared_(false), src_ended_(false), block_frames_(btk_default_block_frames()), hist_len_(0),
      frames_in_(0), cur_T_(0), bsrc_(NULL), src_version_(0), no_stream_feature_(false), npushed_(0), npushed_at_next_(0),
      dWin_(NULL), dBlk_(NULL), win_len_(0), win_pitch_(0), dev_hist_(false), carry_cols_(0), prev_nblocks_(0)
{
  init_(prototype, delayCompensationType);
}

OverSampledDFTSynthesisBank::OverSampledDFTSynthesisBank(gsl_vector* prototype, unsigned M, unsigned m, unsigned r,
                                                         unsigned delayCompensationType, int gainFactor, const String& nm)
    : VectorFloatFeatureStream(M >> r, nm), samp_(NULL), M_(M), m_(m), r_(r), D_(M >> r), gain_(gainFactor), plan_(NULL),
      nblocks_(0), blk_base_(0), prepared_(false), src_ended_(false), block_frames_(btk_default_block_frames()), hist_len_(0),
      frames_in_(0), cur_T_(0), bsrc_(NULL), src_version_(0), no_stream_feature_(true), npushed_(0), npushed_at_next_(0),
      dWin_(NULL), dBlk_(NULL), win_len_(0), win_pitch_(0), dev_hist_(false), carry_cols_(0), prev_nblocks_(0)
{
  init_(prototype, delayCompensationType);
}

void OverSampledDFTSynthesisBank::init_(gsl_vector* prototype, unsigned delayCompensationType)
{
  if (prototype->size != (size_t)M_ * m_)
    throw jconsistency_error("Prototype sizes do not match (%d vs. %d).", (int)prototype->size, (int)(M_ * m_));
  std::vector<double> g(prototype->size);
  for (size_t i = 0; i < g.size(); i++) g[i] = gsl_vector_get(prototype, i);
  check_abi(btk_fb_create(&plan_, (int)M_, (int)m_, (int)r_, (int)delayCompensationType, 1, g.data()));
}

OverSampledDFTSynthesisBank::~OverSampledDFTSynthesisBank() { btk_fb_destroy(plan_); dev_free(dWin_); dev_free(dBlk_); }

// Rows of a round's input window are win_pitch_ frames apart: the next even number, so that every launch of a stream can be an
// ALIGNED one (btk_fb_synthesis_aligned_form: such geometries have a second kernel form for aligned launches, <= 2 ulp from the
// other; a stream cut into rounds gives the same bits as one round only if all its launches take the same form)
static long even_pitch(long frames) { return frames + (frames & 1); }

// The output blocks of one round, from the round's input window on the device: dRound_ complex64 [K][win_pitch_] holds the
// stream frames w0 .. w0 + Lw - 1 (history first), blk_base_ / nblocks_ say which blocks.  The first keep_blocks blocks of the
// round (already handed over) keep their values.
void OverSampledDFTSynthesisBank::run_window_(long Lw, long w0, long b_first, long keep_blocks)
{
  ScopedNs timer(g_device_ns);
  std::vector<float> old;
  const float* prev = static_cast<const float*>(blocks_.get());
  if (keep_blocks > 0 && prev) old.assign(prev, prev + (size_t)std::min<long>(keep_blocks, prev_nblocks_) * D_);
  float* hb = static_cast<float*>(blocks_.ensure(sizeof(float) * (size_t)(nblocks_ ? nblocks_ : 1) * D_));
  if (nblocks_ > 0) {
    // the window starts at stream frame w0: block b of the stream is block b - w0 of the window.  An aligned launch starts at a
    // block whose newest frame has an even index in the window: where the round's first block does not, the launch begins one
    // block earlier (block -1 of a stream: zeros) and that block lands in D samples of slack in front of the others
    const long pd = btk_fb_processing_delay(plan_);
    const long bw = b_first - w0;
    const long lead = (btk_fb_synthesis_aligned_form(plan_) == 1 && ((bw + pd) & 1)) ? 1 : 0;
    const long slack = (D_ + 3) / 4 * 4;
    float* dO = static_cast<float*>(dOut_.ensure(sizeof(float) * ((size_t)(nblocks_ + 1) * D_ + slack)));
    float* first = dO + slack;                                   // (16-byte aligned like dO itself)
    check_abi(btk_fb_synthesis(plan_, dRound_.get(), Lw, win_pitch_, 1, first - lead * (long)D_, (nblocks_ + lead) * D_, bw - lead, nblocks_ + lead, nstream()));
    d2h_async(hb, first, sizeof(float) * (size_t)nblocks_ * D_);
    nsync();
    if (gain_ > 1) for (size_t i = 0; i < (size_t)nblocks_ * D_; i++) hb[i] *= (float)gain_;
  }
  if (!old.empty()) memcpy(hb, old.data(), sizeof(float) * std::min(old.size(), (size_t)nblocks_ * D_));
  prev_nblocks_ = nblocks_;
}

// One round: Yk complex64 [>= K rows][T] are the input frames base .. base + T - 1 of the stream; the output blocks whose newest
// input frame b + pd lies in this round are synthesised from them and from the frames kept of the rounds before (block b reads
// the frames b + pd - (m R - 1) .. b + pd, csrc/fb_kernels.hip; frames before the stream read as zero == the zeroed ring of
// modulated.cc:615-621).  Host form: the source has no device view (a drained stream, a Python object).
void OverSampledDFTSynthesisBank::synthesize_(const std::vector<float>& Yk, long T, long base, bool, long keep_blocks)
{
  const unsigned K = M_ / 2 + 1;
  const long pd = btk_fb_processing_delay(plan_);
  const long b_first = std::max<long>(0, base - pd), b_end = base + T - pd;
  prev_nblocks_ = nblocks_;
  blk_base_ = b_first;
  nblocks_ = b_end > b_first ? b_end - b_first : 0;
  const long Lw = hist_len_ + T, w0 = base - hist_len_, Lp = even_pitch(Lw);
  win_len_ = Lw; win_pitch_ = Lp;
  if (nblocks_ > 0) {
    float* win = static_cast<float*>(hRound_.ensure(sizeof(float) * 2 * K * Lp));
    for (unsigned k = 0; k < K; k++) {
      if (hist_len_) memcpy(&win[2 * (size_t)k * Lp], &hist_[2 * (size_t)k * hist_len_], sizeof(float) * 2 * hist_len_);
      memcpy(&win[2 * ((size_t)k * Lp + hist_len_)], &Yk[2 * (size_t)k * T], sizeof(float) * 2 * T);
      if (Lp > Lw) { win[2 * ((size_t)k * Lp + Lw)] = 0.f; win[2 * ((size_t)k * Lp + Lw) + 1] = 0.f; }
    }
    h2d_async(dRound_.ensure(sizeof(float) * 2 * K * Lp), win, sizeof(float) * 2 * K * Lp);
  }
  run_window_(Lw, w0, b_first, keep_blocks);
}

// The same round from a block that already lies on the device (BlockSource::device_block): the window is put together by
// device copies -- the tail of the round before as history, then the block -- and nothing but the PCM comes back to the host.
void OverSampledDFTSynthesisBank::synthesize_dev_(const void* dYk, long T, long T_stride, long base, long keep_blocks)
{
  const unsigned K = M_ / 2 + 1;
  const long pd = btk_fb_processing_delay(plan_);
  if (carry_cols_ >= 0) {
    // a new round: its window = the last carry_cols_ frames of the window before + the new block.  (The two buffers swap roles
    // every round: both are sized for the larger of the two windows, so that both reach their final size within two rounds.)
    const long keep = carry_cols_, Lw = keep + T, Lp = even_pitch(Lw);
    const size_t need = std::max(sizeof(float) * 2 * K * (Lp ? Lp : 2), sizeof(float) * 2 * K * (size_t)(win_pitch_ ? win_pitch_ : 2));
    char* nw = static_cast<char*>(dRoundNext_.ensure(need));
    if (keep) d2d_2d_async(nw, sizeof(float) * 2 * Lp, static_cast<const char*>(dRound_.get()) + sizeof(float) * 2 * (win_len_ - keep),
                           sizeof(float) * 2 * win_pitch_, sizeof(float) * 2 * keep, K);
    dRound_.swap(dRoundNext_);
    hist_len_ = keep; win_len_ = Lw; win_pitch_ = Lp; carry_cols_ = -1;
  }
  const long Lw = win_len_;
  if (Lw != hist_len_ + T) throw jconsistency_error("OverSampledDFTSynthesisBank: the source's block changed its length within a round (%ld -> %ld frames)\n", Lw - hist_len_, T);
  if (T) d2d_2d_async(static_cast<char*>(dRound_.get()) + sizeof(float) * 2 * hist_len_, sizeof(float) * 2 * win_pitch_, dYk, sizeof(float) * 2 * T_stride,
                      sizeof(float) * 2 * T, K);
  const long b_first = std::max<long>(0, base - pd), b_end = base + T - pd;
  prev_nblocks_ = nblocks_;
  blk_base_ = b_first;
  nblocks_ = b_end > b_first ? b_end - b_first : 0;
  run_window_(Lw, base - hist_len_, b_first, keep_blocks);
}

// the input frames of the current round: the block of an engine node upstream (it is not advanced through next()), or up to
// block_frames() frames pulled from a plain stream
void OverSampledDFTSynthesisBank::prepare_()
{
  const unsigned K = M_ / 2 + 1;
  if (!prepared_) {
    bsrc_ = dynamic_cast<BlockSource*>(samp_.operator->());
    if (bsrc_ && !bsrc_->has_block()) bsrc_ = NULL;
  }
  if (bsrc_) {
    // keep what was already served; next() tells the producer after every block how far a per-frame graph would have pulled, so
    // that a later weight change touches only the frames beyond
    long T = 0, Ts = 0;
    const long keep_blocks = prepared_ ? frame_no_ + 1 - blk_base_ : 0;
    const void* dY = (hist_len_ == 0 || dev_hist_) ? bsrc_->device_block(T, Ts) : NULL;
    if (dY) {
      dev_hist_ = true;
      src_version_ = bsrc_->block_version();
      frames_in_ = bsrc_->block_base();
      cur_T_ = T;
      synthesize_dev_(dY, T, Ts, frames_in_, keep_blocks);
    } else {
      const std::vector<float>& Yk = bsrc_->block(T);
      src_version_ = bsrc_->block_version();
      frames_in_ = bsrc_->block_base();
      cur_T_ = T;
      synthesize_(Yk, T, frames_in_, false, keep_blocks);
    }
  } else {
    std::vector<float> fr;                       // [T][K]
    long T = 0;
    while (!src_ended_ && (block_frames_ == 0 || T < block_frames_)) {
      const gsl_vector_complex* v;
      try { v = samp_->next(); } catch (jiterator_error&) { src_ended_ = true; break; }
      fr.resize((size_t)(T + 1) * K * 2);
      for (unsigned k = 0; k < K; k++) {
        fr[2 * ((size_t)T * K + k)] = (float)v->data[2 * k];
        fr[2 * ((size_t)T * K + k) + 1] = (float)v->data[2 * k + 1];
      }
      T++;
    }
    cur_.assign((size_t)2 * K * T, 0.f);         // [K][T]
    for (long t = 0; t < T; t++)
      for (unsigned k = 0; k < K; k++) {
        cur_[2 * ((size_t)k * T + t)] = fr[2 * ((size_t)t * K + k)];
        cur_[2 * ((size_t)k * T + t) + 1] = fr[2 * ((size_t)t * K + k) + 1];
      }
    cur_T_ = T;
    synthesize_(cur_, T, frames_in_, false, 0);
  }
  prepared_ = true;
}

// the current round is used up: its last frames become the history of the next one.  false: the input stream has ended
bool OverSampledDFTSynthesisBank::fetch_round_()
{
  const unsigned K = M_ / 2 + 1;
  const long pd = btk_fb_processing_delay(plan_);
  const long H = std::max<long>((long)m_ * (1L << r_) + (1L << r_), pd);
  long T = cur_T_;
  const long total = hist_len_ + T, keep = std::min(total, H);
  const long next_base = frames_in_ + T;
  if (bsrc_ && dev_hist_) {
    // every frame of the round has been handed over by now (advance_to), so the device window still is what the source holds
    if (!bsrc_->next_block()) return false;
    carry_cols_ = keep; frames_in_ = next_base;
    nblocks_ = 0;
    prepare_();
    return true;
  }
  const std::vector<float>* Yk = &cur_;
  if (bsrc_) Yk = &bsrc_->block(T);
  std::vector<float> nh((size_t)2 * K * keep);
  for (unsigned k = 0; k < K; k++)
    for (long i = 0; i < keep; i++) {
      const long j = total - keep + i;           // position in [history | this round]
      const float* src = j < hist_len_ ? &hist_[2 * ((size_t)k * hist_len_ + j)] : &(*Yk)[2 * ((size_t)k * T + (j - hist_len_))];
      nh[2 * ((size_t)k * keep + i)] = src[0]; nh[2 * ((size_t)k * keep + i) + 1] = src[1];
    }
  if (bsrc_) { if (!bsrc_->next_block()) return false; }
  else if (src_ended_) return false;
  hist_.swap(nh); hist_len_ = keep; frames_in_ = next_base;
  nblocks_ = 0;
  prepare_();
  return true;
}

// update_buf_ of the source-less form (modulated.cc:551-567): one more subband frame enters the ring
void OverSampledDFTSynthesisBank::input_source_vector(const gsl_vector_complex* block)
{
  if (!block || block->size != M_) throw jdimension_error("Input block length (%d) != fftLen (%d)\n", block ? (int)block->size : 0, M_);
  const unsigned K = M_ / 2 + 1, W = m_ * (1u << r_) + (1u << r_);
  if (ring_.size() >= (size_t)2 * K * W) ring_.erase(ring_.begin(), ring_.begin() + 2 * K);
  for (unsigned k = 0; k < K; k++) {
    ring_.push_back((float)block->data[2 * k * block->stride]);
    ring_.push_back((float)block->data[2 * k * block->stride + 1]);
  }
  npushed_++;
}

// next() of the source-less form: block j of the stream "pd zero frames, then the pushed frames" -- the ring of a per-frame
// graph after j + 1 pushes (update_buffer_ is a no-op and nothing is primed, modulated.cc:536-549, 574-578) -- computed from
// the last m R + R virtual frames, which is everything the polyphase sums of the R overlapping blocks reach
const gsl_vector_float* OverSampledDFTSynthesisBank::next_pushed_()
{
  if (npushed_ - npushed_at_next_ != 1)
    throw jconsistency_error("OverSampledDFTSynthesisBank without a source: exactly one input_source_vector() per next() "
                             "(%ld since the last one)\n", npushed_ - npushed_at_next_);
  const unsigned K = M_ / 2 + 1, R = 1u << r_, W = m_ * R + R;
  const long pd = btk_fb_processing_delay(plan_), Tv = pd + npushed_, Lw = Tv < (long)W ? Tv : (long)W;
  const long nring = (long)(ring_.size() / (2 * K));
  std::vector<float> win((size_t)2 * K * Lw, 0.f);                 // [K][Lw]
  for (long i = 0; i < Lw; i++) {
    const long p = Tv - Lw + i - pd;                               // pushed-frame index of virtual frame Tv - Lw + i
    const long q = p - (npushed_ - nring);                         // its slot in the ring
    if (p < 0 || q < 0) continue;
    for (unsigned k = 0; k < K; k++) {
      win[2 * ((size_t)k * Lw + i)] = ring_[2 * ((size_t)q * K + k)];
      win[2 * ((size_t)k * Lw + i) + 1] = ring_[2 * ((size_t)q * K + k) + 1];
    }
  }
  if (!dWin_) { dWin_ = dev_alloc(sizeof(float) * 2 * K * W); dBlk_ = dev_alloc(sizeof(float) * D_); }
  h2d(dWin_, win.data(), sizeof(float) * win.size());
  check_abi(btk_fb_synthesis(plan_, dWin_, Lw, Lw, 1, (float*)dBlk_, D_, Lw - 1 - pd, 1, nstream()));
  d2h(vector_->data, dBlk_, sizeof(float) * D_);
  if (gain_ > 1) for (unsigned i = 0; i < D_; i++) vector_->data[i] *= (float)gain_;
  npushed_at_next_ = npushed_;
  increment_();
  return vector_;
}

const gsl_vector_float* OverSampledDFTSynthesisBank::next(int frame_no)
{
  if (no_stream_feature_) return next_pushed_();
  if (samp_.is_null()) throw jconsistency_error("OverSampledDFTSynthesisBank: no source stream (no_stream_feature(false) on a source-less bank)\n");
  if (!prepared_ || (bsrc_ && bsrc_->block_version() != src_version_)) prepare_();
  const long idx = frame_no_ + 1;
  while (idx >= blk_base_ + nblocks_)
    if (!fetch_round_()) { is_end_ = true; throw jiterator_error("end of samples!"); }
  memcpy(vector_->data, static_cast<const float*>(blocks_.get()) + (size_t)(idx - blk_base_) * D_, sizeof(float) * D_);
  increment_();
  // block idx of a per-frame graph has pulled the frames 0 .. pd + idx (the priming of modulated.cc:574-578 included)
  if (bsrc_) bsrc_->advance_to((long)btk_fb_processing_delay(plan_) + idx);
  return vector_;
}

long OverSampledDFTSynthesisBank::next_blocks(long max_blocks, const float** blocks)
{
  if (no_stream_feature_) throw jconsistency_error("OverSampledDFTSynthesisBank::next_blocks: a source-less bank is pulled block by block with next()\n");
  if (samp_.is_null()) throw jconsistency_error("OverSampledDFTSynthesisBank: no source stream (no_stream_feature(false) on a source-less bank)\n");
  if (!blocks) throw j_error("OverSampledDFTSynthesisBank::next_blocks: null argument\n");
  if (!prepared_ || (bsrc_ && bsrc_->block_version() != src_version_)) prepare_();
  const long idx = frame_no_ + 1;
  while (idx >= blk_base_ + nblocks_)
    if (!fetch_round_()) { is_end_ = true; *blocks = NULL; return 0; }
  long n = blk_base_ + nblocks_ - idx;
  if (max_blocks > 0 && n > max_blocks) n = max_blocks;
  const float* p = static_cast<const float*>(blocks_.get()) + (size_t)(idx - blk_base_) * D_;
  memcpy(vector_->data, p + (size_t)(n - 1) * D_, sizeof(float) * D_);
  frame_no_ += (int)n;
  if (bsrc_) bsrc_->advance_to((long)btk_fb_processing_delay(plan_) + idx + n - 1);
  *blocks = p;
  return n;
}

void OverSampledDFTSynthesisBank::reset()
{
  if (!no_stream_feature_ && !samp_.is_null()) samp_->reset();
  VectorFloatFeatureStream::reset();
  prepared_ = false; nblocks_ = 0; prev_nblocks_ = 0; blk_base_ = 0; bsrc_ = NULL; src_version_ = 0; src_ended_ = false;
  hist_.clear(); hist_len_ = 0; frames_in_ = 0; cur_.clear(); cur_T_ = 0; win_len_ = 0; win_pitch_ = 0; dev_hist_ = false; carry_cols_ = 0;
  ring_.clear(); npushed_ = 0; npushed_at_next_ = 0;      // buffer_.zero() (modulated.cc:614-622)
}

// ================================================================================ SnapShotArray
SnapShotArray::SnapShotArray(unsigned fftLn, unsigned nChn) : fftLen_(fftLn), nChan_(nChn)
{
  samples_ = new gsl_vector_complex*[nChan_];
  for (unsigned i = 0; i < nChan_; i++) samples_[i] = gsl_vector_complex_calloc(fftLen_);
  snapshots_ = new gsl_vector_complex*[fftLen_];
  for (unsigned i = 0; i < fftLen_; i++) snapshots_[i] = gsl_vector_complex_calloc(nChan_);
}
SnapShotArray::~SnapShotArray()
{
  for (unsigned i = 0; i < nChan_; i++) gsl_vector_complex_free(samples_[i]);
  delete[] samples_;
  for (unsigned i = 0; i < fftLen_; i++) gsl_vector_complex_free(snapshots_[i]);
  delete[] snapshots_;
}
void SnapShotArray::zero()
{
  for (unsigned i = 0; i < nChan_; i++) gsl_vector_complex_set_zero(samples_[i]);
  for (unsigned i = 0; i < fftLen_; i++) gsl_vector_complex_set_zero(snapshots_[i]);
}
void SnapShotArray::set_samples(const gsl_vector_complex* samp, unsigned chanX)
{
  memcpy(samples_[chanX]->data, samp->data, sizeof(double) * 2 * fftLen_);
}
void SnapShotArray::set_snapshots(const gsl_vector_complex* snapshots, unsigned fbinX)
{
  const unsigned fftLen2 = fftLen_ / 2;
  if (fbinX > fftLen2) throw jindex_error("frequency bin %d is out of range (0..%d)\n", (int)fbinX, (int)fftLen2);
  if (snapshots->size != nChan_) throw jdimension_error("snapshot of %d channels, %d expected\n", (int)snapshots->size, (int)nChan_);
  for (unsigned c = 0; c < nChan_; c++) gsl_vector_complex_set(snapshots_[fbinX], c, gsl_vector_complex_get(snapshots, c));
  if (fbinX == 0 || fbinX == fftLen2) return;
  for (unsigned c = 0; c < nChan_; c++)                      // the reference's own target index: fftLen/2 - fbinX (beamformer.cc:88-91)
    gsl_vector_complex_set(snapshots_[fftLen2 - fbinX], c, gsl_complex_conjugate(gsl_vector_complex_get(snapshots, c)));
}
void SnapShotArray::update()
{
  for (unsigned k = 0; k < fftLen_; k++)
    for (unsigned c = 0; c < nChan_; c++) {
      snapshots_[k]->data[2 * c] = samples_[c]->data[2 * k];
      snapshots_[k]->data[2 * c + 1] = samples_[c]->data[2 * k + 1];
    }
}

// ================================================================================ SpectralMatrixArray
SpectralMatrixArray::SpectralMatrixArray(unsigned fftLn, unsigned nChn, float forgetFact)
    : SnapShotArray(fftLn, nChn), mu_((double)forgetFact)
{
  matrices_ = new gsl_matrix_complex*[fftLen_];
  for (unsigned i = 0; i < fftLen_; i++) {
    matrices_[i] = gsl_matrix_complex_alloc(nChan_, nChan_);
    memset(matrices_[i]->data, 0, sizeof(double) * 2 * nChan_ * nChan_);
  }
}
SpectralMatrixArray::~SpectralMatrixArray()
{
  for (unsigned i = 0; i < fftLen_; i++) gsl_matrix_complex_free(matrices_[i]);
  delete[] matrices_;
}
void SpectralMatrixArray::zero()
{
  SnapShotArray::zero();
  for (unsigned i = 0; i < fftLen_; i++) memset(matrices_[i]->data, 0, sizeof(double) * 2 * nChan_ * nChan_);
}
void SpectralMatrixArray::update()
{
  SnapShotArray::update();
  const double alpha = 1.0 - mu_;
  for (unsigned k = 0; k < fftLen_; k++) {
    const double* x = snapshots_[k]->data;
    double* R = matrices_[k]->data;
    for (unsigned i = 0; i < nChan_; i++)
      for (unsigned j = 0; j < nChan_; j++) {
        const double pr = x[2 * i] * x[2 * j] - x[2 * i + 1] * x[2 * j + 1];          // x_i x_j, no conjugate (beamformer.cc:131-139)
        const double pi = x[2 * i] * x[2 * j + 1] + x[2 * i + 1] * x[2 * j];
        R[2 * (i * nChan_ + j)] = mu_ * R[2 * (i * nChan_ + j)] + alpha * pr;
        R[2 * (i * nChan_ + j) + 1] = mu_ * R[2 * (i * nChan_ + j) + 1] + alpha * pi;
      }
  }
}

// ================================================================================ BeamformerWeights
namespace {
gsl_vector_complex* alias_vector(std::complex<double>* p, size_t n)
{
  gsl_vector_complex* v = (gsl_vector_complex*)malloc(sizeof(gsl_vector_complex));
  v->size = n; v->stride = 1; v->data = reinterpret_cast<double*>(p); v->block = NULL; v->owner = 0;
  return v;
}
gsl_matrix_complex* alias_matrix(std::complex<double>* p, size_t n1, size_t n2)
{
  gsl_matrix_complex* m = (gsl_matrix_complex*)malloc(sizeof(gsl_matrix_complex));
  m->size1 = n1; m->size2 = n2; m->tda = n2; m->data = reinterpret_cast<double*>(p); m->block = NULL; m->owner = 0;
  return m;
}
}  // namespace

BeamformerWeights::BeamformerWeights(unsigned fftLen, unsigned chanN, bool halfBandShift, unsigned NC)
    : wq_v((size_t)fftLen * chanN), wl_v((size_t)fftLen * chanN), ta_v((size_t)fftLen * chanN),
      wa_v(chanN > NC ? (size_t)fftLen * (chanN - NC) : 0), B_v(chanN > NC ? (size_t)fftLen * chanN * (chanN - NC) : 0),
      fftLen_(fftLen), chanN_(chanN), NC_(NC), halfBandShift_(halfBandShift), wa_views_(NULL), B_views_(NULL)
{
  // the vectors never change size: the views below stay valid for the life of the object
  const unsigned bs = chanN > NC ? chanN - NC : 0;
  wq_views_ = (gsl_vector_complex**)malloc(sizeof(gsl_vector_complex*) * fftLen);
  wl_views_ = (gsl_vector_complex**)malloc(sizeof(gsl_vector_complex*) * fftLen);
  ta_views_ = (gsl_vector_complex**)malloc(sizeof(gsl_vector_complex*) * fftLen);
  CSDs_ = (gsl_vector_complex**)malloc(sizeof(gsl_vector_complex*) * fftLen);
  if (bs) {
    wa_views_ = (gsl_vector_complex**)malloc(sizeof(gsl_vector_complex*) * fftLen);
    B_views_ = (gsl_matrix_complex**)malloc(sizeof(gsl_matrix_complex*) * fftLen);
  }
  for (unsigned k = 0; k < fftLen; k++) {
    wq_views_[k] = alias_vector(&wq_v[(size_t)k * chanN], chanN);
    wl_views_[k] = alias_vector(&wl_v[(size_t)k * chanN], chanN);
    ta_views_[k] = alias_vector(&ta_v[(size_t)k * chanN], chanN);
    CSDs_[k] = gsl_vector_complex_calloc((size_t)chanN * chanN);
    if (bs) {
      wa_views_[k] = alias_vector(&wa_v[(size_t)k * bs], bs);
      B_views_[k] = alias_matrix(&B_v[(size_t)k * chanN * bs], chanN, bs);
    }
  }
  wp1_ = gsl_vector_complex_calloc(fftLen);
}

BeamformerWeights::~BeamformerWeights()
{
  for (unsigned k = 0; k < fftLen_; k++) {
    free(wq_views_[k]); free(wl_views_[k]); free(ta_views_[k]); gsl_vector_complex_free(CSDs_[k]);
    if (wa_views_) free(wa_views_[k]);
    if (B_views_) free(B_views_[k]);
  }
  free(wq_views_); free(wl_views_); free(ta_views_); free(CSDs_); free(wa_views_); free(B_views_);
  gsl_vector_complex_free(wp1_);
}

void BeamformerWeights::calcMainlobe2(float samplerate, const gsl_vector* delaysT, const gsl_vector* delaysI, bool isGSC)
{
  if (delaysI->size != chanN_)
    throw jdimension_error("The number of delays for an interference signal does not match number of channels (%d vs. %d).\n",
                           (int)delaysI->size, chanN_);
  if (chanN_ < 2) throw jdimension_error("The number of channels must be > 2 but it is %d\n", chanN_);
  gsl_matrix* delaysIs = gsl_matrix_alloc(1, chanN_);
  for (unsigned c = 0; c < chanN_; c++) gsl_matrix_set(delaysIs, 0, c, gsl_vector_get(delaysI, c));
  try { calcMainlobeN(samplerate, delaysT, delaysIs, 2, isGSC); } catch (...) { gsl_matrix_free(delaysIs); throw; }
  gsl_matrix_free(delaysIs);
}

void BeamformerWeights::calcMainlobeN(float samplerate, const gsl_vector* delaysT, const gsl_matrix* delaysIs, unsigned NC, bool isGSC)
{
  if (NC < 2 || NC > chanN_)
    throw jdimension_error("1 < the number of constraints %d <= the number of sensors %d.\n", NC, chanN_);
  if (delaysT->size != chanN_)
    throw jdimension_error("The number of delays does not match number of channels (%d vs. %d).\n", (int)delaysT->size, chanN_);
  if (NC != NC_) throw jdimension_error("The weight object was allocated for %d constraints, not %d\n", NC_, NC);
  if (halfBandShift_) throw j_error("halfBandShift==true with more than one constraint is not supported by this engine\n");
  std::vector<double> dt(chanN_), di((size_t)(NC - 1) * chanN_);
  for (unsigned c = 0; c < chanN_; c++) dt[c] = gsl_vector_get(delaysT, c);
  for (unsigned n = 0; n + 1 < NC; n++)
    for (unsigned c = 0; c < chanN_; c++) di[(size_t)n * chanN_ + c] = gsl_matrix_get(delaysIs, n, c);
  check_abi(btk_weights_mainlobe_n((int)fftLen_, (int)chanN_, samplerate, dt.data(), di.data(), (int)NC, reinterpret_cast<double*>(wq_v.data())));
  // ta_ keeps the plain look-direction manifold calcMainlobe copied before the constraints were applied (beamformer.cc:562)
  check_abi(btk_weights_mainlobe((int)fftLen_, (int)chanN_, samplerate, dt.data(), reinterpret_cast<double*>(ta_v.data())));
  if (isGSC)
    for (unsigned k = 0; k < fftLen_; k++) calcBlockingMatrix(k);
}

void BeamformerWeights::calcSidelobeCancellerU_f(unsigned fbinX, const gsl_vector_complex* w)
{
  const unsigned bs = chanN_ - NC_;
  if (w->size != bs) throw jdimension_error("the size of an active weight vector must be %d but it is %d\n", bs, (int)w->size);
  std::vector<cd> v(bs);
  for (unsigned i = 0; i < bs; i++) { const gsl_complex z = gsl_vector_complex_get(w, i); v[i] = cd(GSL_REAL(z), GSL_IMAG(z)); }
  calcSidelobeCancellerU_f(fbinX, v.data());
}

void BeamformerWeights::setSidelobeCanceller_f(unsigned fbinX, gsl_vector_complex* wl_in)
{
  memcpy(static_cast<void*>(&wl_v[(size_t)fbinX * chanN_]), wl_in->data, sizeof(double) * 2 * chanN_);
}

void BeamformerWeights::setQuiescentVector(unsigned fbinX, gsl_vector_complex* wq_in, bool isGSC)
{
  for (unsigned c = 0; c < chanN_; c++) { const gsl_complex z = gsl_vector_complex_get(wq_in, c); wq_v[(size_t)fbinX * chanN_ + c] = cd(GSL_REAL(z), GSL_IMAG(z)); }
  if (isGSC) calcBlockingMatrix(fbinX);
}

void BeamformerWeights::setQuiescentVectorAll(gsl_complex z, bool isGSC)
{
  for (unsigned k = 0; k < fftLen_; k++) {
    for (unsigned c = 0; c < chanN_; c++) wq_v[(size_t)k * chanN_ + c] = cd(GSL_REAL(z), GSL_IMAG(z));
    if (isGSC) calcBlockingMatrix(k);
  }
}

// reference beamformer.cc:775-828.  The inverse DFT (gsl_fft_complex_radix2_inverse: e^{+j 2 pi k n / M} / M) of the
// Hermitian-extended sequence val[k] = e^{j pi (k+1)} conj(wq_v[k] - wl_v[k]) is taken directly: M^2 / 2 operations per
// channel, one-off.
bool BeamformerWeights::write_fir_coeff(const String& fn, unsigned winType)
{
  const unsigned M = fftLen_, M2 = fftLen_ / 2;
  FILE* fp = fopen(fn.c_str(), "w");
  if (!fp) { printf("could not open %s\n", fn.c_str()); return false; }
  fprintf(fp, "%d %d\n", chanN_, M);
  std::vector<double> window(M);
  for (unsigned i = 0; i < M; i++)
    window[i] = winType == 0 ? 1.0 : winType == 2 ? 0.5 * (1 - cos((2.0 * M_PI * i) / (double)(M - 1)))
                                                  : 0.54 - 0.46 * cos(2. * M_PI / (double)(M - 1) * i);
  std::vector<cd> val(M);
  for (unsigned c = 0; c < chanN_; c++) {
    std::fill(val.begin(), val.end(), cd(0, 0));
    for (unsigned k = 0; k <= M2; k++) {
      const cd wH = std::conj(wq_v[(size_t)k * chanN_ + c] - wl_v[(size_t)k * chanN_ + c]);
      const cd v = std::polar(1.0, M_PI * (k + 1)) * wH;                       // shift fftLen/2
      val[k] = v;
      if (k > 0 && k < M2) val[M - k] = std::conj(v);
    }
    for (unsigned n = 0; n < M; n++) {
      double acc = 0.0;                                                          // real part of the inverse DFT
      for (unsigned k = 0; k < M; k++) {
        const double ph = 2.0 * M_PI * (double)((unsigned long)k * n % M) / M;
        acc += val[k].real() * cos(ph) - val[k].imag() * sin(ph);
      }
      fprintf(fp, "%e ", window[n] * acc / M);
    }
    fprintf(fp, "\n");
  }
  fclose(fp);
  return true;
}

void BeamformerWeights::calcMainlobe(float samplerate, const gsl_vector* delays, bool isGSC)
{
  if (delays->size != chanN_)
    throw jdimension_error("Number of delays does not match number of channels (%d vs. %d).\n", (int)delays->size, chanN_);
  if (isGSC && chanN_ <= 1) throw jdimension_error("The number of channels must be > 1 but it is %d\n", chanN_);
  std::vector<double> d(chanN_);
  for (unsigned c = 0; c < chanN_; c++) d[c] = gsl_vector_get(delays, c);
 // halfBandShift: bin k sits at (k + 0.5) fs / M and every one of the M bins has its own vector (beamformer.cc:515-527)
  if (halfBandShift_) check_abi(btk_weights_mainlobe_halfband((int)fftLen_, (int)chanN_, samplerate, d.data(), reinterpret_cast<double*>(wq_v.data())));
  else check_abi(btk_weights_mainlobe((int)fftLen_, (int)chanN_, samplerate, d.data(), reinterpret_cast<double*>(wq_v.data())));
  ta_v = wq_v;                                                   // setTimeAlignment
  if (isGSC)
    for (unsigned k = 0; k < fftLen_; k++) calcBlockingMatrix(k);
}

void BeamformerWeights::calcBlockingMatrix(unsigned fbinX)
{
  const unsigned bs = chanN_ - NC_;
  check_abi(btk_weights_blocking_matrix(reinterpret_cast<const double*>(&wq_v[(size_t)fbinX * chanN_]), (int)chanN_, (int)NC_,
                                        reinterpret_cast<double*>(&B_v[(size_t)fbinX * chanN_ * bs])));
}

void BeamformerWeights::calcSidelobeCancellerU_f(unsigned fbinX, const cd* w)
{
  if (fbinX >= fftLen_) throw jdimension_error("Must be a frequency bin %d < the length of FFT %d\n", fbinX, fftLen_);
  const unsigned bs = chanN_ - NC_;
  for (unsigned i = 0; i < bs; i++) wa_v[(size_t)fbinX * bs + i] = w[i];
  check_abi(btk_weights_sidelobe(reinterpret_cast<const double*>(&B_v[(size_t)fbinX * chanN_ * bs]),
                                 reinterpret_cast<const double*>(&wa_v[(size_t)fbinX * bs]), (int)chanN_, (int)NC_,
                                 reinterpret_cast<double*>(&wl_v[(size_t)fbinX * chanN_])));
}

void BeamformerWeights::calcSidelobeCancellerP_f(unsigned fbinX, const gsl_vector* packedWeight)
{
  const unsigned bs = chanN_ - NC_;
  if (packedWeight->size != 2 * bs)
    throw jdimension_error("the size of an active weight vector must be %d but it is %d\n", 2 * bs, (int)packedWeight->size);
  std::vector<cd> w(bs);
  for (unsigned i = 0; i < bs; i++) w[i] = cd(gsl_vector_get(packedWeight, 2 * i), gsl_vector_get(packedWeight, 2 * i + 1));
  calcSidelobeCancellerU_f(fbinX, w.data());
}

// ================================================================================ SubbandBeamformer
SubbandBeamformer::SubbandBeamformer(unsigned fftLen, bool halfBandShift, const String& nm)
    : VectorComplexFeatureStream(fftLen, nm), chunk_base_(0), block_frames_(btk_default_block_frames()), chunk_loaded_(false),
      channels_ended_(false), quantum_(1), halfBandShift_(halfBandShift), dXfull_(NULL), snapshot_array_(NULL), fftLen_(fftLen),
      fftLen2_(fftLen / 2), dX_(NULL), T_(0), banks_only_(false), pcm_i16_(false), pcm_f32_valid_(false), pcm_pitch_(0),
      pre_valid_(false), pre_pitch_(0),
      pcm_L_(0), pcm_t0_(0), pcm_valid_(false), snap_valid_(false),
      snapshots_wanted_(false) {}
SubbandBeamformer::~SubbandBeamformer()
{
  if (pre_valid_) (void)hipStreamSynchronize(cstream());
}
// the block state starts over; the device buffers stay (they only ever grow: common/devmem.h)
void SubbandBeamformer::free_device_()
{
  dX_ = NULL; dXfull_ = NULL; T_ = 0; Xhost_.clear();
  chunk_base_ = 0; chunk_loaded_ = false; channels_ended_ = false; pcm_valid_ = false; snap_valid_ = false; pcm_L_ = 0; pcm_t0_ = 0;
  pcm_i16_ = false; pcm_f32_valid_ = false; pcm_pitch_ = 0;
  if (pre_valid_) { (void)hipStreamSynchronize(cstream()); pre_valid_ = false; }     // copies of a block nobody will ask for
}
void SubbandBeamformer::set_channel(VectorComplexFeatureStreamPtr& chan) { channelList_.push_back(chan); }
void SubbandBeamformer::clear_channel() { channelList_.clear(); banks_.clear(); banks_only_ = false; snapshot_array_ = NULL; free_device_(); }

void SubbandBeamformer::reset()
{
  for (ChannelList_::iterator it = channelList_.begin(); it != channelList_.end(); ++it) (*it)->reset();
  if (!snapshot_array_.is_null()) snapshot_array_->zero();
  VectorComplexFeatureStream::reset();
  is_end_ = false;
  free_device_();
}

void SubbandBeamformer::set_block_quantum(long q)
{
  const char* e = getenv("BTK_BLOCK_QUANTUM");
  if (e && *e) q = atol(e);
  quantum_ = q < 1 ? 1 : q;
}

// the snapshots of the current block, launched on the node stream.  Over analysis banks they are computed from the resident
// PCM windows the first time somebody asks; from then on every block brings them along (snapshots_wanted_)
void* SubbandBeamformer::snapshots_()
{
  ensure_chunk_();
  snapshots_wanted_ = true;
  if (!snap_valid_) {
    const unsigned N = chanN(), K = fftLen2_ + 1;
    dX_ = dXBuf_.ensure(sizeof(float) * 2 * K * N * (T_ ? T_ : 1));
    if (T_ > 0) {
      if (!pcm_valid_) throw jconsistency_error("SubbandBeamformer %s: the samples of the current block are gone\n", name().c_str());
      // the windows start at input block b0: stream frame t is frame t - b0 of the window (csrc/fb_kernels.hip: frame t ends at
      // sample (t + laN + 1) D - 1), and no frame of this block reads a sample before it
      // (a 16-bit stream: the bank reads the int16 rows where the geometry has that form -- btk_fb_analysis_i16, the same bits --
      //  and the widening pass with its 6 D bytes per frame and channel falls away; BTK_NODE_I16_STAGED=0: always widen first)
      static const bool direct_off = getenv("BTK_NODE_I16_STAGED") && atoi(getenv("BTK_NODE_I16_STAGED")) == 0;
      if (pcm_i16_ && !direct_off && btk_fb_analysis_i16_direct(banks_[0]->plan()) == 1)
        check_abi(btk_fb_analysis_i16(banks_[0]->plan(), static_cast<const short*>(dPcm16Buf_.get()), pcm_L_, pcm_pitch_ ? pcm_pitch_ : 1, 1, (int)N,
                                      dX_, T_, pcm_t0_, T_, nstream()));
      else
        check_abi(btk_fb_analysis(banks_[0]->plan(), pcm_f32_(), pcm_L_, pcm_pitch_ ? pcm_pitch_ : 1, 1, (int)N, dX_, T_, pcm_t0_, T_, nstream()));
    }
    snap_valid_ = true;
  }
  return dX_;
}

const float* SubbandBeamformer::pcm_f32_()
{
  if (!pcm_i16_) return static_cast<const float*>(dPcmBuf_.get());
  const size_t n = (size_t)chanN() * (size_t)(pcm_pitch_ ? pcm_pitch_ : 1);
  float* d = static_cast<float*>(dPcmBuf_.ensure(sizeof(float) * n));
  if (!pcm_f32_valid_) {
    check_abi(btk_pcm_i16_to_f32(static_cast<const short*>(dPcm16Buf_.get()), d, (long)n, nstream()));
    pcm_f32_valid_ = true;
  }
  return d;
}


bool SubbandBeamformer::i16_stream_possible()
{
  if (chunk_loaded_ || !node_i16_enabled() || !banks_only()) return false;
  for (size_t c = 0; c < banks_.size(); c++) if (!banks_[c]->i16_source_ok()) return false;
  return true;
}

void SubbandBeamformer::begin_i16_stream()
{
  for (size_t c = 0; c < banks_.size(); c++) banks_[c]->begin_i16();
  pcm_i16_ = true;
}

void* SubbandBeamformer::device_snapshots()
{
  void* p = snapshots_();
  nsync();                                                       // the caller may read them from any stream
  return p;
}

bool SubbandBeamformer::next_chunk()
{
  if (!chunk_loaded_) return load_chunk_();
  if (channels_ended_) return false;
  return load_chunk_();
}

// every channel an analysis bank with the same plan -> the channels advance as one batch of sample windows
bool SubbandBeamformer::banks_only()
{
  banks_.clear();
  banks_only_ = !channelList_.empty();
  for (ChannelList_::iterator it = channelList_.begin(); it != channelList_.end(); ++it) {
    OverSampledDFTAnalysisBank* b = dynamic_cast<OverSampledDFTAnalysisBank*>(it->operator->());
    if (!b || (!banks_.empty() && (b->fftlen() != banks_[0]->fftlen() || b->m() != banks_[0]->m() || b->r() != banks_[0]->r() ||
                                   b->delay_compensation_type() != banks_[0]->delay_compensation_type()))) { banks_only_ = false; break; }
    banks_.push_back(b);
  }
  if (!banks_only_) banks_.clear();
  return banks_only_;
}

// every bank pulls one round of input blocks (its block_frames()); the channels advance in lock step, the shortest one ends
// the stream (its zero-padded tail frames included, like a per-frame graph whose first exhausted channel ends it)
void SubbandBeamformer::pull_bank_(size_t c)
{
  const long f0 = chunk_loaded_ ? chunk_base_ + T_ : 0;
  const long want = (f0 / quantum_ + 1) * quantum_;            // a block ends on a multiple of the quantum (or with the stream)
  while (!banks_[c]->at_end() && banks_[c]->frames_ready() < want) banks_[c]->pull_more();
}

// the block the first nbanks banks allow (all of them: the block itself; the first one alone: what the block will be unless a
// later channel turns out shorter)
void SubbandBeamformer::plan_from_pulled_(BlockPlan& p, size_t nbanks) const
{
  const long f0 = chunk_loaded_ ? chunk_base_ + T_ : 0;
  const long laN = btk_fb_lookahead(banks_[0]->plan()), pd = btk_fb_processing_delay(banks_[0]->plan());
  long nblk = -1; bool ended = false;
  for (size_t c = 0; c < nbanks; c++) { const long n = banks_[c]->blocks_pulled(); if (nblk < 0 || n < nblk) nblk = n; }
  for (size_t c = 0; c < nbanks; c++) if (banks_[c]->at_end() && banks_[c]->blocks_pulled() == nblk) ended = true;
  long f1 = ended ? (nblk < laN ? 0 : nblk - laN + pd) : (nblk > laN ? nblk - laN : 0);
  if (!ended) f1 = f1 / quantum_ * quantum_;
  long b0 = banks_[0]->first_block_of_frame(f0);
  for (size_t c = 0; c < nbanks; c++) b0 = std::max(b0, banks_[c]->window_first_block());
  b0 = std::min(b0, nblk);
  p.f0 = f0; p.T = f1 > f0 ? f1 - f0 : 0; p.b0 = b0; p.L = (nblk - b0) * (long)banks_[0]->shiftlen(); p.ended = ended;
}

// banks [c0, c1) pull their round: side by side when every one of them may (SampleFeature sources, bounded rounds)
void SubbandBeamformer::pull_banks_(size_t c0, size_t c1)
{
  ScopedNs timer(g_pull_ns);
  bool par = PullPool::get().threads() > 1 && c1 - c0 > 1;
  for (size_t c = c0; c < c1 && par; c++) par = banks_[c]->parallel_pull_ok();
  if (!par) { for (size_t c = c0; c < c1; c++) pull_bank_(c); return; }
  for (size_t c = c0; c < c1; c++) banks_[c]->reserve_round();                 // (allocation, if any, stays on this thread)
  PullPool::get().parallel_for((int)(c1 - c0), [&](int i) { pull_bank_(c0 + (size_t)i); });
}

void SubbandBeamformer::plan_bank_block(BlockPlan& p)
{
  pull_banks_(0, banks_.size());
  plan_from_pulled_(p, banks_.size());
}

void SubbandBeamformer::commit_bank_block(const BlockPlan& p)
{
  Xhost_.clear();
  chunk_loaded_ = true; chunk_base_ = p.f0; T_ = p.T; channels_ended_ = p.ended;
  pcm_valid_ = false; snap_valid_ = false; dX_ = NULL; dXfull_ = NULL;
  for (size_t c = 0; c < banks_.size(); c++) banks_[c]->release_before(p.f0 + p.T);
}

// 16-bit streams: the block AFTER the one just committed is planned now -- the sources only move on, nothing is copied on the host
// -- and its samples start their way up on the thread's second stream into the other device buffer, under the kernels, the
// download and the serving of the current block: a block then costs max(PCIe, everything else) instead of their sum.  `cur` is the
// block just committed.  The buffer being filled was the current one of the block BEFORE it: the copies wait for what the node
// stream has queued so far.  (The sources run one block further ahead of the frames served than without -- they already ran a
// block ahead.)  BTK_NODE_PREFETCH=0 switches it off.
void SubbandBeamformer::prefetch_next_(const BlockPlan& cur)
{
  static const bool off = getenv("BTK_NODE_PREFETCH") && atoi(getenv("BTK_NODE_PREFETCH")) == 0;
  if (off || !pcm_i16_ || cur.ended || cur.T <= 0) return;
  const unsigned N = chanN();
  BlockPlan pn;
  pull_banks_(0, N);
  plan_from_pulled_(pn, N);
  pre_plan_ = pn;
  pre_pitch_ = (pn.L + 7) / 8 * 8;
  if (pn.T > 0) {
    // the buffer being filled was the current one of the block BEFORE `cur`: that block has been served to its last frame, i.e.
    // the host has waited for everything the node stream did with it -- no stream-side dependency is needed (and none is made:
    // copies that wait on an event of another stream ran at 37 instead of 53 GB/s)
    char* dp = static_cast<char*>(dPcm16Next_.ensure(sizeof(short) * N * (pre_pitch_ ? pre_pitch_ : 8)));
    nsync();                                                       // (cheap: the stream is idle at this point in a pulled graph)
    btk_row_t* tab = static_cast<btk_row_t*>(hRows_.ensure(sizeof(btk_row_t) * N));   // (no upload of this node is in flight here)
    for (unsigned c = 0; c < N; c++) tab[c].src = banks_[c]->window16(pn.b0);
    upload_rows(tab, N, sizeof(short) * pn.L, dp, sizeof(short) * pre_pitch_, cstream());
  }
  pre_valid_ = true;
}

// The block after the current one (the first one when none is loaded): frames chunk_base_ .. chunk_base_ + T_ - 1.
// Returns false when the channels have no further frame (the block is then empty).
bool SubbandBeamformer::load_chunk_()
{
  const unsigned N = chanN(), K = fftLen2_ + 1;
  if (N == 0) throw j_error("set channels first\n");
  if (banks_only()) {
    // The banks keep their windows in pinned memory: N asynchronous copies, one per channel, and the block's samples are resident
    // -- for the fused kernel (SubbandDS::compute_output_) or for snapshots_().  The copy of a channel starts as soon as that
    // channel has pulled its input, under the pulling of the next ones: the first bank says what the block will be, and only
    // if a later channel turns out shorter (the end of a stream with ragged channels) the copies are made again.
    BlockPlan p, p0;
    char* dp = NULL;
    if (pre_valid_) {
      // a 16-bit stream: this block was planned and sent on its way while the one before was computed and served (prefetch_next_)
      ScopedNs timer(g_upload_ns);
      check_hip(hipStreamSynchronize(cstream()), "hipStreamSynchronize");      // (the host waits: a stream-side wait on an event made the copies slower)
      dPcm16Buf_.swap(dPcm16Next_);
      pre_valid_ = false;
      p = pre_plan_;
      commit_bank_block(p);
      pcm_L_ = p.L; pcm_pitch_ = pre_pitch_; pcm_t0_ = p.f0 - p.b0; pcm_valid_ = p.T > 0; pcm_f32_valid_ = false;
      if (snapshots_wanted_) snapshots_();
      prefetch_next_(p);
      return T_ > 0;
    }
    if (!chunk_loaded_ && !pcm_i16_ && i16_stream_possible()) begin_i16_stream();     // a stream begins: 16-bit PCM if every source has it
    // (16-bit streams: rows of int16 a multiple of 16 bytes apart, straight out of the sources' pinned copies -- nothing to pull)
    const size_t es = pcm_i16_ ? sizeof(short) : sizeof(float);
    DeviceBuffer& dbuf = pcm_i16_ ? dPcm16Buf_ : dPcmBuf_;
    auto pitch_of = [&](long L) { return pcm_i16_ ? (L + 7) / 8 * 8 : L; };
    auto row = [&](unsigned c, long b0) {
      return pcm_i16_ ? static_cast<const void*>(banks_[c]->window16(b0)) : static_cast<const void*>(banks_[c]->window(b0));
    };
    // (groups of as many banks as there are helper threads pull side by side; a group's copies start when the group is done)
    const unsigned G = (unsigned)std::max(1, PullPool::get().threads());
    pull_banks_(0, std::min(G, N));
    plan_from_pulled_(p0, 1);
    const long pitch0 = pitch_of(p0.L);
    if (p0.T > 0) dp = static_cast<char*>(dbuf.ensure(es * N * (pitch0 ? pitch0 : 8)));
    // (the table of the rows' addresses: no upload of this node is in flight when a block begins -- the one before ended with a wait)
    btk_row_t* tab = static_cast<btk_row_t*>(hRows_.ensure(sizeof(btk_row_t) * N));
    for (unsigned g0 = 0; g0 < N; g0 += G) {
      const unsigned g1 = std::min(g0 + G, N);
      if (g0 > 0) pull_banks_(g0, g1);
      bool whole = p0.T > 0;
      for (unsigned c = g0; c < g1 && whole; c++)
        whole = banks_[c]->window_first_block() <= p0.b0 && (banks_[c]->blocks_pulled() - p0.b0) * (long)banks_[c]->shiftlen() >= p0.L;
      if (whole) {
        // (a group whose every bank holds the first bank's block: one gather of its rows; any other group waits for the final plan)
        for (unsigned c = g0; c < g1; c++) tab[c].src = row(c, p0.b0);
        upload_rows(tab + g0, g1 - g0, es * p0.L, dp + es * (size_t)g0 * pitch0, es * pitch0, nstream());
      }
    }
    plan_from_pulled_(p, N);
    long pitch = pitch0;
    {
      ScopedNs timer(g_upload_ns);
      if (p.T > 0 && (p0.T <= 0 || p.b0 != p0.b0 || p.L != p0.L)) {
        nsync();
        pitch = pitch_of(p.L);
        dp = static_cast<char*>(dbuf.ensure(es * N * (pitch ? pitch : 8)));
        for (unsigned c = 0; c < N; c++) tab[c].src = row(c, p.b0);
        upload_rows(tab, N, es * p.L, dp, es * pitch, nstream());
      }
      nsync();                                                   // before the banks move their windows on
    }
    commit_bank_block(p);
    pcm_L_ = p.L; pcm_pitch_ = pitch; pcm_t0_ = p.f0 - p.b0; pcm_valid_ = p.T > 0; pcm_f32_valid_ = false;
    if (snapshots_wanted_) snapshots_();
    prefetch_next_(p);
  } else {
    // channels of any other kind are pulled frame by frame, at most block_frames() frames per block; with halfBandShift the
    // reference dots every one of the M snapshots as supplied (beamformer.cc:1113-1128): a generic source owes no conjugate
    // symmetry between its bins, so all M bins go to the device; dX_ keeps the usual bins 0..M/2 for the other consumers
    const long f0 = chunk_loaded_ ? chunk_base_ + T_ : 0;
    Xhost_.clear();
    chunk_loaded_ = true; chunk_base_ = f0; T_ = 0; pcm_valid_ = false; snap_valid_ = false; dX_ = NULL; dXfull_ = NULL;
    const unsigned M = fftLen_, rows = halfBandShift_ ? M : K;
    std::vector<std::vector<float> > fr(N);
    long T = -1; unsigned c = 0;
    bool ended = false;
    for (ChannelList_::iterator it = channelList_.begin(); it != channelList_.end(); ++it, ++c) {
      long t = 0;
      const long per_block = block_frames_ == 0 ? 0 : (block_frames_ + quantum_ - 1) / quantum_ * quantum_;
      while (per_block == 0 || t < per_block) {
        const gsl_vector_complex* v;
        try { v = (*it)->next(); } catch (jiterator_error&) { ended = true; break; }
        fr[c].resize((size_t)(t + 1) * rows * 2);
        for (unsigned k = 0; k < 2 * rows; k++) fr[c][2 * (size_t)t * rows + k] = (float)v->data[k];
        t++;
      }
      if (T < 0 || t < T) T = t;
    }
    channels_ended_ = ended;
    T_ = T;
    const size_t nx = (size_t)2 * rows * N * T_;
    float* Xh = static_cast<float*>(hStage_.ensure(sizeof(float) * (nx ? nx : 1)));
    for (unsigned k = 0; k < rows; k++)
      for (unsigned n = 0; n < N; n++)
        for (long t = 0; t < T_; t++) {
          Xh[2 * (((size_t)k * N + n) * T_ + t)] = fr[n][2 * ((size_t)t * rows + k)];
          Xh[2 * (((size_t)k * N + n) * T_ + t) + 1] = fr[n][2 * ((size_t)t * rows + k) + 1];
        }
    dX_ = dXBuf_.ensure(sizeof(float) * 2 * K * N * (T_ ? T_ : 1));
    if (halfBandShift_) {
      dXfull_ = dXfullBuf_.ensure(sizeof(float) * (nx ? nx : 2));
      h2d_async(dXfull_, Xh, sizeof(float) * nx);
      h2d_async(dX_, Xh, sizeof(float) * 2 * K * N * T_);
    } else {
      h2d_async(dX_, Xh, sizeof(float) * nx);
    }
    nsync();                                                     // the staging buffer is free again
    snap_valid_ = true;
  }
  return T_ > 0;
}

SnapShotArrayPtr SubbandBeamformer::snapshot_array()
{
  const unsigned N = chanN(), K = fftLen2_ + 1;
  if (snapshot_array_.is_null()) snapshot_array_ = new SnapShotArray(fftLen_, N);
  const long lf = frame_no_ - chunk_base_;                     // the frame next() served last, within the current block
  if (chunk_loaded_ && lf >= 0 && lf < T_) {
    if (Xhost_.empty()) { void* dX = snapshots_(); Xhost_.resize((size_t)2 * K * N * T_); d2h(Xhost_.data(), dX, sizeof(float) * Xhost_.size()); }
    gsl_vector_complex** snaps = snapshot_array_->raw_snapshots();
    for (unsigned k = 0; k < K; k++)
      for (unsigned n = 0; n < N; n++) {
        const double re = Xhost_[2 * (((size_t)k * N + n) * T_ + lf)], im = Xhost_[2 * (((size_t)k * N + n) * T_ + lf) + 1];
        snaps[k]->data[2 * n] = re; snaps[k]->data[2 * n + 1] = im;
        if (k > 0 && k < fftLen2_) { snaps[fftLen_ - k]->data[2 * n] = re; snaps[fftLen_ - k]->data[2 * n + 1] = -im; }
      }
  }
  return snapshot_array_;
}

// ================================================================================ SubbandDS
SubbandDS::SubbandDS(unsigned fftLen, bool halfBandShift, const String& nm)
    : SubbandBeamformer(fftLen, halfBandShift, nm), bfweight_(NULL), weights_version_(0), output_version_(0),
      handed_(-1), out_valid_(false), Yhost_valid_(false), w_dev_version_(0), w_dev_valid_(false),
      wq_view_(gsl_vector_complex_calloc(1)) {}
SubbandDS::~SubbandDS() { delete bfweight_; gsl_vector_complex_free(wq_view_); }

void SubbandDS::clear_channel() { SubbandBeamformer::clear_channel(); delete bfweight_; bfweight_ = NULL; }

void SubbandDS::alloc_bfweight_(int NC)
{
  // re-creates the weight object: active weights and post-filter state start over (reference beamformer.cc:1082-1092)
  delete bfweight_;
  bfweight_ = new BeamformerWeights(fftLen_, chanN(), halfBandShift_, (unsigned)NC);
  weights_version_++;
}

void SubbandDS::calc_array_manifold_vectors(float samplerate, const gsl_vector* delays)
{
  alloc_bfweight_(1);
  bfweight_->calcMainlobe(samplerate, delays, false);
}

void SubbandDS::calc_array_manifold_vectors_2(float samplerate, const gsl_vector* delaysT, const gsl_vector* delaysJ)
{
  alloc_bfweight_(2);
  bfweight_->calcMainlobe2(samplerate, delaysT, delaysJ, false);
}

void SubbandDS::calc_array_manifold_vectors_n(float samplerate, const gsl_vector* delaysT, const gsl_matrix* delaysJ, unsigned NC)
{
  alloc_bfweight_((int)NC);
  bfweight_->calcMainlobeN(samplerate, delaysT, delaysJ, NC, false);
}

const gsl_vector_complex* SubbandDS::get_weights(unsigned fbinX)
{
  const unsigned N = chanN();
  gsl_vector_complex_free(wq_view_);
  wq_view_ = gsl_vector_complex_calloc(N);
  memcpy(wq_view_->data, &bfweight_->wq_v[(size_t)fbinX * N], sizeof(double) * 2 * N);
  return wq_view_;
}

void SubbandDS::effective_weights(std::vector<float>& w)
{
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  w.resize((size_t)2 * (fftLen2_ + 1) * chanN());
  check_abi(btk_weights_gsc_effective(reinterpret_cast<const double*>(bfweight_->wq_v.data()), NULL, (int)fftLen_, (int)chanN(), 0, w.data()));
}

void SubbandDS::effective_weights_all_bins(std::vector<float>& w)
{
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  const size_t n = (size_t)fftLen_ * chanN();
  w.resize(2 * n);
  for (size_t i = 0; i < n; i++) { w[2 * i] = (float)bfweight_->wq_v[i].real(); w[2 * i + 1] = (float)bfweight_->wq_v[i].imag(); }
}

void SubbandDS::alignment_vector(bool use_wq, std::vector<float>& d)
{
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  const unsigned N = chanN(), K = fftLen2_ + 1;
  const std::vector<cd>& src = use_wq ? bfweight_->wq_v : bfweight_->ta_v;
  d.resize((size_t)2 * K * N);
  for (size_t i = 0; i < (size_t)K * N; i++) { d[2 * i] = (float)src[i].real(); d[2 * i + 1] = (float)src[i].imag(); }
}

// fused analysis -> apply: the channels are analysis banks of a geometry that has a fused kernel, the block's samples are
// resident, and nobody has asked for the snapshots (then the staged pair runs: the snapshots exist anyway)
bool SubbandDS::fused_path()
{
  ensure_chunk_();
  return banks_only_ && !halfBandShift_ && !snapshots_wanted_ && !snap_valid_ && btk_fb_analysis_bf_fused(banks_[0]->plan()) == 1;
}

// Frames [from_frame, T_) of the current block with the current weights, on the device: Y complex64 [rows][T_] in dYBuf_
// (rows = K, or M with halfBandShift: every bin has its own output).  Frames before from_frame keep what they hold.
void SubbandDS::compute_output_(long from_frame)
{
  ScopedNs timer(g_device_ns);
  const unsigned N = chanN(), K = fftLen2_ + 1, M = fftLen_;
  ensure_chunk_();
  const unsigned rows = halfBandShift_ ? M : K;
  const long Tn = T_ - from_frame;
  float* dY = static_cast<float*>(dYBuf_.ensure(sizeof(float) * 2 * rows * (T_ ? T_ : 1)));
  if (!halfBandShift_) {
    {
      // from the weight object as it is NOW: what a caller wrote through wq() / wa() / B() counts from the next block on even
      // without a call that bumps weights_version_ (131 KB per block at 64 channels x 257 bins)
      std::vector<float> w;
      effective_weights(w);
      h2d(dWBuf_.ensure(sizeof(float) * w.size()), w.data(), sizeof(float) * w.size());
      w_dev_valid_ = true; w_dev_version_ = weights_version_;
    }
    if (Tn > 0) {
      if (fused_path() && pcm_valid_) {
        // SubbandGSC::next over OverSampledDFTAnalysisBank::next (beamformer.cc:1251-1316 over modulated.cc:375-409) in one
        // kernel: the block's samples in, the beamformed frames out
        const long sb = btk_fb_analysis_bf_scratch_bytes(banks_[0]->plan(), 1, (int)N, 0, Tn);
        void* scratch = dScratchBuf_.ensure((size_t)(sb > 0 ? sb : 16));
        if (pcm_i16_ && btk_fb_analysis_bf_i16_fused(banks_[0]->plan()) == 1)
          check_abi(btk_fb_analysis_bf_i16(banks_[0]->plan(), static_cast<const short*>(dPcm16Buf_.get()), pcm_L_, pcm_pitch_ ? pcm_pitch_ : 8, 1,
                                           (int)N, dWBuf_.get(), 0, dY + 2 * from_frame, T_, pcm_t0_ + from_frame, Tn, scratch, sb, nstream()));
        else
          check_abi(btk_fb_analysis_bf(banks_[0]->plan(), pcm_f32_(), pcm_L_, pcm_pitch_ ? pcm_pitch_ : 1, 1, (int)N,
                                       dWBuf_.get(), 0, dY + 2 * from_frame, T_, pcm_t0_ + from_frame, Tn, scratch, sb, nstream()));
      } else {
        // frames [from_frame, T): the frame axis offset by pointer arithmetic, strides stay T_
        const float* dX = static_cast<const float*>(snapshots_());
        check_abi(btk_bf_apply(dWBuf_.get(), 0, dX + 2 * from_frame, dY + 2 * from_frame, 1, (int)K, (int)N, T_, Tn, nstream()));
      }
    }
  } else if (Tn > 0) {
    // reference beamformer.cc:1113-1128 / 1276-1285: y_k = w_k^H x_k for k = 0..M-1 (a rare configuration: host-staged weights)
    std::vector<float> wf;
    effective_weights_all_bins(wf);                              // [M][N]
    w_dev_valid_ = false;
    const float* dX = static_cast<const float*>(snapshots_());
    if (dXfull_) {                                               // pulled sources: all M snapshot bins as supplied, one pass
      h2d(dWBuf_.ensure(sizeof(float) * wf.size()), wf.data(), sizeof(float) * wf.size());
      check_abi(btk_bf_apply(dWBuf_.get(), 0, static_cast<const float*>(dXfull_) + 2 * from_frame, dY + 2 * from_frame, 1, (int)M, (int)N, T_, Tn, nstream()));
    } else {
      // analysis banks of a real signal: x_{M-k} = conj(x_k), so y_{M-k} = conj((conj w_{M-k})^H x_k): a second pass of the same
      // kernel over bins 1 .. M/2-1 with the conjugated upper weights, into rows K .. 2K-1 of the scratch block, mirrored on the host
      std::vector<float> w2((size_t)2 * K * N, 0.f), lo((size_t)2 * K * T_), up((size_t)2 * K * T_), Yfull((size_t)2 * M * T_);
      for (unsigned k = 1; k + 1 < K; k++)
        for (unsigned c = 0; c < N; c++) {
          w2[2 * ((size_t)k * N + c)] = wf[2 * ((size_t)(M - k) * N + c)];
          w2[2 * ((size_t)k * N + c) + 1] = -wf[2 * ((size_t)(M - k) * N + c) + 1];
        }
      float* dT = static_cast<float*>(dScratchBuf_.ensure(sizeof(float) * 2 * K * (T_ ? T_ : 1)));
      h2d(dWBuf_.ensure(sizeof(float) * 2 * M * N), wf.data(), sizeof(float) * 2 * K * N);
      check_abi(btk_bf_apply(dWBuf_.get(), 0, dX, dT, 1, (int)K, (int)N, T_, T_, nstream()));
      d2h(lo.data(), dT, sizeof(float) * lo.size());
      h2d(dWBuf_.get(), w2.data(), sizeof(float) * 2 * K * N);
      check_abi(btk_bf_apply(dWBuf_.get(), 0, dX, dT, 1, (int)K, (int)N, T_, T_, nstream()));
      d2h(up.data(), dT, sizeof(float) * up.size());
      if (from_frame > 0) d2h(Yfull.data(), dY, sizeof(float) * Yfull.size());      // frames already served keep their values
      for (unsigned k = 0; k < K; k++)
        memcpy(&Yfull[2 * ((size_t)k * T_ + from_frame)], &lo[2 * ((size_t)k * T_ + from_frame)], sizeof(float) * 2 * (size_t)Tn);
      for (unsigned k = 1; k + 1 < K; k++)
        for (long t = from_frame; t < T_; t++) {
          Yfull[2 * ((size_t)(M - k) * T_ + t)] = up[2 * ((size_t)k * T_ + t)];
          Yfull[2 * ((size_t)(M - k) * T_ + t) + 1] = -up[2 * ((size_t)k * T_ + t) + 1];
        }
      h2d(dY, Yfull.data(), sizeof(float) * Yfull.size());
    }
  }
  out_valid_ = true; Yhost_valid_ = false;
  output_version_ = weights_version_;
}

// frames of the current block a weight change must leave alone: everything next() served or a block consumer was handed
static long kept_frames(long frame_no, long handed, long chunk_base, long T)
{
  const long k = std::max<long>(frame_no, handed) + 1 - chunk_base;
  return k < 0 ? 0 : (k > T ? T : k);
}

void SubbandDS::ensure_output_()
{
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  ensure_chunk_();
  if (!out_valid_) compute_output_(0);
  else if (output_version_ != weights_version_) compute_output_(kept_frames(frame_no_, handed_, chunk_base_, T_));
}

const float* SubbandDS::host_output_()
{
  ensure_output_();
  if (!Yhost_valid_) {
    const unsigned rows = halfBandShift_ ? fftLen_ : fftLen2_ + 1;
    Yhost_.resize((size_t)2 * rows * T_);
    d2h(Yhost_.data(), dYBuf_.get(), sizeof(float) * Yhost_.size());
    Yhost_valid_ = true;
  }
  return Yhost_.data();
}

const gsl_vector_complex* SubbandDS::next(int frame_no)
{
  if (frame_no == frame_no_) return vector_;
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  ensure_chunk_();
  const long idx = frame_no_ + 1;
  while (idx >= chunk_base_ + T_)
    if (!advance_chunk_()) { is_end_ = true; throw jiterator_error("end of samples!"); }
  if (idx < chunk_base_)      // a block consumer (the synthesis bank, a post-filter) has moved the node past this frame
    throw jconsistency_error("%s: frame %ld was asked for after a block consumer moved the stream on to frame %ld; pull the node "
                             "through one consumer\n", name().c_str(), idx, chunk_base_);
  const float* Yh = host_output_();
  if (halfBandShift_) serve_frame_all_bins(Yh, T_, fftLen_, idx - chunk_base_, vector_);
  else serve_frame(Yh, T_, fftLen_, idx - chunk_base_, vector_);
  increment_();
  return vector_;
}

void SubbandDS::reset() { SubbandBeamformer::reset(); Yhost_.clear(); handed_ = -1; out_valid_ = false; Yhost_valid_ = false; }

bool SubbandDS::advance_chunk_()
{
  out_valid_ = false; Yhost_valid_ = false;
  return next_chunk();
}

const std::vector<float>& SubbandDS::block(long& T)
{
  host_output_();
  T = T_;
  return Yhost_;
}

const void* SubbandDS::device_block(long& T, long& T_stride)
{
  ensure_output_();
  T = T_; T_stride = T_;
  return dYBuf_.get();
}

void SubbandDS::advance_to(long frame_idx)
{
  // frames 0 .. frame_idx (stream indices) count as handed over to the block consumer: a weight change from now on recomputes only
  // the frames beyond.  The mark is the block protocol's own -- frame_no_ stays what next() has served, so a second consumer that
  // pulls this node frame by frame (or the script itself) still gets every frame
  if (frame_idx >= chunk_base_ + T_) frame_idx = chunk_base_ + T_ - 1;
  if (frame_idx > handed_) handed_ = frame_idx;
}

// ================================================================================ SubbandGSC
void SubbandGSC::calc_gsc_weights(float samplerate, const gsl_vector* delaysT)
{
  alloc_bfweight_(1);
  bfweight_->calcMainlobe(samplerate, delaysT, true);
}

void SubbandGSC::calc_gsc_weights_2(float samplerate, const gsl_vector* delaysT, const gsl_vector* delaysJ)
{
  alloc_bfweight_(2);
  bfweight_->calcMainlobe2(samplerate, delaysT, delaysJ, true);
}

void SubbandGSC::calc_gsc_weights_n(float samplerate, const gsl_vector* delaysT, const gsl_matrix* delaysIs, unsigned NC)
{
  alloc_bfweight_((int)NC);
  bfweight_->calcMainlobeN(samplerate, delaysT, delaysIs, NC, true);
}

bool SubbandGSC::write_fir_coeff(const String& fn, unsigned winType)
{
  if (!bfweight_) { fprintf(stderr, "call calc_array_manifold_vectorsX() once\n"); return false; }
  return bfweight_->write_fir_coeff(fn, winType);
}

gsl_matrix_complex* SubbandGSC::blocking_matrix(unsigned srcX, unsigned fbinX)
{
  if (!bfweight_ || srcX != 0) throw j_error("call calc_gsc_weights_x() once\n");
  return bfweight_->B_f(fbinX);
}

void SubbandGSC::set_quiescent_weights_f(unsigned fbinX, const gsl_vector_complex* srcWq)
{
  alloc_bfweight_(1);
  memcpy(static_cast<void*>(&bfweight_->wq_v[(size_t)fbinX * chanN()]), srcWq->data, sizeof(double) * 2 * chanN());
  bfweight_->calcBlockingMatrix(fbinX);
}

void SubbandGSC::set_active_weights_f(unsigned fbinX, const gsl_vector* packedWeight)
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  bfweight_->calcSidelobeCancellerP_f(fbinX, packedWeight);
  weights_version_++;
}

void SubbandGSC::zero_active_weights()
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  std::vector<cd> z(chanN() - bfweight_->NC(), cd(0.0, 0.0));
  for (unsigned k = 0; k < fftLen_; k++) bfweight_->calcSidelobeCancellerU_f(k, z.data());
  weights_version_++;
}

void SubbandGSC::effective_weights(std::vector<float>& w)
{
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  w.resize((size_t)2 * (fftLen2_ + 1) * chanN());
  check_abi(btk_weights_gsc_effective(reinterpret_cast<const double*>(bfweight_->wq_v.data()),
                                      reinterpret_cast<const double*>(bfweight_->wl_v.data()), (int)fftLen_, (int)chanN(),
                                      normalize_weight_ ? 1 : 0, w.data()));
}

void SubbandGSC::effective_weights_all_bins(std::vector<float>& w)
{
  // wq - wl of every bin, normalised like calc_gsc_output (reference beamformer.cc:1208-1243, 1276-1285)
  if (!bfweight_) throw j_error("%s", need_weights_msg_());
  const unsigned N = chanN();
  w.resize((size_t)2 * fftLen_ * N);
  for (unsigned k = 0; k < fftLen_; k++) {
    double nn = 0.0;
    for (unsigned c = 0; c < N; c++) nn += std::norm(bfweight_->wq_v[(size_t)k * N + c] - bfweight_->wl_v[(size_t)k * N + c]);
    const double sc = normalize_weight_ ? 1.0 / (std::sqrt(nn) * N) : 1.0;
    for (unsigned c = 0; c < N; c++) {
      const cd v = (bfweight_->wq_v[(size_t)k * N + c] - bfweight_->wl_v[(size_t)k * N + c]) * sc;
      w[2 * ((size_t)k * N + c)] = (float)v.real(); w[2 * ((size_t)k * N + c) + 1] = (float)v.imag();
    }
  }
}

// ================================================================================ SubbandMVDR
// "linpack" | "exact": see SubbandMVDR::set_svd_rule (beamformer.h); BTK_MVDR_SVD_RULE overrides the default "linpack"
static String default_svd_rule()
{
  const char* e = getenv("BTK_MVDR_SVD_RULE");
  if (!e || !*e) return "linpack";
  if (strcmp(e, "linpack") && strcmp(e, "exact")) throw jparameter_error("BTK_MVDR_SVD_RULE must be linpack or exact, got %s\n", e);
  return e;
}

void SubbandMVDR::set_svd_rule(const String& rule)
{
  if (rule != "linpack" && rule != "exact") throw jparameter_error("svd rule must be linpack or exact, got %s\n", rule.c_str());
  svd_rule_ = rule;
}

SubbandMVDR::SubbandMVDR(unsigned fftLen, bool halfBandShift, const String& nm)
    : SubbandDS(fftLen, halfBandShift, nm), dR_(NULL), have_mvdr_(false), fallbacks_(0), svd_rule_(default_svd_rule()),
      csvdc_not_converged_(0), wm_view_(gsl_vector_complex_calloc(1)), R_view_(NULL)
{
  if (halfBandShift) {                                       // reference beamformer.cc:2283-2285
    gsl_vector_complex_free(wm_view_);
    throw jallocation_error("halfBandShift==true is not yet supported\n");
  }
}
SubbandMVDR::~SubbandMVDR() { dev_free(dR_); gsl_vector_complex_free(wm_view_); gsl_matrix_complex_free(R_view_); }

void SubbandMVDR::divide_all_nondiagonal_elements(float mu)
{
  if (!dR_) throw j_error("Construct first a noise covariance matrix\n");
  check_abi(btk_mvdr_divide_nondiagonal(dR_, (int)(fftLen2_ + 1), (int)chanN(), mu, nstream()));
}

void SubbandMVDR::divide_nondiagonal_elements(unsigned fbinX, float mu)
{
  if (!dR_) throw j_error("Construct first a noise covariance matrix\n");
  if (fbinX > fftLen2_) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)fftLen2_);
  const unsigned N = chanN();
  check_abi(btk_mvdr_divide_nondiagonal(static_cast<float*>(dR_) + (size_t)fbinX * 2 * N * N, 1, (int)N, mu, nstream()));
}

const gsl_matrix_complex* SubbandMVDR::noise_spatial_spectral_matrix(unsigned fbinX)
{
  if (!dR_) return NULL;                                       // the reference returns its NULL R_[fbinX]
  if (fbinX > fftLen2_) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)fftLen2_);
  const unsigned N = chanN();
  if (!R_view_ || R_view_->size1 != N) { gsl_matrix_complex_free(R_view_); R_view_ = gsl_matrix_complex_alloc(N, N); }
  std::vector<float> r((size_t)2 * N * N);
  nsync();
  d2h(r.data(), static_cast<float*>(dR_) + (size_t)fbinX * 2 * N * N, sizeof(float) * r.size());
  for (size_t i = 0; i < (size_t)2 * N * N; i++) R_view_->data[i] = r[i];
  return R_view_;
}
void SubbandMVDR::clear_channel() { SubbandDS::clear_channel(); dev_free(dR_); dR_ = NULL; have_mvdr_ = false; }

void SubbandMVDR::alloc_R_()
{
  if (dR_) return;
  const size_t n = (size_t)2 * (fftLen2_ + 1) * chanN() * chanN();
  dR_ = dev_alloc(sizeof(float) * n);
  dev_zero_async(dR_, sizeof(float) * n);
}

bool SubbandMVDR::set_noise_spatial_spectral_matrix(unsigned fbinX, gsl_matrix_complex* Rnn)
{
  const unsigned N = chanN();
  if (Rnn->size1 != N) { fprintf(stderr, "The number of the rows of the matrix must be %d but it is %lu\n", N, (unsigned long)Rnn->size1); return false; }
  if (Rnn->size2 != N) { fprintf(stderr, "The number of the columns of the matrix must be %d but it is %lu\n", N, (unsigned long)Rnn->size2); return false; }
  if (fbinX > fftLen2_) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)fftLen2_);
  alloc_R_();
  std::vector<float> r((size_t)2 * N * N);
  for (unsigned a = 0; a < N; a++)
    for (unsigned b = 0; b < N; b++) {
      const gsl_complex z = gsl_matrix_complex_get(Rnn, a, b);
      r[2 * ((size_t)a * N + b)] = (float)GSL_REAL(z); r[2 * ((size_t)a * N + b) + 1] = (float)GSL_IMAG(z);
    }
  h2d(static_cast<float*>(dR_) + (size_t)fbinX * 2 * N * N, r.data(), sizeof(float) * r.size());
  return true;
}

bool SubbandMVDR::set_diffuse_noise_model(const gsl_matrix* micPositions, float samplerate, float sspeed)
{
  const unsigned N = chanN();
  if (micPositions->size1 != N) { fprintf(stderr, "The number of microphones must be %d but it is %lu\n", N, (unsigned long)micPositions->size1); return false; }
  if (micPositions->size2 < 3) { fprintf(stderr, "The microphone positions should be described in the three dimensions\n"); return false; }
  alloc_R_();
  std::vector<float> mp((size_t)3 * N);
  for (unsigned a = 0; a < N; a++) for (int j = 0; j < 3; j++) mp[3 * a + j] = (float)gsl_matrix_get(micPositions, a, j);
  void* dmp = dev_alloc(sizeof(float) * mp.size());
  h2d(dmp, mp.data(), sizeof(float) * mp.size());
  check_abi(btk_mvdr_diffuse_model((const float*)dmp, (int)N, (int)fftLen_, samplerate, sspeed, dR_, nstream()));
  nsync();
  dev_free(dmp);
  return true;
}

void SubbandMVDR::set_all_diagonal_loading(float diagonalWeight)
{
  if (!dR_) throw j_error("Construct first a noise covariance matrix\n");
  check_abi(btk_mvdr_diagonal_loading(dR_, (int)(fftLen2_ + 1), (int)chanN(), diagonalWeight, nstream()));
}

void SubbandMVDR::set_diagonal_looading(unsigned fbinX, float diagonalWeight)
{
  if (!dR_) throw j_error("Construct first a noise covariance matrix\n");
  if (fbinX > fftLen2_) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)fftLen2_);
  const unsigned N = chanN();
  check_abi(btk_mvdr_diagonal_loading(static_cast<float*>(dR_) + (size_t)fbinX * 2 * N * N, 1, (int)N, diagonalWeight, nstream()));
}

bool SubbandMVDR::calc_mvdr_weights(float, float dThreshold, bool)
{
  if (!dR_) throw jallocation_error("Set a spatial spectral matrix before calling calc_mvdr_weights()\n");
  if (!bfweight_) throw j_error("call calc_array_manifold_vectorsX() once\n");
  const unsigned N = chanN(), K = fftLen2_ + 1;
  std::vector<float> d;
  alignment_vector(true, d);
  void* dD = dev_alloc(sizeof(float) * d.size());
  void* dW = dev_alloc(sizeof(float) * d.size());
  void* dfb = dev_alloc(sizeof(int));
  const long sbytes = btk_mvdr_scratch_bytes((int)K, (int)N);                   // (only the panel solver copies R)
  void* scratch = sbytes ? dev_alloc((size_t)sbytes) : NULL;
  void* dflags = dev_alloc(sizeof(int) * K);
  h2d(dD, d.data(), sizeof(float) * d.size());
  dev_zero_async(dfb, sizeof(int));
  check_abi(btk_mvdr_weights_flags(dR_, dD, dW, (int)K, (int)N, 0, dThreshold, scratch, (int*)dfb, (int*)dflags, nstream()));
  int rule_counts[2] = {0, 0};
  if (svd_rule_ == "linpack") {
    // pseudoinverse() returns false where the reference's float32 csvdc gives INFO != 0 or a singular value under the
    // threshold -> identity (beamformer.cc:253-270, 2379-2384); those bins also leave the fall-back's list
    void* dcnt = dev_alloc(sizeof(int) * 2);
    void* rs = dev_alloc((size_t)btk_mvdr_linpack_rule_scratch_bytes((int)K, (int)N));
    dev_zero_async(dcnt, sizeof(int) * 2);
    check_abi(btk_mvdr_linpack_rule(dR_, dD, dW, NULL, (int)K, (int)N, 0, 0, 1, dThreshold, (int*)dflags, (int*)dcnt, rs, nstream()));
    nsync();
    d2h(rule_counts, dcnt, sizeof(rule_counts));
    dev_free(dcnt); dev_free(rs);
  }
  nsync();
  csvdc_not_converged_ = rule_counts[0];
  std::vector<int> hflags(K);
  d2h(hflags.data(), dflags, sizeof(int) * K);
  int stopped = 0;
  for (unsigned k = 0; k < K; ++k) stopped += hflags[k] != 0;
  fallbacks_ = 0;
  if (stopped > 0)     // bins the Cholesky solve gave up on and the rule above left open: the pseudo-inverse (beamformer.cc:232-289)
    check_abi(btk_mvdr_pinv_fallback(dR_, dD, dW, (int)K, (int)N, 0, dThreshold, (const int*)dflags, &fallbacks_, nstream()));
  fallbacks_ += rule_counts[0] + rule_counts[1];
  wmvdr_.resize(d.size());
  d2h(wmvdr_.data(), dW, sizeof(float) * wmvdr_.size());
  dev_free(dD); dev_free(dW); dev_free(dfb); dev_free(dflags); dev_free(scratch);
  have_mvdr_ = true;
  weights_version_++;
  return true;
}

const gsl_vector_complex* SubbandMVDR::mvdr_weights(unsigned fbinX)
{
  const unsigned N = chanN();
  gsl_vector_complex_free(wm_view_);
  wm_view_ = gsl_vector_complex_calloc(N);
  for (unsigned c = 0; c < N; c++) {
    wm_view_->data[2 * c] = wmvdr_[2 * ((size_t)fbinX * N + c)];
    wm_view_->data[2 * c + 1] = wmvdr_[2 * ((size_t)fbinX * N + c) + 1];
  }
  return wm_view_;
}

void SubbandMVDR::effective_weights(std::vector<float>& w)
{
  if (!bfweight_) throw j_error("call calc_array_manifold_vectorsX() once\n");
  if (!have_mvdr_) throw j_error("call calc_mvdr_weights() once\n");
  w = wmvdr_;
}

// ================================================================================ SubbandMVDRGSC
void SubbandMVDRGSC::set_active_weights_f(unsigned fbinX, const gsl_vector* packedWeight)
{
  if (!bfweight_) throw j_error("set the quiescent vector once\n");
  bfweight_->calcSidelobeCancellerP_f(fbinX, packedWeight);
  weights_version_++;
}

void SubbandMVDRGSC::zero_active_weights()
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  std::vector<cd> z(chanN() - bfweight_->NC(), cd(0.0, 0.0));
  for (unsigned k = 0; k < fftLen_; k++) bfweight_->calcSidelobeCancellerU_f(k, z.data());
  weights_version_++;
}

// blocking matrix orthogonal to the delay-and-sum weights (beamformer.cc:2637-2642)
bool SubbandMVDRGSC::calc_blocking_matrix1(float samplerate, const gsl_vector* delaysT)
{
  alloc_bfweight_(1);
  bfweight_->calcMainlobe(samplerate, delaysT, true);
  return true;
}

// blocking matrix orthogonal to the MVDR weights (beamformer.cc:2648-2672): bins 1..M/2, the others stay zero
bool SubbandMVDRGSC::calc_blocking_matrix2()
{
  if (!have_mvdr_) return false;
  alloc_bfweight_(1);
  const unsigned N = chanN();
  for (unsigned k = 1; k <= fftLen2_; k++) {
    for (unsigned c = 0; c < N; c++)
      bfweight_->wq_v[(size_t)k * N + c] = cd(wmvdr_[2 * ((size_t)k * N + c)], wmvdr_[2 * ((size_t)k * N + c) + 1]);
    bfweight_->calcBlockingMatrix(k);
  }
  return true;
}

// B <- blocking matrix of the entire vector wq - wl (beamformer.cc:2674-2691)
void SubbandMVDRGSC::upgrade_blocking_matrix()
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  const unsigned N = chanN(), bs = N - bfweight_->NC();
  std::vector<cd> w(N);
  for (unsigned k = 1; k < fftLen_; k++) {
    for (unsigned c = 0; c < N; c++) w[c] = bfweight_->wq_v[(size_t)k * N + c] - bfweight_->wl_v[(size_t)k * N + c];
    check_abi(btk_weights_blocking_matrix(reinterpret_cast<const double*>(w.data()), (int)N, (int)bfweight_->NC(),
                                          reinterpret_cast<double*>(&bfweight_->B_v[(size_t)k * N * bs])));
  }
}

// b_i^H x of the current frame for bins 0..M/2 (beamformer.cc:2693-2717); the other bins keep the beamformer output
const gsl_vector_complex* SubbandMVDRGSC::blocking_matrix_output(int outChanX)
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  const unsigned N = chanN(), bs = N - bfweight_->NC();
  if (outChanX < 0 || (unsigned)outChanX >= bs) throw jdimension_error("blocking matrix has %d columns\n", (int)bs);
  SnapShotArrayPtr snaps = snapshot_array();
  for (unsigned k = 0; k <= fftLen2_; k++) {
    const gsl_vector_complex* x = snaps->snapshot(k);
    cd acc(0.0, 0.0);
    for (unsigned c = 0; c < N; c++)
      acc += std::conj(bfweight_->B_v[((size_t)k * N + c) * bs + outChanX]) * cd(x->data[2 * c], x->data[2 * c + 1]);
    vector_->data[2 * k] = acc.real(); vector_->data[2 * k + 1] = acc.imag();
  }
  return vector_;
}

void SubbandMVDRGSC::effective_weights(std::vector<float>& w)
{
  if (!bfweight_) throw j_error("call calc_array_manifold_vectorsX() once\n");
  if (!have_mvdr_) throw j_error("call calc_mvdr_weights() once\n");
  const unsigned N = chanN(), K = fftLen2_ + 1;
  std::vector<cd> wq((size_t)fftLen_ * N, cd(0.0, 0.0));
  for (size_t i = 0; i < (size_t)K * N; i++) wq[i] = cd(wmvdr_[2 * i], wmvdr_[2 * i + 1]);
  w.resize((size_t)2 * K * N);
  check_abi(btk_weights_gsc_effective(reinterpret_cast<const double*>(wq.data()), reinterpret_cast<const double*>(bfweight_->wl_v.data()),
                                      (int)fftLen_, (int)N, normalize_weight_ ? 1 : 0, w.data()));
}

// ================================================================================ ZelinskiPostFilter
ZelinskiPostFilter::ZelinskiPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha, int type,
                                       int minFrames, const String& nm)
    : VectorComplexFeatureStream(fftLen, nm), fftLen_(fftLen), samp_(output), type_((PostfilterType)type), alpha_(alpha),
      min_frames_(minFrames), has_bf_ptr_(false), Yhost_valid_(false), T_(0), prepared_(false), bf_version_(0), dPhi_(NULL), dPsi_(NULL), dWl_(NULL),
      wp1_(gsl_vector_complex_calloc(fftLen)), hist_start_(0), own_weights_(NULL), manual_frames_(0), base_(0), carry_state_(false),
      handed_(-1)
{
  if (output->size() != fftLen) throw jdimension_error("Input block length (%d) != fftLen (%d)\n", output->size(), fftLen);
}

ZelinskiPostFilter::~ZelinskiPostFilter()
{
  if (has_bf_ptr_ && bf_ptr_->beamformer_weight_object()) bf_ptr_->beamformer_weight_object()->clear_csd_provider(this);
  dev_free(dPhi_); dev_free(dPsi_); dev_free(dWl_); gsl_vector_complex_free(wp1_);
  delete own_weights_;
}

void ZelinskiPostFilter::set_beamformer(SubbandDSPtr& bfptr)
{
  if (!has_bf_ptr_ && own_weights_) { delete own_weights_; own_weights_ = NULL; }      // postfilter.cc:373-382
  has_bf_ptr_ = true;
  bf_ptr_ = bfptr;
  bf_ptr_->set_block_quantum(64);           // the density recursions are scanned in 64-frame chunks (csrc/pf_kernels.hip)
}

void ZelinskiPostFilter::set_snapshot_array(SnapShotArrayPtr& snapShotArray) { snapshot_array_ = snapShotArray; }

// postfilter.cc:393-415: the vector goes into wq (type & TYPE_ZELINSKI2) or the array manifold of a weight object this
// post-filter owns
void ZelinskiPostFilter::set_array_manifold_vector(unsigned fbinX, gsl_vector_complex* v, bool halfBandShift, unsigned NC)
{
  if (fbinX >= size()) throw jdimension_error("fbinX %d must be less than %d\n", fbinX, size());
  if (has_bf_ptr_) throw jconsistency_error("ZelinskiPostFilter: the weights belong to the beamformer given to set_beamformer()\n");
  const unsigned chanN = (unsigned)v->size;
  if (!own_weights_) {
    own_weights_ = new BeamformerWeights(size(), chanN, halfBandShift, NC);
    own_weights_->set_csd_provider([this](gsl_vector_complex** out) { fill_csds_(out); }, this);
  }
  if (chanN != own_weights_->chanN()) throw jdimension_error("array manifold vector of %d channels, %d expected\n", chanN, own_weights_->chanN());
  gsl_vector_complex** dst = (type_ & TYPE_ZELINSKI2) ? own_weights_->wq() : own_weights_->arrayManifold();
  for (unsigned c = 0; c < chanN; c++) gsl_vector_complex_set(dst[fbinX], c, gsl_vector_complex_get(v, c));
}

void ZelinskiPostFilter::bind_csd_provider_()
{
  if (has_bf_ptr_ && bf_ptr_->beamformer_weight_object())
    bf_ptr_->beamformer_weight_object()->set_csd_provider([this](gsl_vector_complex** out) { fill_csds_(out); }, this);
}

// BeamformerWeights::CSDs() on demand (postfilter.cc:77-116): Phi_ij <- a Phi_ij + (1 - a) x'_i conj x'_j for i < j and the same
// recursion on |x'_i|^2, x' = conj(d) x, a = 0 for the post-filter's first two frames, history restarted where the weights were
// recomputed.  The recursion is linear, so the state after frame t of the current block is
//   (prod of a over the block's frames up to t) * [state after the block before]  +  sum_tau fw[tau] x(tau) x(tau)^H,
// fw[tau] = (1 - a_tau) prod_{sigma > tau} a_sigma: ONE weighted covariance launch (btk_cov_accumulate) over the block's raw
// snapshots plus the carried matrices (csd_carry_, refreshed whenever a block is used up).  R: complex64 [K][N][N], unrotated.
void ZelinskiPostFilter::csd_state_(long t, std::vector<float>& R)
{
  const unsigned K = fftLen_ / 2 + 1, N = bf_ptr_->chanN();
  R.assign((size_t)2 * K * N * N, 0.f);
  const long lo = std::max(hist_start_, base_);                      // first frame of this block that belongs to the history
  if (t < lo && csd_carry_.empty()) return;
  const long Tn = t - base_ + 1;                                      // frames of the block up to t
  double tail = 1.0;                                                  // prod of a_sigma over sigma in (tau, t]
  if (Tn > 0) {
    std::vector<float> fw((size_t)Tn, 0.f);
    for (long tau = t; tau >= lo; tau--) {
      const double a_tau = (tau <= 1) ? 0.0 : alpha_;
      fw[(size_t)(tau - base_)] = (float)((1.0 - a_tau) * tail);
      tail *= a_tau;
      if (tail == 0.0) break;
    }
    void* dX = bf_ptr_->device_snapshots();
    const long Tstride = bf_ptr_->num_frames();
    void* dF = dev_alloc(sizeof(float) * Tn);
    void* dR = dev_alloc(sizeof(float) * R.size());
    h2d(dF, fw.data(), sizeof(float) * Tn);
    dev_zero_async(dR, sizeof(float) * R.size());
    check_abi(btk_cov_accumulate(dX, NULL, (const float*)dF, dR, 1, (int)K, (int)N, Tstride, Tn, 0, nstream()));
    nsync();
    d2h(R.data(), dR, sizeof(float) * R.size());
    dev_free(dF); dev_free(dR);
  }
  if (hist_start_ < base_ && csd_carry_.size() == R.size() && tail != 0.0)
    for (size_t i = 0; i < R.size(); i++) R[i] += (float)(tail * csd_carry_[i]);
}

void ZelinskiPostFilter::fill_csds_(gsl_vector_complex** out)
{
  const unsigned K = fftLen_ / 2 + 1;
  const bool manual = !has_bf_ptr_;
  BeamformerWeights* bw = manual ? own_weights_ : bf_ptr_->beamformer_weight_object();
  if (!bw) return;
  const unsigned N = bw->chanN();
  for (unsigned k = 0; k < fftLen_; k++) gsl_vector_complex_set_zero(out[k]);
  std::vector<float> R;
  if (manual) {                                                      // the recursion itself, kept frame by frame (next_manual_)
    if (csd_manual_.empty()) return;
    R.resize(csd_manual_.size() * 2);
    for (size_t i = 0; i < csd_manual_.size(); i++) { R[2 * i] = (float)csd_manual_[i].real(); R[2 * i + 1] = (float)csd_manual_[i].imag(); }
  } else {
    const long t = std::max<long>(frame_no_, handed_);              // last frame a per-frame graph has pulled
    if (!prepared_ || t < 0 || t < hist_start_) return;
    csd_state_(std::min(t, base_ + T_ - 1), R);
  }
  gsl_vector_complex** dvec = align_with_wq_() ? bw->wq() : bw->arrayManifold();
  for (unsigned k = 0; k < K; k++)
    for (unsigned i = 0; i < N; i++) {
      const cd di(dvec[k]->data[2 * i], dvec[k]->data[2 * i + 1]);
      for (unsigned j = i; j < N; j++) {
        const cd dj(dvec[k]->data[2 * j], dvec[k]->data[2 * j + 1]);
        const size_t o = 2 * (((size_t)k * N + i) * N + j);
        const cd v = std::conj(di) * dj * cd(R[o], R[o + 1]);
        out[k]->data[2 * (i * N + j)] = v.real();
        out[k]->data[2 * (i * N + j) + 1] = (i == j) ? 0.0 : v.imag();
      }
    }
}

// next() without a beamformer object (postfilter.cc:424-491 with snapshot_array_ / bf_weights_ given by the caller): the frame of
// samp_ is scaled by the gain the snapshot of this moment yields -- the two kernels of the block path on a one-frame block
const gsl_vector_complex* ZelinskiPostFilter::next_manual_(int frame_no)
{
  const gsl_vector_complex* output;
  if (frame_no >= 0) output = samp_->next(frame_no);
  else output = samp_->next(frame_no_ == frame_reset_no_ ? 0 : frame_no_ + 1);
  if (!own_weights_) throw j_error("set beamformer's weights \n");
  if (lefkimmiatis_or_mccowan_()) throw j_error("%s without a beamformer object is not supported: set_beamformer() first\n", name().c_str());
  if (snapshot_array_.is_null()) throw j_error("ZelinskiPostFilter: set_snapshot_array() first\n");
  if (own_weights_->isHalfBandShift()) throw j_error("post-filters with halfBandShift==true are not supported by this engine\n");
  const unsigned K = fftLen_ / 2 + 1, N = own_weights_->chanN();
  if (snapshot_array_->nChan() != N || snapshot_array_->fftLen() != fftLen_)
    throw jdimension_error("snapshot array is %d x %d, %d x %d expected\n", snapshot_array_->fftLen(), snapshot_array_->nChan(), fftLen_, N);
  gsl_vector_complex** dvec = (type_ & TYPE_ZELINSKI2) ? own_weights_->wq() : own_weights_->arrayManifold();
  std::vector<float> x((size_t)2 * K * N), d((size_t)2 * K * N), y((size_t)2 * K);
  for (unsigned k = 0; k < K; k++) {
    const gsl_vector_complex* sn = snapshot_array_->snapshot(k);
    for (unsigned c = 0; c < N; c++) {
      x[2 * ((size_t)k * N + c)] = (float)sn->data[2 * c]; x[2 * ((size_t)k * N + c) + 1] = (float)sn->data[2 * c + 1];
      d[2 * ((size_t)k * N + c)] = (float)dvec[k]->data[2 * c]; d[2 * ((size_t)k * N + c) + 1] = (float)dvec[k]->data[2 * c + 1];
    }
    y[2 * k] = (float)output->data[2 * k]; y[2 * k + 1] = (float)output->data[2 * k + 1];
  }
  {
    // the spectral densities BeamformerWeights::CSDs() reports (postfilter.cc:77-116), as the recursion itself on the raw
    // snapshots: R <- a R + (1 - a) x x^H, a = 0 for the first two frames (upper triangle; rotated by d when asked for)
    if (csd_manual_.size() != (size_t)K * N * N) csd_manual_.assign((size_t)K * N * N, cd(0, 0));
    const double a = (manual_frames_ <= 1) ? 0.0 : alpha_;
    for (unsigned k = 0; k < K; k++)
      for (unsigned i = 0; i < N; i++) {
        const cd xi(x[2 * ((size_t)k * N + i)], x[2 * ((size_t)k * N + i) + 1]);
        for (unsigned j = i; j < N; j++) {
          const cd xj(x[2 * ((size_t)k * N + j)], x[2 * ((size_t)k * N + j) + 1]);
          cd& r = csd_manual_[((size_t)k * N + i) * N + j];
          r = a * r + (1.0 - a) * xi * std::conj(xj);
        }
      }
  }
  if (!dPhi_) {
    dPhi_ = dev_alloc(sizeof(float) * 2 * K); dPsi_ = dev_alloc(sizeof(float) * K); dWl_ = dev_alloc(sizeof(float) * K);
    dev_zero_async(dPhi_, sizeof(float) * 2 * K);
    dev_zero_async(dPsi_, sizeof(float) * K);
    dev_zero_async(dWl_, sizeof(float) * K);
  }
  void* dXf = dev_alloc(sizeof(float) * x.size());
  void* dD = dev_alloc(sizeof(float) * d.size());
  void* dY = dev_alloc(sizeof(float) * 2 * K);
  void* dC = dev_alloc(sizeof(float) * 2 * K);
  void* dE = dev_alloc(sizeof(float) * K);
  h2d(dXf, x.data(), sizeof(float) * x.size());
  h2d(dD, d.data(), sizeof(float) * d.size());
  check_abi(btk_bf_apply_stats(dD, dD, 0, dXf, dY, dC, (float*)dE, 1, (int)K, (int)N, 1, 1, nstream()));   // only C and E are used
  h2d(dY, y.data(), sizeof(float) * y.size());                                                        // the frame to filter is samp_'s
  check_abi(btk_zelinski_process(dY, dC, (const float*)dE, 1, (int)K, (int)N, 1, 1, alpha_, (int)type_, min_frames_,
                                 manual_frames_, dPhi_, (float*)dPsi_, (float*)dWl_, nstream()));
  nsync();
  d2h(y.data(), dY, sizeof(float) * y.size());
  dev_free(dXf); dev_free(dD); dev_free(dY); dev_free(dC); dev_free(dE);
  for (unsigned k = 0; k < K; k++) {
    vector_->data[2 * k] = y[2 * k]; vector_->data[2 * k + 1] = y[2 * k + 1];
    if (k > 0 && k < fftLen_ / 2) { vector_->data[2 * (fftLen_ - k)] = y[2 * k]; vector_->data[2 * (fftLen_ - k) + 1] = -y[2 * k + 1]; }
  }
  manual_frames_++;
  increment_();
  return vector_;
}

void ZelinskiPostFilter::compute_(long from_frame)
{
  if (!has_bf_ptr_) throw j_error("set beamformer's weights \n");
  SubbandDS* bf = bf_ptr_.operator->();
  // the reference runs its post-filters over all fftLen bins of a half-band-shifted beamformer (postfilter.cc:170-182); this
  // engine's post-filter kernels work on the M/2+1 bins of a non-shifted bank: refuse instead of filtering wrongly
  if (bf->is_half_band_shift()) throw j_error("post-filters over a beamformer with halfBandShift==true are not supported by this engine\n");
  const unsigned N = bf->chanN(), K = fftLen_ / 2 + 1;
  void* dX = bf->device_snapshots();
  T_ = bf->num_frames();
  base_ = bf->chunk_base();
  std::vector<float> w, d;
  bf->effective_weights(w);
  bf->alignment_vector((type_ & TYPE_ZELINSKI2) != 0, d);
  const long Tn = T_ - from_frame;
  const bool carry = carry_state_ && dPhi_ && bf_version_ == bf->weights_version();
  carry_state_ = false;
  if (!dPhi_) { dPhi_ = dev_alloc(sizeof(float) * 2 * K); dPsi_ = dev_alloc(sizeof(float) * K); dWl_ = dev_alloc(sizeof(float) * K); }
  if (!carry) {
    // weights (re)computed: the CSD history restarts, the frame counter keeps counting; the following block
    // of the same stream instead continues from the densities the block before left on the device
    dev_zero_async(dPhi_, sizeof(float) * 2 * K);
    dev_zero_async(dPsi_, sizeof(float) * K);
    dev_zero_async(dWl_, sizeof(float) * K);
  }
  // node-owned blocks that only grow; the frames before from_frame keep what the pass before left in dYb_ (same block, same size)
  void* dY = dYb_.ensure(sizeof(float) * 2 * K * (T_ ? T_ : 1));
  if (Tn > 0) {
    void* dW = dWb_.ensure(sizeof(float) * w.size());
    void* dD = dDb_.ensure(sizeof(float) * d.size());
    void* dC = dCb_.ensure(sizeof(float) * 2 * K * T_);
    void* dE = dEb_.ensure(sizeof(float) * K * T_);
    h2d(dW, w.data(), sizeof(float) * w.size());
    h2d(dD, d.data(), sizeof(float) * d.size());
    // process frames [from_frame, T): offset the frame axis by pointer arithmetic, strides stay T_
    const float* Xo = static_cast<const float*>(dX) + 2 * from_frame;
    float* Yo = static_cast<float*>(dY) + 2 * from_frame;
    float* Co = static_cast<float*>(dC) + 2 * from_frame;
    float* Eo = static_cast<float*>(dE) + from_frame;
    check_abi(btk_bf_apply_stats(dW, dD, 0, Xo, Yo, Co, Eo, 1, (int)K, (int)N, T_, Tn, nstream()));
    check_abi(btk_zelinski_process(Yo, Co, Eo, 1, (int)K, (int)N, T_, Tn, alpha_, (int)type_, min_frames_, base_ + from_frame,
                                   dPhi_, (float*)dPsi_, (float*)dWl_, nstream()));
  }
  Yhost_valid_ = false;
  bf_version_ = bf->weights_version();
  if (!carry) { hist_start_ = base_ + from_frame; csd_carry_.clear(); }
  bind_csd_provider_();
  prepared_ = true;
}

// The block is used up: the densities on the device are those after its last frame (every block is processed to its end), and so
// is the CSD state kept for BeamformerWeights::CSDs(); the beamformer moves on to its next block of snapshots.
bool ZelinskiPostFilter::advance_chunk_()
{
  if (!has_bf_ptr_) return false;
  if (prepared_ && T_ > 0) {
    BeamformerWeights* bw = bf_ptr_->beamformer_weight_object();
    if (bw && base_ + T_ - 1 >= hist_start_) {
      std::vector<float> R;
      csd_state_(base_ + T_ - 1, R);
      csd_carry_.swap(R);
    }
  }
  if (!bf_ptr_->next_block()) return false;
  prepared_ = false; carry_state_ = true; Yhost_valid_ = false;
  return true;
}

// the filtered block as the host sees it (fetched when a frame or the host block is asked for)
const float* ZelinskiPostFilter::host_output_()
{
  if (!Yhost_valid_) {
    Yhost_.resize((size_t)2 * (fftLen_ / 2 + 1) * T_);
    d2h(Yhost_.data(), dYb_.get(), sizeof(float) * Yhost_.size());
    Yhost_valid_ = true;
  }
  return Yhost_.data();
}

const gsl_vector_complex* ZelinskiPostFilter::next(int frame_no)
{
  if (frame_no == frame_no_) return vector_;
  if (!has_bf_ptr_) return next_manual_(frame_no);
  if (!prepared_ || bf_version_ != bf_ptr_->weights_version()) compute_(kept_frames(frame_no_, handed_, bf_ptr_->chunk_base(), bf_ptr_->num_frames()));
  const long idx = frame_no_ + 1;
  while (idx >= base_ + T_) {
    if (!advance_chunk_()) { is_end_ = true; throw jiterator_error("end of samples!"); }
    compute_(0);
  }
  serve_frame(host_output_(), T_, fftLen_, idx - base_, vector_);
  increment_();
  return vector_;
}

const std::vector<float>& ZelinskiPostFilter::block(long& T)
{
  if (!has_bf_ptr_) throw j_error("set beamformer's weights \n");
  if (!prepared_ || bf_version_ != bf_ptr_->weights_version()) compute_(kept_frames(frame_no_, handed_, bf_ptr_->chunk_base(), bf_ptr_->num_frames()));
  host_output_();
  T = T_;
  return Yhost_;
}

const void* ZelinskiPostFilter::device_block(long& T, long& T_stride)
{
  if (!has_bf_ptr_) throw j_error("set beamformer's weights \n");
  if (!prepared_ || bf_version_ != bf_ptr_->weights_version()) compute_(kept_frames(frame_no_, handed_, bf_ptr_->chunk_base(), bf_ptr_->num_frames()));
  T = T_; T_stride = T_;
  return dYb_.get();
}

long ZelinskiPostFilter::block_base()
{
  long T;
  block(T);
  return base_;
}

bool ZelinskiPostFilter::next_block()
{
  if (!advance_chunk_()) return false;
  compute_(0);
  return true;
}

void ZelinskiPostFilter::advance_to(long frame_idx)
{
  if (frame_idx >= base_ + T_) frame_idx = base_ + T_ - 1;
  if (frame_idx > handed_) handed_ = frame_idx;
  if (has_bf_ptr_) bf_ptr_->advance_to(frame_idx);
}

const gsl_vector_complex* ZelinskiPostFilter::postfilter_weights()
{
  if (!dWl_) return NULL;
  const unsigned K = fftLen_ / 2 + 1;
  std::vector<float> wl(K);
  d2h(wl.data(), dWl_, sizeof(float) * K);
  for (unsigned k = 0; k < K; k++) {
    wp1_->data[2 * k] = wl[k]; wp1_->data[2 * k + 1] = 0.0;
    if (k > 0 && k < fftLen_ / 2) { wp1_->data[2 * (fftLen_ - k)] = wl[k]; wp1_->data[2 * (fftLen_ - k) + 1] = 0.0; }
  }
  // the reference keeps these gains in the beamformer's weight object (BeamformerWeights::wp1(), postfilter.cc:447): mirror them
  if (weights_object() && weights_object()->fftLen() == fftLen_)
    memcpy(weights_object()->wp1()->data, wp1_->data, sizeof(double) * 2 * fftLen_);
  return wp1_;
}

void ZelinskiPostFilter::reset()
{
  samp_->reset();
  VectorComplexFeatureStream::reset();
  is_end_ = false;
  prepared_ = false; Yhost_.clear(); Yhost_valid_ = false; T_ = 0; base_ = 0; carry_state_ = false; csd_carry_.clear();
  csd_manual_.clear(); manual_frames_ = 0; hist_start_ = 0; handed_ = -1;
  if (!has_bf_ptr_ && dPhi_) {              // manual mode: the densities of the next utterance start from zero
    const unsigned K = fftLen_ / 2 + 1;
    dev_zero_async(dPhi_, sizeof(float) * 2 * K);
    dev_zero_async(dPsi_, sizeof(float) * K);
    dev_zero_async(dWl_, sizeof(float) * K);
  }
}

// ================================================================================ McCowan / Lefkimmiatis
McCowanPostFilter::McCowanPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha, int type,
                                     int minFrames, float threshold, const String& nm)
    : ZelinskiPostFilter(output, fftLen, alpha, type, minFrames, nm), threshold_of_Rij_(threshold), nChanR_(0), dR_(NULL),
      Rview_(NULL), invR_computed_(false), minSV_(1.0e-8), fbinX1_(0), dU_(NULL), dV_(NULL), svd_rule_(default_svd_rule()),
      lam_valid_(false), lam_version_(0) {}

void McCowanPostFilter::set_svd_rule(const String& rule)
{
  if (rule != "linpack" && rule != "exact") throw jparameter_error("svd rule must be linpack or exact, got %s\n", rule.c_str());
  if (rule != svd_rule_) { svd_rule_ = rule; prepared_ = false; }
}

McCowanPostFilter::~McCowanPostFilter()
{
  dev_free(dR_); dev_free(dU_); dev_free(dV_);
  if (Rview_) gsl_matrix_complex_free(Rview_);
}

void McCowanPostFilter::fetch_R_()
{
  const unsigned K = fftLen_ / 2 + 1;
  Rhost_.resize((size_t)2 * K * nChanR_ * nChanR_);
  if (dR_) d2h(Rhost_.data(), dR_, sizeof(float) * Rhost_.size());
}
void McCowanPostFilter::push_R_()
{
  h2d(dR_, Rhost_.data(), sizeof(float) * Rhost_.size());
  invR_computed_ = false; lam_valid_ = false;
  prepared_ = false;
}

const gsl_matrix_complex* McCowanPostFilter::noise_spatial_spectral_matrix(unsigned fbinX)
{
  if (!dR_) return NULL;
  if (fbinX > fftLen_ / 2) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)(fftLen_ / 2));
  fetch_R_();
  const unsigned N = nChanR_;
  if (Rview_ && Rview_->size1 != N) { gsl_matrix_complex_free(Rview_); Rview_ = NULL; }
  if (!Rview_) Rview_ = gsl_matrix_complex_alloc(N, N);
  for (size_t e = 0; e < (size_t)2 * N * N; e++) Rview_->data[e] = Rhost_[(size_t)2 * fbinX * N * N + e];
  return Rview_;
}

bool McCowanPostFilter::set_noise_spatial_spectral_matrix(unsigned fbinX, gsl_matrix_complex* Rnn)
{
  if (Rnn->size1 != Rnn->size2) { fprintf(stderr, "The noise coherence matrix should be the square matrix\n"); return false; }
  if (fbinX > fftLen_ / 2) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)(fftLen_ / 2));
  const unsigned K = fftLen_ / 2 + 1, N = (unsigned)Rnn->size1;
  if (!dR_ || nChanR_ != N) {
    dev_free(dR_);
    nChanR_ = N;
    dR_ = dev_alloc(sizeof(float) * 2 * K * N * N);
    dev_zero_async(dR_, sizeof(float) * 2 * K * N * N);
  }
  fetch_R_();
  for (unsigned a = 0; a < N; a++)
    for (unsigned b = 0; b < N; b++) {
      const gsl_complex z = gsl_matrix_complex_get(Rnn, a, b);
      Rhost_[2 * (((size_t)fbinX * N + a) * N + b)] = (float)GSL_REAL(z);
      Rhost_[2 * (((size_t)fbinX * N + a) * N + b) + 1] = (float)GSL_IMAG(z);
    }
  push_R_();
  return true;
}

bool McCowanPostFilter::set_diffuse_noise_model(const gsl_matrix* micPositions, double sampleRate, double sspeed)
{
  const unsigned K = fftLen_ / 2 + 1, N = (unsigned)micPositions->size1;
  if (micPositions->size2 < 3) { fprintf(stderr, "The microphone positions should be described in the three dimensions\n"); return false; }
  if (!dR_ || nChanR_ != N) { dev_free(dR_); nChanR_ = N; dR_ = dev_alloc(sizeof(float) * 2 * K * N * N); }
  std::vector<float> mp((size_t)3 * N);
  for (unsigned a = 0; a < N; a++) for (int j = 0; j < 3; j++) mp[3 * a + j] = (float)gsl_matrix_get(micPositions, a, j);
  void* dmp = dev_alloc(sizeof(float) * mp.size());
  h2d(dmp, mp.data(), sizeof(float) * mp.size());
  check_abi(btk_mvdr_diffuse_model((const float*)dmp, (int)N, (int)fftLen_, (float)sampleRate, (float)sspeed, dR_, nstream()));
  nsync();
  dev_free(dmp);
  invR_computed_ = false; lam_valid_ = false;
  prepared_ = false;
  return true;
}

void McCowanPostFilter::set_all_diagonal_loading(float diagonalWeight)
{
  if (!dR_) throw j_error("Construct/set first a noise coherence matrix\n");
  check_abi(btk_mvdr_diagonal_loading(dR_, (int)(fftLen_ / 2 + 1), (int)nChanR_, diagonalWeight, nstream()));
  invR_computed_ = false; lam_valid_ = false;
  prepared_ = false;
}

void McCowanPostFilter::set_diagonal_looading(unsigned fbinX, float diagonalWeight)
{
  if (!dR_) throw j_error("Construct/set first a noise coherence matrix\n");
  if (fbinX > fftLen_ / 2) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)(fftLen_ / 2));
  check_abi(btk_mvdr_diagonal_loading(static_cast<float*>(dR_) + (size_t)2 * fbinX * nChanR_ * nChanR_, 1, (int)nChanR_,
                                      diagonalWeight, nstream()));
  invR_computed_ = false; lam_valid_ = false;
  prepared_ = false;
}

void McCowanPostFilter::divide_nondiagonal_elements(unsigned fbinX, float mu)
{
  if (!dR_) throw j_error("Construct/set first a noise coherence matrix\n");
  if (fbinX > fftLen_ / 2) throw jindex_error("frequency bin %d is out of range: the matrices exist for bins 0..%d\n", (int)fbinX, (int)(fftLen_ / 2));
  fetch_R_();
  const unsigned N = nChanR_;
  for (unsigned a = 0; a < N; a++)
    for (unsigned b = 0; b < N; b++)
      if (a != b) {
        float* z = &Rhost_[2 * (((size_t)fbinX * N + a) * N + b)];
        z[0] = (float)((double)z[0] / (1.0 + mu));
        z[1] = (float)((double)z[1] / (1.0 + mu));
      }
  push_R_();
}

void McCowanPostFilter::divide_all_nondiagonal_elements(float mu)
{
  for (unsigned k = 0; k <= fftLen_ / 2; k++) divide_nondiagonal_elements(k, mu);
}

void McCowanPostFilter::compute_(long from_frame)
{
  if (!has_bf_ptr_) throw j_error("set beamformer's weights \n");
  if (!dR_) throw j_error("%s", no_R_msg_());
  SubbandDS* bf = bf_ptr_.operator->();
  const unsigned N = bf->chanN(), K = fftLen_ / 2 + 1;
  if (N != nChanR_) throw jdimension_error("The noise coherence matrix is %dx%d but there are %d channels\n", nChanR_, nChanR_, N);
  const bool lef = lefkimmiatis_();
  void* dX = bf->device_snapshots();
  T_ = bf->num_frames();
  base_ = bf->chunk_base();
  std::vector<float> w, d;
  bf->effective_weights(w);
  bf->alignment_vector(!lef && (type_ & TYPE_ZELINSKI2) != 0, d);              // postfilter.cc:858-863 vs :1098
  const long Tn = T_ - from_frame;
  const bool carry = carry_state_ && dPhi_ && dU_ && bf_version_ == bf->weights_version();
  carry_state_ = false;
  if (!dPhi_) { dPhi_ = dev_alloc(sizeof(float) * 2 * K); dPsi_ = dev_alloc(sizeof(float) * K); dWl_ = dev_alloc(sizeof(float) * K); }
  if (!dU_) { dU_ = dev_alloc(sizeof(float) * 2 * K); dV_ = dev_alloc(sizeof(float) * 2 * K); }
  if (!carry) {                                                                // (the next block of a stream continues the recursions)
    dev_zero_async(dPhi_, sizeof(float) * 2 * K);
    dev_zero_async(dPsi_, sizeof(float) * K);
    dev_zero_async(dWl_, sizeof(float) * K);
    dev_zero_async(dV_, sizeof(float) * 2 * K);
  }
  void* dY = dYb_.ensure(sizeof(float) * 2 * K * (T_ ? T_ : 1));
  if (Tn > 0) {
    void* dW = dWb_.ensure(sizeof(float) * w.size());
    void* dD = dDb_.ensure(sizeof(float) * d.size());
    void* dUs = dCb_.ensure(sizeof(float) * 2 * K * T_);
    void* dVs = lef ? dVsb_.ensure(sizeof(float) * 2 * K * T_) : NULL;
    void* dE = dEb_.ensure(sizeof(float) * K * T_);
    void* dCs = dCsb_.ensure(sizeof(float) * 2 * K * N * N);
    void* dCv = lef ? dCvb_.ensure(sizeof(float) * 2 * K * N * N) : NULL;
    h2d(dW, w.data(), sizeof(float) * w.size());
    h2d(dD, d.data(), sizeof(float) * d.size());
    check_abi(btk_pf_coherence_coeffs(dR_, threshold_of_Rij_, (int)K, (int)N, dCs, dCv, nstream()));
    const float* Xo = static_cast<const float*>(dX) + 2 * from_frame;
    float* Yo = static_cast<float*>(dY) + 2 * from_frame;
    float* Uo = static_cast<float*>(dUs) + 2 * from_frame;
    float* Vo = lef ? static_cast<float*>(dVs) + 2 * from_frame : NULL;
    float* Eo = static_cast<float*>(dE) + from_frame;
    check_abi(btk_bf_apply_stats2(dW, dD, 0, Xo, Yo, dCs, dCv, Uo, Vo, Eo, 1, (int)K, (int)N, T_, Tn, nstream()));
    if (lef) {
      // Lambda = d^H pinv(R) d of every bin (:967-995) depends on the coherence matrix, the look direction and the SVD rule only:
      // designed once and kept until one of them changes (the csvdc rule alone takes ~0.1 s at 256 channels -- per block it
      // would dominate a stream)
      if (!lam_valid_ || lam_version_ != bf->weights_version() || lam_rule_ != svd_rule_) {
        void* dLam = dLamb_.ensure(sizeof(float) * 2 * K);
        void* dFb = dev_alloc(sizeof(int));
        dev_zero_async(dFb, sizeof(int));
        const long sbytes = btk_mvdr_scratch_bytes((int)K, (int)N);
        void* scratch = sbytes ? dev_alloc((size_t)sbytes) : NULL;
        check_abi(btk_mvdr_lambda(dR_, dD, dLam, (int)K, (int)N, (float)minSV_, scratch, (int*)dFb, nstream()));
        if (svd_rule_ == "linpack") {                      // pseudoinverse() false -> identity, every bin incl. 0 (:971-977)
          void* rs = dev_alloc((size_t)btk_mvdr_linpack_rule_scratch_bytes((int)K, (int)N));
          check_abi(btk_mvdr_linpack_rule(dR_, dD, NULL, dLam, (int)K, (int)N, 0, 0, 0, (float)minSV_, NULL, NULL, rs, nstream()));
          nsync();
          dev_free(rs);
        }
        nsync();
        dev_free(dFb); dev_free(scratch);
        lam_valid_ = true; lam_version_ = bf->weights_version(); lam_rule_ = svd_rule_;
      }
      invR_computed_ = true;
      check_abi(btk_lefkimmiatis_process(Yo, Uo, Vo, dLamb_.get(), (int)fbinX1_, 1, (int)K, (int)N, T_, Tn, alpha_, (int)type_,
                                         min_frames_, base_ + from_frame, dPhi_, dV_, (float*)dWl_, nstream()));
    } else {
      check_abi(btk_zelinski_process(Yo, Uo, Eo, 1, (int)K, (int)N, T_, Tn, alpha_, (int)type_ & 3, min_frames_, base_ + from_frame,
                                     dPhi_, (float*)dPsi_, (float*)dWl_, nstream()));
    }
  }
  Yhost_valid_ = false;
  bf_version_ = bf->weights_version();
  if (!carry) { hist_start_ = base_ + from_frame; csd_carry_.clear(); }
  bind_csd_provider_();
  prepared_ = true;
}

LefkimmiatisPostFilter::LefkimmiatisPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double minSV,
                                               unsigned fbinX1, double alpha, int type, int minFrames, float threshold,
                                               const String& nm)
    : McCowanPostFilter(output, fftLen, alpha, type, minFrames, threshold, nm)
{
  minSV_ = minSV;
  fbinX1_ = fbinX1;
}

void LefkimmiatisPostFilter::calc_inverse_noise_spatial_spectral_matrix()
{
  // calcLambda only needs d^H pinv(R) d, formed with the look direction when the block is computed
  if (!dR_) throw j_error("%s", no_R_msg_());
  prepared_ = false; lam_valid_ = false;
}

// ================================================================================ SubbandGSCRLS
SubbandGSCRLS::SubbandGSCRLS(unsigned fftLen, bool halfBandShift, float mu, float sigma2, const String& nm)
    : SubbandGSC(fftLen, halfBandShift, nm), mu_(mu), diagonal_weight_(sigma2), alpha_(-1.0f), qctype_(NO_QUADRATIC_CONSTRAINT),
      rls_version_(0), is_wa_updated_(true), have_P_(false), dP_(NULL), dW_(NULL), dV_(NULL), dSS_(NULL), dCx_(NULL), dP0_(NULL),
      dW0_(NULL), dSS0_(NULL), uploaded_version_(0), block_ran_(false) {}

SubbandGSCRLS::~SubbandGSCRLS() { dev_free(dP_); dev_free(dW_); dev_free(dV_); dev_free(dSS_); dev_free(dCx_); dev_free(dP0_); dev_free(dW0_); dev_free(dSS0_); }

// quiescent weights and the further blocked directions as the kernels want them, from the weight object as it is NOW
void SubbandGSCRLS::upload_weights_()
{
  const unsigned N = chanN(), K = fftLen2_ + 1, NC = bfweight_->NC();
  dev_free(dCx_); dCx_ = NULL;
  if (NC > 1) {
    // the blocking matrix of calc_gsc_weights_2 / _n keeps N - NC columns: B B^H = I - conj(wq) wq^T / |wq|^2 - sum_j c_j c_j^H;
    // btk_nlms_constraint_vectors returns the directions of conj(B) B^T, i.e. the complex conjugates (include/btkhip.h)
    const unsigned bs = N - NC;
    std::vector<cd> cx((size_t)K * (NC - 1) * N);
    for (unsigned k = 0; k < K; k++) {
      check_abi(btk_nlms_constraint_vectors(reinterpret_cast<const double*>(&bfweight_->wq_v[(size_t)k * N]),
                                            reinterpret_cast<const double*>(&bfweight_->B_v[(size_t)k * N * bs]), (int)N, (int)NC,
                                            reinterpret_cast<double*>(&cx[(size_t)k * (NC - 1) * N])));
    }
    for (size_t i = 0; i < cx.size(); i++) cx[i] = std::conj(cx[i]);
    dCx_ = dev_alloc(sizeof(double) * 2 * cx.size());
    h2d(dCx_, cx.data(), sizeof(double) * 2 * cx.size());
  }
  if (!dV_) dV_ = dev_alloc(sizeof(double) * 2 * K * N);
  h2d(dV_, bfweight_->wq_v.data(), sizeof(double) * 2 * K * N);                 // bins 0..M/2 of wq [M][N]
  wq_uploaded_.assign(bfweight_->wq_v.begin(), bfweight_->wq_v.begin() + (size_t)K * N);
  uploaded_version_ = weights_version_;
}

// New quiescent weights / blocking matrix while the recursion state is kept: the reference's state lives in the space of the
// active weights (Pz_ is (N - NC) x (N - NC), wa has N - NC entries: beamformer.cc:1480-1515) and simply meets the new blocking
// matrix B'; this engine keeps P = B Pz B^H and wl = B wa in channel space, so both change basis: with T = B' B^H (B has
// orthonormal columns) P <- T P T^H, wl <- T wl.  Host float64, bins 1 .. M/2; a rare event (the look direction of an RLS
// canceller moving between blocks).
void SubbandGSCRLS::change_basis_(void* dP, void* dW)
{
  const unsigned N = chanN(), K = fftLen2_ + 1, NC = bfweight_->NC(), bs = N - NC;
  std::vector<cd> P((size_t)K * N * N), W((size_t)K * N), B1((size_t)N * bs), Tm((size_t)N * N), TP((size_t)N * N);
  d2h(P.data(), dP, sizeof(double) * 2 * P.size());
  d2h(W.data(), dW, sizeof(double) * 2 * W.size());
  for (unsigned k = 1; k < K; k++) {
    check_abi(btk_weights_blocking_matrix(reinterpret_cast<const double*>(&wq_uploaded_[(size_t)k * N]), (int)N, (int)NC,
                                          reinterpret_cast<double*>(B1.data())));
    const cd* B2 = &bfweight_->B_v[(size_t)k * N * bs];
    for (unsigned a = 0; a < N; a++)
      for (unsigned b = 0; b < N; b++) {
        cd acc(0, 0);
        for (unsigned j = 0; j < bs; j++) acc += B2[(size_t)a * bs + j] * std::conj(B1[(size_t)b * bs + j]);
        Tm[(size_t)a * N + b] = acc;
      }
    cd* Pk = &P[(size_t)k * N * N];
    for (unsigned a = 0; a < N; a++)
      for (unsigned b = 0; b < N; b++) {
        cd acc(0, 0);
        for (unsigned c = 0; c < N; c++) acc += Tm[(size_t)a * N + c] * Pk[(size_t)c * N + b];
        TP[(size_t)a * N + b] = acc;
      }
    for (unsigned a = 0; a < N; a++)
      for (unsigned b = 0; b < N; b++) {
        cd acc(0, 0);
        for (unsigned c = 0; c < N; c++) acc += TP[(size_t)a * N + c] * std::conj(Tm[(size_t)b * N + c]);
        Pk[(size_t)a * N + b] = acc;
      }
    std::vector<cd> w(N);
    for (unsigned a = 0; a < N; a++) {
      cd acc(0, 0);
      for (unsigned c = 0; c < N; c++) acc += Tm[(size_t)a * N + c] * W[(size_t)k * N + c];
      w[a] = acc;
    }
    for (unsigned a = 0; a < N; a++) W[(size_t)k * N + a] = w[a];
  }
  h2d(dP, P.data(), sizeof(double) * 2 * P.size());
  h2d(dW, W.data(), sizeof(double) * 2 * W.size());
}

void SubbandGSCRLS::alloc_state_()
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  const unsigned N = chanN(), K = fftLen2_ + 1;
  dev_free(dP_); dev_free(dW_); dev_free(dV_); dV_ = NULL; dev_free(dSS_); dev_free(dP0_); dev_free(dW0_); dev_free(dSS0_);
  dP_ = dev_alloc(sizeof(double) * 2 * K * N * N);
  dW_ = dev_alloc(sizeof(double) * 2 * K * N);
  dSS_ = dev_alloc(sizeof(double) * 4);
  dP0_ = dev_alloc(sizeof(double) * 2 * K * N * N);
  dW0_ = dev_alloc(sizeof(double) * 2 * K * N);
  dSS0_ = dev_alloc(sizeof(double) * 4);
  dev_zero_async(dSS_, sizeof(double) * 4);
  upload_weights_();
}

void SubbandGSCRLS::init_precision_matrix(float sigma2)
{
  alloc_state_();
  const unsigned N = chanN(), K = fftLen2_ + 1;
  const float p0 = 1 / sigma2;                                                // float division, beamformer.cc:1491
  check_abi(btk_rls_init_nc(0, dV_, 0, dCx_, (int)bfweight_->NC(), (double)p0, 1, (int)K, (int)N, dP_, dW_, nstream()));
  // the active weights kept in the weight object are the starting point (zeros after calc_gsc_weights)
  h2d(dW_, bfweight_->wl_v.data(), sizeof(double) * 2 * K * N);
  have_P_ = true;
  out_valid_ = false; Yhost_valid_ = false; block_ran_ = false;
}

void SubbandGSCRLS::set_precision_matrix(unsigned fbinX, gsl_matrix_complex* Pz)
{
  if (!have_P_) {
    alloc_state_();
    const unsigned N = chanN(), K = fftLen2_ + 1;
    dev_zero_async(dP_, sizeof(double) * 2 * K * N * N);
    h2d(dW_, bfweight_->wl_v.data(), sizeof(double) * 2 * K * N);
    have_P_ = true;
  }
  const unsigned N = chanN(), bs = N - bfweight_->NC();
  if (fbinX > fftLen2_) return;                                               // only bins 1..M/2 are ever used
  if (Pz->size1 < bs || Pz->size2 < bs) throw jdimension_error("the precision matrix must be at least %dx%d\n", bs, bs);
  // engine basis: P = B Pz B^H
  const cd* B = &bfweight_->B_v[(size_t)fbinX * N * bs];
  std::vector<cd> T((size_t)N * bs), P((size_t)N * N);
  for (unsigned a = 0; a < N; a++)
    for (unsigned j = 0; j < bs; j++) {
      cd acc(0, 0);
      for (unsigned i = 0; i < bs; i++) {
        const gsl_complex z = gsl_matrix_complex_get(Pz, i, j);
        acc += B[(size_t)a * bs + i] * cd(GSL_REAL(z), GSL_IMAG(z));
      }
      T[(size_t)a * bs + j] = acc;
    }
  for (unsigned a = 0; a < N; a++)
    for (unsigned b = 0; b < N; b++) {
      cd acc(0, 0);
      for (unsigned j = 0; j < bs; j++) acc += T[(size_t)a * bs + j] * std::conj(B[(size_t)b * bs + j]);
      P[(size_t)a * N + b] = acc;
    }
  h2d(static_cast<double*>(dP_) + (size_t)2 * fbinX * N * N, P.data(), sizeof(double) * 2 * N * N);
  out_valid_ = false; Yhost_valid_ = false; block_ran_ = false;
}

// The recursion over the current block of snapshots, from the state the block before left (P, w_a and the stream counters move
// to this block's end).  The state at the block's start is kept so that the block can be run again (refresh_block_).
void SubbandGSCRLS::run_block_()
{
  const unsigned N = chanN(), K = fftLen2_ + 1;
  void* dX = snapshots_();
  d2d_async(dP0_, dP_, sizeof(double) * 2 * K * N * N);
  d2d_async(dW0_, dW_, sizeof(double) * 2 * K * N);
  d2d_async(dSS0_, dSS_, sizeof(double) * 4);
  void* dY = dYBuf_.ensure(sizeof(float) * 2 * K * (T_ ? T_ : 1));          // the block's output lives where the static apply's would
  const double params[6] = { (double)mu_, (double)diagonal_weight_, (double)(int)qctype_, (double)alpha_,
                             normalize_weight_ ? 1.0 : 0.0, is_wa_updated_ ? 1.0 : 0.0 };
  void* ws = dWs_.ensure((size_t)btk_rls_workspace_bytes(1, T_ ? T_ : 1));
  check_abi(btk_rls_process_nc(0, params, dV_, 0, dCx_, (int)bfweight_->NC(), dX, dY, 1, (int)fftLen_, (int)N, T_, T_, dP_, dW_, (double*)dSS_, ws, nstream()));
  out_valid_ = true; Yhost_valid_ = false;
  // export wl / wa of bins 1..M/2 as calcSidelobeCancellerU_f leaves them (beamformer.cc:1643)
  std::vector<cd> wl((size_t)K * N);
  d2h(wl.data(), dW_, sizeof(double) * 2 * K * N);
  const unsigned bs = N - bfweight_->NC();
  for (unsigned k = 1; k < K; k++) {
    const cd* B = &bfweight_->B_v[(size_t)k * N * bs];
    for (unsigned c = 0; c < N; c++) bfweight_->wl_v[(size_t)k * N + c] = wl[(size_t)k * N + c];
    for (unsigned i = 0; i < bs; i++) {
      cd acc(0, 0);
      for (unsigned c = 0; c < N; c++) acc += std::conj(B[(size_t)c * bs + i]) * wl[(size_t)k * N + c];
      bfweight_->wa_v[(size_t)k * bs + i] = acc;
    }
  }
  output_version_ = weights_version_;
  block_ran_ = true;
}

const std::vector<float>& SubbandGSCRLS::block(long& T)
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  if (!have_P_) throw j_error("set the precision matrix with init_precision_matrix() or set_precision_matrix()\n");
  if (halfBandShift_) throw j_error("halfBandShift==true is not yet supported\n");
  ensure_chunk_();
  refresh_block_();
  rls_host_output_();
  T = T_;
  return Yhost_;
}

const void* SubbandGSCRLS::device_block(long& T, long& T_stride)
{
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  if (!have_P_) throw j_error("set the precision matrix with init_precision_matrix() or set_precision_matrix()\n");
  if (halfBandShift_) throw j_error("halfBandShift==true is not yet supported\n");
  ensure_chunk_();
  refresh_block_();
  T = T_; T_stride = T_;
  return dYBuf_.get();
}

// the block of the recursion mirrored on the host (SubbandDS::host_output_ would run the static apply)
const float* SubbandGSCRLS::rls_host_output_()
{
  if (!Yhost_valid_) {
    Yhost_.assign((size_t)2 * (fftLen2_ + 1) * T_, 0.f);
    if (T_) d2h(Yhost_.data(), dYBuf_.get(), sizeof(float) * Yhost_.size());
    Yhost_valid_ = true;
  }
  return Yhost_.data();
}

bool SubbandGSCRLS::advance_chunk_()
{
  out_valid_ = false; Yhost_valid_ = false; block_ran_ = false;
  return next_chunk();
}

// The recursion over the current block ran with the weights of that moment, and P / w_a have moved to the block's end.  New
// quiescent weights or a new blocking matrix before any frame OF THIS BLOCK was served: the state at the block's start is put
// back, the new wq / blocked directions are uploaded and the block runs again.  After frames of the block were served the
// per-frame meaning (the recursion continuing from frame t with the new B) would need the state of frame t, which this engine
// does not keep: refuse instead of handing the stale block over
void SubbandGSCRLS::refresh_block_()
{
  if (block_ran_ && rls_version_ == weights_version_) return;
  if (block_ran_) {
    if (kept_frames(frame_no_, handed_, chunk_base_, T_) > 0)
      throw jconsistency_error("SubbandGSCRLS: the weights changed after frames of this block were served; reset() first\n");
    const unsigned N = chanN(), K = fftLen2_ + 1;
    d2d_async(dP_, dP0_, sizeof(double) * 2 * K * N * N);
    d2d_async(dW_, dW0_, sizeof(double) * 2 * K * N);
    d2d_async(dSS_, dSS0_, sizeof(double) * 4);
  }
  if (uploaded_version_ != weights_version_) {
    const bool had_state = !wq_uploaded_.empty();
    if (had_state) change_basis_(dP_, dW_);                 // (uses the wq the state was built with, then ...)
    upload_weights_();                                      // ... the new wq / blocked directions go to the device
  }
  run_block_();
  rls_version_ = weights_version_;
}

const gsl_vector_complex* SubbandGSCRLS::next(int frame_no)
{
  if (frame_no == frame_no_) return vector_;
  if (!bfweight_) throw j_error("call calc_gsc_weights_x() once\n");
  if (!have_P_) throw j_error("set the precision matrix with init_precision_matrix() or set_precision_matrix()\n");
  if (halfBandShift_) throw j_error("halfBandShift==true is not yet supported\n");          // reference beamformer.cc:1528-1530
  ensure_chunk_();
  const long idx = frame_no_ + 1;
  while (idx >= chunk_base_ + T_)
    if (!advance_chunk_()) { is_end_ = true; throw jiterator_error("end of samples!"); }
  refresh_block_();
  serve_frame(rls_host_output_(), T_, fftLen_, idx - chunk_base_, vector_);
  increment_();
  return vector_;
}


// ================================================================================ WPE dereverberation
// (reference dereverberation/dereverberation.cc:40-307, 312-760)
namespace {
unsigned wpe_band_width(unsigned M, double bandWidth, double sampleRate)
{
  if (bandWidth == 0.0) return M / 2;                                      // set_band_width_ (:361-369)
  if (bandWidth > sampleRate / 2.0) throw jdimension_error("Bandwidth is greater than the Nyquist rate.\n");
  return (unsigned)((bandWidth / (sampleRate / 2.0)) * (M / 2));
}
}  // namespace

MultiChannelWPEDereverberation::MultiChannelWPEDereverberation(unsigned subbandsN, unsigned channelsN, unsigned lowerN,
                                                               unsigned upperN, unsigned iterationsN, double loadDb,
                                                               double bandWidth, double diagonal_bias, double sampleRate)
    : subbandsN_(subbandsN), channelsN_(channelsN), lowerN_(lowerN), upperN_(upperN), iterationsN_(iterationsN),
      load_db_(loadDb), diagonal_bias_(diagonal_bias), lower_bw_(wpe_band_width(subbandsN, bandWidth, sampleRate)),
      upper_bw_(subbandsN - lower_bw_), estimated_(false), framesN_(0), dG_(NULL), T_(0), have_out_(false),
      output_(new gsl_vector_complex*[channelsN]), frame_no_(-1)
{
  for (unsigned c = 0; c < channelsN_; c++) output_[c] = gsl_vector_complex_calloc(subbandsN_);
}

MultiChannelWPEDereverberation::~MultiChannelWPEDereverberation()
{
  for (unsigned c = 0; c < channelsN_; c++) gsl_vector_complex_free(output_[c]);
  delete[] output_;
  dev_free(dG_);
}

void MultiChannelWPEDereverberation::set_input(VectorComplexFeatureStreamPtr& samples)
{
  if (sources_.size() == channelsN_) throw jallocation_error("Channel capacity exceeded.");
  sources_.push_back(samples);
}

void MultiChannelWPEDereverberation::reset()
{
  frame_no_ = -1;
  for (size_t c = 0; c < sources_.size(); c++) sources_[c]->reset();
  have_out_ = false;
  out_.clear();
}

void MultiChannelWPEDereverberation::reset_filter() { estimated_ = false; framesN_ = 0; }

void MultiChannelWPEDereverberation::next_speaker()
{
  reset();
  if (dG_) {
    const unsigned K = subbandsN_ / 2 + 1, P = channelsN_ * (upperN_ - lowerN_ + 1);
    dev_zero_async(dG_, sizeof(float) * 2 * channelsN_ * K * P);
  }
}

long MultiChannelWPEDereverberation::snapshots_(void** dX)
{
  const unsigned C = channelsN_, K = subbandsN_ / 2 + 1;
  if (sources_.size() != C) throw jallocation_error("%u of %u input channels are set\n", (unsigned)sources_.size(), C);
  std::vector<std::vector<float> > fr(C);
  long T = -1;
  for (unsigned c = 0; c < C; c++) {
    const long t = drain_complex(sources_[c], subbandsN_, fr[c]);
    if (T < 0 || t < T) T = t;
  }
  std::vector<float> Xh((size_t)2 * K * C * (T > 0 ? T : 0));
  for (unsigned k = 0; k < K; k++)
    for (unsigned c = 0; c < C; c++)
      for (long t = 0; t < T; t++) {
        Xh[2 * (((size_t)k * C + c) * T + t)] = fr[c][2 * ((size_t)t * K + k)];
        Xh[2 * (((size_t)k * C + c) * T + t) + 1] = fr[c][2 * ((size_t)t * K + k) + 1];
      }
  *dX = dev_alloc(sizeof(float) * Xh.size());
  if (!Xh.empty()) h2d(*dX, Xh.data(), sizeof(float) * Xh.size());
  return T;
}

// fill_buffer_ (:506-529) counts frX from 0 and pulls one frame per frX in [start, end) from the inputs' current
// position: the estimate sees the FIRST end - start frames of the inputs (all of them when end < 0).
unsigned MultiChannelWPEDereverberation::estimate_filter(int start_frame_no, int end_frame_no)
{
  const unsigned C = channelsN_, K = subbandsN_ / 2 + 1, P = C * (upperN_ - lowerN_ + 1);
  void* dX = NULL;
  const long T = snapshots_(&dX);
  long n = T;
  if (end_frame_no >= 0) {
    n = (long)end_frame_no - (start_frame_no > 0 ? start_frame_no : 0);
    n = n < 0 ? 0 : (n > T ? T : n);
  }
  if (!dG_) {
    dG_ = dev_alloc(sizeof(float) * 2 * C * K * P);
    dev_zero_async(dG_, sizeof(float) * 2 * C * K * P);
  }
  const long wsb = btk_wpe_workspace_bytes(1, (int)K, (int)C, (int)lowerN_, (int)upperN_, T > 0 ? T : 1);
  void* ws = dev_alloc((size_t)(wsb > 0 ? wsb : 16));
  void* dfail = dev_alloc(sizeof(int));
  dev_zero_async(dfail, sizeof(int));
  int fail = 0;
  try {
    check_abi(btk_wpe_estimate(dX, 1, (int)K, (int)C, T > 0 ? T : 1, n, (int)lowerN_, (int)upperN_, (int)iterationsN_, load_db_,
                               diagonal_bias_, (int)lower_bw_, (int)upper_bw_, dG_, ws, (int*)dfail, nstream()));
    nsync();
    d2h(&fail, dfail, sizeof(int));
  } catch (...) {
    dev_free(dX); dev_free(ws); dev_free(dfail);
    throw;
  }
  dev_free(dX); dev_free(ws); dev_free(dfail);
  if (fail > 0)
    throw jnumeric_error("MultiChannelWPEDereverberation: Cholesky decomposition failed (%d systems).\n"
                         "Some channels may be too similar. Try to increase 'diagonal_bias'", fail);
  for (size_t c = 0; c < sources_.size(); c++) sources_[c]->reset();       // (:428-431)
  framesN_ = (unsigned)n;
  estimated_ = true;
  have_out_ = false;
  out_.clear();
  return framesN_;
}

void MultiChannelWPEDereverberation::prepare_output_()
{
  const unsigned C = channelsN_, K = subbandsN_ / 2 + 1;
  void* dX = NULL;
  T_ = snapshots_(&dX);
  out_.assign((size_t)2 * K * C * (T_ > 0 ? T_ : 0), 0.f);
  if (T_ > 0) {
    void* dO = dev_alloc(sizeof(float) * out_.size());
    try {
      check_abi(btk_wpe_apply(dX, dG_, dO, 1, (int)K, (int)C, T_, T_, (int)lowerN_, (int)upperN_, (int)lower_bw_, (int)upper_bw_, nstream()));
      nsync();
      d2h(out_.data(), dO, sizeof(float) * out_.size());
    } catch (...) { dev_free(dX); dev_free(dO); throw; }
    dev_free(dO);
  }
  dev_free(dX);
  have_out_ = true;
}

const gsl_vector_complex* MultiChannelWPEDereverberation::get_output(unsigned channelX)
{
  if (channelX >= channelsN_)
    throw jindex_error("Invalid channel index: it exceeds the number of channels: %u >= %u\n", channelX, channelsN_);
  return output_[channelX];
}

gsl_vector_complex** MultiChannelWPEDereverberation::calc_every_channel_output(int frame_no)
{
  if (!estimated_) throw jinitialization_error("Call SingleChannelWPEDereverberationFeature::estimate_filter()\n");
  if (frame_no >= 0 && frame_no - 1 != frame_no_)
    throw jindex_error("Problem in 'MultiChannelWPEDereverberation': %d - 1 != %d\n", frame_no, frame_no_);
  if (!have_out_) prepare_output_();
  frame_no_++;
  if (frame_no_ >= T_) throw jiterator_error("end of samples!");
  const unsigned C = channelsN_, K = subbandsN_ / 2 + 1, M = subbandsN_;
  for (unsigned c = 0; c < C; c++)
    for (unsigned k = 0; k < K; k++) {
      const double re = out_[2 * (((size_t)k * C + c) * T_ + frame_no_)], im = out_[2 * (((size_t)k * C + c) * T_ + frame_no_) + 1];
      output_[c]->data[2 * k] = re; output_[c]->data[2 * k + 1] = im;
      if (k > 0 && k < M / 2) { output_[c]->data[2 * (M - k)] = re; output_[c]->data[2 * (M - k) + 1] = -im; }
    }
  return output_;
}

MultiChannelWPEDereverberationFeature::MultiChannelWPEDereverberationFeature(MultiChannelWPEDereverberationPtr& source, unsigned channelX,
                                                                             unsigned primaryChannelX, const String& nm)
    : VectorComplexFeatureStream(source->size(), nm), source_(source), channelX_(channelX), primaryChannelX_(primaryChannelX)
{
  if (channelX >= source->channelsN())
    throw jindex_error("Invalid channel index: it exceeds the number of channels: %u >= %u\n", channelX, source->channelsN());
}

const std::vector<float>& MultiChannelWPEDereverberation::output_block(long* T)
{
  if (!estimated_) throw jinitialization_error("Call SingleChannelWPEDereverberationFeature::estimate_filter()\n");
  if (!have_out_) prepare_output_();
  *T = T_;
  return out_;
}

// In the reference the primary channel's node drives the per-frame estimator and the others read what it computed
// (:716-731).  Here every channel's node serves its own frame counter from the utterance block, so a downstream node
// that drains one channel before the next (the block-served synthesis bank) still sees the right frames.
const gsl_vector_complex* MultiChannelWPEDereverberationFeature::next(int frame_no)
{
  if (frame_no == frame_no_ && frame_no_ >= 0) return vector_;
  if (frame_no >= 0 && frame_no - 1 != frame_no_)
    throw jindex_error("Problem in 'MultiChannelWPEDereverberationFeature': %d - 1 != %d\n", frame_no, frame_no_);
  long T = 0;
  const std::vector<float>& blk = source_->output_block(&T);
  const long idx = frame_no_ + 1;
  if (idx >= T) { is_end_ = true; throw jiterator_error("end of samples!"); }
  const unsigned C = source_->channelsN(), M = size(), K = M / 2 + 1;
  for (unsigned k = 0; k < K; k++) {
    const double re = blk[2 * (((size_t)k * C + channelX_) * T + idx)], im = blk[2 * (((size_t)k * C + channelX_) * T + idx) + 1];
    vector_->data[2 * k] = re; vector_->data[2 * k + 1] = im;
    if (k > 0 && k < M / 2) { vector_->data[2 * (M - k)] = re; vector_->data[2 * (M - k) + 1] = -im; }
  }
  increment_();
  return vector_;
}

void MultiChannelWPEDereverberationFeature::reset()
{
  source_->reset();
  VectorComplexFeatureStream::reset();
}

SingleChannelWPEDereverberationFeature::SingleChannelWPEDereverberationFeature(VectorComplexFeatureStreamPtr& samples, unsigned lowerN,
                                                                               unsigned upperN, unsigned iterationsN, double loadDb,
                                                                               double bandWidth, double sampleRate, const String& nm)
    : VectorComplexFeatureStream(samples->size(), nm),
      core_(new MultiChannelWPEDereverberation(samples->size(), 1, lowerN, upperN, iterationsN, loadDb, bandWidth, 0.0, sampleRate))
{
  core_->set_input(samples);
}

const gsl_vector_complex* SingleChannelWPEDereverberationFeature::next(int frame_no)
{
  if (frame_no == frame_no_ && frame_no_ >= 0) return vector_;
  if (frame_no >= 0 && frame_no - 1 != frame_no_)
    throw jindex_error("Problem in Feature %s: %d != %d\n", name().c_str(), frame_no - 1, frame_no_);
  gsl_vector_complex** out;
  try { out = core_->calc_every_channel_output(-5); }
  catch (jiterator_error&) { is_end_ = true; throw jiterator_error("end of samples!"); }
  increment_();
  memcpy(vector_->data, out[0]->data, sizeof(double) * 2 * size());
  return vector_;
}

void SingleChannelWPEDereverberationFeature::reset()
{
  core_->reset();
  VectorComplexFeatureStream::reset();
}

void SingleChannelWPEDereverberationFeature::next_speaker()
{
  core_->next_speaker();
  VectorComplexFeatureStream::reset();
}

// ================================================================================ SubbandGraphPool
SubbandGraphPool::SubbandGraphPool()
    : rounds_(0), base_(0), prev_T_(0), prev_Lw_(0), prev_Lp_(0), prev_hist_(0), blk_base_(0), out_stride_(0), first_round_(true), i16_(false) {}

SubbandGraphPool::~SubbandGraphPool()
{
  if (pre_.valid) (void)hipStreamSynchronize(cstream());
  for (size_t g = 0; g < graphs_.size(); g++) gsl_vector_float_free(graphs_[g].out);
}

void SubbandGraphPool::add(SubbandDSPtr& beamformer, OverSampledDFTSynthesisBankPtr& synthesis)
{
  if (!first_round_) throw jconsistency_error("SubbandGraphPool: graphs are added before the first next()\n");
  SubbandDS* bf = beamformer.operator->();
  if (!bf->banks_only()) throw jconsistency_error("SubbandGraphPool: the channels of %s must be analysis banks of one geometry\n", bf->name().c_str());
  if (bf->is_half_band_shift()) throw jconsistency_error("SubbandGraphPool: halfBandShift==true is not batched\n");
  if (dynamic_cast<SubbandGSCRLS*>(bf)) throw jconsistency_error("SubbandGraphPool: an adaptive canceller keeps its own recursion per graph and is not batched\n");
  const OverSampledDFTAnalysisBank* a = bf->bank(0);
  if (btk_fb_analysis_bf_fused(a->plan()) != 1)
    throw jconsistency_error("SubbandGraphPool: no fused analysis -> apply kernel for M = %u, m = %u, r = %u\n", a->fftlen(), a->m(), a->r());
  if (synthesis->fftlen() != a->fftlen() || synthesis->r() != a->r())
    throw jdimension_error("SubbandGraphPool: synthesis bank (M = %u, r = %u) does not match the analysis banks (M = %u, r = %u)\n",
                           synthesis->fftlen(), synthesis->r(), a->fftlen(), a->r());
  if (!graphs_.empty()) {
    SubbandDS* b0 = graphs_[0].bf.operator->();
    const OverSampledDFTAnalysisBank* a0 = b0->bank(0);
    const OverSampledDFTSynthesisBank* s0 = graphs_[0].syn.operator->();
    if (bf->chanN() != b0->chanN() || a->fftlen() != a0->fftlen() || a->m() != a0->m() || a->r() != a0->r() ||
        a->delay_compensation_type() != a0->delay_compensation_type() || a->block_frames() != a0->block_frames() ||
        bf->block_quantum() != b0->block_quantum() || synthesis->m() != s0->m() ||
        btk_fb_processing_delay(synthesis->plan()) != btk_fb_processing_delay(s0->plan()))
      throw jdimension_error("SubbandGraphPool: every graph needs the same channel count, filter-bank geometry and block size\n");
  }
  Graph g;
  g.bf = beamformer; g.syn = synthesis; g.live = true; g.T = 0; g.nblocks = 0; g.served = 0;
  g.out = gsl_vector_float_calloc(synthesis->shiftlen()); g.has_out = false; g.round_first = 0; g.round_n = 0;
  graphs_.push_back(g);
}

void SubbandGraphPool::reset()
{
  if (pre_.valid) { (void)hipStreamSynchronize(cstream()); pre_.valid = false; }      // copies of a round nobody will ask for
  for (size_t g = 0; g < graphs_.size(); g++) {
    graphs_[g].syn->reset();                                  // resets the whole graph behind it
    graphs_[g].live = true; graphs_[g].T = 0; graphs_[g].nblocks = 0; graphs_[g].served = 0; graphs_[g].has_out = false;
  }
  rounds_ = 0; base_ = 0; prev_T_ = 0; prev_Lw_ = 0; prev_Lp_ = 0; prev_hist_ = 0; blk_base_ = 0; out_stride_ = 0; first_round_ = true;
  i16_ = false;
}

// The input of one round: every live graph's banks pull a block of input and the sample windows of all graphs go into one block
// [G][N][Lmax] of `buf` (rows of 16 bytes for the fused kernel's vector loads, zero behind a shorter -- ending -- stream), copies
// issued on `stream`.  A graph's windows start their way up as soon as its banks have pulled their input, under the pulling of
// the next graph: the first graph says how long the rows will be, and only if a later one turns out longer the copies are made
// again.  Marks graphs whose stream ends with this round as no longer live.
void SubbandGraphPool::stage_round_(Stage& st, DeviceBuffer& buf, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const size_t G = graphs_.size();
  SubbandDS* b0 = graphs_[0].bf.operator->();
  const unsigned N = b0->chanN();
  const size_t es = i16_ ? sizeof(short) : sizeof(float);
  const long q = i16_ ? 8 : 4;                                 // rows of 16 bytes
  st.plans.assign(G, SubbandBeamformer::BlockPlan());
  st.T.assign(G, 0);
  st.Lmax = 0; st.Tmax = 0; st.t0 = -1; st.f0 = -1; st.valid = true;
  long Lprov = 0;
  char* dPcm = NULL;
  // (the table of the rows' addresses, a segment per graph: no upload of the pool is in flight when a round is staged -- the round
  //  staged before has been waited for, on whichever stream it went)
  btk_row_t* tab = static_cast<btk_row_t*>(hRows_.ensure(sizeof(btk_row_t) * G * N));
  auto upload = [&](size_t g, long pitch) {
    SubbandDS* bf = graphs_[g].bf.operator->();
    const SubbandBeamformer::BlockPlan& p = st.plans[g];
    btk_row_t* t = tab + g * (size_t)N;
    for (unsigned c = 0; c < N; c++)
      t[c].src = i16_ ? static_cast<const void*>(bf->bank(c)->window16(p.b0)) : static_cast<const void*>(bf->bank(c)->window(p.b0));
    upload_rows(t, N, es * p.L, dPcm + es * g * (size_t)N * pitch, es * pitch, stream);
  };
  for (size_t g = 0; g < G; g++) {
    Graph& gr = graphs_[g];
    if (!gr.live) continue;
    SubbandDS* bf = gr.bf.operator->();
    if (!bf->banks_only()) throw jconsistency_error("SubbandGraphPool: the channels of %s changed\n", bf->name().c_str());
    bf->plan_bank_block(st.plans[g]);
    const SubbandBeamformer::BlockPlan& p = st.plans[g];
    if (p.T > 0) {
      if (st.f0 >= 0 && (p.f0 != st.f0 || p.f0 - p.b0 != st.t0))
        throw jconsistency_error("SubbandGraphPool: graph %d is at frame %ld, the others at %ld -- the graphs of a pool advance in lock step\n", (int)g, p.f0, st.f0);
      st.f0 = p.f0; st.t0 = p.f0 - p.b0;
      st.Lmax = std::max(st.Lmax, p.L); st.Tmax = std::max(st.Tmax, p.T);
      if (!dPcm) { Lprov = (p.L + q - 1) / q * q; dPcm = static_cast<char*>(buf.ensure(es * G * N * (Lprov ? Lprov : q))); }
      if (p.L <= Lprov) upload(g, Lprov);
    }
    st.T[g] = p.T;
    if (p.ended) gr.live = false;
  }
  st.Lmax = (st.Lmax + q - 1) / q * q;
  if (st.Tmax > 0 && st.Lmax != Lprov) {
    check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
    dPcm = static_cast<char*>(buf.ensure(es * G * N * st.Lmax));
    for (size_t g = 0; g < G; g++) if (st.T[g] > 0) upload(g, st.Lmax);
  }
}

// One round: the staged input (stage_round_ -- staged ahead in a 16-bit stream), one fused launch, one synthesis launch.
// false: no graph had a frame left.
bool SubbandGraphPool::load_round_()
{
  const size_t G = graphs_.size();
  if (G == 0) return false;
  SubbandDS* b0 = graphs_[0].bf.operator->();
  const OverSampledDFTSynthesisBank* s0 = graphs_[0].syn.operator->();
  const unsigned N = b0->chanN(), M = b0->fftLen(), K = M / 2 + 1, D = s0->shiftlen(), R = 1u << s0->r();
  const long pd = btk_fb_processing_delay(s0->plan());
  const long H = std::max<long>((long)s0->m() * R + R, pd);    // frames of history a round's first block reaches back to
  if (first_round_) {
    // the streams begin: 16-bit PCM if every source of every graph holds it (SampleFeature::pcm16) and the geometry has the entry
    i16_ = btk_fb_analysis_bf_i16_fused(b0->bank(0)->plan()) == 1;
    for (size_t g = 0; g < G && i16_; g++) i16_ = graphs_[g].bf->i16_stream_possible();
    if (i16_) for (size_t g = 0; g < G; g++) graphs_[g].bf->begin_i16_stream();
    pre_.valid = false;
  }
  for (size_t g = 0; g < G; g++) { graphs_[g].T = 0; graphs_[g].nblocks = 0; graphs_[g].served = 0; }
  Stage cur;
  std::chrono::steady_clock::time_point tu0 = std::chrono::steady_clock::now();
  if (pre_.valid) {
    // staged while the round before was served (16-bit streams): the copies run on the thread's second stream
    check_hip(hipStreamSynchronize(cstream()), "hipStreamSynchronize");
    cur = pre_;
    pre_.valid = false;
    dPcm_.swap(dPcmNext_);
  } else {
    stage_round_(cur, dPcm_, nstream());
  }
  if (cur.Tmax == 0) return false;                             // (a plan without frames is the end of its stream)
  const std::vector<SubbandBeamformer::BlockPlan>& plans = cur.plans;
  const long Lmax = cur.Lmax, Tmax = cur.Tmax, t0 = cur.t0, f0 = cur.f0;
  const char* dPcm = static_cast<const char*>(dPcm_.get());
  for (size_t g = 0; g < G; g++) graphs_[g].T = cur.T[g];
  // ---- per-stream weights [G][K][N], from the weight objects as they are now
  float* hW = static_cast<float*>(hW_.ensure(sizeof(float) * 2 * G * K * N));
  std::vector<float> w;
  for (size_t g = 0; g < G; g++) {
    if (graphs_[g].T <= 0) { memset(hW + g * (size_t)2 * K * N, 0, sizeof(float) * 2 * K * N); continue; }
    graphs_[g].bf->effective_weights(w);
    memcpy(hW + g * (size_t)2 * K * N, w.data(), sizeof(float) * 2 * K * N);
  }
  void* dW = dW_.ensure(sizeof(float) * 2 * G * K * N);
  h2d_async(dW, hW, sizeof(float) * 2 * G * K * N);
  nsync();                                                     // the uploads are done: the banks may move their windows on
  std::chrono::steady_clock::time_point tu1 = std::chrono::steady_clock::now();
  g_upload_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(tu1 - tu0).count();
  for (size_t g = 0; g < G; g++) if (graphs_[g].T > 0 || plans[g].ended) graphs_[g].bf->commit_bank_block(plans[g]);
  // ---- the synthesis window of the round, [G][K][keep + Tmax] (rows an even number of frames apart): the last frames of the
  //      window before as history (none in the first round: the stream's first blocks are computed as such, frames before the
  //      stream read as zero), then the new frames -- the same windows, launch alignment included, as a synthesis node on its own
  const long keep = first_round_ ? 0 : std::min(prev_hist_ + prev_T_, H);
  const long Lw = keep + Tmax, Lp = even_pitch(Lw);
  char* win = static_cast<char*>(dWinB_.ensure(std::max(sizeof(float) * 2 * G * K * Lp, sizeof(float) * 2 * G * K * (size_t)prev_Lp_)));
  if (keep) d2d_2d_async(win, sizeof(float) * 2 * Lp, static_cast<const char*>(dWinA_.get()) + sizeof(float) * 2 * (prev_Lw_ - keep),
                         sizeof(float) * 2 * prev_Lp_, sizeof(float) * 2 * keep, G * K);
  base_ = f0;
  dWinA_.swap(dWinB_);
  const long sb = btk_fb_analysis_bf_scratch_bytes(b0->bank(0)->plan(), (int)G, (int)N, 1, Tmax);
  void* scratch = dScratch_.ensure((size_t)(sb > 0 ? sb : 16));
  if (i16_)
    check_abi(btk_fb_analysis_bf_i16(b0->bank(0)->plan(), reinterpret_cast<const short*>(dPcm), Lmax, Lmax, (int)G, (int)N, dW, 1,
                                     win + sizeof(float) * 2 * keep, Lp, t0, Tmax, scratch, sb, nstream()));
  else
    check_abi(btk_fb_analysis_bf(b0->bank(0)->plan(), reinterpret_cast<const float*>(dPcm), Lmax, Lmax, (int)G, (int)N, dW, 1,
                                 win + sizeof(float) * 2 * keep, Lp, t0, Tmax, scratch, sb, nstream()));
  // ---- 16-bit streams: the NEXT round's input is staged now -- its sources only move on, the copies go to the other buffer on the
  //      thread's second stream -- under this round's kernels, download and the serving of its blocks (BTK_NODE_PREFETCH=0: off)
  {
    static const bool off = getenv("BTK_NODE_PREFETCH") && atoi(getenv("BTK_NODE_PREFETCH")) == 0;
    bool more = false;
    for (size_t g = 0; g < G; g++) if (graphs_[g].live) more = true;
    if (i16_ && more && !off) stage_round_(pre_, dPcmNext_, cstream());
  }
  // ---- the output blocks whose newest input frame lies in this round (block b reads the frames b + pd - (m R - 1) .. b + pd)
  const long b_first = std::max<long>(0, base_ - pd), b_end = base_ + Tmax - pd;
  const long nb = b_end > b_first ? b_end - b_first : 0;
  blk_base_ = b_first; out_stride_ = nb * D;
  if (nb > 0) {
    // the window starts at stream frame base_ - keep: block b of the stream is block b - (base_ - keep) of the window; an aligned
    // launch (see OverSampledDFTSynthesisBank::run_window_) may begin one block earlier, that block is dropped
    const long bw = b_first - (base_ - keep);
    const long lead = (btk_fb_synthesis_aligned_form(s0->plan()) == 1 && ((bw + pd) & 1)) ? 1 : 0;
    const long ostride = ((nb + lead) * (long)D + 3) / 4 * 4;
    float* dO = static_cast<float*>(dOut_.ensure(sizeof(float) * G * ostride));
    float* hO = static_cast<float*>(hOut_.ensure(sizeof(float) * G * nb * D));
    check_abi(btk_fb_synthesis(s0->plan(), win, Lw, Lp, (int)G, dO, ostride, bw - lead, nb + lead, nstream()));
    check_hip(hipMemcpy2DAsync(hO, sizeof(float) * nb * D, dO + lead * (long)D, sizeof(float) * ostride, sizeof(float) * nb * D, G,
                               hipMemcpyDeviceToHost, nstream()), "hipMemcpy2DAsync D2H");
    nsync();
    for (size_t g = 0; g < G; g++) {
      const int gain = graphs_[g].syn->gain_factor();
      if (gain > 1) for (long i = 0; i < nb * (long)D; i++) hO[g * (size_t)nb * D + i] *= (float)gain;
    }
  }
  for (size_t g = 0; g < G; g++) {
    const long e = base_ + graphs_[g].T - pd;
    graphs_[g].nblocks = graphs_[g].T > 0 && e > b_first ? e - b_first : 0;
  }
  g_device_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tu1).count();
  prev_T_ = Tmax; prev_Lw_ = Lw; prev_Lp_ = Lp; prev_hist_ = keep; first_round_ = false;
  rounds_++;
  return true;
}

bool SubbandGraphPool::next()
{
  if (graphs_.empty()) return false;
  for (;;) {
    bool any = false;
    for (size_t g = 0; g < graphs_.size(); g++) if (graphs_[g].served < graphs_[g].nblocks) any = true;
    if (any) break;
    // the round is used up (or was too short to complete a block: a first round under the synthesis delay): the next one
    bool more = pre_.valid;                                   // (a staged round may be the last one of streams that are no longer live)
    for (size_t g = 0; g < graphs_.size(); g++) if (graphs_[g].live) more = true;
    if (!more || !load_round_()) {
      for (size_t g = 0; g < graphs_.size(); g++) { graphs_[g].has_out = false; graphs_[g].nblocks = 0; graphs_[g].served = 0; }
      return false;
    }
  }
  const unsigned D = graphs_[0].syn->shiftlen();
  const float* hO = static_cast<const float*>(hOut_.get());
  for (size_t g = 0; g < graphs_.size(); g++) {
    Graph& gr = graphs_[g];
    gr.has_out = gr.served < gr.nblocks;
    if (gr.has_out) {
      memcpy(gr.out->data, hO + g * (size_t)out_stride_ + (size_t)gr.served * D, sizeof(float) * D);
      gr.served++;
    }
  }
  return true;
}

bool SubbandGraphPool::next_round()
{
  if (graphs_.empty()) return false;
  for (size_t g = 0; g < graphs_.size(); g++) { graphs_[g].round_first = 0; graphs_[g].round_n = 0; }
  for (;;) {
    bool any = false;
    for (size_t g = 0; g < graphs_.size(); g++) if (graphs_[g].served < graphs_[g].nblocks) any = true;
    if (any) break;
    bool more = pre_.valid;
    for (size_t g = 0; g < graphs_.size(); g++) if (graphs_[g].live) more = true;
    if (!more || !load_round_()) {
      for (size_t g = 0; g < graphs_.size(); g++) { graphs_[g].has_out = false; graphs_[g].nblocks = 0; graphs_[g].served = 0; }
      return false;
    }
  }
  const unsigned D = graphs_[0].syn->shiftlen();
  const float* hO = static_cast<const float*>(hOut_.get());
  for (size_t g = 0; g < graphs_.size(); g++) {
    Graph& gr = graphs_[g];
    gr.round_first = gr.served; gr.round_n = gr.nblocks - gr.served;
    gr.has_out = gr.round_n > 0;
    if (gr.has_out) memcpy(gr.out->data, hO + g * (size_t)out_stride_ + (size_t)(gr.nblocks - 1) * D, sizeof(float) * D);
    gr.served = gr.nblocks;
  }
  return true;
}

long SubbandGraphPool::round_blocks(unsigned g, const float** blocks) const
{
  if (g >= graphs_.size()) throw jindex_error("SubbandGraphPool: graph %d of %d\n", (int)g, (int)graphs_.size());
  const Graph& gr = graphs_[g];
  if (blocks) *blocks = gr.round_n > 0 ? static_cast<const float*>(hOut_.get()) + g * (size_t)out_stride_ + (size_t)gr.round_first * graphs_[0].syn->shiftlen() : NULL;
  return gr.round_n;
}

const gsl_vector_float* SubbandGraphPool::output(unsigned g) const
{
  if (g >= graphs_.size()) throw jindex_error("SubbandGraphPool: graph %d of %d\n", (int)g, (int)graphs_.size());
  return graphs_[g].has_out ? graphs_[g].out : NULL;
}

bool SubbandGraphPool::is_end(unsigned g) const
{
  if (g >= graphs_.size()) throw jindex_error("SubbandGraphPool: graph %d of %d\n", (int)g, (int)graphs_.size());
  return !graphs_[g].live && graphs_[g].served >= graphs_[g].nblocks;
}
