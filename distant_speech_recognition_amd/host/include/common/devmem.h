// common/devmem.h -- device / pinned-host memory and the HIP stream of the engine's node layer (not part of the reference's
// interface).  The reference's nodes compute on the host, one frame per next(); the engine's nodes compute blocks on the device
// (modulated/modulated.h, BlockSource).  Steady-state rules of the node layer:
//   * every launch and every copy of a host thread's graphs is ordered on ONE non-NULL, non-blocking HIP stream
//     (btk_node_stream(): the reference is single-threaded, one host thread drives one pipeline -- SURVEY 8(b) "Threading");
//   * device and pinned buffers belong to the node that uses them and only ever GROW (DeviceBuffer / PinnedBuffer): once a
//     stream has seen its largest block, a block costs no hipMalloc / hipHostMalloc / hipFree;
//   * uploads come from pinned memory (the analysis banks keep their sample windows there), downloads go to pinned memory,
//     both as hipMemcpyAsync on the node stream.
#pragma once
#include <cstddef>

// the calling thread's node stream as a hipStream_t passed as void* (created on first use; never NULL afterwards)
void* btk_node_stream();
// wait until everything the calling thread's nodes have launched is done (what a caller does before it touches a device pointer
// a node handed out -- device_snapshots(), BlockSource::device_block() -- from another stream)
void btk_node_synchronize();
// allocations the node layer has made so far in this process: {hipMalloc, hipHostMalloc} (tests and the bench use the
// difference across blocks to show that a steady-state block allocates nothing)
void btk_node_alloc_counts(long* device_allocs, long* pinned_allocs);

// where a block's host time goes, summed over the process since the last reset (host/examples/node_api_bench.cc, bench.py
// stages.node_api): pull = the analysis banks drawing their input blocks from the source nodes into the pinned windows;
// upload = the windows' way to the device (asynchronous copies + the wait for them); device = fused / staged launches, synthesis,
// the PCM's way back and the wait for all of it.  What remains of a run's wall time is the per-frame serving of next().
void btk_node_timers(double* pull_s, double* upload_s, double* device_s);
void btk_node_timers_reset();

class DeviceBuffer {
 public:
  DeviceBuffer() : p_(NULL), cap_(0) {}
  ~DeviceBuffer() { release(); }
  // at least `bytes` bytes; the contents are NOT kept when the buffer has to grow (it grows by half as much again at least)
  void* ensure(size_t bytes);
  void* get() const { return p_; }
  size_t capacity() const { return cap_; }
  void release();
  void swap(DeviceBuffer& o) { void* p = p_; p_ = o.p_; o.p_ = p; size_t c = cap_; cap_ = o.cap_; o.cap_ = c; }
 private:
  DeviceBuffer(const DeviceBuffer&);
  DeviceBuffer& operator=(const DeviceBuffer&);
  void* p_;
  size_t cap_;
};

class PinnedBuffer {
 public:
  PinnedBuffer() : p_(NULL), cap_(0) {}
  ~PinnedBuffer() { release(); }
  void* ensure(size_t bytes);                         // contents not kept on growth
  void* ensure_keep(size_t bytes, size_t keep_bytes); // the first keep_bytes bytes survive a growth
  void* get() const { return p_; }
  size_t capacity() const { return cap_; }
  void release();
 private:
  PinnedBuffer(const PinnedBuffer&);
  PinnedBuffer& operator=(const PinnedBuffer&);
  void* p_;
  size_t cap_;
};
