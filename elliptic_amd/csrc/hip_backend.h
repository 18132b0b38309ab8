// ellgpu -- HIP backend for Engine<> (gfx950 / MI355X): device memory, copies
// and the one generic kernel template that runs every functor of engine.h.
//
// Launch geometry: one item per lane, 128-lane workgroups (2 wavefronts), the
// functor's per-lane digit strings staged in LDS as byte columns (lane-major:
// the 64 lanes of a wave read 64 consecutive bytes, conflict-free).
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "engine.h"

namespace ell {

constexpr int BLOCK = 128;
// Chunking of the pipelined host-buffer entry points (Engine::pipelined): the first chunk is one
// quantum of 256 CUs x 4 SIMDs x ELL_PIPE_WAVES x 64 items, the following ones ELL_PIPE_STEP quanta.
// Swept on one box (profiles/r02_host_pipeline_sweep.log): 2 waves x 3 quanta -- chunks of 131 072,
// 393 216 and the rest of a 2^20 batch -- gives 108 M verifies/s from host buffers against 97 M/s
// with the round-1 setting (3 waves x 4 quanta), G*k 512 against 403 M/s.
#ifndef ELL_PIPE_WAVES
#define ELL_PIPE_WAVES 2
#endif
#ifndef ELL_PIPE_STEP
#define ELL_PIPE_STEP 3
#endif

// Fn::MIN_WAVES (optional, default 1) = waves per SIMD the register allocator must
// leave room for (__launch_bounds__'s second argument counts waves per SIMD here).
template <class Fn, class = void>
struct MinWaves { static constexpr int value = 1; };
template <class Fn>
struct MinWaves<Fn, decltype((void)Fn::MIN_WAVES)> { static constexpr int value = Fn::MIN_WAVES; };

// Workgroups are launched with BLOCK lanes on grids that fill the device and with ONE wave (64
// lanes) on smaller ones (HipBackend::launch): the dispatcher places a workgroup's waves together,
// and below about four waves per CU whole two-wave workgroups leave some SIMDs with two waves and
// others with none -- 49 152 verifies took 1.23 ms where 32 768 took 0.95 (profiles/
// r04_workgroup_size_ab.txt).  The LDS columns keep their BLOCK stride either way: a one-wave
// workgroup allocates twice the columns it uses (<= 8.4 KB for the ladders) -- nothing on gfx950,
// the only target (160 KB of LDS per CU: 16 such workgroups take 134 KB); a part with 64 KB per CU
// would want k_run instantiated per lane count.
// CONTRACT for every functor: f(tid, ds) must be IDEMPOTENT per item and free of side effects
// other than its own item's outputs (no atomics, no counters) -- fill_lane makes the idle lanes of
// the launch's last wave repeat the last item's work instead of sitting masked off.
template <class Fn>
__global__ void __launch_bounds__(BLOCK, MinWaves<Fn>::value) k_run(const Fn f, size_t nthreads) {
  __shared__ signed char lds_digits[(Fn::DS_PER_LANE > 0 ? Fn::DS_PER_LANE : 1) * BLOCK];
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  DigitStore ds{lds_digits + threadIdx.x, BLOCK};
  if (!fill_lane(tid, nthreads)) return;       // the launch's last wave works with all its lanes
  f(tid, ds);
}

// The lanes-per-item layer (coop.h): ONE unit per workgroup of one wave.  A field element is one
// DPP row (16 lanes); the wave's four rows hold the same element, or -- inside a step of the group
// law -- four different ones whose products run together (coop.h Q / mulq).  LDS: the digit
// columns and Fn::ROW_BYTES of row memory (every lane's own window table; the rows write the
// same bytes to the same places).
// (at most two waves per SIMD are ever resident: the register allocator may keep every
// loop-invariant row constant -- the shifted copies of p, the lane masks -- in registers)
template <class Fn>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_run_coop(const Fn f, size_t units) {
  __shared__ signed char lds_digits[(Fn::DS_PER_LANE > 0 ? Fn::DS_PER_LANE : 1) * 64];
  __shared__ __attribute__((aligned(16))) unsigned char lds_rows[Fn::ROW_BYTES];
  if ((size_t)blockIdx.x >= units) return;
  DigitStore ds{lds_digits + threadIdx.x, 64};
  f((size_t)blockIdx.x, ds, (void*)lds_rows);
}

struct TimedLaunch {
  const char* name;
  hipEvent_t e0, e1;
};

// defined next to `__constant__ RtField g_rt` (inst.hip, group 16): blocking copy of a
// user-defined curve's parameter block into the current device's constant memory
int rt_upload_device(const RtField* f);

struct HipBackend {
  int device = 0;
  hipStream_t own = nullptr;       // the context's stream
  hipStream_t cur = nullptr;       // stream used by the current call
  hipStream_t own2 = nullptr;      // second compute lane of the host-buffer entry points
  hipStream_t copy = nullptr;      // H2D of the host-buffer entry points
  hipStream_t copy_out = nullptr;  // D2H of the host-buffer entry points
  static constexpr int RING = 16;
  hipEvent_t ring[RING] = {};      // cross-stream dependencies (reused round-robin)
  unsigned ring_i = 0;
  int last = 0;                    // sticky hipError_t of the current call
  bool timing = false;             // record HIP events around every launch
  std::vector<TimedLaunch>* timed = nullptr;

  // The scratch arenas (window tables, Jacobian results, batch-inversion prefixes ...) belong to
  // the context, not to a stream, and the *_dev entry points return without synchronising.  The
  // context has TWO arenas (the compute lanes of the host-buffer pipeline), and a *_dev call takes
  // the arena of ITS stream: a caller that alternates two streams gets two passes in flight -- the
  // latency-bound front of pass i + 1 (s^-1, window tables) runs beside the issue-bound ladder of
  // pass i (131 072 verifies: 1.27 -> 1.08 ms per pass, 2^20: 8.08 -> 7.75 ms;
  // profiles/r04_two_passes_in_flight.jsonl).  A call on a third stream takes the arena used
  // longest ago and first waits (on the device) for the event its previous user recorded after
  // its last launch.  Host-buffer calls use both arenas: they wait for everything the *_dev calls
  // left in flight, and synchronise before they return.
  bool host_synced = false;        // the current call has synchronised `cur` on the host
  struct DevLane {
    hipStream_t stream = nullptr;  // stream of the lane's last *_dev call
    hipEvent_t done = nullptr;     // recorded behind that call's last launch
    bool pending = false;          // ... and possibly not reached yet
  };
  DevLane dlane[2];
  int dcur = 0;                    // lane of the current call
  int dlast = 1;                   // lane of the previous *_dev call
  bool one_dev_lane = false;       // ELLGPU_DEV_LANES=1: every *_dev call takes lane 0 (the old ordering)
  // host-buffer entry points (and anything else that runs on the context's own streams)
  void use_stream(void*) {
    (void)hipSetDevice(device);
    cur = own;
    last = 0;
    host_synced = false;
    dcur = 0;
    for (auto& l : dlane)
      if (l.pending && l.done) {
        if (l.stream != own) note(hipStreamWaitEvent(own, l.done, 0));
        note(hipStreamWaitEvent(own2, l.done, 0));
      }
  }
  // *_dev entry points: returns the scratch lane of this call
  int use_stream_dev(void* s) {
    (void)hipSetDevice(device);
    cur = s ? (hipStream_t)s : own;
    last = 0;
    host_synced = false;
    int L;
    if (one_dev_lane || dlane[0].stream == cur) L = 0;
    else if (dlane[1].stream == cur) L = 1;
    else L = !dlane[0].stream ? 0 : (!dlane[1].stream ? 1 : 1 - dlast);
#ifndef ELL_NO_STREAM_ORDER          // (test switch: shows that test_dev_calls_on_alternating_streams fails without it)
    if (dlane[L].pending && dlane[L].stream != cur && dlane[L].done) note(hipStreamWaitEvent(cur, dlane[L].done, 0));
#endif
    dlane[L].stream = cur;
    dcur = dlast = L;
    return L;
  }
  // end of an entry point: `async` = the call returned without synchronising `cur`
  void end_call(bool async) {
    if (async && dlane[dcur].done) {
      note(hipEventRecord(dlane[dcur].done, cur));
      dlane[dcur].pending = true;
    } else if (host_synced) {
      // the call waited on the host for its streams, which had been ordered after whatever was in
      // flight (use_stream): nothing is left running
      for (auto& l : dlane) l.pending = false;
    }
    // else: a call that neither launched asynchronously nor synchronised (ellgpu_ctx_reserve with
    // nothing to grow): what was in flight before it still is
  }
  // everything this context may have in flight, on whatever stream (before freeing scratch)
  void sync_all() {
    for (auto& l : dlane) {
      if (l.pending && l.stream) note(hipStreamSynchronize(l.stream));
      l.pending = false;
    }
    note(hipStreamSynchronize(cur ? cur : own));
    host_synced = true;
  }
  int device_index() const { return device; }
  int cus = 256;                   // compute units of this device (hipDeviceProp_t::multiProcessorCount)
  // launches of at most this many lanes take one-wave workgroups (see k_run): four waves per SIMD
  // (ELLGPU_ONE_WAVE_GROUPS overrides, read when the context is created; 0 = never)
  size_t one_wave_groups = (size_t)256 * 4 * 64 * 4;
  int compute_units() const { return cus; }
  void* own_stream() const { return (void*)own; }
  void rt_upload(const RtField& f) { note((hipError_t)rt_upload_device(&f)); }
  void note(hipError_t e) {
    if (e != hipSuccess && !last) last = (int)e;
  }
  // one grow-only pinned host buffer (staging of the small host-buffer calls, Engine::pipelined)
  void* pin_buf = nullptr;
  size_t pin_cap = 0;
  void* pinned(size_t bytes) {
    if (bytes <= pin_cap) return pin_buf;
    if (pin_buf) { (void)hipHostFree(pin_buf); pin_buf = nullptr; pin_cap = 0; }
    if (hipHostMalloc(&pin_buf, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pin_buf = nullptr; return nullptr; }
    pin_cap = bytes;
    return pin_buf;
  }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
  }
  void free_(void* p) { note(hipFree(p)); }
  void h2d(void* d, const void* h, size_t bytes) {
    note(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, cur));
  }
  void d2h(void* h, const void* d, size_t bytes) {
    note(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, cur));
  }
  int sync() {
    note(hipStreamSynchronize(cur ? cur : own));
    host_synced = true;
    return last ? E_HIP : E_OK;
  }
  // ---- copy stream for the pipelined host-buffer entry points (Engine::pipelined) ----
  // ELLGPU_PIPE_WAVES (developer override): resident waves per SIMD the quantum assumes
  static size_t pipeline_quantum() {
    static const size_t q = []() {
      const char* e = getenv("ELLGPU_PIPE_WAVES");
      int w = e ? atoi(e) : ELL_PIPE_WAVES;
      if (w < 1 || w > 8) w = ELL_PIPE_WAVES;
      return (size_t)256 * 4 * (size_t)w * 64;
    }();
    return q;
  }
  void h2d_copy(void* d, const void* h, size_t bytes) {
    if (bytes) note(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, copy));
  }
  void d2h_copy(void* h, const void* d, size_t bytes) {       // its own stream: PCIe is full duplex
    if (bytes) note(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, copy_out));
  }
  // the compute stream waits for every copy queued so far
  void copies_before_compute() {
    hipEvent_t e = ring[ring_i++ % RING];
    note(hipEventRecord(e, copy));
    note(hipStreamWaitEvent(cur, e, 0));
  }
  // returns a handle for "everything launched on the compute stream so far"
  int mark_compute() {
    int i = ring_i++ % RING;
    note(hipEventRecord(ring[i], cur));
    return i;
  }
  void copy_after(int ev) { note(hipStreamWaitEvent(copy_out, ring[ev], 0)); }
  // compute lane 0 = the context's stream, lane 1 = a second stream for alternate chunks
  void select_lane(int lane) { cur = lane ? own2 : own; }
  // wait for the copy stream and both lanes
  int sync_lanes() {
    note(hipStreamSynchronize(copy));
    note(hipStreamSynchronize(copy_out));
    note(hipStreamSynchronize(own2));
    note(hipStreamSynchronize(own));
    host_synced = true;                               // `own` had been ordered after whatever was in flight
    return last ? E_HIP : E_OK;
  }
  template <class Fn>
  void launch(const Fn& f, size_t nthreads) {
    if (nthreads == 0) return;
    const unsigned B = nthreads <= one_wave_groups ? 64u : (unsigned)BLOCK;
    unsigned blocks = (unsigned)((nthreads + B - 1) / B);
    TimedLaunch t{Fn::NAME, nullptr, nullptr};
    if (timing && timed) {
      note(hipEventCreate(&t.e0));
      note(hipEventCreate(&t.e1));
      note(hipEventRecord(t.e0, cur));
    }
    hipLaunchKernelGGL(k_run<Fn>, dim3(blocks), dim3(B), 0, cur, f, nthreads);
    note(hipGetLastError());
    if (timing && timed) {
      note(hipEventRecord(t.e1, cur));
      timed->push_back(t);
    }
  }
  template <class Fn>
  void launch_coop(const Fn& f, size_t units) {
    if (units == 0) return;
    TimedLaunch t{Fn::NAME, nullptr, nullptr};
    if (timing && timed) {
      note(hipEventCreate(&t.e0));
      note(hipEventCreate(&t.e1));
      note(hipEventRecord(t.e0, cur));
    }
    hipLaunchKernelGGL(k_run_coop<Fn>, dim3((unsigned)units), dim3(64), 0, cur, f, units);
    note(hipGetLastError());
    if (timing && timed) {
      note(hipEventRecord(t.e1, cur));
      timed->push_back(t);
    }
  }
};

}  // namespace ell
