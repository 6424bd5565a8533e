// The framework's own RCCL layer: one communicator per process group, held directly (ncclComm_t from
// a unique id), with its own prioritised HIP stream, its own events and explicit stream ordering. It
// replaces the implicit runtimes the reference leans on — init_process_group('nccl') plus DDP's
// reducer, ZeroRedundancyOptimizer and FSDP (reference: multi-gpu/ddp/train.py:19-36,284;
// kaggle-zero1.py:1062-1077; kaggle-fsdp.py:1061-1086).
//
// Ordering contract of every collective issued here:
//   1. an event recorded on the caller's current stream makes the communication stream wait for the
//      kernels that produced the input;
//   2. the collective is enqueued on the communication stream;
//   3. an event recorded behind it is returned inside a RcclWork. work.wait() makes the caller's
//      (then current) stream wait for it on the device — the host never blocks; work.synchronize()
//      is the host-blocking variant.
// Buffers are kept alive across the asynchronous use by the work handle: it holds a reference to every
// tensor of the call until wait() has ordered the caller's stream behind the collective (from then on
// any reuse of the memory by that stream is ordered behind it as well). No allocator bookkeeping on
// the communication stream is needed, and memory returns to the pool without an event round trip.
//
// xGMI on an MI355X node is a full point-to-point mesh: RCCL's kernels occupy one workgroup per
// channel, so the channel bounds (min_ctas / max_ctas) are what trades link bandwidth against CUs
// taken from the MFMA kernels of the backward pass; the stream priority decides whether a collective
// that became ready overtakes queued compute kernels.
#include <ATen/ATen.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <chrono>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace {

using at::Tensor;

#define DPA_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    TORCH_CHECK(e_ == hipSuccess, "HIP error in ", #expr, ": ", hipGetErrorString(e_));            \
  } while (0)

#define DPA_RCCL(expr)                                                                             \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    TORCH_CHECK(r_ == ncclSuccess, "RCCL error in ", #expr, ": ", ncclGetErrorString(r_));         \
  } while (0)

ncclDataType_t nccl_dtype(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return ncclFloat32;
    case at::kBFloat16: return ncclBfloat16;
    case at::kHalf: return ncclFloat16;
    case at::kDouble: return ncclFloat64;
    case at::kInt: return ncclInt32;
    case at::kLong: return ncclInt64;
    case at::kByte: return ncclUint8;
    case at::kChar: return ncclInt8;
    case at::kBool: return ncclUint8;
    default: TORCH_CHECK(false, "RcclComm: unsupported dtype ", t.scalar_type());
  }
}

ncclRedOp_t nccl_op(int64_t op) {
  switch (op) {
    case 0: return ncclSum;
    case 1: return ncclProd;
    case 2: return ncclMax;
    case 3: return ncclMin;
    case 4: return ncclAvg;
    default: TORCH_CHECK(false, "RcclComm: reduce op must be 0 sum, 1 prod, 2 max, 3 min or 4 avg");
  }
}

// An event behind one enqueued collective.
struct RcclWork : torch::CustomClassHolder {
  hipEvent_t done = nullptr;
  int device = 0;
  std::vector<Tensor> held;   // operands of the collective, see the ordering contract above
  hipStream_t issue_stream = nullptr;                       // the caller's stream when the collective was issued
  c10::optional<c10::hip::HIPStream> comm_stream_;          // the stream the collective runs on
  explicit RcclWork(int dev) : device(dev) { DPA_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming)); }
  ~RcclWork() override {
    // A handle that was never waited for (an exception unwound past it, a caller dropped it) must not hand its
    // operands back to the allocator while the collective may still be reading or writing them.
    if (done && !held.empty()) (void)hipEventSynchronize(done);
    held.clear();
    if (done) (void)hipEventDestroy(done);
  }
  // device-side: the caller's current stream waits for the collective
  bool wait() {
    const hipStream_t cur = c10::hip::getCurrentHIPStream(device).stream();
    DPA_HIP(hipStreamWaitEvent(cur, done, 0));
    // Ordering `cur` behind the collective protects reuse of the memory by `cur` only. The allocator hands a
    // freed block back to the stream it was allocated on: if that is another stream than the one waiting here,
    // tell the allocator that the communication stream used the block (it then defers the reuse by an event).
    if (cur != issue_stream && comm_stream_.has_value())
      for (const Tensor& t : held)
        if (t.defined() && t.has_storage()) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), *comm_stream_);
    held.clear();
    return true;
  }
  // host-side
  void synchronize() {
    DPA_HIP(hipEventSynchronize(done));
    held.clear();
  }
  bool is_completed() {
    hipError_t e = hipEventQuery(done);
    if (e == hipErrorNotReady) return false;
    DPA_HIP(e);
    held.clear();
    return true;
  }
};

// Communication streams live until the process ends and are shared by successive communicators of the
// same (device, priority): the caching allocator may still have to record an event on one — for a
// block that a collective used — long after the communicator that issued it is gone. High and default
// priority streams come from PyTorch's stream pool (the allocator and the profiler know them); a
// stream below the default priority does not exist in that pool and is made here.
// On ROCm 7.0.2 / RCCL 2.26.6 only the default priority is usable under this communicator: with one rank, RCCL
// calls issued on a pool stream of high priority or on a hand-made low-priority stream stop completing after one
// to five calls (benchmarks/debug/prio_diag.sh; the stall comes and goes between runs and is independent of
// re-recording one 'ready' event, of the explicit group calls and of stream queries after the enqueue, while the
// same event chain without RCCL calls, and c10d's high-priority option, run). The Python layer therefore refuses
// both non-default priorities for this transport (parallel/comm.py); the parameter stays for stacks where it works.
c10::hip::HIPStream comm_stream(int device, int priority) {
  static std::mutex mu;
  static std::vector<std::tuple<int, int, c10::hip::HIPStream>> pool;
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : pool)
    if (std::get<0>(e) == device && std::get<1>(e) == priority) return std::get<2>(e);
  c10::hip::HIPStream s = c10::hip::getDefaultHIPStream((c10::DeviceIndex)device);
  if (priority <= 0) {
    s = c10::hip::getStreamFromPool(/*isHighPriority=*/priority < 0, (c10::DeviceIndex)device);
  } else {
    hipStream_t raw = nullptr;
    DPA_HIP(hipStreamCreateWithPriority(&raw, hipStreamNonBlocking, priority));
    s = c10::hip::getStreamFromExternal(raw, (c10::DeviceIndex)device);
  }
  pool.emplace_back(device, priority, s);
  return s;
}

struct Timed {
  std::string kind;
  int64_t bytes;
  hipEvent_t t0, t1;
};

struct RcclComm : torch::CustomClassHolder {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  c10::optional<c10::hip::HIPStream> cstream;   // the same stream as a PyTorch object (allocator bookkeeping)
  hipEvent_t ready = nullptr;   // "inputs produced" marker, re-recorded per call (stream order makes that safe)
  int rank, world, device;
  int priority = 0;
  int64_t min_ctas = 0, max_ctas = 0;
  bool timing = false;
  std::deque<Timed> timed;
  std::vector<void*> reg_handles;
  int64_t calls = 0;
  std::mutex mu;

  static Tensor unique_id() {
    ncclUniqueId id;
    DPA_RCCL(ncclGetUniqueId(&id));
    Tensor t = at::empty({(int64_t)sizeof(id)}, at::kByte);
    std::memcpy(t.data_ptr(), &id, sizeof(id));
    return t;
  }

  // priority: -1 = highest the device offers, 0 = default, +1 = lowest
  RcclComm(const Tensor& id_bytes, int64_t rank_, int64_t world_, int64_t device_, int64_t priority_,
           int64_t min_ctas_, int64_t max_ctas_)
      : rank((int)rank_), world((int)world_), device((int)device_), min_ctas(min_ctas_), max_ctas(max_ctas_) {
    TORCH_CHECK(id_bytes.device().is_cpu() && id_bytes.scalar_type() == at::kByte && id_bytes.is_contiguous() &&
                    id_bytes.numel() == (int64_t)sizeof(ncclUniqueId),
                "RcclComm: the unique id is a CPU uint8 tensor of ", sizeof(ncclUniqueId), " bytes");
    TORCH_CHECK(world >= 1 && rank >= 0 && rank < world, "RcclComm: rank ", rank, " of ", world);
    ncclUniqueId id;
    std::memcpy(&id, id_bytes.data_ptr(), sizeof(id));
    c10::hip::HIPGuard guard((c10::DeviceIndex)device);
    int lo = 0, hi = 0;   // numerically lower = more urgent: [hi .. lo]
    DPA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    priority = priority_ < 0 ? hi : (priority_ > 0 ? lo : 0);
    if (priority < hi) priority = hi;
    if (priority > lo) priority = lo;
    cstream = comm_stream(device, priority);
    stream = cstream->stream();
    DPA_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 1;
    if (min_ctas > 0) cfg.minCTAs = (int)min_ctas;
    if (max_ctas > 0) cfg.maxCTAs = (int)max_ctas;
    ncclResult_t r = ncclCommInitRankConfig(&comm, world, id, rank, &cfg);
    if (r == ncclInvalidArgument || r == ncclInvalidUsage) {
      // A library older than the header refuses the newer config record before it talks to any peer: only then is
      // the plain call on the same unique id safe (channel bounds then only come from NCCL_MIN_NCHANNELS /
      // NCCL_MAX_NCHANNELS). Any other failure may have left a half-finished bootstrap behind on some ranks;
      // retrying with the same id could hang the rendezvous and would hide the real error.
      TORCH_WARN("RcclComm: ncclCommInitRankConfig refused its config (", ncclGetErrorString(r),
                 "); creating the communicator without per-communicator channel bounds");
      comm = nullptr;
      min_ctas = max_ctas = 0;
      DPA_RCCL(ncclCommInitRank(&comm, world, id, rank));
    } else {
      TORCH_CHECK(r == ncclSuccess, "RCCL error in ncclCommInitRankConfig: ", ncclGetErrorString(r));
    }
  }

  ~RcclComm() override { destroy(); }

  void destroy() {
    std::lock_guard<std::mutex> lk(mu);
    if (comm) {
      if (stream) (void)hipStreamSynchronize(stream);
      for (void* h : reg_handles) (void)ncclCommDeregister(comm, h);
      reg_handles.clear();
      (void)ncclCommDestroy(comm);
      comm = nullptr;
    }
    for (auto& t : timed) {
      (void)hipEventDestroy(t.t0);
      (void)hipEventDestroy(t.t1);
    }
    timed.clear();
    if (ready) (void)hipEventDestroy(ready), ready = nullptr;
    stream = nullptr;   // stays in the process-wide pool
  }

  // Tear the communicator down without waiting for outstanding work (a peer died).
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    if (comm) {
      (void)ncclCommAbort(comm);
      comm = nullptr;
    }
  }

  int64_t get_rank() const { return rank; }
  int64_t get_world() const { return world; }
  int64_t get_priority() const { return priority; }
  int64_t num_calls() const { return calls; }
  int64_t rccl_version() const {
    int v = 0;
    DPA_RCCL(ncclGetVersion(&v));
    return v;
  }
  // what the library reports about itself: ranks in the communicator (not what the caller passed in)
  int64_t comm_count() const {
    int n = 0;
    DPA_RCCL(ncclCommCount(comm, &n));
    return n;
  }
  // "" while healthy, else the asynchronous error RCCL has seen on this communicator
  std::string async_error() const {
    if (!comm) return "communicator destroyed";
    ncclResult_t st = ncclSuccess;
    ncclResult_t r = ncclCommGetAsyncError(comm, &st);
    if (r != ncclSuccess) return ncclGetErrorString(r);
    return st == ncclSuccess ? "" : ncclGetErrorString(st);
  }

  void set_timing(bool on) { timing = on; }

  // (kind, bytes, milliseconds on the communication stream) of the collectives that have finished
  std::vector<std::tuple<std::string, int64_t, double>> drain_timings() {
    std::lock_guard<std::mutex> lk(mu);
    std::vector<std::tuple<std::string, int64_t, double>> out;
    while (!timed.empty()) {
      Timed& t = timed.front();
      if (hipEventQuery(t.t1) == hipErrorNotReady) break;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, t.t0, t.t1) == hipSuccess) out.emplace_back(t.kind, t.bytes, (double)ms);
      (void)hipEventDestroy(t.t0);
      (void)hipEventDestroy(t.t1);
      timed.pop_front();
    }
    return out;
  }

  // Persistent buffers (flat gradient / parameter shards) can be registered once so that RCCL maps
  // them for direct peer access instead of staging through its own FIFO buffers. The memory must stay
  // allocated until deregister_buffers() (or the end of the communicator).
  bool register_buffer(const Tensor& t) {
    check(t, "buffer");
    std::lock_guard<std::mutex> lk(mu);
    void* h = nullptr;
    ncclResult_t r = ncclCommRegister(comm, t.data_ptr(), (size_t)t.nbytes(), &h);
    if (r != ncclSuccess || h == nullptr) return false;
    reg_handles.push_back(h);
    return true;
  }

  int64_t num_registered() {
    std::lock_guard<std::mutex> lk(mu);
    return (int64_t)reg_handles.size();
  }

  int64_t deregister_buffers() {
    std::lock_guard<std::mutex> lk(mu);
    if (stream) DPA_HIP(hipStreamSynchronize(stream));
    int64_t n = 0;
    for (void* h : reg_handles) n += comm && ncclCommDeregister(comm, h) == ncclSuccess;
    reg_handles.clear();
    return n;
  }

  // ---- collectives -----------------------------------------------------------------------------
  c10::intrusive_ptr<RcclWork> all_reduce(Tensor t, int64_t op) {
    check(t, "tensor");
    const ncclRedOp_t rop = nccl_op(op);
    const ncclDataType_t dt = nccl_dtype(t);
    return run("all_reduce", t.nbytes(), {t}, [&] {
      DPA_RCCL(ncclAllReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), dt, rop, comm, stream));
    });
  }

  c10::intrusive_ptr<RcclWork> broadcast(Tensor t, int64_t root) {
    check(t, "tensor");
    TORCH_CHECK(root >= 0 && root < world, "broadcast: root ", root, " of ", world);
    return run("broadcast", t.nbytes(), {t}, [&] {
      DPA_RCCL(ncclBroadcast(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t), (int)root, comm, stream));
    });
  }

  // out (n / world elements) = this rank's slice of the sum; `out` may be the rank's slice of `inp`
  c10::intrusive_ptr<RcclWork> reduce_scatter(Tensor out, Tensor inp, int64_t op) {
    check(out, "out");
    check(inp, "inp");
    TORCH_CHECK(out.scalar_type() == inp.scalar_type() && inp.numel() == out.numel() * world,
                "reduce_scatter: inp must hold world * out.numel() elements of out's dtype");
    return run("reduce_scatter", inp.nbytes(), {out, inp}, [&] {
      DPA_RCCL(ncclReduceScatter(inp.data_ptr(), out.data_ptr(), (size_t)out.numel(), nccl_dtype(inp), nccl_op(op), comm,
                                 stream));
    });
  }

  // out (n * world elements) = every rank's inp, by rank; `inp` may be the rank's slice of `out`
  c10::intrusive_ptr<RcclWork> all_gather(Tensor out, Tensor inp) {
    check(out, "out");
    check(inp, "inp");
    TORCH_CHECK(out.scalar_type() == inp.scalar_type() && out.numel() == inp.numel() * world,
                "all_gather: out must hold world * inp.numel() elements of inp's dtype");
    return run("all_gather", out.nbytes(), {out, inp}, [&] {
      DPA_RCCL(ncclAllGather(inp.data_ptr(), out.data_ptr(), (size_t)inp.numel(), nccl_dtype(inp), comm, stream));
    });
  }

  // Row-wise all-to-all(v) along dim 0: in_splits[p] rows go to peer p, out_splits[p] rows come from it.
  // Empty split lists mean equal splits. One grouped batch of point-to-point transfers: on the xGMI
  // mesh every pair has its own link, so all of them run at once.
  c10::intrusive_ptr<RcclWork> all_to_all(Tensor out, Tensor inp, std::vector<int64_t> out_splits,
                                          std::vector<int64_t> in_splits) {
    check(out, "out");
    check(inp, "inp");
    TORCH_CHECK(out.scalar_type() == inp.scalar_type(), "all_to_all: dtypes differ");
    TORCH_CHECK(out.dim() >= 1 && inp.dim() >= 1, "all_to_all: tensors need a leading dimension");
    const int64_t in_rows = inp.size(0), out_rows = out.size(0);
    const int64_t row_in = in_rows ? inp.numel() / in_rows : 0, row_out = out_rows ? out.numel() / out_rows : 0;
    if (in_splits.empty()) {
      TORCH_CHECK(in_rows % world == 0, "all_to_all: ", in_rows, " rows do not split over ", world, " ranks");
      in_splits.assign(world, in_rows / world);
    }
    if (out_splits.empty()) {
      TORCH_CHECK(out_rows % world == 0, "all_to_all: ", out_rows, " rows do not split over ", world, " ranks");
      out_splits.assign(world, out_rows / world);
    }
    TORCH_CHECK((int)in_splits.size() == world && (int)out_splits.size() == world, "all_to_all: one split per rank");
    int64_t si = 0, so = 0;
    for (int p = 0; p < world; ++p) {
      TORCH_CHECK(in_splits[p] >= 0 && out_splits[p] >= 0, "all_to_all: negative split");
      si += in_splits[p];
      so += out_splits[p];
    }
    TORCH_CHECK(si == in_rows && so == out_rows, "all_to_all: splits (", si, " in, ", so, " out) do not add up to the rows (",
                in_rows, " in, ", out_rows, " out)");
    TORCH_CHECK(in_rows == 0 || out_rows == 0 || row_in == row_out, "all_to_all: row sizes differ");
    const int64_t row = in_rows ? row_in : row_out;
    const size_t esz = (size_t)inp.element_size();
    return run("all_to_all", inp.nbytes(), {out, inp}, [&] {
      const char* src = static_cast<const char*>(inp.data_ptr());
      char* dst = static_cast<char*>(out.data_ptr());
      const ncclDataType_t dt = nccl_dtype(inp);
      int64_t oi = 0, oo = 0;
      for (int p = 0; p < world; ++p) {
        if (in_splits[p] > 0)
          DPA_RCCL(ncclSend(src + (size_t)oi * row * esz, (size_t)(in_splits[p] * row), dt, p, comm, stream));
        if (out_splits[p] > 0)
          DPA_RCCL(ncclRecv(dst + (size_t)oo * row * esz, (size_t)(out_splits[p] * row), dt, p, comm, stream));
        oi += in_splits[p];
        oo += out_splits[p];
      }
    });
  }

  // Host-blocking rendezvous of all ranks (one-element all-reduce, then wait for it).
  void barrier() {
    c10::hip::HIPGuard guard((c10::DeviceIndex)device);
    Tensor one = at::ones({1}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, device));
    all_reduce(one, 0)->synchronize();
  }

  // Host waits until everything enqueued on the communication stream has run.
  void synchronize() { DPA_HIP(hipStreamSynchronize(stream)); }

 private:
  void check(const Tensor& t, const char* name) const {
    TORCH_CHECK(comm != nullptr, "RcclComm: the communicator was destroyed");
    TORCH_CHECK(t.is_cuda() && t.get_device() == device, name, " must live on cuda:", device);
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  }

  template <class F>
  c10::intrusive_ptr<RcclWork> run(const char* kind, int64_t bytes, std::initializer_list<Tensor> used, F&& enqueue) {
    std::lock_guard<std::mutex> lk(mu);
    c10::hip::HIPGuard guard((c10::DeviceIndex)device);
    const c10::hip::HIPStream cur = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device);
    DPA_HIP(hipEventRecord(ready, cur.stream()));
    DPA_HIP(hipStreamWaitEvent(stream, ready, 0));
    Timed tm;
    if (timing) {
      tm.kind = kind;
      tm.bytes = bytes;
      DPA_HIP(hipEventCreate(&tm.t0));
      DPA_HIP(hipEventCreate(&tm.t1));
      DPA_HIP(hipEventRecord(tm.t0, stream));
    }
    // one launch per call, also for the multi-transfer all-to-all; a failed argument check inside must
    // not leave the group open (every later call would then queue behind it and never launch)
    DPA_RCCL(ncclGroupStart());
    try {
      enqueue();
    } catch (...) {
      (void)ncclGroupEnd();
      throw;
    }
    DPA_RCCL(ncclGroupEnd());
    if (timing) {
      DPA_HIP(hipEventRecord(tm.t1, stream));
      timed.push_back(tm);
    }
    auto work = c10::make_intrusive<RcclWork>(device);
    work->held.assign(used.begin(), used.end());
    work->issue_stream = cur.stream();
    work->comm_stream_ = cstream;
    DPA_HIP(hipEventRecord(work->done, stream));
    ++calls;
    return work;
  }
};

}  // namespace

TORCH_LIBRARY_FRAGMENT(dpa, m) {
  m.class_<RcclWork>("RcclWork")
      .def("wait", &RcclWork::wait)
      .def("synchronize", &RcclWork::synchronize)
      .def("is_completed", &RcclWork::is_completed);
  m.class_<RcclComm>("RcclComm")
      .def(torch::init<const Tensor&, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t>())
      .def_static("unique_id", &RcclComm::unique_id)
      .def("rank", &RcclComm::get_rank)
      .def("world", &RcclComm::get_world)
      .def("priority", &RcclComm::get_priority)
      .def("num_calls", &RcclComm::num_calls)
      .def("rccl_version", &RcclComm::rccl_version)
      .def("comm_count", &RcclComm::comm_count)
      .def("async_error", &RcclComm::async_error)
      .def("set_timing", &RcclComm::set_timing)
      .def("drain_timings", &RcclComm::drain_timings)
      .def("register_buffer", &RcclComm::register_buffer)
      .def("num_registered", &RcclComm::num_registered)
      .def("deregister_buffers", &RcclComm::deregister_buffers)
      .def("all_reduce", &RcclComm::all_reduce)
      .def("broadcast", &RcclComm::broadcast)
      .def("reduce_scatter", &RcclComm::reduce_scatter)
      .def("all_gather", &RcclComm::all_gather)
      .def("all_to_all", &RcclComm::all_to_all)
      .def("barrier", &RcclComm::barrier)
      .def("synchronize", &RcclComm::synchronize)
      .def("abort", &RcclComm::abort)
      .def("destroy", &RcclComm::destroy);
}
