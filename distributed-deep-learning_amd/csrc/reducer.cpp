// Bucket reducer of the data-parallel engine (native runtime piece; replaces c10d's C++ Reducer).
//
// Reference: `DDP(model, device_ids=[local_rank])` (ddp.py:127, ddp_n_pp.py:139) - torch's Reducer copies
// every gradient into 25 MiB bucket tensors, all-reduces them on NCCL's stream and copies the result
// back.  Here the gradients already live in ONE flat arena whose contiguous segments ARE the buckets
// (utils/arena.py), so the reducer owns no memory at all: it keeps the segment table and the pending
// counters, and when the last gradient of a segment is final it fences the compute stream
// (bucket-ready event), issues reduce-scatter(AVG) + all-gather on the communication stream through
// the process group (RCCL when the group is "nccl"), and records the grads-ready event that
// `finish()` later makes the compute stream wait for.  One process per GPU; xGMI is fully connected,
// so RS + AG of a bucket use all seven links of a GPU at once.
//
// The Python `DataParallelEngine` (parallel/dp.py) delegates to this class on a GPU; the CPU / gloo
// test transport keeps the pure-Python path (host-staged tensors, no streams).
//
// Two transports: `direct` (an owned communicator, rccl_comm.h) enqueues ncclReduceScatter / ncclAllGather /
// ncclAllReduce on the communication stream ITSELF - the two fences then order the RCCL kernels directly; without
// it the collectives go through the c10d process group (whose internal stream is fenced against the communication
// stream on both sides: two more event hops and a Work object per bucket).
//
// Fence tests (tests/test_gpu_fences.py): on a one-GPU box every collective of a world of one is an identity, so a
// missing fence cannot change a bit.  `test_hook` = 1 enqueues spin(`hook_spin_us`) + "segment *= 0.5" on the stream
// that carried the collective, between the bucket-ready and the grads-ready fence; `skip_fence` deliberately leaves
// one of the two fences out (1 = bucket-ready, 2 = grads-ready) - the negative control those tests must FAIL on.
// Both are constructor arguments (never environment variables) and default to off.
#include "torch_hip.h"
#include "ddl_kernels.h"
#include "rccl_comm.h"

#include <ATen/hip/HIPEvent.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/csrc/distributed/c10d/Types.hpp>

#include <memory>
#include <utility>
#include <vector>

namespace {

using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;
using StreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
using Event = at::cuda::CUDAEvent;

class BucketReducer {
 public:
  BucketReducer(at::Tensor flat_grad, std::vector<int64_t> seg_start, std::vector<int64_t> seg_end,
                std::vector<int64_t> slot_segment, std::vector<int64_t> seg_pending,
                c10::intrusive_ptr<c10d::ProcessGroup> pg, int64_t rank, int64_t world, bool zero1, bool allreduce,
                bool serialize, bool timing, int64_t comm_stream_id, int64_t comm_device_index,
                int64_t comm_device_type, std::shared_ptr<ddl::RcclComm> direct, int64_t test_hook,
                double hook_spin_us, int64_t skip_fence)
      : grad_(std::move(flat_grad)), seg_start_(std::move(seg_start)), seg_end_(std::move(seg_end)),
        slot_segment_(std::move(slot_segment)), pending0_(std::move(seg_pending)), pg_(std::move(pg)),
        rank_(rank), world_(world), zero1_(zero1), allreduce_(allreduce), serialize_(serialize), timing_(timing),
        comm_(c10::Stream::unpack3(comm_stream_id, (c10::DeviceIndex)comm_device_index,
                                   (c10::DeviceType)comm_device_type)),    // the engine's communication stream
        direct_(std::move(direct)), test_hook_(test_hook), hook_spin_us_(hook_spin_us), skip_fence_(skip_fence) {
    TORCH_CHECK(direct_ != nullptr || pg_, "BucketReducer needs a process group or an owned communicator");
    TORCH_CHECK(!test_hook_ || grad_.scalar_type() == at::kFloat, "test hook: fp32 gradient arena");
    if (direct_ != nullptr) {
      TORCH_CHECK(direct_->nranks() == world && direct_->rank() == rank, "owned communicator: rank / world differ from the group's");
      TORCH_CHECK(direct_->device() == grad_.device().index(), "owned communicator lives on another device");
      TORCH_CHECK(grad_.scalar_type() == at::kFloat, "owned communicator path: fp32 gradient arena");
    }
    TORCH_CHECK(comm_.device_index() == grad_.device().index(), "communication stream lives on another device");
    TORCH_CHECK(grad_.is_cuda() && grad_.dim() == 1 && grad_.is_contiguous(), "flat_grad: contiguous 1-D GPU tensor");
    TORCH_CHECK(seg_start_.size() == seg_end_.size() && seg_start_.size() == pending0_.size(), "segment tables differ");
    TORCH_CHECK(world_ >= 1 && rank_ >= 0 && rank_ < world_, "bad rank / world");
    for (size_t k = 0; k < seg_start_.size(); ++k) {
      TORCH_CHECK(0 <= seg_start_[k] && seg_start_[k] < seg_end_[k] && seg_end_[k] <= grad_.numel(), "segment ", k,
                  " out of range");
      TORCH_CHECK((seg_end_[k] - seg_start_[k]) % world_ == 0, "segment ", k, " not divisible by the world size");
    }
    for (int64_t s : slot_segment_) TORCH_CHECK(s >= 0 && s < (int64_t)seg_start_.size(), "slot -> segment out of range");
    pending_ = pending0_;
    launched_.assign(seg_start_.size(), false);
  }

  // `rearm`: also forget what was counted / launched so far (start of a step); without it only the
  // gate changes (the pipeline schedule's no_sync() window around micro-batches 0..M-2)
  void arm(bool armed, bool rearm) {
    armed_ = armed;
    if (rearm) reset();
  }

  // gradient of arena slot `slot` is final (autograd hook / the native engine's per-unit callback)
  void param_ready(int64_t slot) {
    if (!armed_) return;
    TORCH_CHECK(slot >= 0 && slot < (int64_t)slot_segment_.size(), "slot out of range");
    const int64_t k = slot_segment_[slot];
    if (--pending_[k] == 0) segment_ready(k);
  }

  // all gradients of segment k are final: launch its collective(s) behind a bucket-ready fence
  void segment_ready(int64_t k) {
    TORCH_CHECK(k >= 0 && k < (int64_t)seg_start_.size(), "segment out of range");
    if (launched_[k] || !armed_) return;
    launched_[k] = true;
    const int64_t n = (seg_end_[k] - seg_start_[k]) / world_;
    at::Tensor full = grad_.narrow(0, seg_start_[k], seg_end_[k] - seg_start_[k]);
    at::Tensor shard = full.narrow(0, rank_ * n, n);
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(grad_.device().index());
    const Stream where = serialize_ ? main : comm_;       // --debug-sync: everything on the compute stream
    if (!serialize_ && skip_fence_ != 1) {                // bucket-ready: compute -> communication stream
      Event ready;
      ready.record(main);
      ready.block(comm_);
    }
    {
      StreamGuard guard(where);
      std::unique_ptr<Event> t0, t1;
      if (timing_) {
        t0 = std::make_unique<Event>(hipEventDefault);
        t0->record(where);
      }
      if (direct_ != nullptr) {
        // owned communicator: the RCCL kernels are enqueued on `where` itself (in place: shard == full + rank * n)
        float* base = grad_.data_ptr<float>() + seg_start_[k];
        if (allreduce_) {
          direct_->all_reduce(base, (size_t)(n * world_), ncclFloat32, ncclAvg, where.stream());
        } else {
          direct_->reduce_scatter(base, base + rank_ * n, (size_t)n, ncclFloat32, ncclAvg, where.stream());
          if (!zero1_) direct_->all_gather(base + rank_ * n, base, (size_t)n, ncclFloat32, where.stream());
        }
      } else if (allreduce_) {
        c10d::AllreduceOptions opts;
        opts.reduceOp = c10d::ReduceOp::AVG;
        std::vector<at::Tensor> ts{full};
        pg_->allreduce(ts, opts)->wait();
      } else {
        c10d::ReduceScatterOptions opts;
        opts.reduceOp = c10d::ReduceOp::AVG;
        pg_->_reduce_scatter_base(shard, full, opts)->wait();     // in place: shard is this rank's slice of full
        if (!zero1_) pg_->_allgather_base(full, shard)->wait();
      }
      if (test_hook_) {        // visible work of the communication stream between the two fences (fence tests)
        if (hook_spin_us_ > 0) ddl::launch_spin(hook_spin_us_, where.stream());
        ddl::launch_scale(full.data_ptr<float>(), full.numel(), 0.5f, where.stream());
      }
      if (timing_) {
        t1 = std::make_unique<Event>(hipEventDefault);
        t1->record(where);
        spans_.emplace_back(std::move(t0), std::move(t1));
      }
    }
    if (!serialize_) {                                     // grads-ready: communication -> compute stream (finish)
      done_.emplace_back();
      done_.back().record(comm_);
    }
    ++launches_;
  }

  // launch stragglers (segments whose hooks did not all fire), make the compute stream wait for every bucket
  void finish() {
    if (armed_)
      for (size_t k = 0; k < launched_.size(); ++k)
        if (!launched_[k]) segment_ready((int64_t)k);
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(grad_.device().index());
    if (skip_fence_ != 2)
      for (auto& ev : done_) ev.block(main);
    done_.clear();
    reset();
  }

  // RCCL busy time (ms) of the spans recorded since the last call; synchronises on their end events
  double drain_ms() {
    double ms = 0.0;
    for (auto& s : spans_) {
      s.second->synchronize();
      ms += s.first->elapsed_time(*s.second);
    }
    spans_.clear();
    return ms;
  }

  void set_timing(bool on) { timing_ = on; }          // RCCL span events on / off (StepProfiler.set_enabled)
  // per-bucket collective: one in-place all-reduce, or reduce-scatter + all-gather.  The buckets are plain arena
  // ranges and the collectives stay outside the captured graphs, so the form can change between two steps
  // (bench.py measures both on the same captured step: "grad_sync_ab")
  void set_allreduce(bool on) {
    TORCH_CHECK(!zero1_ || !on, "ZeRO-1 needs the scattered form");
    allreduce_ = on;
  }
  int64_t launches() const { return launches_; }
  bool direct() const { return direct_ != nullptr; }

 private:
  void reset() {
    pending_ = pending0_;
    launched_.assign(launched_.size(), false);
  }

  at::Tensor grad_;
  std::vector<int64_t> seg_start_, seg_end_, slot_segment_, pending0_, pending_;
  std::vector<bool> launched_;
  c10::intrusive_ptr<c10d::ProcessGroup> pg_;
  int64_t rank_, world_;
  bool zero1_, allreduce_, serialize_, timing_;
  Stream comm_;
  std::shared_ptr<ddl::RcclComm> direct_;
  int64_t test_hook_;
  double hook_spin_us_;
  int64_t skip_fence_;
  bool armed_ = true;
  std::vector<Event> done_;
  std::vector<std::pair<std::unique_ptr<Event>, std::unique_ptr<Event>>> spans_;
  int64_t launches_ = 0;
};

}  // namespace

void register_reducer(pybind11::module_& m) {
  pybind11::class_<BucketReducer>(m, "BucketReducer")
      .def(pybind11::init<at::Tensor, std::vector<int64_t>, std::vector<int64_t>, std::vector<int64_t>,
                          std::vector<int64_t>, c10::intrusive_ptr<c10d::ProcessGroup>, int64_t, int64_t, bool, bool,
                          bool, bool, int64_t, int64_t, int64_t, std::shared_ptr<ddl::RcclComm>, int64_t, double,
                          int64_t>(),
           pybind11::arg("flat_grad"), pybind11::arg("seg_start"), pybind11::arg("seg_end"),
           pybind11::arg("slot_segment"), pybind11::arg("seg_pending"), pybind11::arg("process_group"),
           pybind11::arg("rank"), pybind11::arg("world"), pybind11::arg("zero1"), pybind11::arg("allreduce"),
           pybind11::arg("serialize"), pybind11::arg("timing"), pybind11::arg("comm_stream_id"),
           pybind11::arg("comm_device_index"), pybind11::arg("comm_device_type"),
           pybind11::arg("direct") = std::shared_ptr<ddl::RcclComm>(), pybind11::arg("test_hook") = 0,
           pybind11::arg("hook_spin_us") = 0.0, pybind11::arg("skip_fence") = 0)
      .def("arm", &BucketReducer::arm, pybind11::arg("armed"), pybind11::arg("rearm"))
      .def("param_ready", &BucketReducer::param_ready)
      .def("segment_ready", &BucketReducer::segment_ready)
      .def("finish", &BucketReducer::finish)
      .def("drain_ms", &BucketReducer::drain_ms)
      .def("set_timing", &BucketReducer::set_timing)
      .def("set_allreduce", &BucketReducer::set_allreduce)
      .def_property_readonly("launches", &BucketReducer::launches)
      .def_property_readonly("direct", &BucketReducer::direct);
}
