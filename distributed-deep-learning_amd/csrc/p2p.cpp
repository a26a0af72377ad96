// Point-to-point engine of the pipeline schedules (native runtime piece).
//
// Reference: `torch.distributed.pipelining` moves stage-boundary activations and their gradients with
// `batch_isend_irecv` from the schedule loop (pp.py:138-150 -> schedules.py:505,758-800); c10d then makes
// RCCL's internal stream wait for EVERYTHING already enqueued on the caller's (compute) stream, and the
// caller's stream wait for the transfer - communication is serialised with compute in both directions.
//
// Here transfers are issued from dedicated P2P streams:
//   * a send group is fenced behind the compute stream with ONE event recorded at the point where its
//     payload was produced (data-ready: compute -> p2p), never behind later compute;
//   * a receive is fenced behind nothing: it can be posted ahead of the compute that will consume it
//     (the schedules pre-post the receives of the next micro-batches into per-micro-batch slots), so the
//     bytes cross xGMI while the previous micro-batch is still computing;
//   * every posted group returns a ticket; `wait(ticket)` makes the CURRENT (compute) stream wait for that
//     group only (arrived: p2p -> compute).  The host never blocks.
// Opposite-direction transfers between one pair stay ONE RCCL group - issued as two calls in different orders
// on the two sides they would deadlock (SURVEY hard part 5).
//
// Two transports.
//   links (owned communicators, rccl_comm.h): ONE 2-rank communicator and ONE hipStream per pipeline neighbour.
//     ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd are enqueued on the neighbour's stream itself, so the
//     fences order the RCCL kernels directly, and a receive pre-posted toward the previous stage can never sit in
//     front of a send toward the next stage (different communicator, different stream, different xGMI link).
//   c10d (fallback): one coalesced group through the process group, whose internal stream is FIFO for every
//     transfer of the stage and fenced against the P2P stream on both sides.
//
// Fence tests (tests/test_gpu_fences.py): `test_hook` = 1 enqueues spin + "received tensors *= 0.5" on the
// transfer's stream between the data-ready and the arrived fence; `skip_fence` leaves one fence out (1 = data-ready,
// 2 = arrived, 3 = the slot-reuse fence) - the negative controls.  Constructor arguments, default off.
//
// `parallel/p2p.py` wraps this class and provides the same interface over gloo (CPU tests, several ranks on
// one GPU), where transfers are host-staged and synchronous.
#include "torch_hip.h"
#include "ddl_kernels.h"
#include "rccl_comm.h"

#include <cstdlib>
#include <ATen/hip/HIPEvent.h>
#include <pybind11/stl.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/csrc/distributed/c10d/Types.hpp>

#include <map>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

namespace {

using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;
using StreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
using Event = at::cuda::CUDAEvent;

ncclDataType_t nccl_dtype_of(const at::Tensor& t) {
  const int code = ddl::rccl_dtype_code((int)t.scalar_type());
  TORCH_CHECK(code >= 0, "p2p: unsupported dtype ", t.scalar_type());
  return (ncclDataType_t)code;
}

// (peer's rank in the pipeline group, communicator, the peer's rank inside it, stream id, device index, device type)
using LinkSpec = std::tuple<int64_t, std::shared_ptr<ddl::RcclComm>, int64_t, int64_t, int64_t, int64_t>;

class P2PEngine {
 public:
  P2PEngine(c10::intrusive_ptr<c10d::ProcessGroup> pg, int64_t stream_id, int64_t device_index,
            int64_t device_type, bool serialize, bool timing, std::vector<LinkSpec> links, int64_t my_rank,
            int64_t world, int64_t test_hook, double hook_spin_us, int64_t skip_fence)
      : pg_(std::move(pg)), p2p_(c10::Stream::unpack3(stream_id, (c10::DeviceIndex)device_index,
                                                      (c10::DeviceType)device_type)),
        serialize_(serialize), timing_(timing), test_hook_(test_hook), hook_spin_us_(hook_spin_us),
        skip_fence_(skip_fence) {
    for (auto& l : links) {
      auto comm = std::get<1>(l);
      TORCH_CHECK(comm != nullptr, "p2p: link without a communicator");
      TORCH_CHECK(comm->device() == p2p_.device_index(), "p2p: link communicator lives on another device");
      const int64_t peer_in_comm = std::get<2>(l);
      TORCH_CHECK(peer_in_comm >= 0 && peer_in_comm < comm->nranks(), "p2p: peer rank outside its communicator");
      Link link{comm, (int)peer_in_comm,
                Stream(c10::Stream::unpack3(std::get<3>(l), (c10::DeviceIndex)std::get<4>(l), (c10::DeviceType)std::get<5>(l)))};
      TORCH_CHECK(link.stream.device_index() == p2p_.device_index(), "p2p: link stream lives on another device");
      links_.emplace(std::get<0>(l), std::move(link));
    }
    if (!links_.empty()) {
      me_ = (int)my_rank;
      world_ = (int)world;
    } else {
      TORCH_CHECK(pg_, "P2PEngine needs a process group or per-neighbour links");
      me_ = pg_->getRank();
      world_ = pg_->getSize();
    }
  }

  // One group: sends[i] -> group rank send_peers[i], recvs[j] <- group rank recv_peers[j].
  // `after_compute`: fence the group behind what the compute stream has enqueued so far (needed when it
  // carries a payload produced there; pure receive groups pass false and start as early as the peer sends).
  int64_t post(std::vector<at::Tensor> sends, std::vector<int64_t> send_peers, std::vector<at::Tensor> recvs,
               std::vector<int64_t> recv_peers, bool after_compute) {
    TORCH_CHECK(sends.size() == send_peers.size() && recvs.size() == recv_peers.size(), "p2p: peers / tensors differ");
    const size_t n = sends.size() + recvs.size();
    TORCH_CHECK(n > 0, "p2p: empty group");
    const int dev = p2p_.device_index();
    for (auto& t : sends) TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.device().index() == dev, "p2p: contiguous tensors on the engine's GPU only");
    for (auto& t : recvs) TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.device().index() == dev, "p2p: contiguous tensors on the engine's GPU only");
    // DDL_P2P_ALLOW_SELF=1: a rank may address itself (RCCL runs a grouped self send + receive as a local copy) -
    // the one-GPU rehearsal of this engine on the real RCCL (bench/p2p_self_probe.py, tests/test_gpu_dist.py)
    static const bool allow_self = [] { const char* e = getenv("DDL_P2P_ALLOW_SELF"); return e && e[0] == '1'; }();
    for (int64_t r : send_peers) TORCH_CHECK(r >= 0 && r < world_ && (r != me_ || allow_self), "p2p: bad send peer ", r);
    for (int64_t r : recv_peers) TORCH_CHECK(r >= 0 && r < world_ && (r != me_ || allow_self), "p2p: bad recv peer ", r);
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev);
    const int64_t ticket = next_++;
    Posted& posted = done_[ticket];
    std::unique_ptr<Event> ready;
    if (!serialize_ && after_compute && skip_fence_ != 1) {      // data-ready: compute -> p2p
      ready = std::make_unique<Event>();
      ready->record(main);
    }
    if (!links_.empty()) {
      // ---- owned communicators: one RCCL group per neighbour, on that neighbour's own stream ----
      std::map<int64_t, std::pair<std::vector<ddl::RcclComm::Xfer>, std::vector<ddl::RcclComm::Xfer>>> by_peer;
      std::map<int64_t, std::vector<at::Tensor>> recv_by_peer;
      for (size_t i = 0; i < sends.size(); ++i) {
        auto it = links_.find(send_peers[i]);
        TORCH_CHECK(it != links_.end(), "p2p: no link toward group rank ", send_peers[i]);
        by_peer[send_peers[i]].first.push_back({sends[i].data_ptr(), (size_t)sends[i].numel(), nccl_dtype_of(sends[i]), it->second.peer});
      }
      for (size_t j = 0; j < recvs.size(); ++j) {
        auto it = links_.find(recv_peers[j]);
        TORCH_CHECK(it != links_.end(), "p2p: no link toward group rank ", recv_peers[j]);
        by_peer[recv_peers[j]].second.push_back({recvs[j].data_ptr(), (size_t)recvs[j].numel(), nccl_dtype_of(recvs[j]), it->second.peer});
        recv_by_peer[recv_peers[j]].push_back(recvs[j]);
      }
      for (auto& kv : by_peer) {
        Link& link = links_.at(kv.first);
        const Stream where = serialize_ ? main : link.stream;
        if (ready) ready->block(where);
        std::unique_ptr<Event> t0, t1;
        if (timing_) {
          t0 = std::make_unique<Event>(hipEventDefault);
          t0->record(where);
        }
        link.comm->group(kv.second.first, kv.second.second, where.stream());
        hook(recv_by_peer[kv.first], where);
        if (timing_) {
          t1 = std::make_unique<Event>(hipEventDefault);
          t1->record(where);
          spans_.emplace_back(std::move(t0), std::move(t1));
        }
        if (!serialize_) {
          posted.arrived.emplace_back();
          posted.arrived.back().record(where);               // arrived: p2p -> compute (wait)
        }
      }
    } else {
      // ---- c10d: one coalesced group (like batch_isend_irecv) on the process group's communicator ----
      const Stream where = serialize_ ? main : p2p_;          // --debug-sync: everything on the compute stream
      if (ready) ready->block(where);
      StreamGuard guard(where);
      std::unique_ptr<Event> t0, t1;
      if (timing_) {
        t0 = std::make_unique<Event>(hipEventDefault);
        t0->record(where);
      }
      pg_->startCoalescing(c10::DeviceType::CUDA);
      c10::intrusive_ptr<c10d::Work> work;
      for (size_t i = 0; i < sends.size(); ++i) {
        std::vector<at::Tensor> ts{sends[i]};
        work = pg_->send(ts, (int)send_peers[i], 0);
      }
      for (size_t j = 0; j < recvs.size(); ++j) {
        std::vector<at::Tensor> ts{recvs[j]};
        work = pg_->recv(ts, (int)recv_peers[j], 0);
      }
      work = pg_->endCoalescing(c10::DeviceType::CUDA);
      if (work) work->wait();                                // stream-level: `where` waits for RCCL's stream
      hook(recvs, where);
      if (timing_) {
        t1 = std::make_unique<Event>(hipEventDefault);
        t1->record(where);
        spans_.emplace_back(std::move(t0), std::move(t1));
      }
      if (!serialize_) {
        posted.arrived.emplace_back();
        posted.arrived.back().record(where);
      }
    }
    // the payloads stay referenced until the group was waited for (the allocator must not hand a send buffer to
    // somebody else while the transfer is still queued on a stream it knows nothing about)
    posted.keep = std::move(sends);
    for (auto& t : recvs) posted.keep.push_back(std::move(t));
    if (serialize_) done_.erase(ticket);
    ++groups_;
    return ticket;
  }

  // the current (compute) stream waits for one posted group
  void wait(int64_t ticket) {
    auto it = done_.find(ticket);
    if (it == done_.end()) return;                            // serialised mode, or already waited for
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(p2p_.device_index());
    if (skip_fence_ != 2)
      for (auto& ev : it->second.arrived) ev.block(main);
    done_.erase(it);
  }

  // ... for every outstanding group (end of a step: send buffers may be overwritten afterwards)
  void drain() {
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(p2p_.device_index());
    if (skip_fence_ != 2)
      for (auto& kv : done_)
        for (auto& ev : kv.second.arrived) ev.block(main);
    done_.clear();
  }

  // Order everything posted from now on behind what the compute stream has enqueued so far.  The schedules call it
  // at the end of every step: the next step pre-posts receives into the SAME per-micro-batch slots, and nothing
  // else keeps such a receive from landing before the forward that still reads the slot has run (an evaluation
  // loop has no backward chain and no host sync; under hipGraph replay the host runs far ahead).
  void fence() {
    if (serialize_ || skip_fence_ == 3) return;
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(p2p_.device_index());
    Event consumed;
    consumed.record(main);
    if (links_.empty()) consumed.block(p2p_);
    for (auto& kv : links_) consumed.block(kv.second.stream);
    ++fences_;
  }

  double drain_ms() {
    double ms = 0.0;
    for (auto& s : spans_) {
      s.second->synchronize();
      ms += s.first->elapsed_time(*s.second);
    }
    spans_.clear();
    return ms;
  }

  void set_timing(bool on) { timing_ = on; }
  int64_t groups() const { return groups_; }
  int64_t outstanding() const { return (int64_t)done_.size(); }
  int64_t num_links() const { return (int64_t)links_.size(); }

 private:
  struct Link {
    std::shared_ptr<ddl::RcclComm> comm;
    int peer;            // the neighbour's rank inside `comm`
    Stream stream;       // this neighbour's transfer stream
  };
  struct Posted {
    std::vector<Event> arrived;
    std::vector<at::Tensor> keep;
  };

  void hook(const std::vector<at::Tensor>& received, const Stream& where) {
    if (!test_hook_) return;
    if (hook_spin_us_ > 0) ddl::launch_spin(hook_spin_us_, where.stream());
    for (const at::Tensor& t : received)
      if (t.scalar_type() == at::kFloat) ddl::launch_scale(t.data_ptr<float>(), t.numel(), 0.5f, where.stream());
  }

  c10::intrusive_ptr<c10d::ProcessGroup> pg_;
  Stream p2p_;
  bool serialize_, timing_;
  int64_t test_hook_;
  double hook_spin_us_;
  int64_t skip_fence_;
  int me_ = 0, world_ = 1;
  std::map<int64_t, Link> links_;
  int64_t next_ = 0, groups_ = 0, fences_ = 0;
  std::map<int64_t, Posted> done_;
  std::vector<std::pair<std::unique_ptr<Event>, std::unique_ptr<Event>>> spans_;
};

}  // namespace

void register_p2p(pybind11::module_& m) {
  pybind11::class_<P2PEngine>(m, "P2PEngine")
      .def(pybind11::init<c10::intrusive_ptr<c10d::ProcessGroup>, int64_t, int64_t, int64_t, bool, bool,
                          std::vector<LinkSpec>, int64_t, int64_t, int64_t, double, int64_t>(),
           pybind11::arg("process_group"), pybind11::arg("stream_id"), pybind11::arg("device_index"),
           pybind11::arg("device_type"), pybind11::arg("serialize"), pybind11::arg("timing"),
           pybind11::arg("links") = std::vector<LinkSpec>(), pybind11::arg("rank") = 0, pybind11::arg("world") = 1,
           pybind11::arg("test_hook") = 0, pybind11::arg("hook_spin_us") = 0.0, pybind11::arg("skip_fence") = 0)
      .def("post", &P2PEngine::post, pybind11::arg("sends"), pybind11::arg("send_peers"), pybind11::arg("recvs"),
           pybind11::arg("recv_peers"), pybind11::arg("after_compute"))
      .def("wait", &P2PEngine::wait)
      .def("drain", &P2PEngine::drain)
      .def("fence", &P2PEngine::fence)
      .def("drain_ms", &P2PEngine::drain_ms)
      .def("set_timing", &P2PEngine::set_timing)
      .def_property_readonly("groups", &P2PEngine::groups)
      .def_property_readonly("outstanding", &P2PEngine::outstanding)
      .def_property_readonly("num_links", &P2PEngine::num_links);
}
