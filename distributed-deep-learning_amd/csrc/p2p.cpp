// Point-to-point engine of the pipeline schedules (native runtime piece).
//
// Reference: `torch.distributed.pipelining` moves stage-boundary activations and their gradients with
// `batch_isend_irecv` from the schedule loop (pp.py:138-150 -> schedules.py:505,758-800); c10d then makes
// RCCL's internal stream wait for EVERYTHING already enqueued on the caller's (compute) stream, and the
// caller's stream wait for the transfer - communication is serialised with compute in both directions.
//
// Here transfers are issued from a dedicated P2P stream:
//   * a send group is fenced behind the compute stream with ONE event recorded at the point where its
//     payload was produced (data-ready: compute -> p2p), never behind later compute;
//   * a receive is fenced behind nothing: it can be posted ahead of the compute that will consume it
//     (the schedules pre-post the receives of the next micro-batches into per-micro-batch slots), so the
//     bytes cross xGMI while the previous micro-batch is still computing;
//   * every posted group returns a ticket; `wait(ticket)` makes the CURRENT (compute) stream wait for that
//     group only (arrived: p2p -> compute).  The host never blocks.
// Opposite-direction transfers between one pair stay ONE RCCL group (start/endCoalescing) - issued as two
// calls in different orders on the two sides they would deadlock (SURVEY hard part 5).
//
// `parallel/p2p.py` wraps this class and provides the same interface over gloo (CPU tests, several ranks on
// one GPU), where transfers are host-staged and synchronous.
#include "torch_hip.h"

#include <cstdlib>
#include <ATen/hip/HIPEvent.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/csrc/distributed/c10d/Types.hpp>

#include <map>
#include <memory>
#include <utility>
#include <vector>

namespace {

using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;
using StreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
using Event = at::cuda::CUDAEvent;

class P2PEngine {
 public:
  P2PEngine(c10::intrusive_ptr<c10d::ProcessGroup> pg, int64_t stream_id, int64_t device_index,
            int64_t device_type, bool serialize, bool timing)
      : pg_(std::move(pg)), p2p_(c10::Stream::unpack3(stream_id, (c10::DeviceIndex)device_index,
                                                      (c10::DeviceType)device_type)),
        serialize_(serialize), timing_(timing) {}

  // One RCCL group: sends[i] -> group rank send_peers[i], recvs[j] <- group rank recv_peers[j].
  // `after_compute`: fence the group behind what the compute stream has enqueued so far (needed when it
  // carries a payload produced there; pure receive groups pass false and start as early as the peer sends).
  int64_t post(std::vector<at::Tensor> sends, std::vector<int64_t> send_peers, std::vector<at::Tensor> recvs,
               std::vector<int64_t> recv_peers, bool after_compute) {
    TORCH_CHECK(sends.size() == send_peers.size() && recvs.size() == recv_peers.size(), "p2p: peers / tensors differ");
    const size_t n = sends.size() + recvs.size();
    TORCH_CHECK(n > 0, "p2p: empty group");
    const int world = pg_->getSize();
    for (auto& t : sends) TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "p2p: contiguous GPU tensors only");
    for (auto& t : recvs) TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "p2p: contiguous GPU tensors only");
    // DDL_P2P_ALLOW_SELF=1: a rank may address itself (RCCL runs a coalesced self send + receive as a local copy) -
    // the one-GPU rehearsal of this engine on the real RCCL (bench/p2p_self_probe.py, tests/test_gpu_dist.py)
    static const bool allow_self = [] { const char* e = getenv("DDL_P2P_ALLOW_SELF"); return e && e[0] == '1'; }();
    const int me = pg_->getRank();
    for (int64_t r : send_peers) TORCH_CHECK(r >= 0 && r < world && (r != me || allow_self), "p2p: bad send peer ", r);
    for (int64_t r : recv_peers) TORCH_CHECK(r >= 0 && r < world && (r != me || allow_self), "p2p: bad recv peer ", r);
    const at::Tensor& any = sends.empty() ? recvs[0] : sends[0];
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(any.device().index());
    const Stream where = serialize_ ? main : p2p_;          // --debug-sync: everything on the compute stream
    if (!serialize_ && after_compute) {
      Event ready;
      ready.record(main);
      ready.block(p2p_);
    }
    {
      StreamGuard guard(where);
      std::unique_ptr<Event> t0, t1;
      if (timing_) {
        t0 = std::make_unique<Event>(hipEventDefault);
        t0->record(where);
      }
      // always one coalesced group (like batch_isend_irecv): every transfer of the pipeline group then uses the
      // group's own communicator and ONE ordered queue - never a lazily created per-pair communicator
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
      if (timing_) {
        t1 = std::make_unique<Event>(hipEventDefault);
        t1->record(where);
        spans_.emplace_back(std::move(t0), std::move(t1));
      }
    }
    const int64_t ticket = next_++;
    if (!serialize_) {
      auto& ev = done_[ticket];
      ev = std::make_unique<Event>();
      ev->record(p2p_);
    }
    ++groups_;
    return ticket;
  }

  // the current (compute) stream waits for one posted group
  void wait(int64_t ticket) {
    auto it = done_.find(ticket);
    if (it == done_.end()) return;                            // serialised mode, or already waited for
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(p2p_.device_index());
    it->second->block(main);
    done_.erase(it);
  }

  // ... for every outstanding group (end of a step: send buffers may be overwritten afterwards)
  void drain() {
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(p2p_.device_index());
    for (auto& kv : done_) kv.second->block(main);
    done_.clear();
  }

  // Order everything posted from now on behind what the compute stream has enqueued so far.  The schedules call it
  // at the end of every step: the next step pre-posts receives into the SAME per-micro-batch slots, and nothing
  // else keeps such a receive from landing before the forward that still reads the slot has run (an evaluation
  // loop has no backward chain and no host sync; under hipGraph replay the host runs far ahead).
  void fence() {
    if (serialize_) return;
    const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(p2p_.device_index());
    Event consumed;
    consumed.record(main);
    consumed.block(p2p_);
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

 private:
  c10::intrusive_ptr<c10d::ProcessGroup> pg_;
  Stream p2p_;
  bool serialize_, timing_;
  int64_t next_ = 0, groups_ = 0, fences_ = 0;
  std::map<int64_t, std::unique_ptr<Event>> done_;
  std::vector<std::pair<std::unique_ptr<Event>, std::unique_ptr<Event>>> spans_;
};

}  // namespace

void register_p2p(pybind11::module_& m) {
  pybind11::class_<P2PEngine>(m, "P2PEngine")
      .def(pybind11::init<c10::intrusive_ptr<c10d::ProcessGroup>, int64_t, int64_t, int64_t, bool, bool>(),
           pybind11::arg("process_group"), pybind11::arg("stream_id"), pybind11::arg("device_index"),
           pybind11::arg("device_type"), pybind11::arg("serialize"), pybind11::arg("timing"))
      .def("post", &P2PEngine::post, pybind11::arg("sends"), pybind11::arg("send_peers"), pybind11::arg("recvs"),
           pybind11::arg("recv_peers"), pybind11::arg("after_compute"))
      .def("wait", &P2PEngine::wait)
      .def("drain", &P2PEngine::drain)
      .def("fence", &P2PEngine::fence)
      .def("drain_ms", &P2PEngine::drain_ms)
      .def("set_timing", &P2PEngine::set_timing)
      .def_property_readonly("groups", &P2PEngine::groups)
      .def_property_readonly("outstanding", &P2PEngine::outstanding);
}
