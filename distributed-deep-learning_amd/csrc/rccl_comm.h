// Owned RCCL communicators ("a C++ extension owning the flat arena, bucket table, streams and events").
//
// Reference: every collective of the reference goes through c10d's ProcessGroupNCCL (ddp.py:29,127;
// pp.py:138-150 `batch_isend_irecv`; ddp_n_pp.py:32-33) - which runs the RCCL kernels on ITS internal stream per
// communicator and fences that stream against the caller's stream on both sides.  A "dedicated communication
// stream" above c10d therefore only carries fences: two extra event hops and a Work object per bucket / group,
// every P2P group of a stage FIFO on one internal stream (a receive pre-posted toward one neighbour sits in front
// of the send toward the other), and collectives that cannot live on a stream the engine controls.
//
// `RcclComm` is one ncclComm_t created with `ncclCommInitRank` from a unique id that rank 0 of the member list
// published through the c10d store (comm.py `direct_comm`).  Collectives and grouped send / receive are enqueued
// DIRECTLY on the hipStream_t the caller names - the bucket reducer's communication stream, one stream per pipeline
// neighbour in the P2P engine - so a fence recorded on that stream really orders the RCCL kernel.
// One process per GPU; xGMI is point-to-point, so a 2-rank communicator per neighbour pair maps onto exactly one link.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ddl {

class RcclComm {
 public:
  // 128 opaque bytes for `ncclCommInitRank`; generated on member 0, shipped to the others by the caller
  static std::string unique_id();
  static int version();

  // collective over the `nranks` members: every member calls it with the same id (blocks until all have joined)
  RcclComm(const std::string& uid, int rank, int nranks, int device);
  ~RcclComm();
  RcclComm(const RcclComm&) = delete;
  RcclComm& operator=(const RcclComm&) = delete;

  int rank() const { return rank_; }
  int nranks() const { return nranks_; }
  int device() const { return device_; }
  int64_t launches() const { return launches_; }

  // all in place capable, all enqueued on `s`; counts in elements of `dt`
  void all_reduce(void* buf, size_t count, ncclDataType_t dt, ncclRedOp_t op, hipStream_t s);
  void reduce_scatter(const void* full, void* shard, size_t shard_count, ncclDataType_t dt, ncclRedOp_t op, hipStream_t s);
  void all_gather(const void* shard, void* full, size_t shard_count, ncclDataType_t dt, hipStream_t s);
  void broadcast(void* buf, size_t count, ncclDataType_t dt, int root, hipStream_t s);

  struct Xfer { void* ptr; size_t count; ncclDataType_t dt; int peer; };
  // ONE group: sends and receives complete together (opposite-direction transfers between one pair issued as two
  // calls in different orders on the two sides would deadlock - SURVEY hard part 5)
  void group(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t s);

  // asynchronous error state of the communicator (a peer that died, a transport error): throws when not ncclSuccess
  void check_async();
  // orderly teardown (flushes, collective over the members) / abort (local, never blocks)
  void destroy();
  void abort();

 private:
  void live() const {
    if (comm_ == nullptr) throw std::runtime_error("RcclComm: communicator already destroyed");
  }
  ncclComm_t comm_ = nullptr;
  int rank_, nranks_, device_;
  int64_t launches_ = 0;
};

// element type of a tensor for RCCL (declared here, defined in rccl_comm.cpp; -1 = unsupported)
int rccl_dtype_code(int at_scalar_type);

}  // namespace ddl
