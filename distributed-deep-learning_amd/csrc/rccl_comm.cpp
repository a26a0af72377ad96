// Owned RCCL communicators: implementation + Python face (see rccl_comm.h for the why).
//
// Reference: the reference never touches NCCL itself - `init_process_group('nccl')` (ddp.py:29, pp.py:28,
// ddp_n_pp.py:31) and c10d own the communicators.  Here the runtime does: ncclGetUniqueId on member 0 ->
// c10d store -> ncclCommInitRank, then collectives / grouped send-recv on the engine's own HIP streams.
#include "torch_hip.h"
#include "rccl_comm.h"

#include <pybind11/stl.h>

#include <cstring>

namespace ddl {

namespace {

#define DDL_NCCL_CHECK(expr)                                                                                   \
  do {                                                                                                         \
    const ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess && r_ != ncclInProgress)                                                             \
      throw std::runtime_error(std::string("RCCL: " #expr " failed: ") + ncclGetErrorString(r_));              \
  } while (0)

struct DeviceScope {      // RCCL launches on the communicator's device; keep the caller's current device intact
  int prev = -1;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev && hipSetDevice(dev) != hipSuccess) throw std::runtime_error("RcclComm: hipSetDevice failed");
    if (prev == dev) prev = -1;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

ncclDataType_t nccl_dtype(const at::Tensor& t) {
  const int code = rccl_dtype_code((int)t.scalar_type());
  TORCH_CHECK(code >= 0, "RcclComm: unsupported dtype ", t.scalar_type());
  return (ncclDataType_t)code;
}

void check_tensor(const RcclComm& c, const at::Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "RcclComm.", what, ": contiguous GPU tensor expected");
  TORCH_CHECK(t.device().index() == c.device(), "RcclComm.", what, ": tensor lives on device ", (int)t.device().index(),
              ", the communicator on ", c.device());
}

}  // namespace

int rccl_dtype_code(int at_scalar_type) {
  switch ((at::ScalarType)at_scalar_type) {
    case at::kFloat: return (int)ncclFloat32;
    case at::kDouble: return (int)ncclFloat64;
    case at::kHalf: return (int)ncclFloat16;
    case at::kBFloat16: return (int)ncclBfloat16;
    case at::kInt: return (int)ncclInt32;
    case at::kLong: return (int)ncclInt64;
    case at::kByte: return (int)ncclUint8;
    case at::kChar: return (int)ncclInt8;
    case at::kBool: return (int)ncclUint8;
    default: return -1;
  }
}

std::string RcclComm::unique_id() {
  ncclUniqueId id;
  DDL_NCCL_CHECK(ncclGetUniqueId(&id));
  return std::string(id.internal, NCCL_UNIQUE_ID_BYTES);
}

int RcclComm::version() {
  int v = 0;
  DDL_NCCL_CHECK(ncclGetVersion(&v));
  return v;
}

RcclComm::RcclComm(const std::string& uid, int rank, int nranks, int device)
    : rank_(rank), nranks_(nranks), device_(device) {
  if (uid.size() != NCCL_UNIQUE_ID_BYTES) throw std::runtime_error("RcclComm: unique id must be 128 bytes");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("RcclComm: bad rank / nranks");
  ncclUniqueId id;
  std::memcpy(id.internal, uid.data(), NCCL_UNIQUE_ID_BYTES);
  DeviceScope scope(device_);
  DDL_NCCL_CHECK(ncclCommInitRank(&comm_, nranks_, id, rank_));
}

RcclComm::~RcclComm() {
  // Never a collective teardown from a destructor: at interpreter exit the peers may be gone already and
  // ncclCommDestroy would wait for them.  `destroy()` is the orderly path (comm.destroy()).
  if (comm_ != nullptr) {
    (void)ncclCommAbort(comm_);
    comm_ = nullptr;
  }
}

void RcclComm::destroy() {
  if (comm_ == nullptr) return;
  DeviceScope scope(device_);
  ncclComm_t c = comm_;
  comm_ = nullptr;
  DDL_NCCL_CHECK(ncclCommDestroy(c));
}

void RcclComm::abort() {
  if (comm_ == nullptr) return;
  ncclComm_t c = comm_;
  comm_ = nullptr;
  (void)ncclCommAbort(c);
}

void RcclComm::check_async() {
  live();
  ncclResult_t st = ncclSuccess;
  DDL_NCCL_CHECK(ncclCommGetAsyncError(comm_, &st));
  if (st != ncclSuccess && st != ncclInProgress)
    throw std::runtime_error(std::string("RCCL communicator in error state: ") + ncclGetErrorString(st));
}

void RcclComm::all_reduce(void* buf, size_t count, ncclDataType_t dt, ncclRedOp_t op, hipStream_t s) {
  live();
  DeviceScope scope(device_);
  DDL_NCCL_CHECK(ncclAllReduce(buf, buf, count, dt, op, comm_, s));
  ++launches_;
}

void RcclComm::reduce_scatter(const void* full, void* shard, size_t shard_count, ncclDataType_t dt, ncclRedOp_t op,
                              hipStream_t s) {
  live();
  DeviceScope scope(device_);
  DDL_NCCL_CHECK(ncclReduceScatter(full, shard, shard_count, dt, op, comm_, s));
  ++launches_;
}

void RcclComm::all_gather(const void* shard, void* full, size_t shard_count, ncclDataType_t dt, hipStream_t s) {
  live();
  DeviceScope scope(device_);
  DDL_NCCL_CHECK(ncclAllGather(shard, full, shard_count, dt, comm_, s));
  ++launches_;
}

void RcclComm::broadcast(void* buf, size_t count, ncclDataType_t dt, int root, hipStream_t s) {
  live();
  if (root < 0 || root >= nranks_) throw std::runtime_error("RcclComm.broadcast: root out of range");
  DeviceScope scope(device_);
  DDL_NCCL_CHECK(ncclBroadcast(buf, buf, count, dt, root, comm_, s));
  ++launches_;
}

void RcclComm::group(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t s) {
  live();
  if (sends.empty() && recvs.empty()) return;
  for (const Xfer& x : sends)
    if (x.peer < 0 || x.peer >= nranks_) throw std::runtime_error("RcclComm.group: send peer out of range");
  for (const Xfer& x : recvs)
    if (x.peer < 0 || x.peer >= nranks_) throw std::runtime_error("RcclComm.group: recv peer out of range");
  DeviceScope scope(device_);
  DDL_NCCL_CHECK(ncclGroupStart());
  ncclResult_t first_error = ncclSuccess;
  for (const Xfer& x : sends) {
    const ncclResult_t r = ncclSend(x.ptr, x.count, x.dt, x.peer, comm_, s);
    if (r != ncclSuccess && r != ncclInProgress && first_error == ncclSuccess) first_error = r;
  }
  for (const Xfer& x : recvs) {
    const ncclResult_t r = ncclRecv(x.ptr, x.count, x.dt, x.peer, comm_, s);
    if (r != ncclSuccess && r != ncclInProgress && first_error == ncclSuccess) first_error = r;
  }
  const ncclResult_t end = ncclGroupEnd();      // always closed, even after a failed enqueue
  if (first_error != ncclSuccess)
    throw std::runtime_error(std::string("RCCL: send / recv enqueue failed: ") + ncclGetErrorString(first_error));
  DDL_NCCL_CHECK(end);
  ++launches_;
}

}  // namespace ddl

// ---- Python face: tensor-level calls on PyTorch's CURRENT stream (tests, the first-contact ladder, the ping) -------
void register_rccl(pybind11::module_& m) {
  using ddl::RcclComm;
  namespace py = pybind11;
  m.def("rccl_unique_id", [] { return py::bytes(RcclComm::unique_id()); },
        "128-byte id for RcclComm (generate on member 0, ship to the others)");
  m.def("rccl_version", &RcclComm::version);
  py::class_<RcclComm, std::shared_ptr<RcclComm>>(m, "RcclComm")
      .def(py::init([](py::bytes uid, int rank, int nranks, int device) {
             const std::string id = uid;              // (converted while the GIL is still held)
             py::gil_scoped_release release;          // ncclCommInitRank blocks until every member has joined
             return std::make_shared<RcclComm>(id, rank, nranks, device);
           }),
           py::arg("unique_id"), py::arg("rank"), py::arg("nranks"), py::arg("device"))
      .def_property_readonly("rank", &RcclComm::rank)
      .def_property_readonly("nranks", &RcclComm::nranks)
      .def_property_readonly("device", &RcclComm::device)
      .def_property_readonly("launches", &RcclComm::launches)
      .def("all_reduce",
           [](RcclComm& c, at::Tensor t, bool avg) {
             ddl::check_tensor(c, t, "all_reduce");
             c.all_reduce(t.data_ptr(), (size_t)t.numel(), ddl::nccl_dtype(t), avg ? ncclAvg : ncclSum, ddl::cur_stream());
           },
           py::arg("tensor"), py::arg("avg") = false)
      .def("all_reduce_max",
           [](RcclComm& c, at::Tensor t) {
             ddl::check_tensor(c, t, "all_reduce_max");
             c.all_reduce(t.data_ptr(), (size_t)t.numel(), ddl::nccl_dtype(t), ncclMax, ddl::cur_stream());
           })
      .def("reduce_scatter",
           [](RcclComm& c, at::Tensor shard, at::Tensor full, bool avg) {
             ddl::check_tensor(c, shard, "reduce_scatter");
             ddl::check_tensor(c, full, "reduce_scatter");
             TORCH_CHECK(full.numel() == shard.numel() * c.nranks() && full.scalar_type() == shard.scalar_type(),
                         "reduce_scatter: full must hold nranks shards");
             c.reduce_scatter(full.data_ptr(), shard.data_ptr(), (size_t)shard.numel(), ddl::nccl_dtype(full),
                              avg ? ncclAvg : ncclSum, ddl::cur_stream());
           },
           py::arg("shard"), py::arg("full"), py::arg("avg") = false)
      .def("all_gather",
           [](RcclComm& c, at::Tensor full, at::Tensor shard) {
             ddl::check_tensor(c, shard, "all_gather");
             ddl::check_tensor(c, full, "all_gather");
             TORCH_CHECK(full.numel() == shard.numel() * c.nranks() && full.scalar_type() == shard.scalar_type(),
                         "all_gather: full must hold nranks shards");
             c.all_gather(shard.data_ptr(), full.data_ptr(), (size_t)shard.numel(), ddl::nccl_dtype(full), ddl::cur_stream());
           },
           py::arg("full"), py::arg("shard"))
      .def("broadcast",
           [](RcclComm& c, at::Tensor t, int root) {
             ddl::check_tensor(c, t, "broadcast");
             c.broadcast(t.data_ptr(), (size_t)t.numel(), ddl::nccl_dtype(t), root, ddl::cur_stream());
           },
           py::arg("tensor"), py::arg("root"))
      .def("group",
           [](RcclComm& c, std::vector<at::Tensor> sends, std::vector<int> send_peers, std::vector<at::Tensor> recvs,
              std::vector<int> recv_peers) {
             TORCH_CHECK(sends.size() == send_peers.size() && recvs.size() == recv_peers.size(), "group: peers / tensors differ");
             std::vector<RcclComm::Xfer> s, r;
             for (size_t i = 0; i < sends.size(); ++i) {
               ddl::check_tensor(c, sends[i], "group");
               s.push_back({sends[i].data_ptr(), (size_t)sends[i].numel(), ddl::nccl_dtype(sends[i]), send_peers[i]});
             }
             for (size_t i = 0; i < recvs.size(); ++i) {
               ddl::check_tensor(c, recvs[i], "group");
               r.push_back({recvs[i].data_ptr(), (size_t)recvs[i].numel(), ddl::nccl_dtype(recvs[i]), recv_peers[i]});
             }
             c.group(s, r, ddl::cur_stream());
           },
           py::arg("sends"), py::arg("send_peers"), py::arg("recvs"), py::arg("recv_peers"))
      .def("check_async", &RcclComm::check_async)
      .def("destroy", &RcclComm::destroy, py::call_guard<py::gil_scoped_release>())
      .def("abort", &RcclComm::abort);
}
