// Direct RCCL binding of librfm_hip.so: the all-reduce(sum) of the dense FM
// gradient buffer over xGMI for callers that do not go through
// torch.distributed.  librccl.so is opened on first use (dlopen), so the library
// itself loads on hosts without RCCL.
#include <dlfcn.h>

#include "rfm_common.h"

namespace {

// the handful of RCCL declarations used (rccl.h: ncclGetUniqueId, ncclCommInitRank,
// ncclAllReduce, ncclAllGather, ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd,
// ncclCommDestroy, ncclGetErrorString; ncclInt8 = 0, ncclFloat64 = 8, ncclSum = 0)
constexpr int kNcclUniqueIdBytes = 128;
struct UniqueId {
  char internal[kNcclUniqueIdBytes];
};
using Comm = void*;
struct Rccl {
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

const Rccl& rccl() {
  static Rccl api = [] {
    Rccl a;
    // an RCCL the process has already mapped (PyTorch's own, bound to PyTorch's HIP runtime)
    // is the one to use: a second copy would bring a second HIP runtime with it, which knows
    // nothing of the caller's allocations.  Only a process without one (no PyTorch) loads
    // the system's; RFM_RCCL_PATH names a specific library.
    void* h = nullptr;
    if (const char* path = getenv("RFM_RCCL_PATH")) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    const bool torch_here = dlopen("libtorch_hip.so", RTLD_NOW | RTLD_NOLOAD) != nullptr ||
                            dlopen("libtorch_cpu.so", RTLD_NOW | RTLD_NOLOAD) != nullptr;
    if (!h && !torch_here) {
      h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h)
      rfm::fail(RFM_ERR_INTERNAL, "cannot bind RCCL (%s)",
                torch_here ? "PyTorch is loaded but its librccl is not mapped" : dlerror());
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.AllGather ||
        !a.Send || !a.Recv || !a.GroupStart || !a.GroupEnd)
      rfm::fail(RFM_ERR_INTERNAL, "librccl.so lacks an expected symbol");
    return a;
  }();
  return api;
}

void check_rccl(int rc, const char* what) {
  if (rc != 0) {
    const Rccl& a = rccl();
    rfm::fail(RFM_ERR_HIP, "%s failed: %s", what, a.GetErrorString ? a.GetErrorString(rc) : "?");
  }
}

}  // namespace

// the collectives of the data-parallel fit loop (rfm_fm_fit_dp) on the ctx's communicator,
// all enqueued on the ctx stream
namespace rfm {

void comm_all_gather(rfm_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank) {
  RFM_REQUIRE(ctx->comm, "rfm_comm_init has not been called");
  if (bytes_per_rank <= 0) return;
  check_rccl(rccl().AllGather(d_send, d_recv, size_t(bytes_per_rank), /*ncclInt8*/ 0, ctx->comm,
                              ctx->stream),
             "ncclAllGather");
}

void comm_all_reduce_sum(rfm_ctx* ctx, double* d_buf, int64_t count) {
  RFM_REQUIRE(ctx->comm, "rfm_comm_init has not been called");
  if (count <= 0) return;
  check_rccl(rccl().AllReduce(d_buf, d_buf, size_t(count), /*ncclFloat64*/ 8, /*ncclSum*/ 0,
                              ctx->comm, ctx->stream),
             "ncclAllReduce");
}

// peer p gets d_send[send_off[p] .. + send_bytes[p]) and its block lands at d_recv + recv_off[p];
// the rank's own block is a device copy.  xGMI is point to point: the sends and receives of
// one exchange are fused in one group so that all links carry traffic at once.
void comm_all_to_all(rfm_ctx* ctx, int rank, const void* d_send, const int64_t* send_off,
                     const int64_t* send_bytes, void* d_recv, const int64_t* recv_off,
                     const int64_t* recv_bytes) {
  RFM_REQUIRE(ctx->comm, "rfm_comm_init has not been called");
  const Rccl& a = rccl();
  const char* src = static_cast<const char*>(d_send);
  char* dst = static_cast<char*>(d_recv);
  RFM_REQUIRE(send_bytes[rank] == recv_bytes[rank], "own block: %lld bytes sent, %lld expected",
              (long long)send_bytes[rank], (long long)recv_bytes[rank]);
  if (send_bytes[rank] > 0 && src + send_off[rank] != dst + recv_off[rank])
    RFM_HIP_CHECK(hipMemcpyAsync(dst + recv_off[rank], src + send_off[rank], size_t(send_bytes[rank]),
                                 hipMemcpyDeviceToDevice, ctx->stream));
  bool any = false;
  for (int p = 0; p < ctx->comm_ranks; ++p)
    any = any || (p != rank && (send_bytes[p] > 0 || recv_bytes[p] > 0));
  if (!any) return;
  check_rccl(a.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < ctx->comm_ranks; ++p) {
    if (p == rank) continue;
    if (send_bytes[p] > 0)
      check_rccl(a.Send(src + send_off[p], size_t(send_bytes[p]), 0, p, ctx->comm, ctx->stream), "ncclSend");
    if (recv_bytes[p] > 0)
      check_rccl(a.Recv(dst + recv_off[p], size_t(recv_bytes[p]), 0, p, ctx->comm, ctx->stream), "ncclRecv");
  }
  check_rccl(a.GroupEnd(), "ncclGroupEnd");
}

}  // namespace rfm

using namespace rfm;

extern "C" {

int32_t rfm_comm_unique_id(uint8_t* h_out128) {
  return guarded([&] {
    RFM_REQUIRE(h_out128, "null pointer");
    UniqueId id;
    check_rccl(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(h_out128, id.internal, kNcclUniqueIdBytes);
  });
}

int32_t rfm_comm_init(rfm_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* h_id128) {
  return guarded([&] {
    RFM_REQUIRE(ctx && h_id128, "null pointer");
    RFM_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank %d of %d", rank, n_ranks);
    RFM_REQUIRE(ctx->comm == nullptr, "communicator already initialised");
    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    UniqueId id;
    std::memcpy(id.internal, h_id128, kNcclUniqueIdBytes);
    Comm comm = nullptr;
    check_rccl(rccl().CommInitRank(&comm, n_ranks, id, rank), "ncclCommInitRank");
    ctx->comm = comm;
    ctx->comm_ranks = n_ranks;
    ctx->comm_rank = rank;
  });
}

int32_t rfm_allreduce_sum(rfm_ctx* ctx, double* d_buf, int64_t count) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_buf, "null pointer");
    RFM_REQUIRE(ctx->comm, "rfm_comm_init has not been called");
    RFM_REQUIRE(count >= 0, "negative count");
    if (count == 0) return;
    check_rccl(rccl().AllReduce(d_buf, d_buf, size_t(count), /*ncclFloat64*/ 8, /*ncclSum*/ 0,
                                ctx->comm, ctx->stream),
               "ncclAllReduce");
  });
}

int32_t rfm_comm_destroy(rfm_ctx* ctx) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    if (!ctx->comm) return;
    (void)hipStreamSynchronize(ctx->stream);
    check_rccl(rccl().CommDestroy(ctx->comm), "ncclCommDestroy");
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
  });
}

}  // extern "C"
