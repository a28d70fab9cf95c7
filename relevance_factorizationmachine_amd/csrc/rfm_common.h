// Shared host-side plumbing of librfm_hip.so: error reporting across the C ABI,
// the context object and small device-memory helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rfm_hip.h"

namespace rfm {

struct Error : std::runtime_error {
  int32_t code;
  Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);

[[noreturn]] inline void fail(int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

#define RFM_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      ::rfm::fail(RFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                  __FILE__, __LINE__);                                                   \
  } while (0)

#define RFM_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) ::rfm::fail(RFM_ERR_BAD_ARG, __VA_ARGS__);      \
  } while (0)

// Runs fn, maps exceptions to the ABI's int32 error classes.
template <class Fn>
inline int32_t guarded(Fn&& fn) {
  try {
    fn();
    return RFM_OK;
  } catch (const Error& e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return RFM_ERR_INTERNAL;
  } catch (...) {
    set_last_error("unknown error");
    return RFM_ERR_INTERNAL;
  }
}

// plan-/ctx-owned device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  void alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    RFM_HIP_CHECK(hipMalloc(&p, n));
    bytes = n;
  }
  void ensure(size_t n) {
    if (n > bytes) alloc(n);
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

// ---------------------------------------------------------------------------
// dispatch on the factor count: a row is handled by `lpr` consecutive lanes, each
// owning `vec` adjacent factors in `nc` chunks (kernels are templates on the three)
// ---------------------------------------------------------------------------
struct Shape {
  int lpr, vec, nc;
};

inline Shape shape_for(int k) {
  RFM_REQUIRE(k >= 1 && k <= RFM_MAX_FACTORS, "n_factors=%d unsupported (1..%d)", k,
              RFM_MAX_FACTORS);
  Shape s;
  s.vec = (k % 2 == 0) ? 2 : 1;
  const int units = (k + s.vec - 1) / s.vec;
  int lpr = 4;
  while (lpr < units && lpr < 64) lpr *= 2;
  s.lpr = lpr;
  int nc = 1;
  while (lpr * nc < units) nc *= 2;
  // three chunks where they cover the row (k = 258 .. 384: the reference's Coat runs use 300):
  // a fourth would be loaded and masked
  if (nc == 4 && lpr * 3 >= units) nc = 3;
  s.nc = nc;
  return s;
}

#define RFM_FOR_SHAPE(S, CALL)                                                         \
  do {                                                                                 \
    const ::rfm::Shape _s = (S);                                                       \
    if (_s.vec == 2) {                                                                 \
      if (_s.nc == 1) {                                                                \
        switch (_s.lpr) {                                                              \
          case 4: CALL(4, 2, 1); break;                                                \
          case 8: CALL(8, 2, 1); break;                                                \
          case 16: CALL(16, 2, 1); break;                                              \
          case 32: CALL(32, 2, 1); break;                                              \
          default: CALL(64, 2, 1); break;                                              \
        }                                                                              \
      } else if (_s.nc == 2) { CALL(64, 2, 2); }                                       \
      else if (_s.nc == 3) { CALL(64, 2, 3); }                                         \
      else if (_s.nc == 4) { CALL(64, 2, 4); }                                         \
      else { CALL(64, 2, 8); }                                                         \
    } else {                                                                           \
      if (_s.nc == 1) {                                                                \
        switch (_s.lpr) {                                                              \
          case 4: CALL(4, 1, 1); break;                                                \
          case 8: CALL(8, 1, 1); break;                                                \
          case 16: CALL(16, 1, 1); break;                                              \
          case 32: CALL(32, 1, 1); break;                                              \
          default: CALL(64, 1, 1); break;                                              \
        }                                                                              \
      } else if (_s.nc == 2) { CALL(64, 1, 2); }                                       \
      else if (_s.nc == 3) { CALL(64, 1, 3); }                                         \
      else if (_s.nc == 4) { CALL(64, 1, 4); }                                         \
      else if (_s.nc == 8) { CALL(64, 1, 8); }                                         \
      else { CALL(64, 1, 16); }                                                        \
    }                                                                                  \
  } while (0)

// the single-chunk shapes only (callers that handle several chunks per lane another way)
#define RFM_FOR_SINGLE_CHUNK_SHAPE(S, CALL)                                            \
  do {                                                                                 \
    const ::rfm::Shape _s = (S);                                                       \
    if (_s.nc != 1) ::rfm::fail(RFM_ERR_INTERNAL, "single-chunk dispatch of nc=%d", _s.nc); \
    if (_s.vec == 2) {                                                                 \
      switch (_s.lpr) {                                                                \
        case 4: CALL(4, 2, 1); break;                                                  \
        case 8: CALL(8, 2, 1); break;                                                  \
        case 16: CALL(16, 2, 1); break;                                                \
        case 32: CALL(32, 2, 1); break;                                                \
        default: CALL(64, 2, 1); break;                                                \
      }                                                                                \
    } else {                                                                           \
      switch (_s.lpr) {                                                                \
        case 4: CALL(4, 1, 1); break;                                                  \
        case 8: CALL(8, 1, 1); break;                                                  \
        case 16: CALL(16, 1, 1); break;                                                \
        case 32: CALL(32, 1, 1); break;                                                \
        default: CALL(64, 1, 1); break;                                                \
      }                                                                                \
    }                                                                                  \
  } while (0)

}  // namespace rfm

struct rfm_ctx;
namespace rfm {
// RCCL collectives on the ctx's communicator and stream (rfm_comm.cpp)
void comm_all_gather(rfm_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank);
void comm_all_reduce_sum(rfm_ctx* ctx, double* d_buf, int64_t count);
void comm_all_to_all(rfm_ctx* ctx, int rank, const void* d_send, const int64_t* send_off,
                     const int64_t* send_bytes, void* d_recv, const int64_t* recv_off,
                     const int64_t* recv_bytes);
}  // namespace rfm

struct rfm_ctx {
  int32_t device = 0;
  hipStream_t stream = nullptr;
  int32_t n_cu = 256;
  rfm::DevBuf loss_partials;  // per-block partial sums of the loss reduction
  void* comm = nullptr;       // RCCL communicator (rfm_comm_init), or null
  int32_t comm_ranks = 0;
  int32_t comm_rank = 0;
  // per-kernel timing (rfm_profile_begin/end): 4 events per recorded step
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;
  void prof_mark() {
    if (!profiling) return;
    hipEvent_t e;
    RFM_HIP_CHECK(hipEventCreate(&e));
    RFM_HIP_CHECK(hipEventRecord(e, stream));
    prof_events.push_back(e);
  }
};
