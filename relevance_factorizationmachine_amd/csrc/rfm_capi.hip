// Context management of librfm_hip.so (rfm_create / rfm_destroy / rfm_sync).
#include "rfm_common.h"

using namespace rfm;

extern "C" {

int32_t rfm_create(int32_t device, void* hip_stream, rfm_ctx** out) {
  return guarded([&] {
    RFM_REQUIRE(out, "null output pointer");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      fail(RFM_ERR_NO_DEVICE, "no HIP device visible (%s)",
           e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    RFM_REQUIRE(device >= 0 && device < count, "device %d out of range (0..%d)", device,
                count - 1);
    RFM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    RFM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    auto* ctx = new rfm_ctx();
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = ctx;
  });
}

int32_t rfm_destroy(rfm_ctx* ctx) {
  return guarded([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)rfm_comm_destroy(ctx);
    delete ctx;
  });
}

int32_t rfm_profile_begin(rfm_ctx* ctx) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    ctx->profiling = true;
  });
}

int32_t rfm_profile_end(rfm_ctx* ctx, double* h_ms, int64_t* h_count) {
  return guarded([&] {
    RFM_REQUIRE(ctx && h_ms && h_count, "null pointer");
    ctx->profiling = false;
    RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) {
      h_ms[i] = 0.0;
      h_count[i] = 0;
    }
    const size_t steps = ctx->prof_events.size() / 4;
    for (size_t s = 0; s < steps; ++s) {
      const hipEvent_t* e = &ctx->prof_events[4 * s];
      for (int ph = 0; ph < 3; ++ph) {
        float ms = 0.f;
        RFM_HIP_CHECK(hipEventElapsedTime(&ms, e[ph], e[ph + 1]));
        h_ms[ph] += ms;
        h_count[ph]++;
      }
      float ms = 0.f;
      RFM_HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[3]));
      h_ms[3] += ms;
      h_count[3]++;
    }
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
  });
}

int32_t rfm_sync(rfm_ctx* ctx) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

// plain copies for callers that hold raw device pointers only (a transport of rfm_fm_fit_dp
// that stages through the host): ordered after the ctx stream's work, complete on return
int32_t rfm_copy_to_host(rfm_ctx* ctx, void* h_dst, const void* d_src, int64_t bytes) {
  return guarded([&] {
    RFM_REQUIRE(ctx && bytes >= 0 && (bytes == 0 || (h_dst && d_src)), "bad arguments");
    if (bytes == 0) return;
    RFM_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, size_t(bytes), hipMemcpyDeviceToHost, ctx->stream));
    RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

int32_t rfm_copy_to_device(rfm_ctx* ctx, void* d_dst, const void* h_src, int64_t bytes) {
  return guarded([&] {
    RFM_REQUIRE(ctx && bytes >= 0 && (bytes == 0 || (d_dst && h_src)), "bad arguments");
    if (bytes == 0) return;
    RFM_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, size_t(bytes), hipMemcpyHostToDevice, ctx->stream));
    RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"
