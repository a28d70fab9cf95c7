// CSR assembly on the device: the step before the training path.
//
// The reference builds its FM design matrix on the host with pandas / SciPy: one-hot user
// (+) one-hot item (+) per-interaction columns (+) the user's feature row (+) the item's
// feature row, `hstack`ed (utils/dataloader/kuairec/_feature.py:54-84,201-207; Coat:
// utils/dataloader/coat/_preparer.py:154-168), then picks rows of it by index for the
// train / val / test splits and the negatively sampled subsets
// (kuairec/_preparer.py:117-136, loader.py:104-115).  Both are the same operation: every
// output row is the concatenation of a few SEGMENTS, each either a one-hot of an id or a
// row of a (small) CSR block chosen by an id, shifted to the segment's first column.
// Here: one pass that adds up the rows' lengths, a scan, one pass that writes the entries.
// Byte / index work bound by HBM: no arithmetic on the values at all.
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "rfm_common.h"

namespace rfm {
namespace {

constexpr int kCsrBlock = 256;
constexpr int kMaxSegments = 16;

struct Seg {            // device view of one rfm_csr_segment
  int32_t kind;         // 0: one-hot of ids[r]; 1: row ids[r] (or r) of a CSR block
  const int32_t* ids;   // nullable for kind 1: the block's row r itself
  const int64_t* indptr;
  const int32_t* indices;
  const double* values;
  int64_t col_offset;
  int64_t n_block_rows;  // kind 1: rows of the block; kind 0: size of the one-hot (ids < it)
};

struct SegList {
  Seg s[kMaxSegments];
  int32_t n;
};

// len[r] = entries of output row r; flags[0] = an id outside its block / one-hot range
__global__ __launch_bounds__(kCsrBlock) void csr_count_kernel(SegList segs, int64_t n_rows,
                                                             int64_t* len, int32_t* flags) {
  for (int64_t r = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x; r < n_rows;
       r += int64_t(gridDim.x) * kCsrBlock) {
    int64_t total = 0;
    for (int i = 0; i < segs.n; ++i) {
      const Seg& s = segs.s[i];
      const int64_t id = s.ids ? int64_t(s.ids[r]) : r;
      if (id < 0 || id >= s.n_block_rows) {
        atomicOr(&flags[0], 1);
        continue;
      }
      total += s.kind == 0 ? 1 : s.indptr[id + 1] - s.indptr[id];
    }
    len[r] = total;
  }
}

__global__ __launch_bounds__(kCsrBlock) void csr_fill_kernel(SegList segs, int64_t n_rows,
                                                            const int64_t* out_indptr,
                                                            int32_t* out_indices,
                                                            double* out_values) {
  for (int64_t r = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x; r < n_rows;
       r += int64_t(gridDim.x) * kCsrBlock) {
    int64_t at = out_indptr[r];
    for (int i = 0; i < segs.n; ++i) {
      const Seg& s = segs.s[i];
      const int64_t id = s.ids ? int64_t(s.ids[r]) : r;
      if (id < 0 || id >= s.n_block_rows) continue;
      if (s.kind == 0) {
        out_indices[at] = int32_t(s.col_offset + id);
        out_values[at] = 1.0;
        ++at;
      } else {
        for (int64_t q = s.indptr[id]; q < s.indptr[id + 1]; ++q, ++at) {
          out_indices[at] = int32_t(s.col_offset + s.indices[q]);
          out_values[at] = s.values[q];
        }
      }
    }
  }
}

SegList to_list(const rfm_csr_segment* h_segments, int32_t n_segments) {
  RFM_REQUIRE(h_segments && n_segments >= 1 && n_segments <= kMaxSegments,
              "n_segments=%d outside 1..%d", n_segments, kMaxSegments);
  SegList l{};
  l.n = n_segments;
  for (int i = 0; i < n_segments; ++i) {
    const rfm_csr_segment& h = h_segments[i];
    RFM_REQUIRE(h.kind == 0 || h.kind == 1, "segment %d: kind %d", i, h.kind);
    RFM_REQUIRE(h.n_block_rows >= 0 && h.col_offset >= 0, "segment %d: negative size / offset", i);
    if (h.kind == 0)
      RFM_REQUIRE(h.d_ids, "segment %d: a one-hot needs ids", i);
    else
      RFM_REQUIRE(h.d_indptr && (h.d_indices || true), "segment %d: null CSR block", i);
    l.s[i] = Seg{h.kind,       h.d_ids,        h.d_indptr,    h.d_indices,
                 h.d_values,   h.col_offset,   h.n_block_rows};
  }
  return l;
}

int grid_for(const rfm_ctx* ctx, int64_t items) {
  return int(std::max<int64_t>(1, std::min<int64_t>((items + kCsrBlock - 1) / kCsrBlock,
                                                    int64_t(ctx->n_cu) * 16)));
}

}  // namespace
}  // namespace rfm

using namespace rfm;

extern "C" {

int32_t rfm_csr_assemble_count(rfm_ctx* ctx, const rfm_csr_segment* h_segments, int32_t n_segments,
                               int64_t n_rows, int64_t* d_out_indptr, int64_t* h_out_nnz) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_out_indptr && h_out_nnz, "null pointer");
    RFM_REQUIRE(n_rows >= 0, "negative n_rows");
    const SegList segs = to_list(h_segments, n_segments);
    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf len, flags, temp;
    len.alloc(size_t(n_rows + 1) * 8);
    flags.alloc(4);
    RFM_HIP_CHECK(hipMemsetAsync(flags.p, 0, 4, st));
    RFM_HIP_CHECK(hipMemsetAsync(len.as<int64_t>() + n_rows, 0, 8, st));
    if (n_rows > 0) {
      hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for(ctx, n_rows)), dim3(kCsrBlock), 0, st, segs,
                         n_rows, len.as<int64_t>(), flags.as<int32_t>());
      RFM_HIP_CHECK(hipGetLastError());
    }
    // indptr = exclusive scan of the lengths (n_rows + 1 values: the last is the total)
    size_t temp_bytes = 0;
    RFM_HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, len.as<int64_t>(), d_out_indptr,
                                          int64_t(0), size_t(n_rows + 1), rocprim::plus<int64_t>(), st));
    temp.alloc(temp_bytes);
    RFM_HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, len.as<int64_t>(), d_out_indptr,
                                          int64_t(0), size_t(n_rows + 1), rocprim::plus<int64_t>(), st));
    int32_t h_flag = 0;
    RFM_HIP_CHECK(hipMemcpyAsync(&h_flag, flags.p, 4, hipMemcpyDeviceToHost, st));
    RFM_HIP_CHECK(hipMemcpyAsync(h_out_nnz, d_out_indptr + n_rows, 8, hipMemcpyDeviceToHost, st));
    RFM_HIP_CHECK(hipStreamSynchronize(st));
    RFM_REQUIRE(!h_flag, "an id lies outside its block (or one-hot range)");
    RFM_REQUIRE(*h_out_nnz < (int64_t(1) << 31), "nnz=%lld does not fit int32 entry offsets",
                (long long)*h_out_nnz);
  });
}

int32_t rfm_csr_assemble_fill(rfm_ctx* ctx, const rfm_csr_segment* h_segments, int32_t n_segments,
                              int64_t n_rows, const int64_t* d_indptr, int32_t* d_out_indices,
                              double* d_out_values) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_indptr, "null pointer");
    RFM_REQUIRE(n_rows >= 0, "negative n_rows");
    const SegList segs = to_list(h_segments, n_segments);
    if (n_rows == 0) return;
    RFM_REQUIRE(d_out_indices && d_out_values, "null output arrays");
    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for(ctx, n_rows)), dim3(kCsrBlock), 0, ctx->stream,
                       segs, n_rows, d_indptr, d_out_indices, d_out_values);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
