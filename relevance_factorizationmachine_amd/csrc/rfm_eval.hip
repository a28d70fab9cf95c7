// Per-iteration validation metric on the device: IPS-DCG@k over the rows of a
// validation frame grouped by user (what the reference's ValEvaluator.evaluate,
// utils/evaluate.py:183-207, computes on the host with pandas + NumPy from the
// scores of src/fm.py:104-110 / src/mf.py:126-132).
//
// One wavefront owns one user's rows.  The reference ranks them with
// `scores.argsort()[::-1]`; the k best rows are found here by k selection rounds
// over the segment: round r picks the greatest (score, position) pair that is
// lexicographically below the pair picked in round r-1, so nothing has to be
// sorted or flagged.  Equal scores therefore rank the LATER row first -- the order
// `argsort(kind="stable")[::-1]` gives.  NumPy's default sort is not stable and
// where it leaves equal scores depends on the host CPU, so the kernel also reports
// the users for whom that order can change the value (a tie that reaches into the
// first k ranks between rows of different label or propensity -- saturated sigmoid
// scores make these common -- or a NaN score); the caller redoes exactly those users
// on the host with NumPy's own sort, which keeps fit() identical to the reference's.
#include "rfm_common.h"

using namespace rfm;

namespace rfm {

constexpr int kEvalBlock = 256;
constexpr int kEvalWave = 64;

struct Pick {
  double s;
  int32_t j;
};

__device__ __forceinline__ bool pick_before(const Pick& a, const Pick& b) {
  // a ranks ahead of b
  return a.s > b.s || (a.s == b.s && a.j > b.j);
}

__global__ __launch_bounds__(kEvalBlock) void val_dcg_users_kernel(
    const double* __restrict__ scores, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ rows, const double* __restrict__ labels,
    const double* __restrict__ pscores, int32_t n_seg, int32_t k, double* __restrict__ user_val,
    double* __restrict__ user_ok, double* __restrict__ user_amb) {
  const int lane = threadIdx.x % kEvalWave;
  const int u = int(blockIdx.x) * (kEvalBlock / kEvalWave) + threadIdx.x / kEvalWave;
  if (u >= n_seg) return;
  const int32_t b = seg_ptr[u], e = seg_ptr[u + 1];

  // users without a positive label are left out of the mean (evaluate.py:199-200)
  double ysum = 0.0;
  for (int32_t j = b + lane; j < e; j += kEvalWave) ysum += labels[j];
#pragma unroll
  for (int off = kEvalWave / 2; off >= 1; off >>= 1) ysum += __shfl_xor(ysum, off);
  if (e <= b || ysum == 0.0) {
    if (lane == 0) {
      user_val[u] = 0.0;
      user_ok[u] = 0.0;
      user_amb[u] = 0.0;
    }
    return;
  }

  Pick last{__builtin_huge_val(), 0x7fffffff};
  double last_y = 0.0, last_p = 1.0;
  double head = 0.0, tail = 0.0;
  bool amb = false;
  const int rounds = min(k, e - b);
  for (int r = 0; r < rounds; ++r) {
    Pick best{-__builtin_huge_val(), -1};
    for (int32_t j = b + lane; j < e; j += kEvalWave) {
      const Pick c{scores[rows ? rows[j] : j], j};
      if (pick_before(last, c) && (best.j < 0 || pick_before(c, best))) best = c;
    }
#pragma unroll
    for (int off = kEvalWave / 2; off >= 1; off >>= 1) {
      Pick o;
      o.s = __shfl_xor(best.s, off);
      o.j = __shfl_xor(best.j, off);
      if (o.j >= 0 && (best.j < 0 || pick_before(o, best))) best = o;
    }
    if (best.j < 0) break;  // only NaN scores are left (the user is reported below)
    // utils/metrics.py:70-78: y[0]/p[0] + sum_{r>=1} y[r] / (p[r] * log2(r+1))
    const double y = labels[best.j];
    const double p = pscores ? pscores[best.j] : 1.0;
    // a tie inside the first k ranks between rows that differ
    amb = amb || (r > 0 && best.s == last.s && (y != last_y || p != last_p));
    last = best;
    last_y = y;
    last_p = p;
    if (r == 0)
      head = y / p;
    else
      tail += y / (p * log2(double(r + 1)));
  }
  // a tie with the last ranked row that reaches past rank k, or a NaN score
  bool mine = false;
  for (int32_t j = b + lane; j < e; j += kEvalWave) {
    const double sc = scores[rows ? rows[j] : j];
    const double p = pscores ? pscores[j] : 1.0;
    mine = mine || sc != sc || (sc == last.s && (labels[j] != last_y || p != last_p));
  }
  amb = amb || __any(mine);
  if (lane == 0) {
    user_val[u] = head + tail;
    user_ok[u] = 1.0;
    user_amb[u] = amb ? 1.0 : 0.0;
  }
}

// mean over the users that count, summed in a fixed order; out[1] = number of users
// whose value depends on how equal scores are ordered
__global__ __launch_bounds__(1024) void val_dcg_mean_kernel(const double* __restrict__ user_val,
                                                            const double* __restrict__ user_ok,
                                                            const double* __restrict__ user_amb,
                                                            int32_t n_seg,
                                                            double* __restrict__ out) {
  __shared__ double sv[1024], sc[1024], sa[1024];
  double v = 0.0, c = 0.0, a = 0.0;
  for (int i = threadIdx.x; i < n_seg; i += 1024) {
    const double ok = user_ok[i];
    v += ok != 0.0 ? user_val[i] : 0.0;
    c += ok;
    a += user_amb[i];
  }
  sv[threadIdx.x] = v;
  sc[threadIdx.x] = c;
  sa[threadIdx.x] = a;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if (int(threadIdx.x) < w) {
      sv[threadIdx.x] += sv[threadIdx.x + w];
      sc[threadIdx.x] += sc[threadIdx.x + w];
      sa[threadIdx.x] += sa[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sv[0] / sc[0];  // no user counts -> nan, as np.mean([])
    out[1] = sa[0];
  }
}

// The k best rows of every user, for the metrics of the reference's TestEvaluator
// (utils/evaluate.py:80-127: DCG@K, Recall, MAP, mean exposure, CatalogCoverage, Gini all
// read the first K rows of `scores.argsort()[::-1]` per user).  Same selection rounds and
// the same tie rule as above; out_pos[u][r] = position (in the grouped order) of the row
// ranked r, or -1 past the user's rows.  out_flags[u]: bit 0 = the user has a positive
// label (the others are left out of every metric, evaluate.py:99-100), bit 1 = the ranking
// of the first k rows depends on how equal scores are ordered (a tie, inside the first k
// ranks or across rank k, between rows that differ in label, propensity or item; or a NaN
// score) -- the caller redoes those users with NumPy's own sort.
__global__ __launch_bounds__(kEvalBlock) void topk_users_kernel(
    const double* __restrict__ scores, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ rows, const double* __restrict__ labels,
    const double* __restrict__ pscores, const int32_t* __restrict__ items, int32_t n_seg,
    int32_t k, int32_t* __restrict__ out_pos, int32_t* __restrict__ out_flags) {
  const int lane = threadIdx.x % kEvalWave;
  const int u = int(blockIdx.x) * (kEvalBlock / kEvalWave) + threadIdx.x / kEvalWave;
  if (u >= n_seg) return;
  const int32_t b = seg_ptr[u], e = seg_ptr[u + 1];
  double ysum = 0.0;
  for (int32_t j = b + lane; j < e; j += kEvalWave) ysum += labels[j];
#pragma unroll
  for (int off = kEvalWave / 2; off >= 1; off >>= 1) ysum += __shfl_xor(ysum, off);
  int32_t* pos = out_pos + int64_t(u) * k;
  for (int r = lane; r < k; r += kEvalWave) pos[r] = -1;

  Pick last{__builtin_huge_val(), 0x7fffffff};
  double last_y = 0.0, last_p = 1.0;
  int32_t last_item = -1;
  bool amb = false;
  const int rounds = min(k, e - b);
  for (int r = 0; r < rounds; ++r) {
    Pick best{-__builtin_huge_val(), -1};
    for (int32_t j = b + lane; j < e; j += kEvalWave) {
      const Pick c{scores[rows ? rows[j] : j], j};
      if (pick_before(last, c) && (best.j < 0 || pick_before(c, best))) best = c;
    }
#pragma unroll
    for (int off = kEvalWave / 2; off >= 1; off >>= 1) {
      Pick o;
      o.s = __shfl_xor(best.s, off);
      o.j = __shfl_xor(best.j, off);
      if (o.j >= 0 && (best.j < 0 || pick_before(o, best))) best = o;
    }
    if (best.j < 0) break;  // only NaN scores are left (reported below)
    const double y = labels[best.j];
    const double p = pscores ? pscores[best.j] : 1.0;
    const int32_t it = items ? items[best.j] : 0;
    amb = amb || (r > 0 && best.s == last.s && (y != last_y || p != last_p || it != last_item));
    last = best;
    last_y = y;
    last_p = p;
    last_item = it;
    if (lane == 0) pos[r] = best.j;
  }
  bool mine = false;
  for (int32_t j = b + lane; j < e; j += kEvalWave) {
    const double sc = scores[rows ? rows[j] : j];
    const double p = pscores ? pscores[j] : 1.0;
    const int32_t it = items ? items[j] : 0;
    mine = mine || sc != sc ||
           (rounds > 0 && sc == last.s && (labels[j] != last_y || p != last_p || it != last_item));
  }
  amb = amb || __any(mine);
  if (lane == 0) out_flags[u] = ((e > b && ysum != 0.0) ? 1 : 0) | (amb ? 2 : 0);
}

}  // namespace rfm

extern "C" {

int32_t rfm_val_dcg(rfm_ctx* ctx, const double* d_scores, const int32_t* d_seg_ptr,
                    const int32_t* d_rows, const double* d_labels, const double* d_pscores,
                    int32_t n_segments, int32_t k, double* d_user_scratch, double* d_out) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    RFM_REQUIRE(n_segments >= 0, "n_segments=%d", n_segments);
    RFM_REQUIRE(k >= 1, "k=%d (ranking positions) must be >= 1", k);
    RFM_REQUIRE(d_out && d_user_scratch, "null output pointer");
    if (n_segments > 0)
      RFM_REQUIRE(d_scores && d_seg_ptr && d_labels, "null pointer");
    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    double* val = d_user_scratch;
    double* ok = d_user_scratch + n_segments;
    double* amb = d_user_scratch + 2 * int64_t(n_segments);
    if (n_segments > 0) {
      const int per_block = kEvalBlock / kEvalWave;
      const int grid = (n_segments + per_block - 1) / per_block;
      hipLaunchKernelGGL(val_dcg_users_kernel, dim3(grid), dim3(kEvalBlock), 0, ctx->stream,
                         d_scores, d_seg_ptr, d_rows, d_labels, d_pscores, n_segments, k, val,
                         ok, amb);
      RFM_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(val_dcg_mean_kernel, dim3(1), dim3(1024), 0, ctx->stream, val, ok, amb,
                       n_segments, d_out);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_topk_users(rfm_ctx* ctx, const double* d_scores, const int32_t* d_seg_ptr,
                       const int32_t* d_rows, const double* d_labels, const double* d_pscores,
                       const int32_t* d_items, int32_t n_segments, int32_t k, int32_t* d_out_pos,
                       int32_t* d_out_flags) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    RFM_REQUIRE(n_segments >= 0, "n_segments=%d", n_segments);
    RFM_REQUIRE(k >= 1, "k=%d (ranking positions) must be >= 1", k);
    if (n_segments == 0) return;
    RFM_REQUIRE(d_scores && d_seg_ptr && d_labels && d_out_pos && d_out_flags, "null pointer");
    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    const int per_block = kEvalBlock / kEvalWave;
    const int grid = (n_segments + per_block - 1) / per_block;
    hipLaunchKernelGGL(topk_users_kernel, dim3(grid), dim3(kEvalBlock), 0, ctx->stream, d_scores,
                       d_seg_ptr, d_rows, d_labels, d_pscores, d_items, n_segments, k, d_out_pos,
                       d_out_flags);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
