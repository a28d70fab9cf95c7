// Touched-row form of the FM gradient and its exchange (option 1).
//
// The reference's V gradient is row-sparse by construction: only rows with a non-zero
// gradient are written (src/fm.py:183-187), i.e. the columns the batch touches.  The
// gradient kernels of rfm_fm_kernels.hpp (grad mode) write those rows into a plan-owned
// table indexed by column, [G_V (n*k) | g_w (n) | g_w0], and stamp touch[col] with the
// step id; nothing is cleared between steps.  The kernels here
//   * compact the stamped columns, ascending, into a list of records
//       [column (as f64), G_V[column, 0..k), g_w[column]]            (k+2 doubles)
//     (count -> scan -> gather; per-range lower bounds for the owner exchange),
//   * apply such a list (theta -= lr * g: utils/optimizer.py:56-64 on the touched rows),
//   * reduce the lists the ranks sent to the owner of a column range: records of one
//     column are added in rank order, the owner's row is updated and emitted as
//       [column, V_new[column, 0..k), w_new[column]],
//   * store such updated rows into a replica.
#pragma once

#include "rfm_fm_kernels.hpp"

namespace rfm {

constexpr int kTouchChunk = 2048;  // columns per workgroup of the compaction
constexpr int kMaxRanges = 64;     // owner ranges (= ranks) the bounds kernel serves

__device__ inline int block_sum_int(int v, int* lds /*[kBlock/kWave]*/) {
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  __syncthreads();
  if (threadIdx.x % kWave == 0) lds[threadIdx.x / kWave] = v;
  __syncthreads();
  int s = 0;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i) s += lds[i];
  return s;
}

// chunk_cnt[b] = stamped columns among [b*kTouchChunk, (b+1)*kTouchChunk)
__global__ __launch_bounds__(kBlock) void touch_count_kernel(const int32_t* touch, int32_t id,
                                                            int64_t n, int32_t* chunk_cnt) {
  __shared__ int lds[kBlock / kWave];
  const int64_t c0 = int64_t(blockIdx.x) * kTouchChunk;
  int cnt = 0;
  for (int i = threadIdx.x; i < kTouchChunk; i += kBlock) {
    const int64_t c = c0 + i;
    cnt += (c < n && touch[c] == id) ? 1 : 0;
  }
  const int s = block_sum_int(cnt, lds);
  if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = s;
}

// Chunk b's stamped columns, ascending, become the column fields of the records
// chunk_off(b) .. : the workgroup adds up the counts of the chunks before it (a few
// thousand ints at most), ranks its own stamped columns by ballots, and writes
// rows[pos][0] = column.  The workgroup whose chunk holds a range start lo_i also owns
// range_bounds[i] (= records with a column below lo_i); the last one writes the total.
__global__ __launch_bounds__(kBlock) void touch_list_kernel(const int32_t* touch, int32_t id,
                                                           int64_t n, int k,
                                                           const int32_t* chunk_cnt, int n_chunks,
                                                           const double* table, double* rows,
                                                           int64_t cap_rows, int32_t* n_touched,
                                                           double* out_gw0,
                                                           const int32_t* range_lo, int n_ranges,
                                                           int32_t* range_bounds) {
  __shared__ int lds[kBlock / kWave];
  __shared__ int wave_cnt[kBlock / kWave];
  const int b = blockIdx.x;
  const int64_t c0 = int64_t(b) * kTouchChunk;
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int width = k + 2;
  if (b == 0 && threadIdx.x == 0 && out_gw0) out_gw0[0] = table[n * k + n];
  int part = 0;
  for (int i = threadIdx.x; i < b; i += kBlock) part += chunk_cnt[i];
  const int first = block_sum_int(part, lds);
  int m = 0;
  for (int r0 = 0; r0 < kTouchChunk; r0 += kBlock) {
    const int64_t c = c0 + r0 + threadIdx.x;
    const bool on = c < n && touch[c] == id;
    const unsigned long long mask = __ballot(on);
    __syncthreads();
    if (lane == 0) wave_cnt[wv] = __popcll(mask);
    __syncthreads();
    int pos = first + m;
    int round = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
      if (w < wv) pos += wave_cnt[w];
      round += wave_cnt[w];
    }
    pos += __popcll(mask & ((1ull << lane) - 1ull));
    if (on && pos < cap_rows) rows[int64_t(pos) * width] = double(c);
    m += round;
  }
  if (b == n_chunks - 1 && threadIdx.x == 0) {
    n_touched[0] = first + m;
    if (range_lo) range_bounds[n_ranges] = first + m;
  }
  // owner-range starts that fall into this chunk (the last chunk also takes those at or past n)
  for (int i = 0; range_lo && i < n_ranges; ++i) {
    int64_t lo = range_lo[i];
    lo = lo < 0 ? 0 : (lo > n ? n : lo);
    const bool mine = (lo >= c0 && lo < c0 + kTouchChunk) || (b == n_chunks - 1 && lo >= c0);
    if (!mine) continue;  // uniform over the workgroup
    int cnt = 0;
    for (int64_t c = c0 + threadIdx.x; c < lo; c += kBlock) cnt += touch[c] == id ? 1 : 0;
    const int below = block_sum_int(cnt, lds);
    if (threadIdx.x == 0) range_bounds[i] = first + below;
  }
}

// fills the records whose column field touch_list_kernel wrote: one wave per record
__global__ __launch_bounds__(kBlock) void rows_fill_kernel(const double* table, double* rows,
                                                          const int32_t* n_rows, int64_t cap_rows,
                                                          int64_t n, int k) {
  const int lane = threadIdx.x % kWave;
  const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = int64_t(gridDim.x) * (kBlock / kWave);
  int64_t cnt = n_rows[0];
  if (cnt > cap_rows) cnt = cap_rows;
  const int width = k + 2;
  for (int64_t i = wave; i < cnt; i += n_waves) {
    double* row = rows + i * width;
    const int64_t col = int64_t(row[0]);
    for (int f = lane; f < k; f += kWave) row[1 + f] = table[col * k + f];
    if (lane == 0) row[k + 1] = table[n * k + col];
  }
}

// theta -= lr * g over a record list; *n_rows records (device count), one wave per record
__global__ __launch_bounds__(kBlock) void rows_apply_kernel(const double* rows, const int32_t* n_rows,
                                                           int64_t cap_rows, const double* gw0,
                                                           double* w0, double* w, double* V,
                                                           int64_t n, int k, double lr) {
  const int lane = threadIdx.x % kWave;
  const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = int64_t(gridDim.x) * (kBlock / kWave);
  int64_t cnt = n_rows[0];
  if (cnt > cap_rows) cnt = cap_rows;
  const int width = k + 2;
  for (int64_t i = wave; i < cnt; i += n_waves) {
    const double* row = rows + i * width;
    const int64_t col = int64_t(row[0]);
    if (col < 0 || col >= n) continue;
    for (int f = lane; f < k; f += kWave) V[col * k + f] -= lr * row[1 + f];
    if (lane == 0) w[col] -= lr * row[k + 1];
  }
  if (gw0 && blockIdx.x == 0 && threadIdx.x == 0) w0[0] -= lr * gw0[0];
}

// Owner side.  rows holds n_seg record lists back to back (seg_ptr[s] .. seg_ptr[s+1]: what
// rank s sent, ascending by column).  One wave per record: the record of the lowest rank
// holding a column leads; it adds the other ranks' records of that column in rank order,
// updates the owner's row and emits [column, V_new, w_new] at its own position; every
// other record's position gets column -1 (so positions, and the list size, are known
// before the kernel runs).
// (w0 given: a record with column == n carries a rank's g_w0 in its g_w field; it is reduced
// like any other -- rank order -- and emitted as [n, 0.., w0_new])
__global__ __launch_bounds__(kBlock) void rows_reduce_kernel(const double* rows,
                                                            const int32_t* seg_ptr, int n_seg,
                                                            const double* w, const double* V,
                                                            int64_t n, int k, double lr,
                                                            double* out_rows,
                                                            const double* w0 = nullptr) {
  const int lane = threadIdx.x % kWave;
  const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = int64_t(gridDim.x) * (kBlock / kWave);
  const int width = k + 2;
  const int64_t total = seg_ptr[n_seg];
  for (int64_t i = wave; i < total; i += n_waves) {
    const double colf = rows[i * width];
    const int64_t col = int64_t(colf);
    // lane s looks the column up in segment s (n_seg <= 64)
    int64_t at = -1;
    int own = -1;
    if (lane < n_seg) {
      const int64_t lo = seg_ptr[lane], hi = seg_ptr[lane + 1];
      if (i >= lo && i < hi) {
        at = i;
      } else {
        int64_t a = lo, b = hi;
        while (a < b) {
          const int64_t mid = (a + b) >> 1;
          if (rows[mid * width] < colf)
            a = mid + 1;
          else
            b = mid;
        }
        at = (a < hi && rows[a * width] == colf) ? a : -1;
      }
      if (i >= lo && i < hi) own = lane;
    }
    const unsigned long long has = __ballot(at >= 0);
    const int my_seg = __ffsll((long long)__ballot(own >= 0)) - 1;
    const bool is_w0 = w0 != nullptr && col == n;
    const bool leader = (has & ((1ull << my_seg) - 1ull)) == 0ull && col >= 0 && (col < n || is_w0);
    double* out = out_rows + i * width;
    if (!leader) {
      if (lane == 0) out[0] = -1.0;
      continue;
    }
    for (int f0 = 0; f0 < k + 1; f0 += kWave) {
      const int f = f0 + lane;  // f < k: G_V component; f == k: g_w
      double sum = 0.0;
      for (int s = my_seg; s < n_seg; ++s) {
        const int64_t j = __shfl(at, s, kWave);
        if (j >= 0 && f <= k) sum += rows[j * width + 1 + f];
      }
      if (f < k)
        out[1 + f] = is_w0 ? 0.0 : V[col * k + f] - lr * sum;
      else if (f == k)
        out[1 + k] = (is_w0 ? w0[0] : w[col]) - lr * sum;
    }
    if (lane == 0) out[0] = colf;
  }
}

// store updated rows into a replica; w0 -= lr * (sum of the ranks' partial g_w0, in rank order)
// (w0_records: a record with column == n holds the new w0 in its w field)
__global__ __launch_bounds__(kBlock) void rows_set_kernel(const double* rows, int64_t n_rows,
                                                         const double* gw0_parts, int n_parts,
                                                         int64_t part_stride, double* w0,
                                                         double* w, double* V, int64_t n, int k,
                                                         double lr, bool w0_records = false) {
  const int lane = threadIdx.x % kWave;
  const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = int64_t(gridDim.x) * (kBlock / kWave);
  const int width = k + 2;
  for (int64_t i = wave; i < n_rows; i += n_waves) {
    const double* row = rows + i * width;
    const int64_t col = int64_t(row[0]);
    if (w0_records && col == n) {
      if (lane == 0) w0[0] = row[k + 1];
      continue;
    }
    if (col < 0 || col >= n) continue;
    for (int f = lane; f < k; f += kWave) V[col * k + f] = row[1 + f];
    if (lane == 0) w[col] = row[k + 1];
  }
  if (gw0_parts && blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int r = 0; r < n_parts; ++r) s += gw0_parts[r * part_stride];
    w0[0] -= lr * s;
  }
}

// ---------------------------------------------------------------------------
// the data-parallel fit loop (rfm_fm_fit_dp): transfer sizes ahead of the loop
// ---------------------------------------------------------------------------
// Which columns a shard's gradient records name depends on the row ids alone: the columns of
// the shard's entries, plus -- when the shard is not empty -- every on-chip (hot) column of the
// plan (their workgroups always write a row).  One thread per (row, entry position): stamps
// touch[column] = id, exactly what the gradient kernels will stamp.
__global__ __launch_bounds__(kBlock) void rows_mark_kernel(const Entry* ent, const RowRec* rows,
                                                          const char* ell, int64_t ell_stride,
                                                          int lpr, const int32_t* row_ids,
                                                          int64_t n_rows, int32_t* touch,
                                                          int32_t id, const int32_t* hot_cols,
                                                          int n_hot) {
  const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  if (ell) {
    for (int64_t i = tid; i < n_rows * lpr; i += stride) {
      const Entry e = reinterpret_cast<const Entry*>(ell + int64_t(row_ids[i / lpr]) * ell_stride)[i % lpr];
      if (e.slot != kNilSlot) touch[e.col] = id;
    }
  } else {
    // a wave per row: rows of the plain records can be of any length
    const int lane = threadIdx.x % kWave;
    for (int64_t t = tid / kWave; t < n_rows; t += stride / kWave) {
      const RowRec rec = rows[row_ids[t]];
      for (int64_t j = lane; j < rec.len; j += kWave) touch[ent[rec.begin + j].col] = id;
    }
  }
  if (n_rows > 0)
    for (int64_t h = tid; h < n_hot; h += stride) touch[hot_cols[h]] = id;
}

// appends the record that carries the shard's g_w0: [n, 0.., g_w0] at position *n_rows
__global__ void rows_append_w0_kernel(double* rows, const int32_t* n_rows, int64_t cap_rows,
                                      const double* gw0, int64_t n, int k) {
  const int64_t at = n_rows[0];
  if (at >= cap_rows) return;  // (the bounds check below reports the overflow)
  double* row = rows + at * (k + 2);
  for (int f = threadIdx.x; f < k + 2; f += blockDim.x)
    row[f] = f == 0 ? double(n) : (f == k + 1 ? gw0[0] : 0.0);
}

// the record list a step produced must be what the count pass promised: range bounds equal
// (the last one + 1 for the g_w0 record) and within capacity; otherwise flag[0] = iteration + 1
__global__ void bounds_check_kernel(const int32_t* actual, const int32_t* planned, int n_ranges,
                                    int64_t cap_rows, int32_t it, int32_t* flag) {
  const int i = threadIdx.x;
  if (i > n_ranges) return;
  const int32_t want = planned[i] - (i == n_ranges ? 1 : 0);
  if (actual[i] != want || planned[n_ranges] > cap_rows) atomicCAS(&flag[0], 0, it + 1);
}

// raw sums of the per-workgroup loss partials of a run of launches (block b: launch b), and
// their scaling once the ranks' sums have been combined: out = -sum / n_rows
__global__ __launch_bounds__(kBlock) void loss_sum_many_kernel(const double* partial, int64_t stride,
                                                              int n_partial, double* out) {
  __shared__ double lds[kBlock];
  const double* row = partial + int64_t(blockIdx.x) * stride;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += kBlock) acc += row[i];
  const double s = block_sum<kBlock>(acc, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void loss_scale_kernel(const double* sums, int64_t count, double n_rows, double* out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < count) out[i] = -sums[i] / n_rows;
}

}  // namespace rfm
