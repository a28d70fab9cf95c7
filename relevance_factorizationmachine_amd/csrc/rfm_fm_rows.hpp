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

// one workgroup: exclusive scan of the chunk counts (in place: chunk_cnt[b] becomes the
// position of chunk b's first record), the total, and for every range start range_lo[i]
// the number of stamped columns below it (= position of the range's first record).
__global__ __launch_bounds__(kBlock) void touch_scan_kernel(const int32_t* touch, int32_t id,
                                                           int64_t n, int32_t* chunk_cnt,
                                                           int n_chunks, int32_t* n_touched,
                                                           const int32_t* range_lo, int n_ranges,
                                                           int32_t* range_bounds) {
  __shared__ int lds[kBlock / kWave];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n_chunks; b0 += kBlock) {
    const int b = b0 + threadIdx.x;
    const int v = b < n_chunks ? chunk_cnt[b] : 0;
    // inclusive scan inside the wave, then over the waves
    int x = v;
    for (int o = 1; o < kWave; o <<= 1) {
      const int y = __shfl_up(x, o, kWave);
      if (int(threadIdx.x % kWave) >= o) x += y;
    }
    __syncthreads();
    if (threadIdx.x % kWave == kWave - 1) lds[threadIdx.x / kWave] = x;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < int(threadIdx.x / kWave); ++w) before += lds[w];
    if (b < n_chunks) chunk_cnt[b] = before + x - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) n_touched[0] = carry;
  // positions of the range starts: whole chunks below come from the scan, the rest is counted
  for (int i = 0; i <= n_ranges && range_lo; ++i) {
    const int64_t lo = i < n_ranges ? int64_t(range_lo[i]) : n;  // wave-uniform
    const int64_t clamped = lo < 0 ? 0 : (lo > n ? n : lo);
    const int b = int(clamped / kTouchChunk);
    int cnt = 0;
    for (int64_t c = int64_t(b) * kTouchChunk + threadIdx.x; c < clamped; c += kBlock)
      cnt += touch[c] == id ? 1 : 0;
    const int part = block_sum_int(cnt, lds);
    if (threadIdx.x == 0) range_bounds[i] = (b < n_chunks ? chunk_cnt[b] : carry) + part;
    __syncthreads();
  }
}

// records of chunk b, ascending by column, at rows[chunk_off[b] ...]
__global__ __launch_bounds__(kBlock) void touch_gather_kernel(const int32_t* touch, int32_t id,
                                                             int64_t n, int k,
                                                             const int32_t* chunk_off,
                                                             const double* table, double* rows,
                                                             int64_t cap_rows, double* out_gw0) {
  __shared__ int32_t list[kTouchChunk];
  __shared__ int wave_cnt[kBlock / kWave];
  __shared__ int m_sh;
  const int64_t c0 = int64_t(blockIdx.x) * kTouchChunk;
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  if (threadIdx.x == 0) m_sh = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_gw0) out_gw0[0] = table[n * k + n];
  __syncthreads();
  for (int r0 = 0; r0 < kTouchChunk; r0 += kBlock) {
    const int64_t c = c0 + r0 + threadIdx.x;
    const bool on = c < n && touch[c] == id;
    const unsigned long long mask = __ballot(on);
    if (lane == 0) wave_cnt[wv] = __popcll(mask);
    __syncthreads();
    int pos = m_sh;
    for (int w = 0; w < wv; ++w) pos += wave_cnt[w];
    pos += __popcll(mask & ((1ull << lane) - 1ull));
    if (on) list[pos] = int32_t(c);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < kBlock / kWave; ++w) t += wave_cnt[w];
      m_sh += t;
    }
    __syncthreads();
  }
  const int m = m_sh;
  const int width = k + 2;
  const int64_t first = chunk_off[blockIdx.x];
  for (int idx = threadIdx.x; idx < m * width; idx += kBlock) {
    const int r = idx / width, f = idx - r * width;
    const int64_t col = list[r];
    if (first + r >= cap_rows) continue;  // the caller sees the true count and can tell
    double v;
    if (f == 0)
      v = double(col);
    else if (f <= k)
      v = table[col * k + (f - 1)];
    else
      v = table[n * k + col];
    rows[(first + r) * width + f] = v;
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
__global__ __launch_bounds__(kBlock) void rows_reduce_kernel(const double* rows,
                                                            const int32_t* seg_ptr, int n_seg,
                                                            const double* w, const double* V,
                                                            int64_t n, int k, double lr,
                                                            double* out_rows) {
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
    const bool leader = (has & ((1ull << my_seg) - 1ull)) == 0ull && col >= 0 && col < n;
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
        out[1 + f] = V[col * k + f] - lr * sum;
      else if (f == k)
        out[1 + k] = w[col] - lr * sum;
    }
    if (lane == 0) out[0] = colf;
  }
}

// store updated rows into a replica; w0 -= lr * (sum of the ranks' partial g_w0, in rank order)
__global__ __launch_bounds__(kBlock) void rows_set_kernel(const double* rows, int64_t n_rows,
                                                         const double* gw0_parts, int n_parts,
                                                         int64_t part_stride, double* w0,
                                                         double* w, double* V, int64_t n, int k,
                                                         double lr) {
  const int lane = threadIdx.x % kWave;
  const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = int64_t(gridDim.x) * (kBlock / kWave);
  const int width = k + 2;
  for (int64_t i = wave; i < n_rows; i += n_waves) {
    const double* row = rows + i * width;
    const int64_t col = int64_t(row[0]);
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

}  // namespace rfm
