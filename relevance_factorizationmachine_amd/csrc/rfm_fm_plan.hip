// Builder of the FM training plan (rfm_fm_plan_create / _create_device): the one-time,
// per-fit layout of the training log (train["features"], ["labels"], ["pscores"] of
// src/fm.py:55-79) that the step kernels of rfm_fm_kernels.hpp read.
//
// Everything of size nnz happens on the device, from the device copy of the caller's CSR:
//   1. plan_rows_kernel     one thread per row: RowRec {first entry, length, label,
//                           propensity}; checks indptr / column ranges; row_of[entry]
//   2. rocprim radix sort   entries by column (stable: CSR order, hence row order, is
//                           kept inside a column) -- the column-major view
//   3. plan_starts_kernel   first sorted position of every column; a column named twice
//                           by one row shows up as two neighbours of the same row
//   (host, O(n_features + windows): hot class, slot base of every sparse column, window
//    descriptors, crossing-column lists -- the only part that comes back from the device
//    is the n_features + 1 column starts)
//   4. plan_scatter_kernel  Entry {column, slot | hot rank, value} in CSR order and
//                           SlotRec {value, column} in slot order
// The slot order inside a column is the row order, as in the host builder this replaces,
// so the fixed-order sums of fm_consume_kernel are bit-identical to what they were.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstring>
#include <memory>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "rfm_fm_plan.h"
#include "rfm_fm_records.h"

namespace rfm {
namespace {

constexpr int32_t kHotTag = INT32_MIN;  // colinfo: kHotTag + hot rank; else slot - sorted position

// flags[0] indptr not monotone / out of range, [1] column index out of range,
// [2] a row names a column twice, [3] longest row
__global__ __launch_bounds__(kBlock) void plan_rows_kernel(const int64_t* indptr,
                                                          const int32_t* indices,
                                                          const double* y, const double* p,
                                                          int64_t n_rows, int64_t nnz, int64_t n,
                                                          RowRec* rows, int32_t* row_of,
                                                          int32_t* flags) {
  for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < n_rows;
       r += int64_t(gridDim.x) * kBlock) {
    const int64_t b = indptr[r], e = indptr[r + 1];
    if (b < 0 || e < b || e > nnz) {
      atomicOr(&flags[0], 1);
      rows[r] = RowRec{0, 0, y[r], p[r]};
      continue;
    }
    rows[r] = RowRec{b, e - b, y[r], p[r]};
    atomicMax(&flags[3], int32_t(e - b > INT32_MAX ? INT32_MAX : e - b));
    for (int64_t q = b; q < e; ++q) {
      const int32_t c = indices[q];
      if (c < 0 || c >= n) atomicOr(&flags[1], 1);
      row_of[q] = int32_t(r);
    }
  }
}

// cstart[c] = first sorted position whose column is >= c (cstart[n] = nnz)
__global__ __launch_bounds__(kBlock) void plan_starts_kernel(const int32_t* key, const int32_t* pos,
                                                            const int32_t* row_of, int64_t nnz,
                                                            int64_t n, int32_t* cstart,
                                                            int32_t* flags) {
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i <= nnz;
       i += int64_t(gridDim.x) * kBlock) {
    const int64_t prev = i == 0 ? -1 : int64_t(key[i - 1]);
    const int64_t cur = i == nnz ? n : int64_t(key[i]);
    for (int64_t c = prev + 1; c <= cur; ++c) cstart[c] = int32_t(i);
    if (i > 0 && i < nnz && prev == cur && row_of[pos[i]] == row_of[pos[i - 1]])
      atomicOr(&flags[2], 1);
  }
}

__global__ __launch_bounds__(kBlock) void plan_scatter_kernel(const int32_t* key, const int32_t* pos,
                                                             const double* values,
                                                             const int32_t* colinfo, int64_t nnz,
                                                             Entry* ent, SlotRec* slots) {
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nnz;
       i += int64_t(gridDim.x) * kBlock) {
    const int32_t c = key[i], q = pos[i];
    const double x = values[q];
    const int32_t info = colinfo[c];
    if (info < kHotTag + (1 << 20)) {  // hot column: -1 - rank
      ent[q] = Entry{c, -1 - (info - kHotTag), x};
    } else {
      const int32_t sl = int32_t(i + info);
      ent[q] = Entry{c, sl, x};
      slots[sl] = SlotRec{x, c, 0};
    }
  }
}

// padded row blocks Entry[lpr] (rfm_fm_records.h) + the rows' {label, propensity} pairs
__global__ __launch_bounds__(kBlock) void plan_ell_kernel(const RowRec* rows, const Entry* ent,
                                                         int64_t n_rows, int lpr, char* ell,
                                                         int64_t stride, double2* ell_yp) {
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n_rows * lpr;
       i += int64_t(gridDim.x) * kBlock) {
    const int64_t r = i / lpr;
    const int j = int(i % lpr);
    const RowRec rec = rows[r];
    if (j == 0) ell_yp[r] = make_double2(rec.y, rec.p);
    Entry e{0, kNilSlot, 0.0};
    if (rec.len > 0) {
      e = ent[rec.begin + (j < rec.len ? j : rec.len - 1)];
      if (j >= rec.len) {
        e.slot = kNilSlot;
        e.x = 0.0;
      }
    }
    reinterpret_cast<Entry*>(ell + r * stride)[j] = e;
  }
}

// A log in the form the sliced loss forward reads (rfm_fm_sliced.hpp): 2^mll records per row,
// cached columns first.  A row belongs to 2^mll consecutive lanes of one wavefront (entry e
// in lane e); a record's place is the count of the records of its kind before it (ballots).
__global__ __launch_bounds__(kBlock) void sl_translate_kernel(const int64_t* indptr, const int32_t* indices,
                                                            const double* values, int64_t n_rows,
                                                            const int32_t* rank, int mll, int n_cached,
                                                            int row_bytes, SlEnt* out) {
  const int ML = 1 << mll;
  const int64_t total = n_rows << mll;
  const int lane = threadIdx.x & (kWave - 1);
  const int shift = lane & ~(ML - 1);  // first lane of this row's group
  const unsigned long long group = ML == 64 ? ~0ull : ((1ull << ML) - 1ull);
  const int zoff = n_cached * row_bytes;
  for (int64_t s0 = int64_t(blockIdx.x) * kBlock; s0 < total; s0 += int64_t(gridDim.x) * kBlock) {
    const int64_t s = s0 + threadIdx.x;  // (total is a multiple of 2^mll: a group is whole or absent)
    const bool live = s < total;
    const int64_t r = live ? s >> mll : 0;
    const int e = int(s & (ML - 1));
    const int64_t b0 = indptr[r], len = indptr[r + 1] - b0;
    SlEnt o{zoff, kSlPad, 0.0};
    int kind = 2;  // 0 cached, 1 not cached, 2 padding
    if (live && len > ML) {
      if (e == 0) o.col = kSlLong;
    } else if (live && e < len) {
      const int32_t col = indices[b0 + e];
      const int32_t rk = rank[col];
      kind = rk >= 0 ? 0 : 1;
      if (rk >= 0) o.off = rk * row_bytes;
      o.col = col;
      o.x = values[b0 + e];
    }
    const unsigned long long mc = (__ballot(kind == 0) >> shift) & group;
    const unsigned long long mu = (__ballot(kind == 1) >> shift) & group;
    const unsigned long long below = (1ull << e) - 1ull;
    int pos;
    if (kind == 0)
      pos = __popcll(mc & below);
    else if (kind == 1)
      pos = __popcll(mc) + __popcll(mu & below);
    else
      pos = __popcll(mc) + __popcll(mu) + __popcll(~(mc | mu) & group & below);
    if (live) out[(r << mll) + pos] = o;
  }
}

// a log as the records the forward kernel's REC form reads (labels / propensities unused: 0 / 1)
__global__ __launch_bounds__(kBlock) void log_records_kernel(const int64_t* indptr, const int32_t* indices,
                                                           const double* values, int64_t n_rows,
                                                           RowRec* rows, Entry* ent) {
  for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < n_rows; r += int64_t(gridDim.x) * kBlock) {
    const int64_t b = indptr[r], e = indptr[r + 1];
    rows[r] = RowRec{b, e - b, 0.0, 1.0};
    for (int64_t q = b; q < e; ++q) ent[q] = Entry{indices[q], 0, values[q]};
  }
}

void upload(DevBuf& dst, const void* src, size_t bytes, hipStream_t stream) {
  dst.alloc(bytes);
  if (bytes) RFM_HIP_CHECK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, stream));
}

int grid_for(const rfm_ctx* ctx, int64_t items) {
  return int(std::max<int64_t>(1, std::min<int64_t>((items + kBlock - 1) / kBlock,
                                                    int64_t(ctx->n_cu) * 16)));
}

rfm_fm_plan* build_plan(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                        const double* d_values, const double* d_y, const double* d_pscore,
                        int64_t n_rows, int64_t n_features, int32_t n_factors, int64_t max_batch,
                        int32_t hot_min_count) {
  RFM_REQUIRE(ctx && d_indptr && d_y && d_pscore, "null pointer");
  RFM_REQUIRE(n_rows >= 1 && n_features >= 1 && max_batch >= 1, "bad shape");
  RFM_REQUIRE(max_batch * int64_t(n_factors) < (int64_t(1) << 31),
              "max_batch * n_factors = %lld does not fit the 31-bit offsets of the batch's Q rows",
              (long long)(max_batch * int64_t(n_factors)));
  RFM_REQUIRE(n_rows < (int64_t(1) << 31), "too many rows");
  RFM_REQUIRE(n_features < (int64_t(1) << 31) - 2, "n_features too large");
  RFM_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  const bool timing = env_int("RFM_PLAN_TIMING", 0) != 0;
  auto t_prev = std::chrono::steady_clock::now();
  const auto lap = [&](const char* what) {
    if (!timing) return;
    RFM_HIP_CHECK(hipStreamSynchronize(st));
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[plan] %s: %.2f ms\n", what,
            std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };

  int64_t nnz = 0;
  RFM_HIP_CHECK(hipMemcpyAsync(&nnz, d_indptr + n_rows, 8, hipMemcpyDeviceToHost, st));
  RFM_HIP_CHECK(hipStreamSynchronize(st));
  RFM_REQUIRE(nnz >= 0 && nnz < (int64_t(1) << 31) - kWave - (1 << 20), "nnz=%lld unsupported",
              (long long)nnz);
  RFM_REQUIRE(nnz == 0 || (d_indices && d_values), "null CSR arrays");
  const size_t nz = size_t(nnz), nf = size_t(n_features), nr = size_t(n_rows);

  auto plan = std::make_unique<rfm_fm_plan>();
  plan->device = ctx->device;
  plan->n_rows = n_rows;
  plan->n_features = n_features;
  plan->nnz = nnz;
  plan->max_batch = max_batch;
  plan->k = n_factors;

  // ---- device passes over the entries ------------------------------------------------
  DevBuf flags, row_of, key, pos, cstart, temp;
  flags.alloc(16);
  RFM_HIP_CHECK(hipMemsetAsync(flags.p, 0, 16, st));
  row_of.alloc(std::max<size_t>(nz, 1) * 4);
  plan->rows.alloc(nr * sizeof(RowRec));
  hipLaunchKernelGGL(plan_rows_kernel, dim3(grid_for(ctx, n_rows)), dim3(kBlock), 0, st, d_indptr,
                     d_indices, d_y, d_pscore, n_rows, nnz, n_features, plan->rows.as<RowRec>(),
                     row_of.as<int32_t>(), flags.as<int32_t>());
  RFM_HIP_CHECK(hipGetLastError());
  int32_t h_flags[4] = {0, 0, 0, 0};
  RFM_HIP_CHECK(hipMemcpyAsync(h_flags, flags.p, 16, hipMemcpyDeviceToHost, st));
  RFM_HIP_CHECK(hipStreamSynchronize(st));
  RFM_REQUIRE(!h_flags[0], "indptr not monotone / out of range (nnz %lld)", (long long)nnz);
  RFM_REQUIRE(!h_flags[1], "a column index lies outside 0..%lld", (long long)n_features - 1);
  lap("row records + checks");

  key.alloc(std::max<size_t>(nz, 1) * 4);
  pos.alloc(std::max<size_t>(nz, 1) * 4);
  cstart.alloc((nf + 1) * 4);
  if (nnz > 0) {
    int bits = 1;
    while ((int64_t(1) << bits) < n_features) ++bits;
    size_t temp_bytes = 0;
    const auto positions = rocprim::counting_iterator<int32_t>(0);
    RFM_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_indices, key.as<int32_t>(),
                                            positions, pos.as<int32_t>(), nz, 0u, unsigned(bits),
                                            st));
    temp.alloc(temp_bytes);
    RFM_HIP_CHECK(rocprim::radix_sort_pairs(temp.p, temp_bytes, d_indices, key.as<int32_t>(),
                                            positions, pos.as<int32_t>(), nz, 0u, unsigned(bits),
                                            st));
  }
  lap("sort by column");
  hipLaunchKernelGGL(plan_starts_kernel, dim3(grid_for(ctx, nnz + 1)), dim3(kBlock), 0, st,
                     key.as<int32_t>(), pos.as<int32_t>(), row_of.as<int32_t>(), nnz, n_features,
                     cstart.as<int32_t>(), flags.as<int32_t>());
  RFM_HIP_CHECK(hipGetLastError());
  std::vector<int32_t> h_start(nf + 1);
  RFM_HIP_CHECK(hipMemcpyAsync(h_start.data(), cstart.p, (nf + 1) * 4, hipMemcpyDeviceToHost, st));
  RFM_HIP_CHECK(hipMemcpyAsync(h_flags, flags.p, 16, hipMemcpyDeviceToHost, st));
  RFM_HIP_CHECK(hipStreamSynchronize(st));
  RFM_REQUIRE(!h_flags[2], "a row names a column twice: sum duplicate entries first");
  lap("column starts");

  // ---- host: classes, slot bases, windows, crossing columns (O(n_features + windows)) --
  const auto len = [&](size_t c) { return int64_t(h_start[c + 1]) - int64_t(h_start[c]); };
  // hot class: expected entries per batch >= hot_min, most frequent first, LDS budget
  std::vector<int32_t> hot_cols;
  std::vector<int32_t> hot_rank(nf, -1);
  // hot_min_count = -2: the hot class at its default threshold with FIXED-ORDER sums, where
  // the forward shapes this plan can take are built for them; otherwise as -1 (no hot class)
  const bool fixed = hot_min_count == -2 && forward_fixed_order_ok(ctx, max_batch, n_factors) &&
                     env_int("RFM_NO_FIXED_HOT", 0) == 0;
  // Factor counts of more than one chunk per lane (k > 128, or odd k > 64) have NO on-chip
  // class: a column's LDS sums are k + 2 doubles (17 columns fit at k = 400) and every forward
  // workgroup -- four rows each at that width -- would leave a slab of them; measured at the
  // reference's published point (k = 400, B = 2 000; profiles/r3b): 76 us per step with the
  // class, 52 us without.  All sums then have a fixed order: such fits are bitwise
  // reproducible whatever hot_min_count says.  (RFM_HOT_MULTI_CHUNK=1: the old behaviour.)
  const bool chunked = shape_for(n_factors).nc > 1 && env_int("RFM_HOT_MULTI_CHUNK", 0) == 0;
  if ((hot_min_count >= 0 || fixed) && !chunked) {
    const int64_t hot_min = hot_min_count > 0 ? hot_min_count : kDefaultHotMinCount;
    for (size_t c = 0; c < nf; ++c)
      if (len(c) * max_batch >= hot_min * n_rows) hot_cols.push_back(int32_t(c));
    std::stable_sort(hot_cols.begin(), hot_cols.end(),
                     [&](int32_t x, int32_t y) { return len(size_t(x)) > len(size_t(y)); });
    const size_t per_col = size_t(n_factors + 2) * 8;
    // RFM_HOT_LDS_KB / RFM_MAX_HOT override the budget (tuning experiments only)
    // (the fixed-order form parks a trip's Q rows, row sets and cell sums in LDS as well)
    const size_t budget = size_t(env_int("RFM_HOT_LDS_KB", int((fixed ? kHotLdsBudgetFixed : kHotLdsBudget) >> 10))) << 10;
    const size_t cap = std::min<size_t>(size_t(env_int("RFM_MAX_HOT", kMaxHot)), budget / per_col);
    if (hot_cols.size() > cap) hot_cols.resize(cap);
    // (ranks descend by frequency: the forward deals the columns to its lane groups by rank)
    for (size_t h = 0; h < hot_cols.size(); ++h) hot_rank[size_t(hot_cols[h])] = int32_t(h);
  }
  // tasks of fm_consume_kernel: the slot view is cut into tasks of task_words x 64 slots, one
  // per lane group, GPB of them per workgroup.  The sparse-class columns are laid out in
  // ascending order; a column may run over several tasks, but it never crosses the edge of a
  // workgroup's slots unless it is longer than all of them (then it starts on such an edge,
  // takes whole workgroups -- each leaves one partial row for fm_finalize_kernel -- and the
  // rest of its last workgroup stays empty).
  const Shape shp = shape_for(n_factors);
  const int64_t GPB = kBlock / shp.lpr;
  const double density = double(max_batch) / double(n_rows);  // marked fraction of a column
  int64_t W = 1;
  // (several chunks per lane: twice the marks per task -- measured at k = 400, B = 2 000:
  // 81 / 51 / 49 / 55 us per step at 2 / 4 / 8 / 16 words, profiles/r3c)
  const int task_marks = shp.nc > 1 ? 2 * kTaskMarks : kTaskMarks;
  while (W * 2 * 64 * density <= 1.5 * task_marks && W * 2 <= int64_t(kTaskTrips) * shp.lpr) W *= 2;
  if (const int forced = env_int("RFM_TASK_WORDS", 0))  // tuning experiments only
    W = std::max<int64_t>(1, std::min<int64_t>(forced, int64_t(kTaskTrips) * shp.lpr));
  const int64_t C = W * 64;     // slots of a task
  const int64_t BC = GPB * C;   // slots of a workgroup
  std::vector<int64_t> cptr(nf, 0);  // first slot of every sparse-class column
  std::vector<int32_t> colinfo(nf);
  std::vector<SplitCol> split_short, split_long;
  std::vector<std::pair<int64_t, int32_t>> piece_blocks;  // (workgroup, partial row)
  int64_t at = 0;
  int32_t n_parts = 0;
  for (size_t c = 0; c < nf; ++c) {
    if (hot_rank[c] >= 0) {
      colinfo[c] = kHotTag + hot_rank[c];
      continue;
    }
    const int64_t lc = len(c);
    if (lc == 0) {
      cptr[c] = at;
      colinfo[c] = 0;
      continue;
    }
    if (lc <= BC) {
      const int64_t room = BC - at % BC;
      if (lc > room) at += room;
      cptr[c] = at;
      at += lc;
    } else {
      at = (at + BC - 1) / BC * BC;
      cptr[c] = at;
      const int64_t n_blk = (lc + BC - 1) / BC;
      SplitCol sc{int32_t(c), n_parts, int32_t(n_blk), 0};
      for (int64_t b = 0; b < n_blk; ++b) piece_blocks.emplace_back(at / BC + b, n_parts++);
      at += n_blk * BC;
      (sc.part_count <= kShortSplit ? split_short : split_long).push_back(sc);
    }
    colinfo[c] = int32_t(cptr[c] - int64_t(h_start[c]));
  }
  const int64_t n_blocks = std::max<int64_t>(1, (at + BC - 1) / BC);
  const int64_t n_slots = n_blocks * BC;
  RFM_REQUIRE(n_slots < (int64_t(1) << 31) - 512, "slot space too large");
  const size_t ns = size_t(n_slots);
  // task descriptions: the columns of the first and last occupied slot of every task, and
  // whether they continue from / into the neighbouring task of the same workgroup
  std::vector<TaskRec> tasks(size_t(n_blocks * GPB), TaskRec{-1, -1, 0, -1});
  {
    size_t c = 0;  // cursor: first sparse column whose slots end after the task's start
    const auto cend = [&](size_t cc) { return cptr[cc] + (hot_rank[cc] >= 0 ? 0 : len(cc)); };
    for (int64_t t = 0; t < n_blocks * GPB; ++t) {
      const int64_t b0 = t * C, e0 = b0 + C;
      while (c < nf && (hot_rank[c] >= 0 || len(c) == 0 || cend(c) <= b0)) ++c;
      if (c >= nf || cptr[c] >= e0) continue;  // nothing in this task
      size_t cl = c;  // last column with a slot in the task
      for (size_t nx = c + 1; nx < nf && cptr[nx] < e0; ++nx)
        if (hot_rank[nx] < 0 && len(nx) > 0) cl = nx;
      TaskRec& tr = tasks[size_t(t)];
      tr.first_col = int32_t(c);
      tr.last_col = int32_t(cl);
      const bool first_in_block = t % GPB == 0, last_in_block = t % GPB == GPB - 1;
      if (cptr[c] < b0 && !first_in_block) tr.flags |= 1;
      if (cend(cl) > e0 && !last_in_block) tr.flags |= 2;
    }
  }
  for (const auto& pb : piece_blocks) tasks[size_t(pb.first * GPB)].part = pb.second;
  std::vector<SplitCol> split(split_short);
  split.insert(split.end(), split_long.begin(), split_long.end());
  lap("classes, tasks, split columns (host)");

  // ---- entry and slot records ----------------------------------------------------------
  plan->n_slots = n_slots;
  plan->n_task_blocks = int32_t(n_blocks);
  plan->task_words = int32_t(W);
  plan->n_split_short = int32_t(split_short.size());
  plan->n_split_long = int32_t(split_long.size());
  plan->n_parts = n_parts;
  plan->n_hot = int32_t(hot_cols.size());
  plan->hot_fixed = fixed && !hot_cols.empty();
  plan->h_hot_cols = hot_cols;
  plan->fwd_grid_max = forward_grid(ctx, max_batch, n_factors);
  DevBuf d_colinfo;
  upload(d_colinfo, colinfo.data(), nf * 4, st);
  plan->ent.alloc((nz + 1) * sizeof(Entry));  // +1: clamp target of empty logs
  RFM_HIP_CHECK(hipMemsetAsync(plan->ent.as<Entry>() + nz, 0, sizeof(Entry), st));
  // (slots between tasks -- the padding to whole bitmap words -- are never marked)
  plan->slots.alloc((ns + 256) * sizeof(SlotRec));
  RFM_HIP_CHECK(hipMemsetAsync(plan->slots.p, 0, plan->slots.bytes, st));
  if (nnz > 0) {
    hipLaunchKernelGGL(plan_scatter_kernel, dim3(grid_for(ctx, nnz)), dim3(kBlock), 0, st,
                       key.as<int32_t>(), pos.as<int32_t>(), d_values, d_colinfo.as<int32_t>(),
                       nnz, plan->ent.as<Entry>(), plan->slots.as<SlotRec>());
    RFM_HIP_CHECK(hipGetLastError());
  }
  // every row fits one round of a lane group: the many-rows forward reads padded row blocks
  // (one dependent load less per row: 59.2 vs 61.0 us per step at B = 65 536 on config 3; the
  // one-row shape of small batches measured 0.9 us slower with them, so plans for small
  // batches keep the plain records); the plain records are then not kept
  const int64_t max_len = h_flags[3];
  plan->max_row_len = int32_t(max_len);
  plan->hot_rounds = int32_t(std::max<int64_t>(1, (max_len + shp.lpr - 1) / shp.lpr));
  // ... and, with RFM_PREP=1 (an experiment that is measured and NOT the default: it removes a
  // dependent level from each of the step's launches and changes nothing measurable --
  // profiles/r3i), plans for small batches keep the row blocks as the source
  // of PREPARED steps (rfm_fm_prep.hpp)
  const bool many_rows = forward_many_rows(ctx, max_batch, n_factors);
  const int prep_mode = many_rows ? 0 : env_int("RFM_PREP", 0);
  const bool want_prep = prep_mode == 1;
  if (nnz > 0 && max_len <= shp.lpr && (many_rows || want_prep) && env_int("RFM_NO_ELL", 0) == 0) {
    plan->ell_stride = int64_t(shp.lpr) * int64_t(sizeof(Entry));
    plan->ell.alloc(nr * size_t(plan->ell_stride));
    plan->ell_yp.alloc(nr * 16);
    hipLaunchKernelGGL(plan_ell_kernel, dim3(grid_for(ctx, n_rows * shp.lpr)), dim3(kBlock), 0, st,
                       plan->rows.as<RowRec>(), plan->ent.as<Entry>(), n_rows, shp.lpr,
                       plan->ell.as<char>(), plan->ell_stride, plan->ell_yp.as<double2>());
    RFM_HIP_CHECK(hipGetLastError());
    RFM_HIP_CHECK(hipStreamSynchronize(st));
    plan->ent.release();
    plan->rows.release();
    if (want_prep) {
      // iterations per chunk: what 384 MiB hold twice (two chunks are alive at a time)
      plan->n_tasks = int32_t(n_blocks * GPB);
      const size_t per_iter = size_t(max_batch) * (size_t(plan->ell_stride) + 16) +
                              size_t(plan->n_tasks) * (size_t(kPrepCap) * 32 + 4);
      const size_t fit = (size_t(env_int("RFM_PREP_MB", 384)) << 20) / std::max<size_t>(per_iter, 1);
      plan->prep_iters = int32_t(std::min<size_t>(fit, 64));
      plan->prep_ok = plan->prep_iters >= 8;
    }
  }
  if (prep_mode == 2 && nnz > 0) {  // records only: nothing but the tasks' buckets per iteration
    plan->n_tasks = int32_t(n_blocks * GPB);
    const size_t per_iter = size_t(plan->n_tasks) * (size_t(kPrepCap) * 32 + 4);
    const size_t fit = (size_t(env_int("RFM_PREP_MB", 384)) << 20) / std::max<size_t>(per_iter, 1);
    plan->prep_iters = int32_t(std::min<size_t>(fit, 64));
    plan->prep_ok = plan->prep_iters >= 8;
    plan->prep_records_only = true;
  }
  // sliced loss forwards (rfm_fm_sliced.hpp): factor counts of several chunks per lane, even
  // (16-byte loads).  Slices: the power of two that covers k in 256-factor pieces (so that the
  // slices divide the eight XCDs), evenly wide.  Cached columns: at least one row in 64 holds
  // them, most frequent first, as many as the LDS left beside the staged rows takes.
  std::vector<int32_t> sl_cols, sl_rank;
  if (shp.nc > 1 && n_factors % 2 == 0 && n_factors <= 1024 && nnz > 0) {
    int ns = 1;  // (a wavefront covers 256 factors of a row: four per lane)
    while (ns * 256 < n_factors) ns *= 2;
    const int sw = ((n_factors + ns - 1) / ns + 1) & ~1;
    // (... and while the translated training log stays below 4 GiB)
    if (ns <= 8 && int64_t(ns - 1) * sw < n_factors && (n_rows << 6) * 16 <= (int64_t(4) << 30)) {
      plan->sl_ns = ns;
      plan->sl_sw = sw;
      int mll = 4;
      while ((int64_t(1) << mll) < max_len && mll < 6) ++mll;
      plan->sl_ml_log2 = mll;
      for (size_t c = 0; c < nf; ++c)
        if (len(c) * 64 >= n_rows) sl_cols.push_back(int32_t(c));
      std::stable_sort(sl_cols.begin(), sl_cols.end(),
                       [&](int32_t x, int32_t y) { return len(size_t(x)) > len(size_t(y)); });
      const size_t cap = size_t(kSlicedLds - kSlicedLdsSlack - 1024) / size_t(sliced_row_bytes(sw)) - 1;  // (+ the zero row)
      if (sl_cols.size() > cap) sl_cols.resize(cap);
      sl_rank.assign(nf, -1);
      for (size_t h = 0; h < sl_cols.size(); ++h) sl_rank[size_t(sl_cols[h])] = int32_t(h);
      plan->sl_n_cached = int32_t(sl_cols.size());
      if (sl_cols.empty())
        plan->sl_cols.alloc(4);
      else
        upload(plan->sl_cols, sl_cols.data(), sl_cols.size() * 4, st);
      upload(plan->sl_rank, sl_rank.data(), nf * 4, st);
      const SlEnt pad_record{plan->sl_n_cached * sliced_row_bytes(sw), kSlPad, 0.0};
      upload(plan->sl_pad, &pad_record, sizeof(SlEnt), st);
      // the training log in its translated form (rows of 2^mll records)
      plan->sl_train.alloc((nr << mll) * sizeof(SlEnt));
      hipLaunchKernelGGL(sl_translate_kernel, dim3(grid_for(ctx, n_rows << mll)), dim3(kBlock), 0, st,
                         d_indptr, d_indices, d_values, n_rows, plan->sl_rank.as<int32_t>(), mll,
                         plan->sl_n_cached, sliced_row_bytes(sw), plan->sl_train.as<SlEnt>());
      RFM_HIP_CHECK(hipGetLastError());
    }
  }
  upload(plan->tasks, tasks.data(), tasks.size() * sizeof(TaskRec), st);
  upload(plan->split, split.data(), split.size() * sizeof(SplitCol), st);
  upload(plan->hot_cols, hot_cols.data(), hot_cols.size() * 4, st);
  plan->slot_t.alloc((ns + 256) * sizeof(SlotMark));
  RFM_HIP_CHECK(hipMemsetAsync(plan->slot_t.p, 0, plan->slot_t.bytes, st));
  // (two bitmaps, used by alternate steps: the chunked form of fm_consume_kernel clears the
  // other step's; the single-chunk form only ever uses the first)
  plan->bits_words = int64_t(ns / 64 + 8);
  plan->slot_bits.alloc(size_t(plan->bits_words) * 2 * 8);
  RFM_HIP_CHECK(hipMemsetAsync(plan->slot_bits.p, 0, plan->slot_bits.bytes, st));
  // partial rows [n_parts][k+3]; stamp 0 never matches a step id (they start at 1)
  plan->parts.alloc(std::max<size_t>(size_t(n_parts), 1) * size_t(n_factors + 3) * 8);
  RFM_HIP_CHECK(hipMemsetAsync(plan->parts.p, 0, plan->parts.bytes, st));
  plan->hot_slab.alloc(size_t(kMaxFwdGrid) * std::max<size_t>(hot_cols.size(), 1) *
                       size_t(n_factors + 2) * 8);
  plan->err_partial.alloc(size_t(kMaxFwdGrid) * 8);
  plan->Q.alloc(size_t(max_batch) * size_t(n_factors) * 8);
  plan->err.alloc(size_t(max_batch) * 8);
  // the host vectors and the transient device buffers die at scope exit: wait for the stream
  RFM_HIP_CHECK(hipStreamSynchronize(st));
  lap("entry / slot records, uploads, scratch");
  return plan.release();
}

}  // namespace

void sliced_translate(rfm_ctx* ctx, const rfm_fm_plan* plan, const int64_t* d_indptr,
                      const int32_t* d_indices, const double* d_values, int64_t n_rows, DevBuf& out) {
  const int mll = plan->sl_ml_log2;
  out.ensure((size_t(std::max<int64_t>(n_rows, 1)) << mll) * sizeof(SlEnt));
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(sl_translate_kernel, dim3(grid_for(ctx, n_rows << mll)), dim3(kBlock), 0,
                     ctx->stream, d_indptr, d_indices, d_values, n_rows,
                     plan->sl_rank.as<int32_t>(), mll, plan->sl_n_cached,
                     sliced_row_bytes(plan->sl_sw), out.as<SlEnt>());
  RFM_HIP_CHECK(hipGetLastError());
}

}  // namespace rfm

using namespace rfm;

extern "C" {

int32_t rfm_fm_plan_create_device(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                                  const double* d_values, const double* d_y,
                                  const double* d_pscore, int64_t n_rows, int64_t n_features,
                                  int32_t n_factors, int64_t max_batch, int32_t hot_min_count,
                                  rfm_fm_plan** out) {
  return guarded([&] {
    RFM_REQUIRE(out, "null output pointer");
    *out = build_plan(ctx, d_indptr, d_indices, d_values, d_y, d_pscore, n_rows, n_features,
                      n_factors, max_batch, hot_min_count);
  });
}

int32_t rfm_fm_plan_create(rfm_ctx* ctx, const int64_t* h_indptr, const int32_t* h_indices,
                           const double* h_values, const double* h_y, const double* h_pscore,
                           int64_t n_rows, int64_t n_features, int32_t n_factors,
                           int64_t max_batch, int32_t hot_min_count, rfm_fm_plan** out) {
  return guarded([&] {
    RFM_REQUIRE(ctx && h_indptr && h_y && h_pscore && out, "null pointer");
    RFM_REQUIRE(n_rows >= 1 && n_features >= 1 && max_batch >= 1, "bad shape");
    const int64_t nnz = h_indptr[n_rows];
    RFM_REQUIRE(nnz >= 0 && nnz < (int64_t(1) << 31), "nnz=%lld unsupported", (long long)nnz);
    RFM_REQUIRE(nnz == 0 || (h_indices && h_values), "null CSR arrays");
    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    // a transient device copy of the log; the plan is built from it on the device
    DevBuf indptr, indices, values, y, p;
    const auto up = [&](DevBuf& dst, const void* src, size_t bytes) {
      dst.alloc(bytes);
      if (bytes) RFM_HIP_CHECK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    };
    up(indptr, h_indptr, size_t(n_rows + 1) * 8);
    up(indices, h_indices, size_t(nnz) * 4);
    up(values, h_values, size_t(nnz) * 8);
    up(y, h_y, size_t(n_rows) * 8);
    up(p, h_pscore, size_t(n_rows) * 8);
    *out = build_plan(ctx, indptr.as<int64_t>(), indices.as<int32_t>(), values.as<double>(),
                      y.as<double>(), p.as<double>(), n_rows, n_features, n_factors, max_batch,
                      hot_min_count);
  });
}

int32_t rfm_fm_plan_destroy(rfm_fm_plan* plan) {
  return guarded([&] {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    delete plan;
  });
}

int32_t rfm_fm_plan_info(const rfm_fm_plan* plan, int64_t* h_out8) {
  return guarded([&] {
    RFM_REQUIRE(plan && h_out8, "null pointer");
    h_out8[0] = int64_t(plan->n_task_blocks) * (rfm::kBlock / rfm::shape_for(plan->k).lpr);
    h_out8[1] = plan->n_split_short + plan->n_split_long;
    h_out8[2] = plan->n_hot;
    h_out8[3] = plan->nnz;
    h_out8[4] = int64_t(plan->device_bytes());
    h_out8[5] = plan->fwd_grid_max;
    h_out8[6] = plan->n_slots;
    h_out8[7] = plan->task_words;
  });
}

int32_t rfm_fm_plan_register_log(rfm_ctx* ctx, rfm_fm_plan* plan, int32_t slot, const int64_t* d_indptr,
                                 const int32_t* d_indices, const double* d_values, int64_t n_rows) {
  return guarded([&] {
    RFM_REQUIRE(ctx && plan && (slot == 0 || slot == 1), "null pointer / slot outside 0..1");
    rfm_fm_plan::SlLog& log = plan->sl_log[slot];
    log.rows = -1;
    log.records = false;
    if (n_rows <= 0 || !d_indptr) return;  // (nothing to keep)
    if (plan->sl_ns > 0) sliced_translate(ctx, plan, d_indptr, d_indices, d_values, n_rows, log.tr);
    if (slot == 0 && n_rows < (int64_t(1) << 31)) {  // the validation log as records (it may ride, rfm_fm_train)
      int64_t nnz = 0;
      RFM_HIP_CHECK(hipMemcpyAsync(&nnz, d_indptr + n_rows, 8, hipMemcpyDeviceToHost, ctx->stream));
      RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (nnz >= 0 && nnz < (int64_t(1) << 31) - 2) {
        log.rows_rec.ensure(size_t(n_rows) * sizeof(RowRec));
        log.ent_rec.ensure(size_t(nnz + 1) * sizeof(Entry));
        RFM_HIP_CHECK(hipMemsetAsync(log.ent_rec.as<Entry>() + nnz, 0, sizeof(Entry), ctx->stream));
        hipLaunchKernelGGL(log_records_kernel, dim3(grid_for(ctx, n_rows)), dim3(kBlock), 0, ctx->stream,
                           d_indptr, d_indices, d_values, n_rows, log.rows_rec.as<RowRec>(),
                           log.ent_rec.as<Entry>());
        RFM_HIP_CHECK(hipGetLastError());
        log.records = true;
      }
    }
    log.indptr = d_indptr;
    log.indices = d_indices;
    log.values = d_values;
    log.rows = n_rows;
  });
}

int32_t rfm_fm_plan_sliced(const rfm_fm_plan* plan, int32_t* h_out4) {
  return guarded([&] {
    RFM_REQUIRE(plan && h_out4, "null pointer");
    h_out4[0] = plan->sl_ns;
    h_out4[1] = plan->sl_ns > 0 ? plan->sl_sw : 0;
    h_out4[2] = plan->sl_ns > 0 ? plan->sl_n_cached : 0;
    h_out4[3] = plan->sl_ns > 0 ? 1 << plan->sl_ml_log2 : 0;
  });
}

int32_t rfm_fm_plan_layout(const rfm_fm_plan* plan, int32_t* h_out4) {
  return guarded([&] {
    RFM_REQUIRE(plan && h_out4, "null pointer");
    h_out4[0] = plan->ell.p ? 1 : 0;
    h_out4[1] = plan->ell.p ? int32_t(plan->ell_stride) : 0;
    h_out4[2] = rfm::shape_for(plan->k).lpr;
    h_out4[3] = plan->max_row_len;
  });
}

int32_t rfm_fm_plan_hot_columns(const rfm_fm_plan* plan, int32_t* h_out, int32_t capacity) {
  return guarded([&] {
    RFM_REQUIRE(plan && (h_out || capacity == 0), "null pointer");
    RFM_REQUIRE(capacity >= plan->n_hot, "capacity %d < %d hot columns", capacity, plan->n_hot);
    std::copy(plan->h_hot_cols.begin(), plan->h_hot_cols.end(), h_out);
  });
}

}  // extern "C"
