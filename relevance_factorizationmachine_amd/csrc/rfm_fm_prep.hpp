// Prepared steps of the fit() loop (rfm_fm_train): the part of a step that depends on the row
// ids alone -- which rows, hence which entries, hence which slots of which tasks -- laid out
// AHEAD of the loop, a chunk of iterations per launch, so that the step's two launches start
// at their data instead of discovering it (src/fm.py:72-79: the batch of iteration `epoch` is
// resample(..., random_state=epoch): known before anything is trained).
//   fm_prep_gather_kernel  per (iteration, batch position): the row's padded block + {label,
//                          propensity} copied into batch order; every sparse-class entry
//                          dropped into its task's bucket (unordered, atomic cursor)
//   fm_prep_sort_kernel    per (iteration, task): the bucket ranked by slot -> PrepRec records
//                          in slot order (the order fm_consume_kernel adds in)
// The forward then reads its rows at E[t] (no row id -> row block indirection, no marks to
// leave), the gradient launch reads its records at R[task] (no bitmap to scan).
#pragma once

#include "rfm_fm_kernels.hpp"

namespace rfm {

// grid-stride over (iteration, batch position, entry position of the row block)
__global__ __launch_bounds__(kBlock) void fm_prep_gather_kernel(
    const char* ell, int64_t ell_stride, const double2* ell_yp, int lpr, const int32_t* ids,
    int64_t ids_stride, int64_t batch, int n_it, int32_t task_slots, int32_t n_tasks, Entry* E,
    double2* YP, PrepTmp* tmp, int32_t* cnt) {
  const int64_t total = int64_t(n_it) * batch * lpr;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * kBlock) {
    const int j = int(i % lpr);
    const int64_t row = i / lpr;  // it * batch + t
    const int it = int(row / batch);
    const int32_t t = int32_t(row % batch);
    const int32_t r = ids[int64_t(it) * ids_stride + t];
    const Entry e = reinterpret_cast<const Entry*>(ell + int64_t(r) * ell_stride)[j];
    E[i] = e;
    if (j == 0) YP[row] = ell_yp[r];
    if (e.slot >= 0) {  // sparse class (hot entries carry -1 - rank, padding kNilSlot)
      const int32_t task = e.slot / task_slots;
      const int64_t bucket = int64_t(it) * n_tasks + task;
      const int pos = atomicAdd(&cnt[bucket], 1);
      if (pos < kPrepCap) tmp[bucket * kPrepCap + pos] = PrepTmp{e.slot, t, e.x};
    }
  }
}

// The same buckets from the plan's plain records (RowRec / Entry), without copying the rows
// (RFM_PREP=2: "records only" -- the forward keeps reading the plan's records by row id and
// merely stops leaving marks): a wave per (iteration, batch position).
__global__ __launch_bounds__(kBlock) void fm_prep_bucket_kernel(
    const RowRec* rows, const Entry* ent, const int32_t* ids, int64_t ids_stride, int64_t batch,
    int n_it, int32_t task_slots, int32_t n_tasks, PrepTmp* tmp, int32_t* cnt) {
  const int lane = threadIdx.x % kWave;
  const int64_t n_rows = int64_t(n_it) * batch;
  for (int64_t row = (int64_t(blockIdx.x) * kBlock + threadIdx.x) / kWave; row < n_rows;
       row += int64_t(gridDim.x) * (kBlock / kWave)) {
    const int it = int(row / batch);
    const int32_t t = int32_t(row % batch);
    const RowRec rec = rows[ids[int64_t(it) * ids_stride + t]];
    for (int64_t j = lane; j < rec.len; j += kWave) {
      const Entry e = ent[rec.begin + j];
      if (e.slot >= 0) {
        const int64_t bucket = int64_t(it) * n_tasks + e.slot / task_slots;
        const int pos = atomicAdd(&cnt[bucket], 1);
        if (pos < kPrepCap) tmp[bucket * kPrepCap + pos] = PrepTmp{e.slot, t, e.x};
      }
    }
  }
}

// one 16-lane group per (iteration, task); flags[it] = 1 when a bucket overflowed (that
// iteration then takes the unprepared path)
__global__ __launch_bounds__(kBlock) void fm_prep_sort_kernel(const PrepTmp* tmp, const int32_t* cnt,
                                                             const SlotRec* slots, int32_t n_tasks,
                                                             int n_it, PrepRec* rec, int32_t* flags) {
  constexpr int G = 16, GPB = kBlock / G;
  __shared__ PrepTmp lds[GPB][kPrepCap];
  const int l = threadIdx.x % G, g = threadIdx.x / G;
  const int64_t n_buckets = int64_t(n_it) * n_tasks;
  for (int64_t b = int64_t(blockIdx.x) * GPB + g; b < n_buckets; b += int64_t(gridDim.x) * GPB) {
    const int have = cnt[b];
    if (have == 0) continue;
    if (have > kPrepCap) {
      if (l == 0) flags[b / n_tasks] = 1;
      continue;
    }
    for (int i = l; i < have; i += G) lds[g][i] = tmp[b * kPrepCap + i];
    // (same 16 lanes of one wave wrote and read: no barrier needed)
    for (int i = l; i < have; i += G) {
      const PrepTmp me = lds[g][i];
      int rank = 0;
      for (int j = 0; j < have; ++j) rank += lds[g][j].slot < me.slot ? 1 : 0;  // slots are distinct
      rec[b * kPrepCap + rank] = PrepRec{me.t, slots[me.slot].col, me.x};
    }
  }
}

}  // namespace rfm
