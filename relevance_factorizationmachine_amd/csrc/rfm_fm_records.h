// Records of the FM training plan (shared by the kernels and the plan builder).
#pragma once

#include <cstdint>

namespace rfm {

constexpr int kBlock = 256;
constexpr int kWave = 64;

struct Entry {       // one CSR entry of the training log, 16 B
  int32_t col;       // feature column
  int32_t slot;      // >= 0: slot of the sparse class; < 0: hot column -1-slot
  double x;          // feature value
};
struct RowRec {      // one row of the training log, 32 B
  int64_t begin;     // first entry
  int64_t len;       // number of entries
  double y;          // label
  double p;          // propensity (already raised to pow_used by the loader)
};
// The padded form of the log (plans whose longest row fits one round of a lane group): row
// blocks Entry[LPR] at a fixed stride (256 B at k = 32: two whole cache lines), so that a row's
// entries are ONE dependent load behind the row id (not row record -> entries).  The entries
// past the row's end repeat its last entry with x = 0 and slot kNilSlot: the row's length is
// the count of the others.  {label, propensity} pairs sit in a plain array by row.
constexpr int32_t kNilSlot = INT32_MIN;
struct SlotRec {  // one slot of the column-major view, 16 B
  double x;       // feature value
  int32_t col;    // feature column
  int32_t pad;
};
struct SlotMark {  // what the forward leaves at a marked slot, 16 B
  int32_t t;       // batch position of the slot's row
  int32_t pad;
  double err;      // the row's residual y/p - sigmoid(logit)
};
struct TaskRec {      // one task of fm_consume_kernel: task_words x 64 consecutive slots; 16 B
  int32_t first_col;  // column of the task's first occupied slot
  int32_t last_col;   // column of its last occupied slot
  int32_t flags;      // bit0: first_col continues from the previous task (same workgroup)
                      // bit1: last_col continues into the next task (same workgroup)
  int32_t part;       // >= 0: the task starts a workgroup that lies inside a column longer than a
                      // workgroup's tasks; the workgroup's sums go to this partial row
};
#ifndef RFM_TASK_TRIPS
#define RFM_TASK_TRIPS 1
#endif
constexpr int kTaskTrips = RFM_TASK_TRIPS;  // bitmap words of its task a lane loads (task_words <= lanes of a group)
struct SplitCol {      // a sparse-class column longer than a whole workgroup's tasks
  int32_t col;
  int32_t part_begin;  // its partial rows: parts[part_begin .. +part_count), in slot order
  int32_t part_count;
  int32_t pad;
};

// Prepared steps (rfm_fm_train): what a step needs that depends on the row ids alone is laid
// out ahead of the loop, many iterations per launch -- the batch's row blocks in batch order
// and, per task, the batch's entries of the task's slots in slot order (PrepRec), at a fixed
// place: a task's records are then the gradient launch's FIRST load.
constexpr int kPrepCap = 64;  // records of a task kept in place (more: the iteration is not prepared)
struct PrepTmp {  // as gathered, unordered
  int32_t slot;
  int32_t t;
  double x;
};
struct PrepRec {  // in slot order
  int32_t t;      // batch position of the entry's row
  int32_t col;    // feature column
  double x;       // feature value
};


// Sliced loss forwards (rfm_fm_sliced.hpp): LDS of a workgroup (the cached columns' slices)
constexpr int kSlicedLds = 160 << 10;
constexpr int kSlicedLdsSlack = 2048;  // readable bytes after the last row (unclamped reads of idle lanes)
struct SlEnt {   // one entry of a translated log, 16 B.  A row's records: the cached columns'
                 // entries first (in row order), then the others, then padding
  int32_t off;   // byte offset of the column's slice in the workgroup's LDS copy; the ZERO row's
                 // (n_cached rows in) for a column that is not cached and for padding
  int32_t col;   // the column; kSlPad: no entry; kSlLong (record 0): the row is longer than the
                 // records of a row -- read it from the CSR arrays
  double x;
};
constexpr int32_t kSlPad = -1;
constexpr int32_t kSlLong = -2;
// bytes of a cached column's LDS row: its slice, its squared norm, 8 bytes of padding
constexpr int sliced_row_bytes(int sw) { return sw * 8 + 16; }

}  // namespace rfm
