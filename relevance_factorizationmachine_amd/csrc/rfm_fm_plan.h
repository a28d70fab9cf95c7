// The FM training plan: the one-time (per fit) device layout of a training log that the
// step kernels read.  Built by rfm_fm_plan.hip (on the device), used by rfm_fm.hip.
#pragma once

#include <vector>

#include "rfm_common.h"

struct rfm_fm_plan {
  int32_t device = 0;
  int64_t n_rows = 0, n_features = 0, nnz = 0, n_slots = 0, max_batch = 0;
  int32_t k = 0;
  int32_t n_task_blocks = 0;  // workgroups of fm_consume_kernel that process tasks
  int32_t task_words = 1;     // 64-slot words per task
  int32_t n_split_short = 0, n_split_long = 0, n_parts = 0, n_hot = 0;
  int32_t max_row_len = 0;  // entries of the longest row of the log
  int32_t hot_rounds = 1;  // rounds of a lane group's entries that cover the longest row
  bool hot_fixed = false;  // hot-class sums in a fixed order (hot_min_count = -2)
  int64_t step = 0;  // stamps the partial rows of a step
  int64_t bits_words = 0;  // 64-slot words of ONE slot bitmap (slot_bits holds two)
  int32_t fwd_grid_max = 0;  // forward workgroups of a max_batch step (= hot-sum slabs)
  rfm::DevBuf ell;            // padded row blocks (every row <= lanes-per-group entries), else empty
  rfm::DevBuf ell_yp;         // ... with the rows' {label, propensity} pairs
  int64_t ell_stride = 0;
  rfm::DevBuf ent, rows, slot_t, slot_bits, slots, tasks, split, parts, Q, err, hot_cols, hot_slab,
      err_partial;
  rfm::DevBuf loss_rows;  // rfm_fm_train: per-workgroup loss partials of a run of iterations
  // sliced loss forwards of rfm_fm_train (rfm_fm_sliced.hpp; k > 128 and even): slices of the
  // factors, factors per slice, records per translated row (log2), and the log's most frequent columns --
  // the ones a workgroup keeps in LDS -- as a list and as a rank per column
  int32_t sl_ns = 0, sl_sw = 0, sl_ml_log2 = 4, sl_n_cached = 0;
  rfm::DevBuf sl_cols, sl_rank;
  rfm::DevBuf sl_train;  // the training log translated (rows of 2^sl_ml_log2 SlEnt records)
  rfm::DevBuf sl_pad;    // one padding record
  // ... of the validation log of the current rfm_fm_train call (slot 0) and of the log that
  // rfm_fm_plan_forward scores (slot 1).  rfm_fm_plan_register_log: the arrays a slot was
  // translated from -- calls that name the same arrays skip the translation (a fit() that trains
  // one iteration per call and scores an evaluation log after each)
  struct SlLog {
    rfm::DevBuf tr;
    rfm::DevBuf rows_rec, ent_rec;  // slot 0: the log as RowRec / Entry records (what the step's
                                    // forward launch reads when the validation rows ride in it)
    bool records = false;
    const void* indptr = nullptr;
    const void* indices = nullptr;
    const void* values = nullptr;
    int64_t rows = -1;  // -1: nothing registered
    bool holds(const void* ip, const void* ix, const void* v, int64_t n) const {
      return rows == n && indptr == ip && indices == ix && values == v;
    }
  } sl_log[2];
  rfm::DevBuf sl_z;      // partial logits [iterations of a run][slices][rows]
  rfm::DevBuf sl_zf;     // ... of rfm_fm_plan_forward's rows
  std::vector<int32_t> h_hot_cols;  // host copy of hot_cols (rfm_fm_plan_hot_columns)
  // touched-row gradients (rfm_fm_grad_rows), allocated on first use: the gradient table
  // [G_V | g_w | g_w0] indexed by column, never cleared; touch[col] == touch_seq marks the
  // rows of the current step
  rfm::DevBuf row_table, touch, chunk_cnt;
  int32_t touch_seq = 0;
  // rfm_fm_fit_dp (grown on demand, kept between calls): dense gradient, loss sums, record
  // lists (mine / received / everybody's updated rows), transfer-plan arrays, small scratch
  rfm::DevBuf dp_grad, dp_sums, dp_rows, dp_recv, dp_all, dp_bounds, dp_all_bounds, dp_seg,
      dp_range_lo, dp_small;
  // prepared steps of rfm_fm_train (rfm_fm_prep.hpp): two chunk buffers, each holding the
  // batch-ordered row blocks E / YP and the tasks' records (tmp -> rec, cnt) of prep_iters
  // iterations, and the overflow flags of a chunk (device + pinned host copy + its event)
  bool prep_ok = false;       // the plan's layout allows prepared steps (padded row blocks, small batches)
  bool prep_records_only = false;  // RFM_PREP=2: the tasks' records only, rows stay where they are
  int32_t prep_iters = 0;     // iterations per chunk
  int32_t n_tasks = 0;        // tasks of the plan (n_task_blocks x lane groups per workgroup)
  struct PrepChunk {
    rfm::DevBuf E, YP, tmp, rec, cnt, flags;
    int32_t* h_flags = nullptr;  // pinned
    hipEvent_t ready = nullptr;
  } prep[2];
  ~rfm_fm_plan() {
    for (auto& c : prep) {
      if (c.h_flags) (void)hipHostFree(c.h_flags);
      if (c.ready) (void)hipEventDestroy(c.ready);
    }
  }
  rfm::DevBuf ids_seen, ids_flags;  // RFM_CHECK_IDS=1: validation of the steps' row ids
  int32_t ids_stamp = 0;
  size_t device_bytes() const {
    return ell.bytes + ell_yp.bytes + ent.bytes + rows.bytes + slot_t.bytes + slot_bits.bytes + slots.bytes + tasks.bytes +
           split.bytes + parts.bytes + Q.bytes + err.bytes + hot_cols.bytes + hot_slab.bytes +
           err_partial.bytes + sl_cols.bytes + sl_rank.bytes + sl_train.bytes + sl_log[0].tr.bytes +
           sl_log[1].tr.bytes + sl_z.bytes + sl_zf.bytes;
  }
};

namespace rfm {

// bytes of LDS a forward workgroup spends on the hot class sums (next to 8 KiB of
// reduction scratch and the 33 KiB entry buffer of the 1024-thread shape: gfx950 gives a
// workgroup up to 160 KiB)
constexpr size_t kHotLdsBudget = 56 << 10;
constexpr size_t kHotLdsBudgetFixed = 40 << 10;  // ... when the fixed-order form shares the LDS
constexpr int kMaxHot = 160;  // beyond this the slab traffic outweighs what the class saves
constexpr int32_t kDefaultHotMinCount = 32;
constexpr int32_t kShortSplit = 8;   // split columns up to this many partial rows: one lane group
// fm_consume_kernel's tasks: task_words is the power of two for which a task expects about
// kTaskMarks marked slots per max_batch step (at most kTaskTrips words per lane of a group)
constexpr int kTaskMarks = 8;
constexpr int kMaxFwdGrid = 2048;   // upper bound of the forward's grid, sizes scratch

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// a log translated for the sliced loss forward of `plan` (rfm_fm_sliced.hpp), on ctx's stream
// (defined in rfm_fm_plan.hip)
void sliced_translate(rfm_ctx* ctx, const rfm_fm_plan* plan, const int64_t* d_indptr,
                      const int32_t* d_indices, const double* d_values, int64_t n_rows, DevBuf& out);

// workgroups the training forward launches for `rows` batch rows (defined in rfm_fm.hip)
int forward_grid(const rfm_ctx* ctx, int64_t rows, int n_factors);
// whether that forward takes the many-rows-in-flight shape
bool forward_many_rows(const rfm_ctx* ctx, int64_t rows, int n_factors);
// the forward shapes a plan with this max_batch can take are built for fixed-order hot sums
bool forward_fixed_order_ok(const rfm_ctx* ctx, int64_t max_batch, int n_factors);

}  // namespace rfm
