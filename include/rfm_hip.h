/*
 * rfm_hip.h -- C ABI of librfm_hip.so, the MI355X (gfx950) implementation of the
 * FM / MF mini-batch SGD path of tatsuki1107/Relevance-FactorizationMachine.
 *
 * The reference has no FFI of its own (it is pure Python); the boundary it
 * offers is the Python class surface of src/base.py, src/fm.py, src/mf.py and
 * utils/optimizer.py.  Each entry point below names the reference code
 * (file:line, relative to the reference root) whose arithmetic it replaces.
 * The Python mirror of that surface (relevance_factorizationmachine_amd/fm.py,
 * mf.py) calls these through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every call returns int32: RFM_OK or an RFM_ERR_* class; the message is
 *    fetched with rfm_last_error() (thread-local).  No C++ exception crosses.
 *  - pointers named d_* are DEVICE pointers owned by the caller (torch tensors
 *    in the Python mirror); h_* are HOST pointers.  The library allocates only
 *    ctx-/plan-owned scratch and frees it in the matching destroy call.
 *  - calls that take a ctx are asynchronous on the ctx's HIP stream; use
 *    rfm_sync() (or synchronise the stream yourself) before reading results.
 *  - a CSR row names a column at most once (SciPy sums duplicate entries before
 *    it squares them, src/fm.py:127; the Python mirror canonicalises its inputs the
 *    same way); the order of a row's entries is free, explicit zeros are fine.
 *  - all parameters and activations are float64 (the reference computes in
 *    NumPy float64); CSR column indices and row ids are int32, CSR row
 *    pointers int64, labels are passed as float64 {0,1}.
 *  - one ctx per (host thread, device); no global mutable state.
 */
#ifndef RFM_HIP_H
#define RFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFM_VERSION 100 /* 0.1.0 */

enum {
  RFM_OK = 0,
  RFM_ERR_BAD_ARG = 1, /* caller error: shapes, null pointers, unsupported k */
  RFM_ERR_HIP = 2,     /* a HIP runtime call or kernel launch failed */
  RFM_ERR_NO_DEVICE = 3,
  RFM_ERR_INTERNAL = 4
};

/* limits of this build */
#define RFM_MAX_FACTORS 1024
/* levels the sequential MF kernel reads an example's rows ahead of its own level */
#define RFM_MF_READ_AHEAD 4
/* rfm_mf_schedule_ex: gap of an example whose user row no earlier example of the batch writes */
#define RFM_MF_NO_WRITER (1 << 30)

typedef struct rfm_ctx rfm_ctx;
typedef struct rfm_fm_plan rfm_fm_plan;

/* ---- diagnostics ------------------------------------------------------- */
int32_t rfm_version(void);
/* copies the calling thread's last error message (NUL-terminated) */
int32_t rfm_last_error(char* buf, size_t n);

/* ---- context ----------------------------------------------------------- */
/* hip_stream: a hipStream_t to run on (NULL = the device's default stream). */
int32_t rfm_create(int32_t device, void* hip_stream, rfm_ctx** out);
int32_t rfm_destroy(rfm_ctx* ctx);
int32_t rfm_sync(rfm_ctx* ctx);
/* Synchronous copies ordered after the work on the ctx stream (for a host-staged
 * rfm_transport, which is handed raw device pointers). */
int32_t rfm_copy_to_host(rfm_ctx* ctx, void* h_dst, const void* d_src, int64_t bytes);
int32_t rfm_copy_to_device(rfm_ctx* ctx, void* d_dst, const void* h_src, int64_t bytes);

/* ---- per-kernel timing (HIP events on the ctx stream) ---------------------
 * Between rfm_profile_begin and rfm_profile_end every FM training step records
 * events around its launches.  rfm_profile_end synchronises and returns, per
 * phase, the summed milliseconds and the number of launches:
 *   phase 0 forward (+residual, Q, marks)   phase 1 column gradient/update
 *   phase 2 finalize (long columns, w0)     phase 3 whole step
 * h_ms[4], h_count[4]. */
int32_t rfm_profile_begin(rfm_ctx* ctx);
int32_t rfm_profile_end(rfm_ctx* ctx, double* h_ms, int64_t* h_count);

/* ---- mini-batch selection (host) ---------------------------------------
 * Replaces sklearn.utils.resample(X, y, p, replace=False, n_samples=B,
 * random_state=epoch) as called at src/fm.py:72-79 and src/mf.py:88-95:
 * legacy MT19937 seeded with the iteration number, Fisher-Yates shuffle of
 * arange(n_rows) with NumPy's masked-rejection interval draw, first B ids.
 * h_out_ids[(e - epoch_begin) * batch_size + t], bit-identical to NumPy.
 * RFM_ERR_BAD_ARG when batch_size > n_rows (the reference raises ValueError). */
int32_t rfm_sample_batches(int64_t n_rows, int64_t batch_size, int64_t epoch_begin,
                           int64_t n_epochs, int32_t* h_out_ids, int32_t n_threads);

/* ---- content fingerprint of a host buffer (host) -------------------------
 * Not on the reference's path: the reference reads its inputs on every fit()
 * (src/fm.py:72-79 gathers from train["features"] each iteration); this library keeps device
 * copies of a split between fits and re-uploads when this 64-bit hash of ALL bytes of the
 * host array differs.  Independent of n_threads. */
int32_t rfm_hash_bytes(const void* h_data, int64_t n_bytes, int32_t n_threads, uint64_t* h_out);

/* ---- FM: forward / scores ----------------------------------------------
 * Replaces FactorizationMachines.predict (src/fm.py:114-133) with _sigmoid
 * (src/base.py:63-66): out[t] = sigmoid(clip(w0 + sum_i w_i x_ti
 *   + 0.5 * sum_f[(sum_i v_if x_ti)^2 - sum_i v_if^2 x_ti^2], +-700))
 * for row r = d_row_ids ? d_row_ids[t] : t of the CSR matrix.  d_indices and
 * d_values must be readable for at least one element even if the matrix has
 * no entries (rows without entries score sigmoid(w0)). */
int32_t rfm_fm_forward(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                       const double* d_values, const int32_t* d_row_ids, int64_t n_rows,
                       const double* d_w0, const double* d_w, const double* d_V,
                       int64_t n_features, int32_t n_factors, double* d_out_pred);

/* Replaces _cross_entropy_loss (src/base.py:37-61):
 * *d_out = -(1/n) * sum_t[(y/p) log(pred_t + eps) + (1 - y/p) log(1 - pred_t + eps)]
 * with y, p taken at row d_row_ids[t] (or t) and pred at t.  Deterministic
 * two-pass reduction. */
int32_t rfm_ips_logloss(rfm_ctx* ctx, const double* d_y, const double* d_pred,
                        const double* d_pscore, const int32_t* d_row_ids, int64_t n_rows,
                        double eps, double* d_out_loss);

/* Fused forward + loss of the rows (src/fm.py:90-102): scores never leave the
 * chip.  d_out_pred may be NULL. */
int32_t rfm_fm_forward_loss(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                            const double* d_values, const double* d_y, const double* d_pscore,
                            const int32_t* d_row_ids, int64_t n_rows, const double* d_w0,
                            const double* d_w, const double* d_V, int64_t n_features,
                            int32_t n_factors, double eps, double* d_out_pred,
                            double* d_out_loss);

/* ---- FM: training plan --------------------------------------------------
 * One-time (per fit) device layout of the training log (train["features"],
 * ["labels"], ["pscores"] of src/fm.py:55-79) for the gradient without global
 * atomics: row records, entry records and a column-major view (entries sorted
 * by column, row order kept inside a column), built ON THE DEVICE and stored in
 * plan-owned device memory.  rfm_fm_plan_create takes the caller's HOST arrays
 * (and uploads a transient copy); rfm_fm_plan_create_device takes the DEVICE copy
 * of the log the caller already holds (same dtypes as everywhere: indptr int64,
 * indices int32, values / labels / propensities float64).  Both reject an indptr
 * that is not monotone, a column index outside 0..n_features-1 and a row that
 * names a column twice (RFM_ERR_BAD_ARG).
 * max_batch bounds the batch size of later steps (max_batch * n_factors < 2^31).
 * hot_min_count: a column
 * whose expected number of entries per batch (its training frequency *
 * max_batch / n_rows) reaches this value is accumulated on chip by the forward
 * workgroups instead of through its column list (0 = library default).  The on-chip
 * sums of the default form are LDS float atomics: their last bits depend on arrival
 * order.  -1 = no such columns: every sum of a step has a fixed order (bitwise
 * reproducible results).  -2 = the default columns, summed in a fixed order as well
 * (each column's rows of a forward workgroup in row order, the workgroups in block
 * order): bitwise reproducible at a fraction of -1's cost; where the factor count has
 * no such kernel (more than 128 factors, or fewer than 17 at batches of the many-rows
 * shape) it means -1. */
int32_t rfm_fm_plan_create(rfm_ctx* ctx, const int64_t* h_indptr, const int32_t* h_indices,
                           const double* h_values, const double* h_y, const double* h_pscore,
                           int64_t n_rows, int64_t n_features, int32_t n_factors,
                           int64_t max_batch, int32_t hot_min_count, rfm_fm_plan** out);
int32_t rfm_fm_plan_create_device(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                                  const double* d_values, const double* d_y,
                                  const double* d_pscore, int64_t n_rows, int64_t n_features,
                                  int32_t n_factors, int64_t max_batch, int32_t hot_min_count,
                                  rfm_fm_plan** out);
int32_t rfm_fm_plan_destroy(rfm_fm_plan* plan);
/* h_out8[0]=tasks of the gradient launch, [1]=columns longer than a workgroup's
 * tasks (split), [2]=hot columns, [3]=nnz, [4]=device bytes owned by the plan,
 * [5]=forward workgroups of a max_batch step (= hot-sum slabs per step), [6]=slots
 * of the sparse class (padded to whole tasks), [7]=64-slot bitmap words per task */
int32_t rfm_fm_plan_info(const rfm_fm_plan* plan, int64_t* h_out8);
/* how the plan keeps the log for the forward: h_out4[0]=1 if as padded row blocks (every
 * row fits one round of a lane group and max_batch asks for the many-rows shape), else 0 =
 * row + entry records; [1]=bytes of a row block (0 if none); [2]=lanes per row of this
 * factor count; [3]=entries of the longest row */
int32_t rfm_fm_plan_layout(const rfm_fm_plan* plan, int32_t* h_out4);
/* the sliced loss forwards of rfm_fm_train (even factor counts above 128): h_out4[0]=slices of
 * the factors (0: this plan has none), [1]=factors per slice, [2]=columns whose slice a workgroup
 * keeps in LDS, [3]=records per row of the translated logs */
int32_t rfm_fm_plan_sliced(const rfm_fm_plan* plan, int32_t* h_out4);
/* Optional: a log that coming calls on this plan will name (device CSR arrays): slot 0 = the
 * validation log of rfm_fm_train, slot 1 = the log rfm_fm_plan_forward scores.  A plan with sliced
 * loss forwards keeps the log's translated form, and calls that pass the SAME three pointers and row
 * count skip the per-call translation (a fit() that trains one iteration per call and scores an
 * evaluation log after each).  The caller promises that the arrays do not change until it registers
 * the slot again (n_rows = 0: nothing); any other arrays are translated per call. */
int32_t rfm_fm_plan_register_log(rfm_ctx* ctx, rfm_fm_plan* plan, int32_t slot, const int64_t* d_indptr,
                                 const int32_t* d_indices, const double* d_values, int64_t n_rows);
/* rfm_fm_forward through the plan: scores of the rows of a CSR with the plan's column count, by the
 * sliced forward (rfm_fm_train's loss forward, even factor counts above 128) when the plan has one
 * and the rows are enough for it, else by the plain forward -- the same scores up to the order of
 * the sums.  Replaces src/fm.py:114-133 inside a fit() that scores an evaluation log every
 * iteration (utils/search_params.py:96-111). */
int32_t rfm_fm_plan_forward(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                            const int32_t* d_indices, const double* d_values, int64_t n_rows,
                            const double* d_w0, const double* d_w, const double* d_V,
                            double* d_out_pred);
/* the hot columns (ascending), h_out[0 .. info[2]); capacity = room in h_out */
int32_t rfm_fm_plan_hot_columns(const rfm_fm_plan* plan, int32_t* h_out, int32_t capacity);

/* ---- FM: one training step ----------------------------------------------
 * Replaces lines src/fm.py:80-88 with _update_w0/_update_w/_update_V
 * (src/fm.py:135-187) and SGD.update (utils/optimizer.py:56-64) for the batch
 * rows d_row_ids[0..batch): residual e = y/p - predict(old params); batch-SUM
 * gradients (no 1/|B|); w0, w, V updated in place with lr.  The CSR / label /
 * propensity arrays are the device copy of the log the plan was built from
 * (the step reads the plan's own records of it).
 * PRECONDITION of every call that takes the row ids of a step (rfm_fm_step,
 * rfm_fm_grad, rfm_fm_grad_rows, rfm_fm_train, rfm_fm_train_dp, rfm_fm_fit_dp): the ids of one
 * step lie in 0 .. n_rows-1 of the plan's log and are DISTINCT -- what
 * resample(replace=False) yields (src/fm.py:72-79).  A row's batch position is
 * recorded with a plain store, so a repeated id would silently lose one of its
 * contributions, and an id outside the log would index the records out of
 * bounds.  The calls do not check this by default; with the environment
 * variable RFM_CHECK_IDS=1 they validate the ids on the device first (this
 * synchronises the stream) and return RFM_ERR_BAD_ARG.
 * d_w0 / d_w / d_V: ordinary device memory (hipMalloc -- what a torch device tensor is): every
 * element has ONE writer per step, which adds to w0 / w[col] with a hardware f64 atomic add that
 * returns nothing (no wait for the old value); fine-grained / host-mapped memory is not supported. */
int32_t rfm_fm_step(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                    const int32_t* d_indices, const double* d_values, const double* d_y,
                    const double* d_pscore, const int32_t* d_row_ids, int64_t batch,
                    double* d_w0, double* d_w, double* d_V, double lr);

/* Same gradients, not applied: d_grad = [G_V (n*k) | g_w (n) | g_w0 (1)] for
 * the given rows (a rank's shard of the batch).  Every element of d_grad is
 * written.  For the data-parallel path: all-reduce d_grad, then rfm_fm_apply. */
int32_t rfm_fm_grad(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                    const int32_t* d_indices, const double* d_values, const double* d_y,
                    const double* d_pscore, const int32_t* d_row_ids, int64_t batch,
                    const double* d_w0, const double* d_w, const double* d_V, double* d_grad);
/* theta -= lr * grad over the same [V | w | w0] layout (utils/optimizer.py:56-64). */
int32_t rfm_fm_apply(rfm_ctx* ctx, double* d_w0, double* d_w, double* d_V,
                     const double* d_grad, int64_t n_features, int32_t n_factors, double lr);

/* ---- FM: touched-row gradients -----------------------------
 * The reference writes only the rows of V whose gradient is non-zero
 * (src/fm.py:183-187): the columns the batch touches.  rfm_fm_grad_rows computes
 * the same batch-SUM gradients as rfm_fm_grad for rows d_row_ids[0..batch) (a
 * rank's shard; batch may be 0) but returns them as a list of RECORDS, ascending
 * by column, of n_factors + 2 doubles each:
 *     [column (exact in a double), G_V[column, 0..k), g_w[column]]
 * d_rows has room for cap_rows records; *d_n_rows receives the true number (a
 * caller that sees *d_n_rows > cap_rows got only the first cap_rows), *d_gw0
 * the shard's g_w0.  Optional owner ranges for the exchange: d_range_lo[n_ranges]
 * (ascending first columns, n_ranges <= 64) -> d_range_bounds[n_ranges + 1]:
 * position in the list of the first record with column >= range_lo[i], and the
 * total.  Nothing of size n_features is cleared or moved per call: the kernels
 * stamp the columns they write, and a count / scan / gather over the stamps
 * builds the list. */
int32_t rfm_fm_grad_rows(rfm_ctx* ctx, rfm_fm_plan* plan, const int32_t* d_row_ids, int64_t batch,
                         const double* d_w0, const double* d_w, const double* d_V, double* d_rows,
                         int64_t cap_rows, int32_t* d_n_rows, double* d_gw0,
                         const int32_t* d_range_lo, int32_t n_ranges, int32_t* d_range_bounds);
/* theta -= lr * g for the records of such a list (utils/optimizer.py:56-64 with a
 * row index): V[column,:] -= lr * G_V row, w[column] -= lr * g_w; w0 -= lr * *d_gw0
 * when d_gw0 is given.  The count is read on the device (min(*d_n_rows, cap_rows)). */
int32_t rfm_fm_apply_rows(rfm_ctx* ctx, const double* d_rows, const int32_t* d_n_rows,
                          int64_t cap_rows, const double* d_gw0, double* d_w0, double* d_w,
                          double* d_V, int64_t n_features, int32_t n_factors, double lr);
/* Owner side of the exchange.  d_rows holds n_segments lists back to back, segment
 * s = records d_seg_ptr[s] .. d_seg_ptr[s+1] (what rank s sent; ascending by column;
 * total_rows = d_seg_ptr[n_segments], passed on the host too to size the launch).
 * Records of one column are added in segment (= rank) order, the row is updated from
 * the owner's d_V / d_w (read only), and d_out_rows -- same number of records -- holds
 * [column, V_new[column,:], w_new[column]] at the position of the column's first
 * record and column = -1 at the positions of its other records. */
int32_t rfm_fm_reduce_rows(rfm_ctx* ctx, const double* d_rows, const int32_t* d_seg_ptr,
                           int32_t n_segments, int64_t total_rows, const double* d_w,
                           const double* d_V, int64_t n_features, int32_t n_factors, double lr,
                           double* d_out_rows);
/* Store updated rows ([column, V_new row, w_new]; column < 0 = skip; columns distinct)
 * into a replica, and w0 -= lr * (d_gw0_parts[0] + d_gw0_parts[part_stride] + ...,
 * n_parts terms in that order) when n_parts > 0. */
int32_t rfm_fm_set_rows(rfm_ctx* ctx, const double* d_rows, int64_t n_rows,
                        const double* d_gw0_parts, int32_t n_parts, int64_t part_stride,
                        double* d_w0, double* d_w, double* d_V, int64_t n_features,
                        int32_t n_factors, double lr);

/* ---- FM: the fit() loop --------------------------------------------------
 * Replaces the body of FactorizationMachines.fit (src/fm.py:71-102) for
 * iterations [0, n_iters): step on batch d_ids[it*batch ...], then the train
 * loss of the SAME batch with the new parameters, then the validation loss.
 * d_out_train_loss / d_out_val_loss receive one value per iteration (either
 * may be NULL to skip that forward).  Everything is enqueued on the ctx
 * stream; nothing synchronises.  For even factor counts above 128 (every published run of the
 * reference) the two loss forwards are ONE launch sliced by factors that keeps the log's frequent
 * columns in LDS, once the batch and the validation log together have RFM_SLICED_MIN_ROWS rows
 * (default 4 096; RFM_SLICED_LOSS=0: never) -- same losses up to the order of the sums.  The
 * other loss forwards leave their rows' scores and one launch per run of iterations takes the
 * logarithms (RFM_DEFER_LOSS=0: inside the forward); at small batches the train-loss rows of an
 * iteration are scored inside the NEXT iteration's forward launch (RFM_RIDE_LOSS=0: apart), and so
 * are the rows of a small validation log registered with rfm_fm_plan_register_log (RFM_RIDE_VAL=0).
 * (Environment, experiments only: RFM_PREP=1 at plan creation
 * makes calls of 8 or more iterations lay their batches out ahead of the loop -- same results bit
 * for bit, one wait on an event per chunk of iterations; RFM_TRAIN_GRAPH=1 replays the call's
 * launches as one hipGraph.) */
int32_t rfm_fm_train(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                     const int32_t* d_indices, const double* d_values, const double* d_y,
                     const double* d_pscore, const int32_t* d_ids, int64_t batch,
                     int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                     const int64_t* d_val_indptr, const int32_t* d_val_indices,
                     const double* d_val_values, const double* d_val_y,
                     const double* d_val_pscore, int64_t n_val, double eps,
                     double* d_out_train_loss, double* d_out_val_loss);
/* rfm_fm_train with the evaluator hook of the reference's search loop inside
 * (utils/search_params.py:96-111: fit(..., evaluator=ValEvaluator) -- after every iteration the
 * scores of the evaluation log, src/fm.py:104-110, and their IPS-DCG@k, utils/evaluate.py:160-207):
 * iteration i of the call leaves the scores of the n_ev rows of the evaluation CSR in
 * d_scores + (slot_first + i) * scores_stride (through rfm_fm_plan_forward), the per-group values
 * of rfm_val_dcg in d_user_scratch + (slot_first + i) * user_stride (>= 3 * n_segments doubles) and
 * its two results in d_dcg_out[2 i], [2 i + 1].  d_seg_ptr / d_rows / d_labels / d_ev_pscores /
 * n_segments / k: as rfm_val_dcg.  One enqueue for the whole run of iterations; nothing synchronises. */
int32_t rfm_fm_train_eval(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                          const int32_t* d_indices, const double* d_values, const double* d_y,
                          const double* d_pscore, const int32_t* d_ids, int64_t batch,
                          int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                          const int64_t* d_val_indptr, const int32_t* d_val_indices,
                          const double* d_val_values, const double* d_val_y,
                          const double* d_val_pscore, int64_t n_val, double eps,
                          double* d_out_train_loss, double* d_out_val_loss,
                          const int64_t* d_ev_indptr, const int32_t* d_ev_indices,
                          const double* d_ev_values, int64_t n_ev, const int32_t* d_seg_ptr,
                          const int32_t* d_rows, const double* d_labels, const double* d_ev_pscores,
                          int32_t n_segments, int32_t k, double* d_scores, int64_t scores_stride,
                          double* d_user_scratch, int64_t user_stride, int64_t slot_first,
                          double* d_dcg_out);

/* ---- FM: the data-parallel fit() loop ------------------------
 * The reference has no multi-process mode; this is the body of FactorizationMachines.fit
 * (src/fm.py:71-102) for one rank of n_ranks replicas: per iteration the rank computes the
 * batch-SUM gradient of ITS contiguous shard of the global batch d_ids[it*global_batch ..)
 * (rank r takes rows [r*q + min(r, m), ...) with q, m = divmod(global_batch, n_ranks): the
 * first m ranks one row more), the shards' gradients are combined, every replica applies
 * the same update, then the train loss of the same global batch with the new parameters
 * (src/fm.py:90-96; every rank its shard) and the validation loss (src/fm.py:98-102; the
 * validation rows cut the same way) are formed as sums, combined over the ranks and
 * written -- the same value on every rank -- to d_out_train_loss / d_out_val_loss
 * (n_iters doubles each, either may be NULL).
 *   exchange 0 (dense, what the north star names): all-reduce(sum) of [G_V | g_w | g_w0],
 *     n_features*(n_factors+1)+1 doubles per iteration, then rfm_fm_apply.
 *   exchange 1 (touched rows, feature-range ownership): rank r owns the columns
 *     [n_features*r/n_ranks, n_features*(r+1)/n_ranks); per iteration the gradient records
 *     of rfm_fm_grad_rows (+ one record for g_w0, owned by the last rank) go to the owners,
 *     an owner adds the records of a column in RANK ORDER, updates its row and sends the
 *     updated rows to everybody (rfm_fm_reduce_rows / rfm_fm_set_rows): each row is
 *     computed once and copied, the replicas stay bit-identical.  Which columns a shard
 *     touches depends on the row ids only, so the sizes of every transfer of the whole
 *     call are derived BEFORE the loop (one count pass per iteration, one all-gather of the
 *     counts, one host synchronisation per call); inside the loop nothing synchronises and
 *     nothing is allocated.
 * Transport: `transport` == NULL -> RCCL on the communicator of rfm_comm_init (n_ranks and the
 * rank are its; no communicator = one rank, nothing is exchanged), everything on the ctx
 * stream.  Otherwise the caller's functions move the (device) buffers -- e.g. staged through
 * the host over another fabric, or ranks that share one GPU in a test; each must be ordered
 * after the work already enqueued on the ctx stream and complete before it returns (or
 * be enqueued on that stream).  Offsets and sizes are bytes; entry p of an array belongs to
 * peer p (the rank's own entry included: a local copy).  Return 0 for success.
 * The call ends with one synchronisation (it reads back an error flag: a transfer plan that
 * does not match what the gradient kernels produced is RFM_ERR_INTERNAL, never a silent
 * truncation). */
typedef struct rfm_transport {
  void* user;
  int32_t n_ranks, rank;
  int32_t (*all_gather)(void* user, const void* d_send, void* d_recv, int64_t bytes_per_rank);
  int32_t (*all_reduce_sum)(void* user, double* d_buf, int64_t count);
  int32_t (*all_to_all)(void* user, const void* d_send, const int64_t* h_send_off,
                        const int64_t* h_send_bytes, void* d_recv, const int64_t* h_recv_off,
                        const int64_t* h_recv_bytes);
} rfm_transport;
int32_t rfm_fm_fit_dp(rfm_ctx* ctx, rfm_fm_plan* plan, const rfm_transport* transport,
                      int32_t exchange, const int32_t* d_ids, int64_t global_batch,
                      int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                      const int64_t* d_val_indptr, const int32_t* d_val_indices,
                      const double* d_val_values, const double* d_val_y,
                      const double* d_val_pscore, int64_t n_val, double eps,
                      double* d_out_train_loss, double* d_out_val_loss);

/* ---- MF ------------------------------------------------------------------
 * Replaces LogisticMatrixFactorization.predict/_predict_pair
 * (src/mf.py:136-170): out[t] = sigmoid(clip(P_u . Q_i + b_u + b_i + b)). */
int32_t rfm_mf_predict(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                       const int32_t* d_row_ids, int64_t n_rows, const double* d_P,
                       const double* d_Q, const double* d_bu, const double* d_bi, double b,
                       int32_t n_factors, double* d_out_pred);
int32_t rfm_mf_predict_loss(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                            const double* d_y, const double* d_pscore,
                            const int32_t* d_row_ids, int64_t n_rows, const double* d_P,
                            const double* d_Q, const double* d_bu, const double* d_bi,
                            double b, int32_t n_factors, double eps, double* d_out_pred,
                            double* d_out_loss);

/* Order-preserving schedule of one batch (host): the reference updates the
 * batch strictly sequentially (src/mf.py:97-108).  Example s must run after
 * the latest earlier example sharing its user or its item; level(s) = 1 +
 * max(level of those).  Examples of one level touch disjoint rows, so running
 * the levels in order reproduces the sequential result exactly.
 * h_users/h_items: ids of the batch in batch order.  h_order[batch] receives
 * the batch positions grouped by level (ascending position inside a level),
 * h_level_ptr[n_levels+1] (capacity batch+1) the group boundaries;
 * *h_n_levels the number of levels. */
int32_t rfm_mf_schedule(const int32_t* h_users, const int32_t* h_items, int64_t batch,
                        int32_t n_users, int32_t n_items, int32_t* h_order,
                        int32_t* h_level_ptr, int32_t* h_n_levels);

/* Replaces the inner loop src/mf.py:97-108 with _update_P/_Q/_b_u/_b_i
 * (src/mf.py:172-216): for each level in order, for each example of the level
 * (independent): err = y/p - predict; P_u -= lr(-err Q_i + reg P_u); Q_i -=
 * lr(-err P_u(new) + reg Q_i); b_u, b_i likewise.  d_order/h_level_ptr from
 * rfm_mf_schedule (level_ptr both on the host, to size the launches, and on
 * the device); d_pos_rows[s] is the training row of batch position s. */
int32_t rfm_mf_sgd_levels(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                          const double* d_y, const double* d_pscore,
                          const int32_t* d_pos_rows, const int32_t* d_order,
                          const int32_t* h_level_ptr, const int32_t* d_level_ptr,
                          int32_t n_levels, double* d_P,
                          double* d_Q, double* d_bu, double* d_bi, double b,
                          int32_t n_factors, double lr, double reg);

/* The same schedule in the form the fast kernels read: the batch's examples as
 * 24-byte records {int32 user, int32 item, int32 cache_slot, int32 gap, double
 * label/propensity} grouped by level (ascending batch position inside a level).
 * h_users/h_items/h_y/h_pscore are the batch in batch order.  Up to cache_cap items
 * that occur more than once in the batch (most frequent first) are listed in
 * h_cache_items: the sequential kernel keeps their rows in LDS (cache_slot >= 0;
 * -1 = the item occurs once, -2 = repeated without a slot); gap = levels between the
 * example and the previous writer of its user row (RFM_MF_NO_WRITER if none), i.e.
 * how far ahead of its level the row is final and may be read.  The cached rows must
 * fit the kernel's LDS: cache_cap <= rfm_mf_cache_capacity(n_factors) (32 KiB of rows, at most 1024).  Capacities: h_ex batch records, h_level_ptr batch+1,
 * h_cache_items cache_cap. */
/* Item rows (with their bias) the sequential kernel may keep in LDS for one batch: the largest
 * cache_cap rfm_mf_schedule_ex / rfm_mf_sgd_levels_ex accept for this factor count. */
int32_t rfm_mf_cache_capacity(int32_t n_factors, int32_t* h_out);
int32_t rfm_mf_schedule_ex(const int32_t* h_users, const int32_t* h_items, const double* h_y,
                           const double* h_pscore, int64_t batch, int32_t n_users,
                           int32_t n_items, int32_t cache_cap, void* h_ex,
                           int32_t* h_level_ptr, int32_t* h_n_levels, int32_t* h_cache_items,
                           int32_t* h_n_cached);
/* rfm_mf_sgd_levels on those records (d_ex = device copy of h_ex, d_cache_items of
 * h_cache_items; n_cached rows of n_factors+2 doubles must fit 32 KiB of LDS). */
int32_t rfm_mf_sgd_levels_ex(rfm_ctx* ctx, const void* d_ex, const int32_t* h_level_ptr,
                             const int32_t* d_level_ptr, int32_t n_levels,
                             const int32_t* d_cache_items, int32_t n_cached, double* d_P,
                             double* d_Q, double* d_bu, double* d_bi, double b,
                             int32_t n_factors, double lr, double reg);

/* ---- MF across GPUs: user-range partition ------------------
 * NOT the reference's semantics (its batch is strictly sequential, src/mf.py:97-108;
 * exact mode = one GPU or independent replicas).  Throughput mode: rank r runs the exact
 * sequential SGD (rfm_mf_sgd_levels_ex) over the batch's examples whose USER lies in its
 * range, in batch order, on its own replica of Q / b_i; P / b_u rows are only ever
 * written by their owner.  After the batch each rank forms what its examples changed,
 * rfm_mf_delta: d_out = d_cur - d_sync, the deltas are all-reduced (sum), and
 * rfm_mf_merge: d_sync += d_total, d_cur = d_sync brings every replica to the same
 * values.  With one rank nothing is exchanged and the result is the reference's. */
int32_t rfm_mf_delta(rfm_ctx* ctx, const double* d_cur, const double* d_sync, double* d_out,
                     int64_t count);
int32_t rfm_mf_merge(rfm_ctx* ctx, double* d_cur, double* d_sync, const double* d_total,
                     int64_t count);

/* ---- multi-GPU exchange (RCCL over xGMI) ----------------------------------
 * The data-parallel FM step all-reduces the dense gradient buffer of rfm_fm_grad
 * (SURVEY.md 8e).  The Python mirror does that through torch.distributed (backend
 * "nccl" is RCCL); these entry points are the same collective for callers
 * without torch: rank 0 draws an id (128 bytes) and hands it to the other ranks
 * by any means, every rank calls rfm_comm_init on its own ctx (one process per
 * GPU), then rfm_allreduce_sum sums d_buf[0..count) in place over all ranks on
 * the ctx stream.  librccl.so is loaded on first use. */
int32_t rfm_comm_unique_id(uint8_t* h_out128);
int32_t rfm_comm_init(rfm_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* h_id128);
int32_t rfm_allreduce_sum(rfm_ctx* ctx, double* d_buf, int64_t count);
int32_t rfm_comm_destroy(rfm_ctx* ctx);

/* The data-parallel fit() loop of one rank, enqueued on the ctx stream without any
 * host synchronisation: for iteration it in [0, n_iters) the rank computes the
 * gradient of rows d_ids[it*global_batch + shard_lo .. + shard_hi) (its contiguous
 * shard of the global batch; an empty shard contributes zeros), all-reduces d_grad
 * over the ranks of rfm_comm_init (skipped when no communicator or one rank), and
 * applies theta -= lr * grad -- the same update on every rank.  d_grad is scratch
 * of n_features*(n_factors+1)+1 doubles. */
int32_t rfm_fm_train_dp(rfm_ctx* ctx, rfm_fm_plan* plan, const int32_t* d_ids,
                        int64_t global_batch, int64_t shard_lo, int64_t shard_hi,
                        int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                        double* d_grad);

/* HOGWILD-style variant of the same batch (NOT the reference's semantics): all
 * examples of the batch are updated concurrently without ordering, so examples
 * sharing a user or an item race.  Throughput mode for very large batches; its
 * results match the reference only at the loss / ranking level, not 1e-5 on the
 * parameters.  d_pos_rows[s] is the training row of batch position s. */
int32_t rfm_mf_sgd_hogwild(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                           const double* d_y, const double* d_pscore,
                           const int32_t* d_pos_rows, int64_t batch, double* d_P, double* d_Q,
                           double* d_bu, double* d_bi, double b, int32_t n_factors, double lr,
                           double reg);

/* ---- per-iteration validation metric ---------------------
 * IPS-DCG@k of ValEvaluator.evaluate (utils/evaluate.py:183-207 with
 * utils/metrics.py:53-80), the value src/fm.py:104-110 and src/mf.py:126-132
 * append to val_metrics every iteration.  The frame's rows are grouped by user
 * (groups in ascending user order, rows of a group in frame order -- what
 * DataFrame.groupby("user").agg(list) yields): group g holds positions
 * d_seg_ptr[g] .. d_seg_ptr[g+1]; d_labels / d_pscores are indexed by position,
 * the score of position j is d_scores[d_rows[j]] (d_rows == NULL: d_scores[j]).
 * Per group: rows ranked by descending score, groups whose labels sum to zero are
 * left out, value = y0/p0 + sum_{r=1..k-1} y_r / (p_r * log2(r+1)); d_out[0] = mean
 * over the groups that count (nan if none).  Equal scores rank the later position
 * first, i.e. argsort(kind="stable")[::-1]; NumPy's default sort, which the
 * reference calls, is not stable and orders ties differently from CPU to CPU, so
 * d_out[1] = number of counted groups whose value depends on that order (a tie
 * reaching into the first k ranks between rows of different label or propensity,
 * or a NaN score, which is never ranked here).  With d_out[1] == 0 the value is the
 * reference's; otherwise the caller decides (the Python mirror redoes exactly those
 * groups on the host with NumPy's own sort, from the per-group flags below).  d_pscores == NULL means all ones
 * (the Naive estimator's ones_pscore column; also calc_dcg_at_k of
 * utils/metrics.py:83-107).  d_user_scratch: 3*n_segments doubles; after the call
 * [0, n) holds the per-group values, [n, 2n) 1.0 / 0.0 for counted / left out and
 * [2n, 3n) 1.0 for the order-dependent groups. */
int32_t rfm_val_dcg(rfm_ctx* ctx, const double* d_scores, const int32_t* d_seg_ptr,
                    const int32_t* d_rows, const double* d_labels, const double* d_pscores,
                    int32_t n_segments, int32_t k, double* d_user_scratch, double* d_out);

/* ---- test-set metrics -------------------------------------
 * The ranking half of TestEvaluator.evaluate (utils/evaluate.py:80-127): per user
 * group (same grouping arrays as rfm_val_dcg; d_items = item id per position, may
 * be NULL) the positions of the k best-scored rows, d_out_pos[g*k + r] = position
 * of rank r or -1 past the group's rows, same tie rule (later position first).
 * d_out_flags[g]: bit 0 = the group has a positive label (the others are left out
 * of every metric, evaluate.py:99-100); bit 1 = the first k ranks depend on how
 * equal scores are ordered (a tie between rows of different label, propensity or
 * item inside the first k ranks or across rank k, or a NaN score) -- the Python
 * mirror ranks exactly those groups again with NumPy's own sort.  DCG@K, Recall,
 * MAP, mean exposure, CatalogCoverage and Gini (utils/metrics.py:9-166) are sums
 * over these n_segments x k positions. */
int32_t rfm_topk_users(rfm_ctx* ctx, const double* d_scores, const int32_t* d_seg_ptr,
                       const int32_t* d_rows, const double* d_labels, const double* d_pscores,
                       const int32_t* d_items, int32_t n_segments, int32_t k, int32_t* d_out_pos,
                       int32_t* d_out_flags);

/* ---- CSR assembly -------------------------------------------
 * The step before the training path: the FM design matrix of the reference's loaders
 * -- one-hot user (+) one-hot item (+) per-interaction columns (+) the user's feature row
 * (+) the item's feature row, hstack'ed (utils/dataloader/kuairec/_feature.py:54-84,
 * 201-207; coat/_preparer.py:154-168) -- and the row selections made from it
 * (features[indices]: kuairec/_preparer.py:117-136, loader.py:104-115).  Output row r
 * is the concatenation, in segment order, of
 *   kind 0: the single entry (col_offset + d_ids[r], 1.0)          -- a one-hot of an id
 *   kind 1: row d_ids[r] (d_ids == NULL: row r) of a CSR block, its columns shifted by
 *           col_offset                            -- a feature table gathered by id
 * n_block_rows = rows of the block (kind 1) / size of the one-hot (kind 0); an id outside
 * it is RFM_ERR_BAD_ARG.  Blocks use the ABI's dtypes (indptr int64, indices int32,
 * values float64); stored zeros are kept as they are.  Call rfm_csr_assemble_count (fills
 * d_out_indptr[n_rows + 1], returns the number of entries on the host -- it synchronises),
 * allocate indices / values, then rfm_csr_assemble_fill. */
typedef struct rfm_csr_segment {
  int32_t kind;
  const int32_t* d_ids;
  const int64_t* d_indptr;
  const int32_t* d_indices;
  const double* d_values;
  int64_t col_offset;
  int64_t n_block_rows;
} rfm_csr_segment;
int32_t rfm_csr_assemble_count(rfm_ctx* ctx, const rfm_csr_segment* h_segments, int32_t n_segments,
                               int64_t n_rows, int64_t* d_out_indptr, int64_t* h_out_nnz);
int32_t rfm_csr_assemble_fill(rfm_ctx* ctx, const rfm_csr_segment* h_segments, int32_t n_segments,
                              int64_t n_rows, const int64_t* d_indptr, int32_t* d_out_indices,
                              double* d_out_values);

#ifdef __cplusplus
}
#endif
#endif /* RFM_HIP_H */
