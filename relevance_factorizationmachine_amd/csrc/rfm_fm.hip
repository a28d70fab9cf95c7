// FM path of librfm_hip.so: launch side and C-ABI entry points of the forward, the training
// step, its dense and touched-row gradient forms, and the fit() loop.  The kernels are in
// rfm_fm_kernels.hpp / rfm_fm_rows.hpp, the training plan is built by rfm_fm_plan.hip
// (layout: rfm_fm_plan.h, rfm_fm_records.h; DESIGN.md sections 3 and 4).
//
// One training step (rfm_fm_step) is two launches on one stream:
//   1. fm_forward_kernel   rows of the batch in parallel: q_t = V^T x_t, logit, residual;
//                          writes Q, leaves {t, residual} marks + bitmap bits at the slots of
//                          the row's sparse-class entries and adds the hot entries'
//                          err_t x_tj [q_t, 1, x_tj] into the workgroup's LDS sums.
//   2. fm_consume_kernel   one task (a fixed number of 64-slot bitmap words) per lane group:
//                          lists the marked slots, accumulates err_t x_tj [Q[t,:], 1, x_tj]
//                          IN SLOT ORDER, updates the columns inside the task in place and
//                          combines columns that run over several tasks of the workgroup in
//                          LDS; trailing workgroups reduce the hot columns' slabs and w0.
//  (3. fm_finalize_kernel  only for columns longer than a whole workgroup's tasks.)
// No global float atomics.  Sparse-class sums have a fixed order (bitwise reproducible);
// hot-class sums inside one workgroup are LDS atomics, so their last bits may vary from run
// to run (hot_min_count < 0 turns the class off).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <string>
#include <thread>

#include "rfm_common.h"
#include "rfm_fm_kernels.hpp"
#include "rfm_fm_plan.h"
#include "rfm_fm_prep.hpp"
#include "rfm_fm_rows.hpp"
#include "rfm_fm_sliced.hpp"

static_assert(RFM_MAX_FACTORS <= 1024, "fm_finalize_kernel's LDS totals hold 1024+2 values");

namespace rfm {

// forward launch geometry: 1024-thread workgroups whose lane groups keep several
// rows in flight once the batch fills the chip with them (one per CU: as few
// hot-sum slabs as possible), 256-thread / one-row ones otherwise.  RFM_FWD_PER_CU
// overrides the workgroups per CU of the big shape (tuning experiments only).
struct FwdGeom {
  int block, grid;
};
#ifndef RFM_FWD_SMALL_BLOCK
#define RFM_FWD_SMALL_BLOCK 256
#endif
constexpr int kSmallBlock = RFM_FWD_SMALL_BLOCK;  // threads of the one-row-per-group shape
constexpr int kMaxDevices = 64;  // devices of one process whose launch attributes are remembered

inline FwdGeom forward_geom(const rfm_ctx* ctx, int64_t n_rows, const Shape& s, bool records) {
  static const int per_cu = std::max(1, env_int("RFM_FWD_PER_CU", kBigBlock >= 1024 ? 1 : 2));
  static const int force = env_int("RFM_FWD_BLOCK", 0);
  FwdGeom g;
  // (the plain forward -- the caller's CSR arrays: predict, validation loss -- takes the same two
  // shapes with its own number of rows per lane group)
  const int64_t rows_big = int64_t(kBigBlock / s.lpr) * (records ? rows_in_flight(s.nc) : rows_in_flight_plain(s.nc));
  const int64_t blocks_big = (n_rows + rows_big - 1) / rows_big;
  // (measured on config 3: the many-rows shape wins from about a third of a chip of such
  // workgroups: 16 384 rows 26 vs 30 us, 8 192 rows 21 vs 20 us)
  if (force != 256 && (force == 512 || blocks_big * 2 >= int64_t(ctx->n_cu) * per_cu)) {
    g.block = kBigBlock;
    g.grid = int(std::max<int64_t>(1, std::min<int64_t>(blocks_big, int64_t(ctx->n_cu) * per_cu)));
  } else {
    g.block = kSmallBlock;
    const int gpb = kSmallBlock / s.lpr;
    const int64_t want = (n_rows + gpb - 1) / gpb;
    g.grid = int(std::max<int64_t>(1, std::min<int64_t>(want, int64_t(ctx->n_cu) * 8)));
  }
  g.grid = std::min(g.grid, 2048);
  return g;
}

int forward_grid(const rfm_ctx* ctx, int64_t rows, int n_factors) {
  return forward_geom(ctx, rows, shape_for(n_factors), true).grid;
}

bool forward_many_rows(const rfm_ctx* ctx, int64_t rows, int n_factors) {
  return forward_geom(ctx, rows, shape_for(n_factors), true).block == kBigBlock;
}

// one instantiation: raises its dynamic-LDS limit when a launch needs more than the default
template <int L, int Vv, int N, int BLOCK, int R, bool REC, bool ELL, bool DET, bool SEG = false, bool XTRA = false>
void launch_forward_as(rfm_ctx* ctx, const FwdArgs& a, const FwdGeom& geom, size_t lds) {
  const auto kern = &fm_forward_kernel<L, Vv, N, BLOCK, R, REC, ELL, DET, SEG, XTRA>;
  // (the attribute belongs to the function ON a device: kept per device; atomics because
  // contexts of different host threads share the instantiation)
  static std::atomic<size_t> lds_allowed[kMaxDevices];
  const int dev = ctx->device >= 0 && ctx->device < kMaxDevices ? ctx->device : -1;
  if (lds > (64u << 10) && (dev < 0 || lds > lds_allowed[dev].load(std::memory_order_relaxed))) {
    RFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    if (dev >= 0) lds_allowed[dev].store(lds, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(geom.grid), dim3(BLOCK), lds, ctx->stream, a);
}

// the instantiation a launch takes: shape of the workgroups (geom), source of the records,
// and -- for a plan that asks for them -- fixed-order hot sums
template <int L, int Vv, int N>
void launch_forward_shape(rfm_ctx* ctx, const FwdArgs& a, const FwdGeom& geom, size_t lds,
                          bool recs, bool fixed) {
  constexpr int R = rows_in_flight(N);
  if (geom.block == kBigBlock) {
    if (!recs) return launch_forward_as<L, Vv, N, kBigBlock, rows_in_flight_plain(N), false, false, false>(ctx, a, geom, lds);
    if constexpr (hot_fixed_order(L, N, kBigBlock, R)) {
      if (fixed) {
        if (a.ell) return launch_forward_as<L, Vv, N, kBigBlock, R, true, true, true>(ctx, a, geom, lds);
        return launch_forward_as<L, Vv, N, kBigBlock, R, true, false, true>(ctx, a, geom, lds);
      }
    }
    RFM_REQUIRE(!fixed, "fixed-order hot sums are not built for this factor count at this batch");
    if (a.ell) return launch_forward_as<L, Vv, N, kBigBlock, R, true, true, false>(ctx, a, geom, lds);
    RFM_REQUIRE(a.ent && a.rows, "the plan holds no row records");
    return launch_forward_as<L, Vv, N, kBigBlock, R, true, false, false>(ctx, a, geom, lds);
  }
  if (!recs) return launch_forward_as<L, Vv, N, kSmallBlock, 1, false, false, false>(ctx, a, geom, lds);
  // (a plan made for many-rows batches keeps only the padded row blocks: a step of fewer rows
  // on it reads those too)
  if constexpr (hot_fixed_order(L, N, kSmallBlock, 1)) {
    if (fixed) {
      if (a.ell) return launch_forward_as<L, Vv, N, kSmallBlock, 1, true, true, true>(ctx, a, geom, lds);
      return launch_forward_as<L, Vv, N, kSmallBlock, 1, true, false, true>(ctx, a, geom, lds);
    }
  }
  RFM_REQUIRE(!fixed, "fixed-order hot sums are not built for this factor count");
  if (a.n_rows_x > 0) {  // (XTRA form: the last workgroups only score another batch's rows)
    if (a.ell) return launch_forward_as<L, Vv, N, kSmallBlock, 1, true, true, false, false, true>(ctx, a, geom, lds);
    RFM_REQUIRE(a.ent && a.rows, "the plan holds no row records");
    return launch_forward_as<L, Vv, N, kSmallBlock, 1, true, false, false, false, true>(ctx, a, geom, lds);
  }
  if (a.ell) return launch_forward_as<L, Vv, N, kSmallBlock, 1, true, true, false>(ctx, a, geom, lds);
  RFM_REQUIRE(a.ent && a.rows, "the plan holds no row records");
  return launch_forward_as<L, Vv, N, kSmallBlock, 1, true, false, false>(ctx, a, geom, lds);
}

bool forward_fixed_order_ok(const rfm_ctx* ctx, int64_t max_batch, int n_factors) {
  const Shape s = shape_for(n_factors);
  if (!hot_fixed_order(s.lpr, s.nc, kSmallBlock, 1)) return false;  // (steps of fewer rows)
  return !forward_many_rows(ctx, max_batch, n_factors) ||
         hot_fixed_order(s.lpr, s.nc, kBigBlock, rows_in_flight(s.nc));
}

void launch_forward(rfm_ctx* ctx, FwdArgs a, FwdGeom geom) {
  if (a.n_rows <= 0) return;
  const Shape s = shape_for(a.k);
  const bool recs = a.ent != nullptr || a.ell != nullptr;  // the plan's records
  const bool fixed = recs && a.hot_fixed && a.n_hot > 0;
  const size_t lds =
      forward_lds_bytes(geom.block, s.lpr, s.vec, s.nc,
                        geom.block == kBigBlock ? (recs ? rows_in_flight(s.nc) : rows_in_flight_plain(s.nc)) : 1,
                        a.n_hot, a.k, fixed);
  RFM_REQUIRE(lds <= (160u << 10), "forward kernel: %zu bytes of LDS", lds);
  RFM_REQUIRE(!fixed || a.hot_rounds >= 1, "hot_rounds unset");
#define RFM_CALL_FWD(L, Vv, N) launch_forward_shape<L, Vv, N>(ctx, a, geom, lds, recs, fixed)
  RFM_FOR_SHAPE(s, RFM_CALL_FWD);
#undef RFM_CALL_FWD
  RFM_HIP_CHECK(hipGetLastError());
}

void launch_forward(rfm_ctx* ctx, FwdArgs a) {
  if (a.n_rows <= 0) return;
  launch_forward(ctx, a, forward_geom(ctx, a.n_rows, shape_for(a.k), a.ent != nullptr || a.ell != nullptr));
}

// forward with loss: partials in ctx scratch, finished into d_out_loss
void forward_loss(rfm_ctx* ctx, FwdArgs a, double* d_out_loss) {
  RFM_REQUIRE(a.n_rows > 0, "loss of zero rows");
  const FwdGeom geom = forward_geom(ctx, a.n_rows, shape_for(a.k), a.ent != nullptr || a.ell != nullptr);
  ctx->loss_partials.ensure(size_t(std::max(kMaxFwdGrid, ctx->n_cu * 8)) * sizeof(double));
  a.loss_partial = ctx->loss_partials.as<double>();
  launch_forward(ctx, a, geom);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, ctx->stream,
                     ctx->loss_partials.as<double>(), geom.grid, a.n_rows, d_out_loss);
  RFM_HIP_CHECK(hipGetLastError());
}

// The two loss forwards of a fit() iteration in ONE launch (fm_forward_kernel's SEG form): rows
// a.row_ids[0 .. n_rows_a) of the training log (loss partials -> row_a) and every row of the
// validation log (-> row_b).  Returns the number of partials per row, or -1 when the rows
// together do not take the many-rows shape (the caller then launches the two separately).
int forward_loss_pair_deferred(rfm_ctx* ctx, FwdArgs a, double* row_a, double* row_b) {
  const Shape s = shape_for(a.k);
  const FwdGeom geom = forward_geom(ctx, a.n_rows, s, false);
  if (geom.block != kBigBlock || geom.grid > kMaxFwdGrid) return -1;
  a.loss_partial = row_a;
  a.loss_partial2 = row_b;
  const size_t lds = forward_lds_bytes(geom.block, s.lpr, s.vec, s.nc, rows_in_flight_plain(s.nc), 0, a.k, false);
#define RFM_CALL_SEG(L, Vv, N) \
  launch_forward_as<L, Vv, N, kBigBlock, rows_in_flight_plain(N), false, false, false, true>(ctx, a, geom, lds)
  RFM_FOR_SHAPE(s, RFM_CALL_SEG);
#undef RFM_CALL_SEG
  RFM_HIP_CHECK(hipGetLastError());
  return geom.grid;
}

// forward with loss whose partials go to `partial_row` (kMaxFwdGrid doubles) and are
// finished later, many launches at once; returns the number of partials written
int forward_loss_deferred(rfm_ctx* ctx, FwdArgs a, double* partial_row) {
  RFM_REQUIRE(a.n_rows > 0, "loss of zero rows");
  const FwdGeom geom = forward_geom(ctx, a.n_rows, shape_for(a.k), a.ent != nullptr || a.ell != nullptr);
  RFM_REQUIRE(geom.grid <= kMaxFwdGrid, "forward grid %d exceeds %d", geom.grid, kMaxFwdGrid);
  a.loss_partial = partial_row;
  launch_forward(ctx, a, geom);
  return geom.grid;
}

// Sliced loss forward (rfm_fm_sliced.hpp) over `rows` rows: workgroups per slice (one
// workgroup per CU, a multiple of the XCDs a slice is dealt to), rows per workgroup and per
// staged chunk, LDS.  ok = false: the plan has no slices, or the rows are too few to pay for
// every workgroup's copy of the cached columns (RFM_SLICED_MIN_ROWS, default 4 096;
// RFM_SLICED_LOSS=0: never).
struct SlicedGeom {
  bool ok = false;
  int grid = 0, rows_per_wg = 0;
  size_t lds = 0;
};
SlicedGeom sliced_geom(const rfm_ctx* ctx, const rfm_fm_plan* plan, int64_t rows) {
  SlicedGeom g;
  if (plan->sl_ns <= 0 || env_int("RFM_SLICED_LOSS", 1) == 0) return g;
  if (rows < std::max(1, env_int("RFM_SLICED_MIN_ROWS", 4096))) return g;
  if (rows >= (int64_t(1) << 31) - 1) return g;  // (the kernel numbers a log's rows in 31 bits)
  const int xs = 8 / plan->sl_ns;
  int wps = std::max(xs, ctx->n_cu / plan->sl_ns / xs * xs);
  // (few rows: fewer, fuller workgroups -- each fills its own copy of the cached columns)
  const int64_t min_rows = kSlWaves * 4;
  while (wps > xs && rows / wps < min_rows) wps -= xs;
  g.rows_per_wg = int(std::min<int64_t>((rows + wps - 1) / wps, INT32_MAX));
  g.lds = sliced_lds_bytes(plan->sl_n_cached, plan->sl_sw);
  if (g.lds > size_t(kSlicedLds)) return g;
  g.grid = 8 * (wps / xs);
  g.ok = true;
  return g;
}

void launch_sliced(rfm_ctx* ctx, const rfm_fm_plan* plan, const SlicedGeom& g, SlicedArgs a) {
  a.k = plan->k;
  a.ns = plan->sl_ns;
  a.sw = plan->sl_sw;
  a.n_cached = plan->sl_n_cached;
  a.cached_cols = plan->sl_cols.as<int32_t>();
  a.cached_rank = plan->sl_rank.as<int32_t>();
  a.ml_log2 = plan->sl_ml_log2;
  a.rows_per_wg = g.rows_per_wg;
  static std::atomic<size_t> lds_allowed[kMaxDevices];
  const int dev = ctx->device >= 0 && ctx->device < kMaxDevices ? ctx->device : -1;
  if (g.lds > (64u << 10) && (dev < 0 || g.lds > lds_allowed[dev].load(std::memory_order_relaxed))) {
    RFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fm_logit_slices_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(g.lds)));
    if (dev >= 0) lds_allowed[dev].store(g.lds, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(fm_logit_slices_kernel, dim3(g.grid), dim3(kSlBlock), g.lds, ctx->stream, a);
  RFM_HIP_CHECK(hipGetLastError());
}

}  // namespace rfm

using namespace rfm;

namespace {

void check_step_args(const rfm_fm_plan* plan, const void* indptr, const void* indices,
                     const void* values, const void* y, const void* p, const void* ids,
                     int64_t batch) {
  RFM_REQUIRE(plan, "null plan");
  // the step reads the plan's own records of the log; the caller's arrays are
  // only checked for presence (they are what the plan was built from)
  RFM_REQUIRE(indptr && indices && values && y && p && ids, "null pointer");
  RFM_REQUIRE(batch >= 1 && batch <= plan->max_batch, "batch=%lld outside 1..max_batch=%lld",
              (long long)batch, (long long)plan->max_batch);
}

// RFM_CHECK_IDS=1 (debugging aid; synchronises): the ids of every one of the n_iters
// batches must be rows of the plan's log and distinct within their batch
void validate_ids(rfm_ctx* ctx, rfm_fm_plan* plan, const int32_t* d_ids, int64_t batch,
                  int64_t n_iters) {
  if (env_int("RFM_CHECK_IDS", 0) == 0 || batch <= 0 || n_iters <= 0) return;
  if (!plan->ids_seen.p || plan->ids_stamp > INT32_MAX - n_iters - 2) {
    plan->ids_seen.ensure(size_t(plan->n_rows) * 4);
    plan->ids_flags.ensure(8);
    RFM_HIP_CHECK(hipMemsetAsync(plan->ids_seen.p, 0, plan->ids_seen.bytes, ctx->stream));
    plan->ids_stamp = 0;
  }
  RFM_HIP_CHECK(hipMemsetAsync(plan->ids_flags.p, 0, 8, ctx->stream));
  const int64_t total = batch * n_iters;
  const int grid = int(std::min<int64_t>((total + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 8));
  hipLaunchKernelGGL(ids_check_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_ids, batch,
                     n_iters, plan->n_rows, plan->ids_seen.as<int32_t>(), plan->ids_stamp + 1,
                     plan->ids_flags.as<int32_t>());
  plan->ids_stamp += int32_t(n_iters);
  int32_t flags[2] = {0, 0};
  RFM_HIP_CHECK(hipMemcpyAsync(flags, plan->ids_flags.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  RFM_REQUIRE(!flags[0], "a row id lies outside the plan's log (0..%lld)", (long long)plan->n_rows - 1);
  RFM_REQUIRE(!flags[1], "a row id occurs twice in one batch: the ids of a step must be distinct");
}

// a prepared step's inputs (rfm_fm_prep.hpp): the batch's row blocks in batch order and the
// tasks' records of this iteration
struct PrepView {
  const Entry* E;
  const double2* YP;
  const PrepRec* rec;
  const int32_t* cnt;
};

// the three launches of one step; grad == nullptr -> update in place
// rows of the plan's log that a step's forward launch scores on the side (fm_forward_kernel, XTRA)
struct XtraRows {
  const int32_t* ids;  // rows ids[0 .. n) of the plan's log ...
  int64_t n;
  double* out_pred;
  const RowRec* rows_y;  // ... and every row of another log given as records (n_y = 0: none)
  const Entry* ent_y;
  int64_t n_y;
  double* out_pred_y;
};
// whether a step of `batch` rows can take them along: the one-row forward shape, arrival-order hot
// sums, rows through the plan's records
bool step_takes_extra_rows(const rfm_ctx* ctx, const rfm_fm_plan* plan, int64_t batch) {
  return forward_geom(ctx, batch, shape_for(plan->k), true).block == kSmallBlock &&
         !(plan->hot_fixed && plan->n_hot > 0);
}

void enqueue_step(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                  const int32_t* d_indices, const double* d_values, const double* d_y,
                  const double* d_pscore, const int32_t* d_row_ids, int64_t batch, double* d_w0,
                  double* d_w, double* d_V, double lr, double* d_grad, int32_t* d_touch = nullptr,
                  int32_t touch_id = 0, const PrepView* prep = nullptr, const XtraRows* xtra = nullptr) {
  const int k = plan->k;
  const Shape s = shape_for(k);
  (void)d_indptr;
  (void)d_indices;
  (void)d_values;
  (void)d_y;
  (void)d_pscore;
  FwdArgs f{};
  f.ent = plan->ent.as<Entry>();
  f.rows = plan->rows.as<RowRec>();
  f.ell = plan->ell.as<char>();
  f.ell_stride = plan->ell_stride;
  f.ell_yp = plan->ell_yp.as<double2>();
  f.row_ids = d_row_ids;
  f.n_rows = batch;
  f.w0 = d_w0;
  f.w = d_w;
  f.V = d_V;
  f.k = k;
  f.out_err = plan->err.as<double>();
  f.out_Q = plan->Q.as<double>();
  f.slot_mark = plan->slot_t.as<SlotMark>();
  // more than one chunk per lane: the slot bitmap alternates between two buffers by step
  // parity (see fm_consume_kernel, CH form)
  const bool chunked = s.nc > 1;
  const int64_t parity = chunked ? ((plan->step + 1) & 1) : 0;
  unsigned long long* bits = plan->slot_bits.as<unsigned long long>() + parity * plan->bits_words;
  unsigned long long* bits_other =
      plan->slot_bits.as<unsigned long long>() + (1 - parity) * plan->bits_words;
  f.slot_bits = bits;
  if (prep) {  // nothing to mark ...
    f.slot_mark = nullptr;
    f.slot_bits = nullptr;
    if (prep->E) {  // ... and the rows at their batch position
      f.ent = nullptr;
      f.rows = nullptr;
      f.ell = reinterpret_cast<const char*>(prep->E);
      f.ell_yp = prep->YP;
      f.row_ids = nullptr;
    }
  }
  f.n_hot = plan->n_hot;
  f.hot_rounds = plan->hot_rounds;
  f.hot_fixed = plan->hot_fixed ? 1 : 0;
  f.hot_slab = plan->hot_slab.as<double>();
  f.err_partial = plan->err_partial.as<double>();
#ifdef RFM_ABLATE
  f.ablate = env_int("RFM_ABLATE_MASK", 0);
#endif
  const FwdGeom geom = forward_geom(ctx, batch, s, true);  // (the step's own workgroups: geom.grid slabs)
  FwdGeom launch = geom;
  if (xtra && xtra->n > 0) {
    RFM_REQUIRE(geom.block == kSmallBlock && !(f.hot_fixed && f.n_hot > 0) && !(prep && prep->E),
                "this step cannot score extra rows");
    const FwdGeom gx = forward_geom(ctx, xtra->n, s, true);
    RFM_REQUIRE(gx.block == kSmallBlock, "extra rows: unexpected geometry");
    f.grid_main = geom.grid;
    f.grid_x = gx.grid;
    f.row_ids_x = xtra->ids;
    f.n_rows_x = xtra->n;
    f.out_pred_x = xtra->out_pred;
    launch.grid = geom.grid + gx.grid;
    if (xtra->n_y > 0) {
      const FwdGeom gy = forward_geom(ctx, xtra->n_y, s, true);
      RFM_REQUIRE(gy.block == kSmallBlock && f.ent && f.rows, "extra log: unexpected geometry / plan form");
      f.rows_y = xtra->rows_y;
      f.ent_y = xtra->ent_y;
      f.n_rows_y = xtra->n_y;
      f.out_pred_y = xtra->out_pred_y;
      launch.grid += gy.grid;
    }
  }
#ifdef RFM_FWD_STAMPS
  // timing builds: clock readings of the many-rows forward, printed for the 60th step of the plan
  static DevBuf fwd_stamps;
  const size_t stamp_count = size_t(launch.grid) * (launch.block / kWave) * 2 * 8;
  if (env_int("RFM_FWD_STAMPS", 0)) {
    fwd_stamps.ensure(stamp_count * 8);
    RFM_HIP_CHECK(hipMemsetAsync(fwd_stamps.p, 0, stamp_count * 8, ctx->stream));
    f.stamps = static_cast<long long*>(fwd_stamps.p);
  }
#endif
  ctx->prof_mark();
#ifdef RFM_ABLATE
  if (!(f.ablate & 64))
#endif
    launch_forward(ctx, f, launch);
  ctx->prof_mark();
#ifdef RFM_FWD_STAMPS
  if (f.stamps && plan->step == 59) {
    std::vector<long long> h(stamp_count);
    RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    RFM_HIP_CHECK(hipMemcpy(h.data(), fwd_stamps.p, stamp_count * 8, hipMemcpyDeviceToHost));
    const int waves = launch.block / kWave;
    for (int wv : {0, waves / 2, waves - 1}) {
      double d[2][8] = {{0}}, n = 0;
      for (int b = 0; b < launch.grid; ++b) {
        const long long* t0 = &h[((size_t(b) * waves + wv) * 2 + 0) * 8];
        const long long* t1 = t0 + 8;
        if (!t0[0] || !t1[6]) continue;
        for (int i = 0; i < 8; ++i) {
          d[0][i] += t0[i] ? double(t0[i] - t0[0]) : 0.0;
          d[1][i] += t1[i] ? double(t1[i] - t0[0]) : 0.0;
        }
        n += 1;
      }
      if (n > 0)
        fprintf(stderr, "[forward stamps] wave %2d over %.0f workgroups, clocks since entry -- trip 0: rows in %.0f, gathers summed %.0f, "
                        "scores out %.0f, marks + hot adds %.0f | trip 1: start %.0f, rows in %.0f, gathers %.0f, scores %.0f, hot %.0f | "
                        "trips done %.0f, end %.0f\n",
                wv, n, d[0][1] / n, d[0][2] / n, d[0][3] / n, d[0][4] / n, d[1][0] / n, d[1][1] / n, d[1][2] / n, d[1][3] / n,
                d[1][4] / n, d[1][5] / n, d[1][6] / n);
    }
  }
#endif

  if (d_grad && !d_touch) {  // dense gradient: every element is written
    const size_t bytes = (size_t(plan->n_features) * (k + 1) + 1) * sizeof(double);
    RFM_HIP_CHECK(hipMemsetAsync(d_grad, 0, bytes, ctx->stream));
  }
  const double stamp = double(++plan->step);
  {
    ConsArgs c{};
    c.tasks = plan->tasks.as<TaskRec>();
    c.task_words = plan->task_words;
    c.slot_bits = bits;
    c.slot_bits_other = bits_other;
    c.slot_mark = plan->slot_t.as<SlotMark>();
    c.slots = plan->slots.as<SlotRec>();
    c.Q = plan->Q.as<double>();
    c.k = k;
    c.n = plan->n_features;
    c.w0 = d_w0;
    c.V = d_V;
    c.w = d_w;
    c.lr = lr;
    c.parts = plan->parts.as<double>();
    c.stamp = stamp;
    c.grad = d_grad;
    c.touch = d_touch;
    c.touch_id = touch_id;
    c.nb_tasks = plan->n_task_blocks;  // one task per lane group
    c.n_hot = plan->n_hot;
    c.hot_cols = plan->hot_cols.as<int32_t>();
    c.hot_slab = plan->hot_slab.as<double>();
    c.n_slabs = geom.grid;
    c.err_partial = plan->err_partial.as<double>();
    if (prep) {
      c.prep_rec = prep->rec;
      c.prep_cnt = prep->cnt;
      c.err = plan->err.as<double>();
    }
    const int grid = c.nb_tasks + c.n_hot + 1;  // tasks, then the hot columns, then w0
#ifdef RFM_CONS_STAMPS
    static DevBuf cons_stamps;
    const size_t cstamp_count = size_t(grid) * 8 * (kBlock / kWave) * 8;  // (x chunks at most 8)
    if (env_int("RFM_CONS_STAMPS", 0)) {
      cons_stamps.ensure(cstamp_count * 8);
      RFM_HIP_CHECK(hipMemsetAsync(cons_stamps.p, 0, cstamp_count * 8, ctx->stream));
      c.stamps = static_cast<long long*>(cons_stamps.p);
    }
#endif
    // LDS: the groups' lists + parked records + head rows, or the hot workgroups' scratch
    const int gpb = kBlock / s.lpr;
    const int win = s.lpr >= 32 ? 64 : 4 * s.lpr;  // WinShape<LPR>::WIN
    const size_t lds = std::max<size_t>(size_t(gpb) * size_t(win) * (sizeof(WinRec) + 8) +
                                            size_t(gpb) * size_t(k + 3) * 8,
                                        size_t(kBlock + 1024 + 2) * 8);
#ifdef RFM_ABLATE
    if (!(f.ablate & 128))
#endif
    {
      if (chunked) {
        // one workgroup per (four tasks, chunk of 64 lanes x vec factors)
        const int n_chunks = ((k + s.vec - 1) / s.vec + 63) / 64;
        c.n_chunks = n_chunks;
        // chunks dealt to XCDs (see the kernel): every chunk gets at least 8 / n_chunks XCDs
        static const bool xcd_on = env_int("RFM_XCD_CHUNKS", 1) != 0;
        c.xcd_chunks = xcd_on && n_chunks <= 8 ? 1 : 0;
        const int per_chunk = 8 / n_chunks;  // (the fewest XCDs a chunk gets)
        const dim3 g2 = c.xcd_chunks ? dim3(8 * ((grid + per_chunk - 1) / per_chunk)) : dim3(grid, n_chunks);
        if (s.vec == 2 && prep)
          hipLaunchKernelGGL((fm_consume_kernel<64, 2, 1, true, true>), g2, dim3(kBlock), lds, ctx->stream, c);
        else if (s.vec == 2)
          hipLaunchKernelGGL((fm_consume_kernel<64, 2, 1, true, false>), g2, dim3(kBlock), lds, ctx->stream, c);
        else if (prep)
          hipLaunchKernelGGL((fm_consume_kernel<64, 1, 1, true, true>), g2, dim3(kBlock), lds, ctx->stream, c);
        else
          hipLaunchKernelGGL((fm_consume_kernel<64, 1, 1, true, false>), g2, dim3(kBlock), lds, ctx->stream, c);
      } else if (prep) {
#define RFM_CALL_CONS(L, Vv, N)                                                                      \
  hipLaunchKernelGGL((fm_consume_kernel<L, Vv, N, false, true>), dim3(grid), dim3(kBlock), lds, \
                     ctx->stream, c)
        RFM_FOR_SINGLE_CHUNK_SHAPE(s, RFM_CALL_CONS);
#undef RFM_CALL_CONS
      } else {
#define RFM_CALL_CONS(L, Vv, N) \
  hipLaunchKernelGGL((fm_consume_kernel<L, Vv, N>), dim3(grid), dim3(kBlock), lds, ctx->stream, c)
        RFM_FOR_SINGLE_CHUNK_SHAPE(s, RFM_CALL_CONS);
#undef RFM_CALL_CONS
      }
      RFM_HIP_CHECK(hipGetLastError());
    }
#ifdef RFM_CONS_STAMPS
    if (c.stamps && plan->step == 60) {  // timing builds: the task workgroups' clock readings of the 60th step
      std::vector<long long> h(cstamp_count);
      RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      RFM_HIP_CHECK(hipMemcpy(h.data(), cons_stamps.p, cstamp_count * 8, hipMemcpyDeviceToHost));
      const int waves = kBlock / kWave;
      double d[8] = {0}, n = 0, life_max = 0;
      long long t_first = 0, t_last = 0;
      for (size_t b = 0; b < cstamp_count / 8 / waves; ++b) {
        for (int wv = 0; wv < waves; ++wv) {
          const long long* t = &h[(b * waves + wv) * 8];
          if (!t[0] || !t[4]) continue;
          for (int i = 1; i < 5; ++i) d[i] += double(t[i] - t[0]);
          life_max = std::max(life_max, double(t[4] - t[0]));
          t_first = t_first ? std::min(t_first, t[0]) : t[0];
          t_last = std::max(t_last, t[4]);
          n += 1;
        }
      }
      if (n > 0)
        fprintf(stderr, "[consume stamps] %.0f task wavefronts, clocks since their entry: slots listed %.0f, chain run %.0f, "
                        "barrier %.0f, combined %.0f (longest life %.0f)\n",
                n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, life_max);
      (void)t_first;
      (void)t_last;
      // ... and of the wavefronts that live longer than 0.7 of the longest: which workgroups, which phase
      double e[8] = {0}, m = 0, bsum = 0, bmin = 1e18, bmax = 0;
      for (size_t b = 0; b < cstamp_count / 8 / waves; ++b) {
        for (int wv = 0; wv < waves; ++wv) {
          const long long* t = &h[(b * waves + wv) * 8];
          if (!t[0] || !t[4] || double(t[4] - t[0]) < 0.7 * life_max) continue;
          for (int i = 1; i < 5; ++i) e[i] += double(t[i] - t[0]);
          bsum += double(b);
          bmin = std::min(bmin, double(b));
          bmax = std::max(bmax, double(b));
          m += 1;
        }
      }
      if (m > 0)
        fprintf(stderr, "[consume stamps]   the %.0f longest-lived: listed %.0f, chain run %.0f, barrier %.0f, combined %.0f; "
                        "workgroups %.0f .. %.0f (mean %.0f) of %d\n",
                m, e[1] / m, e[2] / m, e[3] / m, e[4] / m, bmin, bmax, bsum / m, grid);
    }
#endif
  }
  ctx->prof_mark();
  // columns cut into several tasks (none on most plans): their partial rows
  if (plan->n_split_short + plan->n_split_long > 0) {
    FinArgs fa{};
    fa.split = plan->split.as<SplitCol>();
    fa.n_split_short = plan->n_split_short;
    fa.n_split_long = plan->n_split_long;
    fa.parts = plan->parts.as<double>();
    fa.stamp = stamp;
    fa.k = k;
    fa.n = plan->n_features;
    fa.w = d_w;
    fa.V = d_V;
    fa.lr = lr;
    fa.grad = d_grad;
    fa.touch = d_touch;
    fa.touch_id = touch_id;
    const int gpb = kBlock / s.lpr;
    const int nb_short = (plan->n_split_short + gpb - 1) / gpb;
    const int grid = nb_short + plan->n_split_long;
#ifdef RFM_ABLATE
    if (f.ablate & 256) return;
#endif
    if (chunked) {
      const dim3 g2(grid, ((k + s.vec - 1) / s.vec + 63) / 64);
      if (s.vec == 2)
        hipLaunchKernelGGL(fm_finalize_chunk_kernel<2>, g2, dim3(kBlock), 0, ctx->stream, fa, nb_short);
      else
        hipLaunchKernelGGL(fm_finalize_chunk_kernel<1>, g2, dim3(kBlock), 0, ctx->stream, fa, nb_short);
    } else {
#define RFM_CALL_FIN(L, Vv, N)                                                                 \
  hipLaunchKernelGGL((fm_finalize_kernel<L, Vv, N>), dim3(grid), dim3(kBlock), 0, ctx->stream, \
                     fa, nb_short)
      RFM_FOR_SINGLE_CHUNK_SHAPE(s, RFM_CALL_FIN);
#undef RFM_CALL_FIN
    }
    RFM_HIP_CHECK(hipGetLastError());
  }
  ctx->prof_mark();
}

// The prepared steps of one rfm_fm_train call: chunks of plan->prep_iters iterations, two
// chunk buffers alive (the chunk being trained on, the next one already laid out).
struct PrepRun {
  rfm_ctx* ctx;
  rfm_fm_plan* plan;
  const int32_t* d_ids;
  int64_t batch, n_iters;
  int per_chunk = 0, n_chunks = 0;
  bool on = false, records_only = false;

  PrepRun(rfm_ctx* c, rfm_fm_plan* p, const int32_t* ids, int64_t b, int64_t n, bool allowed)
      : ctx(c), plan(p), d_ids(ids), batch(b), n_iters(n) {
    // (a short call -- an evaluator between iterations -- would pay the three launches of a
    // chunk for a handful of steps)
    records_only = plan->prep_records_only;
    on = allowed && plan->prep_ok && n_iters >= 8 && (records_only ? plan->rows.p != nullptr : plan->ell.p != nullptr);
    if (!on) return;
    per_chunk = plan->prep_iters;
    n_chunks = int((n_iters + per_chunk - 1) / per_chunk);
    enqueue(0);
    if (n_chunks > 1) enqueue(1);
  }

  void enqueue(int c) {
    auto& ch = plan->prep[c % 2];
    const size_t cap_it = size_t(plan->prep_iters), nt = size_t(plan->n_tasks), mb = size_t(plan->max_batch);
    if (!ch.tmp.p) {
      if (!records_only) {
        ch.E.alloc(cap_it * mb * size_t(plan->ell_stride));
        ch.YP.alloc(cap_it * mb * 16);
      }
      ch.tmp.alloc(cap_it * nt * kPrepCap * sizeof(PrepTmp));
      ch.rec.alloc(cap_it * nt * kPrepCap * sizeof(PrepRec));
      ch.cnt.alloc(cap_it * nt * 4);
      ch.flags.alloc(cap_it * 4);
      RFM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ch.h_flags), cap_it * 4, hipHostMallocDefault));
      RFM_HIP_CHECK(hipEventCreateWithFlags(&ch.ready, hipEventDisableTiming));
    }
    const int64_t first = int64_t(c) * per_chunk;
    const int n_it = int(std::min<int64_t>(per_chunk, n_iters - first));
    hipStream_t st = ctx->stream;
    RFM_HIP_CHECK(hipMemsetAsync(ch.cnt.p, 0, size_t(n_it) * nt * 4, st));
    RFM_HIP_CHECK(hipMemsetAsync(ch.flags.p, 0, size_t(n_it) * 4, st));
    const Shape s = shape_for(plan->k);
    if (records_only) {
      const int64_t waves = int64_t(n_it) * batch;
      const int grid_a = int(std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, int64_t(ctx->n_cu) * 32)));
      hipLaunchKernelGGL(fm_prep_bucket_kernel, dim3(grid_a), dim3(kBlock), 0, st, plan->rows.as<RowRec>(),
                         plan->ent.as<Entry>(), d_ids + first * batch, batch, batch, n_it,
                         int32_t(plan->task_words * 64), plan->n_tasks, ch.tmp.as<PrepTmp>(),
                         ch.cnt.as<int32_t>());
    } else {
      const int64_t items = int64_t(n_it) * batch * s.lpr;
      const int grid_a = int(std::max<int64_t>(1, std::min<int64_t>((items + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 32)));
      hipLaunchKernelGGL(fm_prep_gather_kernel, dim3(grid_a), dim3(kBlock), 0, st, plan->ell.as<char>(),
                         plan->ell_stride, plan->ell_yp.as<double2>(), s.lpr, d_ids + first * batch, batch,
                         batch, n_it, int32_t(plan->task_words * 64), plan->n_tasks, ch.E.as<Entry>(),
                         ch.YP.as<double2>(), ch.tmp.as<PrepTmp>(), ch.cnt.as<int32_t>());
    }
    const int64_t buckets = int64_t(n_it) * plan->n_tasks;
    const int grid_b = int(std::max<int64_t>(1, std::min<int64_t>((buckets + 15) / 16, int64_t(ctx->n_cu) * 32)));
    hipLaunchKernelGGL(fm_prep_sort_kernel, dim3(grid_b), dim3(kBlock), 0, st, ch.tmp.as<PrepTmp>(),
                       ch.cnt.as<int32_t>(), plan->slots.as<SlotRec>(), plan->n_tasks, n_it,
                       ch.rec.as<PrepRec>(), ch.flags.as<int32_t>());
    RFM_HIP_CHECK(hipGetLastError());
    RFM_HIP_CHECK(hipMemcpyAsync(ch.h_flags, ch.flags.p, size_t(n_it) * 4, hipMemcpyDeviceToHost, st));
    RFM_HIP_CHECK(hipEventRecord(ch.ready, st));
  }

  // the view of iteration `it`, or false when the iteration is not prepared (a task with more
  // than kPrepCap marks: it takes the bitmap path)
  bool view(int64_t it, PrepView& v) {
    if (!on) return false;
    const int c = int(it / per_chunk), j = int(it % per_chunk);
    auto& ch = plan->prep[c % 2];
    if (j == 0) RFM_HIP_CHECK(hipEventSynchronize(ch.ready));  // its flags are on the host
    if (ch.h_flags[j]) return false;
    const size_t nt = size_t(plan->n_tasks);
    v.E = records_only ? nullptr
                       : reinterpret_cast<const Entry*>(ch.E.as<char>() + size_t(j) * size_t(batch) * size_t(plan->ell_stride));
    v.YP = records_only ? nullptr : ch.YP.as<double2>() + size_t(j) * size_t(batch);
    v.rec = ch.rec.as<PrepRec>() + size_t(j) * nt * kPrepCap;
    v.cnt = ch.cnt.as<int32_t>() + size_t(j) * nt;
    return true;
  }

  // after the last step of chunk c has been enqueued: its buffer is free for chunk c + 2
  void done(int64_t it) {
    if (!on) return;
    const int c = int(it / per_chunk);
    if ((it + 1) % per_chunk == 0 && c + 2 < n_chunks) enqueue(c + 2);
  }
};

// Timing experiment (RFM_TRAIN_GRAPH): everything enqueued on the context's stream while this
// object lives is captured into one hipGraph on a stream of its own (the legacy default stream
// cannot be captured) and run by replay().  Whatever happens in between -- an RFM_REQUIRE, a HIP
// error -- the destructor ends the capture, restores ctx->stream and frees the stream.
struct GraphCapture {
  rfm_ctx* ctx;
  hipStream_t user_stream = nullptr, cap_stream = nullptr;
  bool capturing = false;
  GraphCapture(rfm_ctx* c, bool on) : ctx(c) {
    if (!on) return;
    user_stream = ctx->stream;
    RFM_HIP_CHECK(hipStreamSynchronize(user_stream));
    RFM_HIP_CHECK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    if (hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipStreamDestroy(cap_stream);
      cap_stream = nullptr;
      fail(RFM_ERR_HIP, "hipStreamBeginCapture failed");
    }
    ctx->stream = cap_stream;
    capturing = true;
  }
  GraphCapture(const GraphCapture&) = delete;
  GraphCapture& operator=(const GraphCapture&) = delete;
  hipGraph_t end() {
    hipGraph_t graph = nullptr;
    if (capturing) {
      capturing = false;
      ctx->stream = user_stream;
      if (hipStreamEndCapture(cap_stream, &graph) != hipSuccess) graph = nullptr;
    }
    return graph;
  }
  void replay(int64_t n_iters) {
    if (!cap_stream) return;
    hipGraph_t graph = end();
    RFM_REQUIRE(graph, "hipStreamEndCapture failed");
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
      (void)hipGraphDestroy(graph);
      fail(RFM_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    const bool timed = env_int("RFM_TRAIN_GRAPH", 0) > 1;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t run = hipGraphLaunch(exec, cap_stream);
    if (run == hipSuccess) run = hipStreamSynchronize(cap_stream);
    if (timed && run == hipSuccess)
      fprintf(stderr, "[rfm] graph of %lld iterations: %.2f us per iteration (launch to drain)\n",
              (long long)n_iters,
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() /
                  double(std::max<int64_t>(n_iters, 1)));
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    RFM_HIP_CHECK(run);
  }
  ~GraphCapture() {
    if (hipGraph_t graph = end()) (void)hipGraphDestroy(graph);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
  }
};

FwdArgs forward_args(const int64_t* d_indptr, const int32_t* d_indices, const double* d_values,
                     const int32_t* d_row_ids, int64_t n_rows, const double* d_w0,
                     const double* d_w, const double* d_V, int32_t k) {
  FwdArgs f{};
  f.indptr = d_indptr;
  f.indices = d_indices;
  f.values = d_values;
  f.row_ids = d_row_ids;
  f.n_rows = n_rows;
  f.w0 = d_w0;
  f.w = d_w;
  f.V = d_V;
  f.k = k;
#ifdef RFM_ABLATE
  f.ablate = env_int("RFM_ABLATE_MASK", 0);  // (timing experiments: bit 4 = every gather reads row 0 of V)
#endif
  return f;
}

}  // namespace

#include "rfm_fm_dp.hpp"

extern "C" {

int32_t rfm_fm_forward(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                       const double* d_values, const int32_t* d_row_ids, int64_t n_rows,
                       const double* d_w0, const double* d_w, const double* d_V,
                       int64_t n_features, int32_t n_factors, double* d_out_pred) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    RFM_REQUIRE(n_rows >= 0 && n_rows < (int64_t(1) << 31) && n_features >= 1, "bad shape");
    if (n_rows == 0) return;  // nothing to score (empty inputs carry null pointers)
    RFM_REQUIRE(d_indptr && d_w0 && d_w && d_V && d_out_pred, "null pointer");
    RFM_REQUIRE(d_indices && d_values, "null CSR arrays");
    FwdArgs f = forward_args(d_indptr, d_indices, d_values, d_row_ids, n_rows, d_w0, d_w, d_V,
                             n_factors);
    f.out_pred = d_out_pred;
    launch_forward(ctx, f);
  });
}

int32_t rfm_ips_logloss(rfm_ctx* ctx, const double* d_y, const double* d_pred,
                        const double* d_pscore, const int32_t* d_row_ids, int64_t n_rows,
                        double eps, double* d_out_loss) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_y && d_pred && d_pscore && d_out_loss, "null pointer");
    RFM_REQUIRE(n_rows >= 1, "loss of zero rows");
    const int grid =
        int(std::min<int64_t>((n_rows + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 8));
    ctx->loss_partials.ensure(size_t(std::max(kMaxFwdGrid, ctx->n_cu * 8)) * sizeof(double));
    hipLaunchKernelGGL(logloss_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_y, d_pred,
                       d_pscore, d_row_ids, n_rows, eps, ctx->loss_partials.as<double>());
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, ctx->stream,
                       ctx->loss_partials.as<double>(), grid, n_rows, d_out_loss);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_fm_forward_loss(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                            const double* d_values, const double* d_y, const double* d_pscore,
                            const int32_t* d_row_ids, int64_t n_rows, const double* d_w0,
                            const double* d_w, const double* d_V, int64_t n_features,
                            int32_t n_factors, double eps, double* d_out_pred,
                            double* d_out_loss) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_indptr && d_indices && d_values && d_y && d_pscore && d_w0 && d_w &&
                    d_V && d_out_loss,
                "null pointer");
    RFM_REQUIRE(n_rows >= 1 && n_rows < (int64_t(1) << 31) && n_features >= 1, "bad shape");
    FwdArgs f = forward_args(d_indptr, d_indices, d_values, d_row_ids, n_rows, d_w0, d_w, d_V,
                             n_factors);
    f.y = d_y;
    f.pscore = d_pscore;
    f.eps = eps;
    f.out_pred = d_out_pred;
    forward_loss(ctx, f, d_out_loss);
  });
}

int32_t rfm_fm_step(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                    const int32_t* d_indices, const double* d_values, const double* d_y,
                    const double* d_pscore, const int32_t* d_row_ids, int64_t batch,
                    double* d_w0, double* d_w, double* d_V, double lr) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V, "null pointer");
    check_step_args(plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_row_ids, batch);
    validate_ids(ctx, plan, d_row_ids, batch, 1);
    enqueue_step(ctx, plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_row_ids, batch,
                 d_w0, d_w, d_V, lr, nullptr);
  });
}

int32_t rfm_fm_grad(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                    const int32_t* d_indices, const double* d_values, const double* d_y,
                    const double* d_pscore, const int32_t* d_row_ids, int64_t batch,
                    const double* d_w0, const double* d_w, const double* d_V, double* d_grad) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V && d_grad, "null pointer");
    check_step_args(plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_row_ids, batch);
    validate_ids(ctx, plan, d_row_ids, batch, 1);
    enqueue_step(ctx, plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_row_ids, batch,
                 const_cast<double*>(d_w0), const_cast<double*>(d_w), const_cast<double*>(d_V),
                 0.0, d_grad);
  });
}

int32_t rfm_fm_apply(rfm_ctx* ctx, double* d_w0, double* d_w, double* d_V,
                     const double* d_grad, int64_t n_features, int32_t n_factors, double lr) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V && d_grad, "null pointer");
    RFM_REQUIRE(n_features >= 1 && n_factors >= 1, "bad shape");
    const int64_t nk = n_features * int64_t(n_factors);
    const int64_t total = nk + n_features + 1;
    const int grid =
        int(std::min<int64_t>((total + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 16));
    hipLaunchKernelGGL(fm_apply_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_V, d_w, d_w0,
                       d_grad, nk, n_features, lr);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------------------
// touched-row gradients and their exchange (SURVEY.md 8e option 1)
// ---------------------------------------------------------------------------
int32_t rfm_fm_grad_rows(rfm_ctx* ctx, rfm_fm_plan* plan, const int32_t* d_row_ids, int64_t batch,
                         const double* d_w0, const double* d_w, const double* d_V, double* d_rows,
                         int64_t cap_rows, int32_t* d_n_rows, double* d_gw0,
                         const int32_t* d_range_lo, int32_t n_ranges, int32_t* d_range_bounds) {
  return guarded([&] {
    RFM_REQUIRE(ctx && plan && d_w0 && d_w && d_V && d_rows && d_n_rows && d_gw0, "null pointer");
    RFM_REQUIRE(batch >= 0 && batch <= plan->max_batch, "batch=%lld outside 0..max_batch=%lld",
                (long long)batch, (long long)plan->max_batch);
    RFM_REQUIRE(batch == 0 || d_row_ids, "null row ids");
    RFM_REQUIRE(cap_rows >= 0, "negative capacity");
    RFM_REQUIRE(n_ranges >= 0 && n_ranges <= kMaxRanges, "n_ranges=%d outside 0..%d", n_ranges,
                kMaxRanges);
    RFM_REQUIRE(n_ranges == 0 || (d_range_lo && d_range_bounds), "null range arrays");
    const int32_t id = next_touch_ids(ctx, plan, 1);
    validate_ids(ctx, plan, d_row_ids, batch, 1);
    enqueue_grad_rows(ctx, plan, id, d_row_ids, batch, d_w0, d_w, d_V, d_rows, cap_rows, d_n_rows,
                      d_gw0, d_range_lo, n_ranges, d_range_bounds);
  });
}

int32_t rfm_fm_apply_rows(rfm_ctx* ctx, const double* d_rows, const int32_t* d_n_rows,
                          int64_t cap_rows, const double* d_gw0, double* d_w0, double* d_w,
                          double* d_V, int64_t n_features, int32_t n_factors, double lr) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_rows && d_n_rows && d_w0 && d_w && d_V, "null pointer");
    RFM_REQUIRE(n_features >= 1 && n_factors >= 1 && cap_rows >= 0, "bad shape");
    const int wpb = kBlock / kWave;
    const int grid = int(std::max<int64_t>(
        1, std::min<int64_t>((cap_rows + wpb - 1) / wpb, int64_t(ctx->n_cu) * 8)));
    hipLaunchKernelGGL(rows_apply_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_rows,
                       d_n_rows, cap_rows, d_gw0, d_w0, d_w, d_V, n_features, int(n_factors), lr);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_fm_reduce_rows(rfm_ctx* ctx, const double* d_rows, const int32_t* d_seg_ptr,
                           int32_t n_segments, int64_t total_rows, const double* d_w,
                           const double* d_V, int64_t n_features, int32_t n_factors, double lr,
                           double* d_out_rows) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_seg_ptr && d_w && d_V, "null pointer");
    RFM_REQUIRE(n_segments >= 1 && n_segments <= kWave, "n_segments=%d outside 1..%d", n_segments,
                kWave);
    RFM_REQUIRE(n_features >= 1 && n_factors >= 1 && total_rows >= 0, "bad shape");
    if (total_rows == 0) return;
    RFM_REQUIRE(d_rows && d_out_rows, "null record lists");
    const int wpb = kBlock / kWave;
    const int grid = int(std::min<int64_t>((total_rows + wpb - 1) / wpb, int64_t(ctx->n_cu) * 8));
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_rows,
                       d_seg_ptr, int(n_segments), d_w, d_V, n_features, int(n_factors), lr,
                       d_out_rows);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_fm_set_rows(rfm_ctx* ctx, const double* d_rows, int64_t n_rows,
                        const double* d_gw0_parts, int32_t n_parts, int64_t part_stride,
                        double* d_w0, double* d_w, double* d_V, int64_t n_features,
                        int32_t n_factors, double lr) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V, "null pointer");
    RFM_REQUIRE(n_features >= 1 && n_factors >= 1 && n_rows >= 0 && n_parts >= 0, "bad shape");
    RFM_REQUIRE(n_rows == 0 || d_rows, "null record list");
    RFM_REQUIRE(n_parts == 0 || d_gw0_parts, "null g_w0 partials");
    if (n_rows == 0 && n_parts == 0) return;
    const int wpb = kBlock / kWave;
    const int grid = int(std::max<int64_t>(
        1, std::min<int64_t>((n_rows + wpb - 1) / wpb, int64_t(ctx->n_cu) * 8)));
    hipLaunchKernelGGL(rows_set_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_rows, n_rows,
                       n_parts ? d_gw0_parts : nullptr, int(n_parts), part_stride, d_w0, d_w, d_V,
                       n_features, int(n_factors), lr);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_fm_train_dp(rfm_ctx* ctx, rfm_fm_plan* plan, const int32_t* d_ids,
                        int64_t global_batch, int64_t shard_lo, int64_t shard_hi,
                        int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                        double* d_grad) {
  return guarded([&] {
    RFM_REQUIRE(ctx && plan && d_ids && d_w0 && d_w && d_V && d_grad, "null pointer");
    RFM_REQUIRE(n_iters >= 0 && global_batch >= 1, "bad shape");
    RFM_REQUIRE(0 <= shard_lo && shard_lo <= shard_hi && shard_hi <= global_batch, "bad shard");
    RFM_REQUIRE(shard_hi - shard_lo <= plan->max_batch, "shard larger than the plan's max_batch");
    for (int64_t it = 0; it < n_iters; ++it)
      validate_ids(ctx, plan, d_ids + it * global_batch + shard_lo, shard_hi - shard_lo, 1);
    const int64_t count = plan->n_features * int64_t(plan->k + 1) + 1;
    const int64_t nk = plan->n_features * int64_t(plan->k);
    const int apply_grid =
        int(std::min<int64_t>((count + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 16));
    for (int64_t it = 0; it < n_iters; ++it) {
      if (shard_hi > shard_lo) {
        enqueue_step(ctx, plan, nullptr, nullptr, nullptr, nullptr, nullptr,
                     d_ids + it * global_batch + shard_lo, shard_hi - shard_lo, d_w0, d_w, d_V,
                     0.0, d_grad);
      } else {
        RFM_HIP_CHECK(hipMemsetAsync(d_grad, 0, size_t(count) * sizeof(double), ctx->stream));
      }
      if (ctx->comm && ctx->comm_ranks > 1) {
        const int32_t rc = rfm_allreduce_sum(ctx, d_grad, count);
        if (rc != RFM_OK) throw Error(rc, "all-reduce of the gradient failed (see above)");
      }
      hipLaunchKernelGGL(fm_apply_kernel, dim3(apply_grid), dim3(kBlock), 0, ctx->stream, d_V,
                         d_w, d_w0, d_grad, nk, plan->n_features, lr);
      RFM_HIP_CHECK(hipGetLastError());
    }
  });
}

}  // extern "C"

namespace {

// scores of the rows of a CSR through the plan (rfm_fm_plan_forward): the sliced forward where
// the plan has one and the rows are enough for it, else the plain one
void plan_forward(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr, const int32_t* d_indices,
                  const double* d_values, int64_t n_rows, const double* d_w0, const double* d_w,
                  const double* d_V, double* d_out_pred) {
  if (n_rows == 0) return;
  const SlicedGeom sliced = sliced_geom(ctx, plan, n_rows);
  if (!sliced.ok) {  // the plain forward (rfm_fm_forward)
    FwdArgs f = forward_args(d_indptr, d_indices, d_values, nullptr, n_rows, d_w0, d_w, d_V, plan->k);
    f.out_pred = d_out_pred;
    launch_forward(ctx, f);
    return;
  }
  rfm_fm_plan::SlLog& log = plan->sl_log[1];
  if (!log.holds(d_indptr, d_indices, d_values, n_rows)) {
    log.rows = -1;
    sliced_translate(ctx, plan, d_indptr, d_indices, d_values, n_rows, log.tr);
  }
  plan->sl_zf.ensure(size_t(plan->sl_ns) * size_t(n_rows) * 8);
  SlicedArgs f{};
  f.tr_a = plan->sl_train.as<SlEnt>();
  f.tr_b = log.tr.as<SlEnt>();
  f.pad = plan->sl_pad.as<SlEnt>();
  f.indptr_b = d_indptr;
  f.indices_b = d_indices;
  f.values_b = d_values;
  f.n_b = n_rows;
  f.w0 = d_w0;
  f.w = d_w;
  f.V = d_V;
  f.zpart = plan->sl_zf.as<double>();
  launch_sliced(ctx, plan, sliced, f);
  const int grid = int(std::min<int64_t>((n_rows + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 8));
  hipLaunchKernelGGL(scores_from_slices_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream,
                     plan->sl_zf.as<double>(), plan->sl_ns, n_rows, d_out_pred);
  RFM_HIP_CHECK(hipGetLastError());
}

// what a fit() loop with a ValEvaluator does after every iteration (utils/search_params.py:96-111,
// utils/evaluate.py:160-207): score the evaluation log, take its IPS-DCG@k (rfm_val_dcg)
struct EvalHook {
  const int64_t* indptr;
  const int32_t* indices;
  const double* values;
  int64_t n_rows;
  const int32_t* seg_ptr;
  const int32_t* rows;
  const double* labels;
  const double* pscores;
  int32_t n_segments, k;
  double* scores;        // [n_slots][scores_stride]; iteration i of the call -> slot slot_first + i
  int64_t scores_stride;
  double* user_scratch;  // [n_slots][user_stride]
  int64_t user_stride;
  int64_t slot_first;
  double* dcg_out;       // [n_iters][2]
};

void train_loop(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr, const int32_t* d_indices,
                const double* d_values, const double* d_y, const double* d_pscore, const int32_t* d_ids,
                int64_t batch, int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                const int64_t* d_val_indptr, const int32_t* d_val_indices, const double* d_val_values,
                const double* d_val_y, const double* d_val_pscore, int64_t n_val, double eps,
                double* d_out_train_loss, double* d_out_val_loss, const EvalHook* hook);

}  // namespace

extern "C" {

int32_t rfm_fm_plan_forward(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                            const int32_t* d_indices, const double* d_values, int64_t n_rows,
                            const double* d_w0, const double* d_w, const double* d_V,
                            double* d_out_pred) {
  return guarded([&] {
    RFM_REQUIRE(ctx && plan && d_indptr && d_w0 && d_w && d_V && d_out_pred, "null pointer");
    RFM_REQUIRE(n_rows >= 0, "negative n_rows");
    plan_forward(ctx, plan, d_indptr, d_indices, d_values, n_rows, d_w0, d_w, d_V, d_out_pred);
  });
}

int32_t rfm_fm_train(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                     const int32_t* d_indices, const double* d_values, const double* d_y,
                     const double* d_pscore, const int32_t* d_ids, int64_t batch,
                     int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                     const int64_t* d_val_indptr, const int32_t* d_val_indices,
                     const double* d_val_values, const double* d_val_y,
                     const double* d_val_pscore, int64_t n_val, double eps,
                     double* d_out_train_loss, double* d_out_val_loss) {
  return guarded([&] {
    train_loop(ctx, plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_ids, batch, n_iters, d_w0, d_w,
               d_V, lr, d_val_indptr, d_val_indices, d_val_values, d_val_y, d_val_pscore, n_val, eps,
               d_out_train_loss, d_out_val_loss, nullptr);
  });
}

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
                          double* d_dcg_out) {
  return guarded([&] {
    RFM_REQUIRE(d_ev_indptr && d_seg_ptr && d_rows && d_labels && d_scores && d_user_scratch && d_dcg_out,
                "null pointer (evaluation)");
    RFM_REQUIRE(n_ev >= 1 && scores_stride >= n_ev && user_stride >= 3 * int64_t(n_segments) && slot_first >= 0,
                "bad evaluation shape");
    const EvalHook hook{d_ev_indptr, d_ev_indices, d_ev_values, n_ev, d_seg_ptr, d_rows, d_labels,
                        d_ev_pscores, n_segments, k, d_scores, scores_stride, d_user_scratch, user_stride,
                        slot_first, d_dcg_out};
    train_loop(ctx, plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_ids, batch, n_iters, d_w0, d_w,
               d_V, lr, d_val_indptr, d_val_indices, d_val_values, d_val_y, d_val_pscore, n_val, eps,
               d_out_train_loss, d_out_val_loss, &hook);
  });
}

}  // extern "C"

namespace {

void train_loop(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr, const int32_t* d_indices,
                const double* d_values, const double* d_y, const double* d_pscore, const int32_t* d_ids,
                int64_t batch, int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                const int64_t* d_val_indptr, const int32_t* d_val_indices, const double* d_val_values,
                const double* d_val_y, const double* d_val_pscore, int64_t n_val, double eps,
                double* d_out_train_loss, double* d_out_val_loss, const EvalHook* hook) {
  {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V, "null pointer");
    RFM_REQUIRE(n_iters >= 0, "negative n_iters");
    if (n_iters == 0) return;
    check_step_args(plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_ids, batch);
    validate_ids(ctx, plan, d_ids, batch, n_iters);
    if (d_out_val_loss)
      RFM_REQUIRE(d_val_indptr && d_val_indices && d_val_values && d_val_y && d_val_pscore &&
                      n_val >= 1,
                  "validation arrays missing");
    // the losses' per-workgroup partials of a run of iterations are finished by one launch
    // per run (fixed order inside an iteration, as loss_finish_kernel does it)
    constexpr int64_t kRun = 128;
    if (d_out_train_loss || d_out_val_loss)
      plan->loss_rows.ensure(size_t(2 * kRun) * size_t(kMaxFwdGrid) * sizeof(double));
    double* train_rows = plan->loss_rows.as<double>();
    double* val_rows = train_rows + kRun * kMaxFwdGrid;
    int train_parts = 0, val_parts = 0;
    // factor counts of several chunks per lane: the loss forwards sliced by factors, ONE launch
    // per iteration that leaves partial logits; the scores and logarithms of a whole run of
    // iterations are then computed together (rfm_fm_sliced.hpp)
    const int64_t sl_a = d_out_train_loss ? batch : 0, sl_b = d_out_val_loss ? n_val : 0;
    const SlicedGeom sliced = sliced_geom(ctx, plan, sl_a + sl_b);
    // RFM_TRAIN_GRAPH=1 (timing experiment): the run's launches captured into one hipGraph
    // and replayed once (GraphCapture restores the context's stream on every way out)
    static const bool as_graph = env_int("RFM_TRAIN_GRAPH", 0) != 0;
    // (prepared steps wait for an event on the host at every chunk: not inside a capture)
    PrepRun prepared(ctx, plan, d_ids, batch, n_iters, !as_graph);
    // one decision for the whole call (the partials of a run are finished together).  Only for
    // factor counts of several chunks per lane: there it saves a launch (k = 400, B = 2 000:
    // 0.098 -> 0.090 ms per iteration of fit()); at one chunk per lane the batch's rows are
    // faster through the plan's padded row blocks than through the CSR arrays (config 3:
    // 104.7 vs 111.6 us per iteration at B = 65 536).  RFM_MERGE_LOSS=0 / 2: never / always.
    const int merge_mode = env_int("RFM_MERGE_LOSS", 1);
    const bool merge_call = !sliced.ok && merge_mode != 0 && (merge_mode == 2 || shape_for(plan->k).nc > 1) &&
                            d_out_train_loss && d_out_val_loss && (!prepared.on || prepared.records_only) &&
                            forward_geom(ctx, batch + n_val, shape_for(plan->k), false).block == kBigBlock;
    // The plain loss forwards (neither sliced nor merged) leave their rows' SCORES and take no
    // logarithms: the two logs of a row's term are ~200 dependent f64 instructions, and a whole
    // run of iterations' terms are computed by one launch instead (RFM_DEFER_LOSS=0: in the
    // forward, staged through LDS).
    // (a call of a few iterations -- a fit() with a host evaluator trains one per call -- would only
    // add the run's launches)
    const bool scores_only = !sliced.ok && !merge_call && sl_a + sl_b > 0 && n_iters >= 4 &&
                             env_int("RFM_DEFER_LOSS", 1) != 0;
    const bool deferred = sliced.ok || scores_only;
    const int zns = sliced.ok ? plan->sl_ns : 0;  // (0: the buffer holds scores)
    int64_t run_len = kRun;
    const int64_t z_per_iter = int64_t(std::max(zns, 1)) * (sl_a + sl_b);
    if (deferred) {
      run_len = std::max<int64_t>(1, std::min<int64_t>(kRun, (int64_t(256) << 20) / (z_per_iter * 8)));
      run_len = std::min(run_len, n_iters);
      plan->sl_z.ensure(size_t(run_len) * size_t(z_per_iter) * 8);
      // (the validation log is only known here: translated once per call)
      // (... unless the caller has registered these arrays: rfm_fm_plan_register_log, slot 0)
      if (sliced.ok && sl_b > 0 && !plan->sl_log[0].holds(d_val_indptr, d_val_indices, d_val_values, sl_b)) {
        plan->sl_log[0].rows = -1;  // (the slot is about to hold another log)
        sliced_translate(ctx, plan, d_val_indptr, d_val_indices, d_val_values, sl_b, plan->sl_log[0].tr);
      }
    }
#ifdef RFM_SLICED_STAMPS
    DevBuf stamps;
    if (sliced.ok && env_int("RFM_SLICED_STAMPS", 0)) {
      stamps.alloc(size_t(sliced.grid) * kSlWaves * 8 * 8);
      RFM_HIP_CHECK(hipMemsetAsync(stamps.p, 0, stamps.bytes, ctx->stream));
    }
#endif
    const auto finish = [&](int64_t first, int64_t count) {
      if (count <= 0) return;
      if (deferred) {
        const auto shares = [&](int64_t rows) { return int(std::min<int64_t>(64, (rows + kBlock - 1) / kBlock)); };
        if (sl_a > 0) {
          train_parts = shares(sl_a);
          hipLaunchKernelGGL(loss_from_slices_kernel, dim3(train_parts, int(count)), dim3(kBlock), 0,
                             ctx->stream, plan->sl_z.as<double>(), z_per_iter, zns,
                             sl_a + sl_b, int64_t(0), sl_a, d_ids + first * batch, batch, d_y,
                             d_pscore, eps, train_rows, int64_t(kMaxFwdGrid));
        }
        if (sl_b > 0) {
          val_parts = shares(sl_b);
          hipLaunchKernelGGL(loss_from_slices_kernel, dim3(val_parts, int(count)), dim3(kBlock), 0,
                             ctx->stream, plan->sl_z.as<double>(), z_per_iter, zns,
                             sl_a + sl_b, sl_a, sl_b, static_cast<const int32_t*>(nullptr),
                             int64_t(0), d_val_y, d_val_pscore, eps, val_rows, int64_t(kMaxFwdGrid));
        }
      }
      if (d_out_train_loss)
        hipLaunchKernelGGL(loss_finish_many_kernel, dim3(int(count)), dim3(kBlock), 0, ctx->stream,
                           train_rows, int64_t(kMaxFwdGrid), train_parts, batch,
                           d_out_train_loss + first);
      if (d_out_val_loss)
        hipLaunchKernelGGL(loss_finish_many_kernel, dim3(int(count)), dim3(kBlock), 0, ctx->stream,
                           val_rows, int64_t(kMaxFwdGrid), val_parts, n_val, d_out_val_loss + first);
      RFM_HIP_CHECK(hipGetLastError());
    };
    GraphCapture capture(ctx, as_graph && !ctx->profiling);
    // Small batches: the train-loss forward of iteration it - 1 reads the parameters that step
    // it's forward reads -- it RIDES in that launch (extra workgroups that only score the previous
    // batch's rows; fm_forward_kernel's XTRA form), and only the last iteration's is a launch of
    // its own.  A run's logarithms then wait for the next step's launch.  (RFM_RIDE_LOSS=0: never.)
    const bool ride = scores_only && d_out_train_loss && step_takes_extra_rows(ctx, plan, batch) &&
                      !(prepared.on && !prepared.records_only) && env_int("RFM_RIDE_LOSS", 1) != 0;
    // ... and so may the validation rows (the same parameters again), when the caller has registered
    // the log (rfm_fm_plan_register_log keeps it as records), it takes the one-row shape too, and the
    // plan holds plain records (RFM_RIDE_VAL=0: never)
    const rfm_fm_plan::SlLog& vlog = plan->sl_log[0];
    const bool ride_val = ride && d_out_val_loss && vlog.records &&
                          vlog.holds(d_val_indptr, d_val_indices, d_val_values, n_val) && plan->ent.p &&
                          !plan->ell.p && forward_geom(ctx, n_val, shape_for(plan->k), true).block == kSmallBlock &&
                          env_int("RFM_RIDE_VAL", 1) != 0;
    int64_t run_first = 0, pending_first = -1, pending_count = 0;
    for (int64_t it = 0; it < n_iters; ++it) {
      const int32_t* ids = d_ids + it * batch;
      const int64_t slot = it - run_first;
      PrepView pv{};
      const bool is_prepared = prepared.view(it, pv);
      XtraRows prev{};
      if (ride && it > 0) {
        const int64_t prev_slot = pending_count > 0 ? pending_count - 1 : slot - 1;
        double* zs = plan->sl_z.as<double>() + prev_slot * z_per_iter;
        prev = XtraRows{ids - batch, batch, zs, nullptr, nullptr, 0, nullptr};
        if (ride_val)
          prev = XtraRows{ids - batch, batch, zs, vlog.rows_rec.as<RowRec>(), vlog.ent_rec.as<Entry>(),
                          n_val, zs + sl_a};
      }
      enqueue_step(ctx, plan, d_indptr, d_indices, d_values, d_y, d_pscore, ids, batch, d_w0,
                   d_w, d_V, lr, nullptr, nullptr, 0, is_prepared ? &pv : nullptr,
                   prev.n > 0 ? &prev : nullptr);
      if (pending_count > 0) {  // (before this iteration's forwards reuse the run's first slots)
        finish(pending_first, pending_count);
        pending_count = 0;
      }
      // both losses asked for: ONE launch over the batch's rows of the training log and the
      // validation log (RFM_MERGE_LOSS=0: two launches)
      bool merged = false;
      if (sliced.ok) {
        SlicedArgs f{};
        f.tr_a = plan->sl_train.as<SlEnt>();
        f.tr_b = plan->sl_log[0].tr.as<SlEnt>();
        f.pad = plan->sl_pad.as<SlEnt>();
        f.indptr_a = d_indptr;
        f.indices_a = d_indices;
        f.values_a = d_values;
        f.row_ids = ids;
        f.n_a = sl_a;
        f.indptr_b = d_val_indptr;
        f.indices_b = d_val_indices;
        f.values_b = d_val_values;
        f.n_b = sl_b;
        f.w0 = d_w0;
        f.w = d_w;
        f.V = d_V;
        f.zpart = plan->sl_z.as<double>() + slot * z_per_iter;
#ifdef RFM_SLICED_STAMPS
        f.stamps = static_cast<long long*>(stamps.p);
#endif
        launch_sliced(ctx, plan, sliced, f);
        merged = true;
      } else if (merge_call) {
        FwdArgs f = forward_args(d_indptr, d_indices, d_values, ids, batch + n_val, d_w0, d_w, d_V, plan->k);
        f.n_rows_a = batch;
        f.y = d_y;
        f.pscore = d_pscore;
        f.indptr2 = d_val_indptr;
        f.indices2 = d_val_indices;
        f.values2 = d_val_values;
        f.y2 = d_val_y;
        f.pscore2 = d_val_pscore;
        f.eps = eps;
        const int parts = forward_loss_pair_deferred(ctx, f, train_rows + slot * kMaxFwdGrid,
                                                     val_rows + slot * kMaxFwdGrid);
        RFM_REQUIRE(parts > 0, "merged loss forward: unexpected geometry");
        train_parts = val_parts = parts;
        merged = true;
      }
      if (d_out_train_loss && !merged && !(ride && it + 1 < n_iters)) {
        // same batch, new parameters (src/fm.py:90-96), through the plan's records
        FwdArgs f{};
        f.ent = plan->ent.as<Entry>();
        f.rows = plan->rows.as<RowRec>();
        const bool rows_prepared = is_prepared && pv.E != nullptr;
        f.ell = rows_prepared ? reinterpret_cast<const char*>(pv.E) : plan->ell.as<char>();
        f.ell_stride = plan->ell_stride;
        f.ell_yp = rows_prepared ? pv.YP : plan->ell_yp.as<double2>();
        f.row_ids = rows_prepared ? nullptr : ids;
        f.n_rows = batch;
        f.w0 = d_w0;
        f.w = d_w;
        f.V = d_V;
        f.k = plan->k;
        f.eps = eps;
        if (scores_only) {
          f.out_pred = plan->sl_z.as<double>() + slot * z_per_iter;
          launch_forward(ctx, f);
        } else {
          train_parts = forward_loss_deferred(ctx, f, train_rows + slot * kMaxFwdGrid);
        }
      }
      if (d_out_val_loss && !merged && !(ride_val && it + 1 < n_iters)) {
        FwdArgs f = forward_args(d_val_indptr, d_val_indices, d_val_values, nullptr, n_val,
                                 d_w0, d_w, d_V, plan->k);
        f.eps = eps;
        if (scores_only) {
          f.out_pred = plan->sl_z.as<double>() + slot * z_per_iter + sl_a;
          launch_forward(ctx, f);
        } else {
          f.y = d_val_y;
          f.pscore = d_val_pscore;
          val_parts = forward_loss_deferred(ctx, f, val_rows + slot * kMaxFwdGrid);
        }
      }
      if (hook) {  // the evaluator's scores and their IPS-DCG@k, this iteration's parameters
        double* sc = hook->scores + (hook->slot_first + it) * hook->scores_stride;
        plan_forward(ctx, plan, hook->indptr, hook->indices, hook->values, hook->n_rows, d_w0, d_w, d_V, sc);
        const int32_t rc = rfm_val_dcg(ctx, sc, hook->seg_ptr, hook->rows, hook->labels, hook->pscores,
                                       hook->n_segments, hook->k,
                                       hook->user_scratch + (hook->slot_first + it) * hook->user_stride,
                                       hook->dcg_out + 2 * it);
        if (rc != RFM_OK) throw Error(rc, "the evaluator's DCG failed (see above)");
      }
      if (slot + 1 == run_len) {
        if (ride && it + 1 < n_iters) {  // (its last train scores arrive with the next step)
          pending_first = run_first;
          pending_count = run_len;
        } else {
          finish(run_first, run_len);
        }
        run_first = it + 1;
      }
      prepared.done(it);
    }
    finish(run_first, n_iters - run_first);
    capture.replay(n_iters);
#ifdef RFM_SLICED_STAMPS
    if (sliced.ok && stamps.p) {  // the LAST iteration's clock readings, averaged over the workgroups
      std::vector<long long> h(size_t(sliced.grid) * kSlWaves * 8);
      RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      RFM_HIP_CHECK(hipMemcpy(h.data(), stamps.p, h.size() * 8, hipMemcpyDeviceToHost));
      for (int wv : {0, kSlWaves - 1}) {
        double d[7] = {0}, n = 0;
        for (int b = 0; b < sliced.grid; ++b) {
          const long long* t = &h[(size_t(b) * kSlWaves + wv) * 8];
          if (!t[0] || !t[6]) continue;
          for (int i = 1; i < 7; ++i) d[i] += double(t[i] - t[0]);
          n += 1;
        }
        fprintf(stderr, "[sliced stamps] wave %d over %.0f workgroups (clocks since entry): fill issued+summed %.0f, barrier %.0f, prologue %.0f, row 1 %.0f, row 8 %.0f, end %.0f\n",
                wv, n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n);
      }
    }
#endif
  }
}

}  // namespace
