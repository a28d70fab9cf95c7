// The data-parallel fit() loop of one rank (rfm_fm_fit_dp).  Included by
// rfm_fm.hip (it enqueues that file's launches).  The reference has no multi-process mode;
// what is reproduced is src/fm.py:71-102 with the batch's rows dealt to the ranks:
//   dense        gradient of the shard -> all-reduce(sum) of [G_V | g_w | g_w0] -> apply
//   touched rows gradient records of the shard -> owners (all-to-all) -> rank-ordered sum +
//                update by the owner -> updated rows to everybody (all-to-all) -> store
// and the two losses as per-rank sums combined once per call.  Nothing inside the loop
// synchronises with the host or allocates: every transfer size of the call is derived from
// the row ids before the loop (plan_transfers).
#pragma once

namespace rfm {
namespace {

// contiguous shard of `total` items for `rank` (the first total % world ranks take one more)
inline void shard_of(int64_t total, int world, int rank, int64_t& lo, int64_t& hi) {
  const int64_t q = total / world, m = total % world;
  lo = rank * q + std::min<int64_t>(rank, m);
  hi = lo + q + (rank < m ? 1 : 0);
}

struct DpExchange {
  rfm_ctx* ctx;
  const rfm_transport* cb;
  int world, rank;
  static void ok(int32_t rc, const char* what) {
    if (rc != 0) fail(RFM_ERR_INTERNAL, "transport %s failed (%d)", what, rc);
  }
  void all_gather(const void* d_send, void* d_recv, int64_t bytes) {
    if (cb)
      ok(cb->all_gather(cb->user, d_send, d_recv, bytes), "all_gather");
    else
      comm_all_gather(ctx, d_send, d_recv, bytes);
  }
  void all_reduce_sum(double* d_buf, int64_t count) {
    if (cb)
      ok(cb->all_reduce_sum(cb->user, d_buf, count), "all_reduce_sum");
    else
      comm_all_reduce_sum(ctx, d_buf, count);
  }
  void all_to_all(const void* d_send, const int64_t* soff, const int64_t* sbytes, void* d_recv,
                  const int64_t* roff, const int64_t* rbytes) {
    if (cb)
      ok(cb->all_to_all(cb->user, d_send, soff, sbytes, d_recv, roff, rbytes), "all_to_all");
    else
      comm_all_to_all(ctx, rank, d_send, soff, sbytes, d_recv, roff, rbytes);
  }
};

// table / stamps of the touched-row gradients, allocated on first use; returns the next stamp
// (`ids` of them are reserved: the caller uses id .. id + ids - 1)
int32_t next_touch_ids(rfm_ctx* ctx, rfm_fm_plan* plan, int64_t ids) {
  const int64_t n = plan->n_features;
  const int k = plan->k;
  RFM_REQUIRE(ids >= 1 && ids < INT32_MAX / 2, "too many iterations in one call");
  if (!plan->row_table.p) {
    plan->row_table.alloc((size_t(n) * size_t(k + 1) + 1) * sizeof(double));
    plan->touch.alloc(size_t(n) * 4);
    plan->chunk_cnt.alloc(size_t((n + kTouchChunk - 1) / kTouchChunk) * 4);
    plan->touch_seq = 0;
  }
  if (plan->touch_seq == 0 || int64_t(plan->touch_seq) + ids >= INT32_MAX) {
    RFM_HIP_CHECK(hipMemsetAsync(plan->touch.p, 0, plan->touch.bytes, ctx->stream));
    plan->touch_seq = 0;
  }
  const int32_t first = plan->touch_seq + 1;
  plan->touch_seq += int32_t(ids);
  return first;
}

// the launches of rfm_fm_grad_rows: gradient of `batch` rows into the plan's table, stamped
// with `id`, then the stamped columns as records (count -> list -> fill)
void enqueue_grad_rows(rfm_ctx* ctx, rfm_fm_plan* plan, int32_t id, const int32_t* d_row_ids,
                       int64_t batch, const double* d_w0, const double* d_w, const double* d_V,
                       double* d_rows, int64_t cap_rows, int32_t* d_n_rows, double* d_gw0,
                       const int32_t* d_range_lo, int32_t n_ranges, int32_t* d_range_bounds) {
  const int64_t n = plan->n_features;
  const int k = plan->k;
  double* table = plan->row_table.as<double>();
  int32_t* touch = plan->touch.as<int32_t>();
  if (batch > 0) {
    enqueue_step(ctx, plan, nullptr, nullptr, nullptr, nullptr, nullptr, d_row_ids, batch,
                 const_cast<double*>(d_w0), const_cast<double*>(d_w), const_cast<double*>(d_V), 0.0,
                 table, touch, id);
  } else {  // an empty shard touches nothing
    RFM_HIP_CHECK(hipMemsetAsync(table + n * (k + 1), 0, sizeof(double), ctx->stream));
  }
  const int n_chunks = int((n + kTouchChunk - 1) / kTouchChunk);
  int32_t* chunk = plan->chunk_cnt.as<int32_t>();
  hipLaunchKernelGGL(touch_count_kernel, dim3(n_chunks), dim3(kBlock), 0, ctx->stream, touch, id, n,
                     chunk);
  hipLaunchKernelGGL(touch_list_kernel, dim3(n_chunks), dim3(kBlock), 0, ctx->stream, touch, id, n,
                     k, chunk, n_chunks, table, d_rows, cap_rows, d_n_rows, d_gw0,
                     n_ranges ? d_range_lo : nullptr, int(n_ranges), d_range_bounds);
  if (cap_rows > 0) {
    const int wpb = kBlock / kWave;
    const int64_t most = std::min<int64_t>(cap_rows, n);
    const int grid = int(std::max<int64_t>(
        1, std::min<int64_t>((most + wpb - 1) / wpb, int64_t(ctx->n_cu) * 8)));
    hipLaunchKernelGGL(rows_fill_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, table, d_rows,
                       d_n_rows, cap_rows, n, k);
  }
  RFM_HIP_CHECK(hipGetLastError());
}

// Sizes of every transfer of a call of `n_iters` iterations, from the row ids alone.
struct TransferPlan {
  int world = 1, rank = 0;
  int64_t n_iters = 0;
  std::vector<int32_t> bounds;  // [world][n_iters][world + 1]: rank s's record list cut by owner
  std::vector<int32_t> seg;     // [n_iters][world + 1]: what this rank receives, by source
  std::vector<int64_t> out_off; // [n_iters][world + 1]: the updated rows, by owner
  int64_t cap_rows = 0, cap_recv = 0, cap_all = 0;
  const int32_t* of(int s, int64_t it) const { return bounds.data() + (size_t(s) * n_iters + it) * (world + 1); }
};

void plan_transfers(rfm_ctx* ctx, rfm_fm_plan* plan, DpExchange& ex, const int32_t* d_ids,
                    int64_t global_batch, int64_t lo, int64_t hi, int64_t n_iters,
                    TransferPlan& tp) {
  const int W = ex.world, nb = W + 1;
  const int64_t n = plan->n_features;
  const Shape shp = shape_for(plan->k);
  tp.world = W;
  tp.rank = ex.rank;
  tp.n_iters = n_iters;
  std::vector<int32_t> range_lo(static_cast<size_t>(W));
  for (int r = 0; r < W; ++r) range_lo[size_t(r)] = int32_t(n * r / W);
  plan->dp_range_lo.ensure(size_t(W) * 4);
  RFM_HIP_CHECK(hipMemcpyAsync(plan->dp_range_lo.p, range_lo.data(), size_t(W) * 4,
                               hipMemcpyHostToDevice, ctx->stream));
  plan->dp_bounds.ensure(size_t(n_iters) * nb * 4);
  plan->dp_all_bounds.ensure(size_t(W) * size_t(n_iters) * nb * 4);
  plan->dp_small.ensure(8 * 4 + 64 * 4 + 16);
  int32_t* d_n_rows = plan->dp_small.as<int32_t>();  // (layout: rfm_fm_fit_dp)
  const int32_t id0 = next_touch_ids(ctx, plan, n_iters);
  const int n_chunks = int((n + kTouchChunk - 1) / kTouchChunk);
  const int64_t batch = hi - lo;
  for (int64_t it = 0; it < n_iters; ++it) {
    const int32_t id = id0 + int32_t(it);
    if (batch > 0) {
      const int64_t items = plan->ell.p ? batch * shp.lpr : batch * kWave;
      const int grid = int(std::max<int64_t>(1, std::min<int64_t>((items + kBlock - 1) / kBlock,
                                                                  int64_t(ctx->n_cu) * 16)));
      hipLaunchKernelGGL(rows_mark_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream,
                         plan->ent.as<Entry>(), plan->rows.as<RowRec>(), plan->ell.as<char>(),
                         plan->ell_stride, shp.lpr, d_ids + it * global_batch + lo, batch,
                         plan->touch.as<int32_t>(), id, plan->hot_cols.as<int32_t>(), plan->n_hot);
    }
    hipLaunchKernelGGL(touch_count_kernel, dim3(n_chunks), dim3(kBlock), 0, ctx->stream,
                       plan->touch.as<int32_t>(), id, n, plan->chunk_cnt.as<int32_t>());
    hipLaunchKernelGGL(touch_list_kernel, dim3(n_chunks), dim3(kBlock), 0, ctx->stream,
                       plan->touch.as<int32_t>(), id, n, plan->k, plan->chunk_cnt.as<int32_t>(),
                       n_chunks, plan->row_table.as<double>(), static_cast<double*>(nullptr),
                       int64_t(0), d_n_rows, static_cast<double*>(nullptr),
                       plan->dp_range_lo.as<int32_t>(), W, plan->dp_bounds.as<int32_t>() + it * nb);
  }
  RFM_HIP_CHECK(hipGetLastError());
  // every rank's bounds to every rank: the one exchange whose result the host reads
  ex.all_gather(plan->dp_bounds.p, plan->dp_all_bounds.p, n_iters * nb * 4);
  tp.bounds.resize(size_t(W) * size_t(n_iters) * nb);
  RFM_HIP_CHECK(hipMemcpyAsync(tp.bounds.data(), plan->dp_all_bounds.p, tp.bounds.size() * 4,
                               hipMemcpyDeviceToHost, ctx->stream));
  RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  // the record that carries a rank's g_w0 (column n) closes its list: owned by the last rank
  for (int s = 0; s < W; ++s)
    for (int64_t it = 0; it < n_iters; ++it) tp.bounds[(size_t(s) * n_iters + it) * nb + W] += 1;
  tp.seg.assign(size_t(n_iters) * nb, 0);
  tp.out_off.assign(size_t(n_iters) * nb, 0);
  for (int64_t it = 0; it < n_iters; ++it) {
    int32_t* seg = tp.seg.data() + it * nb;
    int64_t* off = tp.out_off.data() + it * nb;
    for (int s = 0; s < W; ++s) {
      const int32_t* b = tp.of(s, it);
      for (int r = 0; r < W; ++r) RFM_REQUIRE(b[r + 1] >= b[r] && b[0] == 0, "corrupt bounds");
      seg[s + 1] = seg[s] + (b[ex.rank + 1] - b[ex.rank]);
    }
    for (int r = 0; r < W; ++r) {
      int64_t cnt = 0;
      for (int s = 0; s < W; ++s) cnt += tp.of(s, it)[r + 1] - tp.of(s, it)[r];
      off[r + 1] = off[r] + cnt;
    }
    tp.cap_rows = std::max<int64_t>(tp.cap_rows, tp.of(ex.rank, it)[W]);
    tp.cap_recv = std::max<int64_t>(tp.cap_recv, seg[W]);
    tp.cap_all = std::max<int64_t>(tp.cap_all, off[W]);
  }
  // on the device: this rank's planned bounds (what bounds_check_kernel compares with) and the
  // segment pointers of rfm_fm_reduce_rows, for all iterations
  plan->dp_seg.ensure(size_t(n_iters) * nb * 4);
  RFM_HIP_CHECK(hipMemcpyAsync(plan->dp_bounds.p, tp.of(ex.rank, 0), size_t(n_iters) * nb * 4,
                               hipMemcpyHostToDevice, ctx->stream));
  RFM_HIP_CHECK(hipMemcpyAsync(plan->dp_seg.p, tp.seg.data(), size_t(n_iters) * nb * 4,
                               hipMemcpyHostToDevice, ctx->stream));
  const size_t wb = size_t(plan->k + 2) * 8;
  plan->dp_rows.ensure(size_t(std::max<int64_t>(tp.cap_rows, 1)) * wb);
  plan->dp_recv.ensure(size_t(std::max<int64_t>(tp.cap_recv, 1)) * wb);
  plan->dp_all.ensure(size_t(std::max<int64_t>(tp.cap_all, 1)) * wb);
  // (the host vectors uploaded above outlive the copies: tp is the caller's, range_lo was
  // consumed before the synchronisation)
}

}  // namespace
}  // namespace rfm

extern "C" int32_t rfm_fm_fit_dp(rfm_ctx* ctx, rfm_fm_plan* plan, const rfm_transport* transport,
                                 int32_t exchange, const int32_t* d_ids, int64_t global_batch,
                                 int64_t n_iters, double* d_w0, double* d_w, double* d_V,
                                 double lr, const int64_t* d_val_indptr,
                                 const int32_t* d_val_indices, const double* d_val_values,
                                 const double* d_val_y, const double* d_val_pscore, int64_t n_val,
                                 double eps, double* d_out_train_loss, double* d_out_val_loss) {
  using namespace rfm;
  return guarded([&] {
    RFM_REQUIRE(ctx && plan && d_w0 && d_w && d_V, "null pointer");
    RFM_REQUIRE(exchange == 0 || exchange == 1, "exchange=%d (0 dense, 1 touched rows)", exchange);
    RFM_REQUIRE(n_iters >= 0 && global_batch >= 1 && n_val >= 0, "bad shape");
    if (n_iters == 0) return;
    RFM_REQUIRE(d_ids, "null row ids");
    DpExchange ex{ctx, transport, 1, 0};
    if (transport) {
      RFM_REQUIRE(transport->all_gather && transport->all_reduce_sum && transport->all_to_all,
                  "transport lacks a function");
      ex.world = transport->n_ranks;
      ex.rank = transport->rank;
    } else if (ctx->comm) {
      ex.world = ctx->comm_ranks;
      ex.rank = ctx->comm_rank;
    }
    const int W = ex.world, nb = W + 1;
    RFM_REQUIRE(W >= 1 && W <= kMaxRanges - 1 && ex.rank >= 0 && ex.rank < W, "bad rank %d of %d",
                ex.rank, W);
    int64_t lo, hi, vlo, vhi;
    shard_of(global_batch, W, ex.rank, lo, hi);
    shard_of(n_val, W, ex.rank, vlo, vhi);
    const int64_t batch = hi - lo, n_my_val = vhi - vlo;
    RFM_REQUIRE(batch <= plan->max_batch, "shard of %lld rows exceeds the plan's max_batch %lld",
                (long long)batch, (long long)plan->max_batch);
    const bool want_val = d_out_val_loss && n_val > 0;
    if (want_val)
      RFM_REQUIRE(d_val_indptr && d_val_indices && d_val_values && d_val_y && d_val_pscore,
                  "validation arrays missing");
    for (int64_t it = 0; it < n_iters && batch > 0; ++it)
      validate_ids(ctx, plan, d_ids + it * global_batch + lo, batch, 1);
    const int64_t n = plan->n_features;
    const int k = plan->k;
    const int64_t count = n * int64_t(k + 1) + 1;
    const int64_t nk = n * int64_t(k);
    const size_t wb = size_t(k + 2) * 8;
    hipStream_t st = ctx->stream;

    // per-iteration loss SUMS of this rank: [train (n_iters) | val (n_iters)]
    constexpr int64_t kRun = 128;
    plan->dp_sums.ensure(size_t(2 * n_iters) * 8);
    RFM_HIP_CHECK(hipMemsetAsync(plan->dp_sums.p, 0, size_t(2 * n_iters) * 8, st));
    double* sums_train = plan->dp_sums.as<double>();
    double* sums_val = sums_train + n_iters;
    if (d_out_train_loss || want_val)
      plan->loss_rows.ensure(size_t(2 * kRun) * size_t(kMaxFwdGrid) * sizeof(double));
    double* train_rows = plan->loss_rows.as<double>();
    double* val_rows = train_rows + kRun * kMaxFwdGrid;
    int train_parts = 0, val_parts = 0;
    const auto finish = [&](int64_t first, int64_t cnt) {
      if (cnt <= 0) return;
      if (d_out_train_loss && batch > 0)
        hipLaunchKernelGGL(loss_sum_many_kernel, dim3(int(cnt)), dim3(kBlock), 0, st, train_rows,
                           int64_t(kMaxFwdGrid), train_parts, sums_train + first);
      if (want_val && n_my_val > 0)
        hipLaunchKernelGGL(loss_sum_many_kernel, dim3(int(cnt)), dim3(kBlock), 0, st, val_rows,
                           int64_t(kMaxFwdGrid), val_parts, sums_val + first);
      RFM_HIP_CHECK(hipGetLastError());
    };

    TransferPlan tp;
    // (RFM_DP_FORCE_EXCHANGE=1: a single rank goes through the exchange too -- every collective
    // with itself -- so that the whole multi-rank loop, RCCL calls included, can be run and
    // checked on one GPU)
    const bool exchange_on = W > 1 || env_int("RFM_DP_FORCE_EXCHANGE", 0) != 0;
    const bool rows_mode = exchange == 1 && exchange_on;
    int32_t id0 = 0;
    if (rows_mode) {
      plan_transfers(ctx, plan, ex, d_ids, global_batch, lo, hi, n_iters, tp);
      id0 = next_touch_ids(ctx, plan, n_iters);
    } else if (exchange_on) {
      plan->dp_grad.ensure(size_t(count) * 8);
    }
    // small device scratch: [0] record count | [1..2] error flag | [8 .. 8+64) a step's real
    // bounds | then the shard's g_w0
    plan->dp_small.ensure(8 * 4 + 64 * 4 + 16);
    int32_t* d_n_rows = plan->dp_small.as<int32_t>();
    int32_t* d_flag = d_n_rows + 1;
    int32_t* d_chk = d_n_rows + 8;
    double* d_gw0 = reinterpret_cast<double*>(plan->dp_small.as<char>() + 8 * 4 + 64 * 4);
    RFM_HIP_CHECK(hipMemsetAsync(d_flag, 0, 8, st));
    const int apply_grid = int(std::min<int64_t>((count + kBlock - 1) / kBlock, int64_t(ctx->n_cu) * 16));
    std::vector<int64_t> soff(static_cast<size_t>(W)), sbytes(static_cast<size_t>(W)),
        roff(static_cast<size_t>(W)), rbytes(static_cast<size_t>(W));

    int64_t run_first = 0;
    for (int64_t it = 0; it < n_iters; ++it) {
      const int32_t* ids = d_ids + it * global_batch + lo;
      if (!exchange_on) {
        enqueue_step(ctx, plan, nullptr, nullptr, nullptr, nullptr, nullptr, ids, batch, d_w0, d_w,
                     d_V, lr, nullptr);
      } else if (!rows_mode) {
        double* grad = plan->dp_grad.as<double>();
        if (batch > 0)
          enqueue_step(ctx, plan, nullptr, nullptr, nullptr, nullptr, nullptr, ids, batch, d_w0, d_w,
                       d_V, 0.0, grad);
        else
          RFM_HIP_CHECK(hipMemsetAsync(grad, 0, size_t(count) * 8, st));
        ex.all_reduce_sum(grad, count);
        hipLaunchKernelGGL(fm_apply_kernel, dim3(apply_grid), dim3(kBlock), 0, st, d_V, d_w, d_w0,
                           grad, nk, n, lr);
      } else {
        const int32_t* mine = tp.of(ex.rank, it);
        const int32_t* seg = tp.seg.data() + it * nb;
        const int64_t* off = tp.out_off.data() + it * nb;
        double* rows = plan->dp_rows.as<double>();
        enqueue_grad_rows(ctx, plan, id0 + int32_t(it), ids, batch, d_w0, d_w, d_V, rows, tp.cap_rows,
                          d_n_rows, d_gw0, plan->dp_range_lo.as<int32_t>(), W, d_chk);
        hipLaunchKernelGGL(rows_append_w0_kernel, dim3(1), dim3(kWave), 0, st, rows, d_n_rows,
                           tp.cap_rows, d_gw0, n, k);
        hipLaunchKernelGGL(bounds_check_kernel, dim3(1), dim3(kWave * 2), 0, st, d_chk,
                           plan->dp_bounds.as<int32_t>() + it * nb, W, tp.cap_rows, int32_t(it), d_flag);
        // records to their owners
        for (int p = 0; p < W; ++p) {
          soff[size_t(p)] = int64_t(mine[p]) * int64_t(wb);
          sbytes[size_t(p)] = int64_t(mine[p + 1] - mine[p]) * int64_t(wb);
          roff[size_t(p)] = int64_t(seg[p]) * int64_t(wb);
          rbytes[size_t(p)] = int64_t(seg[p + 1] - seg[p]) * int64_t(wb);
        }
        ex.all_to_all(rows, soff.data(), sbytes.data(), plan->dp_recv.p, roff.data(), rbytes.data());
        // the owner's ordered sums and updated rows, written where they sit in the list of all
        double* all = plan->dp_all.as<double>();
        double* out = all + off[ex.rank] * (k + 2);
        const int64_t total = seg[W];
        if (total > 0) {
          const int wpb = kBlock / kWave;
          const int grid = int(std::min<int64_t>((total + wpb - 1) / wpb, int64_t(ctx->n_cu) * 8));
          hipLaunchKernelGGL(rows_reduce_kernel, dim3(grid), dim3(kBlock), 0, st,
                             plan->dp_recv.as<double>(), plan->dp_seg.as<int32_t>() + it * nb, W, d_w,
                             d_V, n, k, lr, out, d_w0);
        }
        // every owner's updated rows to everybody
        for (int p = 0; p < W; ++p) {
          soff[size_t(p)] = off[ex.rank] * int64_t(wb);
          sbytes[size_t(p)] = (off[ex.rank + 1] - off[ex.rank]) * int64_t(wb);
          roff[size_t(p)] = off[p] * int64_t(wb);
          rbytes[size_t(p)] = (off[p + 1] - off[p]) * int64_t(wb);
        }
        ex.all_to_all(all, soff.data(), sbytes.data(), all, roff.data(), rbytes.data());
        if (off[W] > 0) {
          const int wpb = kBlock / kWave;
          const int grid = int(std::max<int64_t>(
              1, std::min<int64_t>((off[W] + wpb - 1) / wpb, int64_t(ctx->n_cu) * 8)));
          hipLaunchKernelGGL(rows_set_kernel, dim3(grid), dim3(kBlock), 0, st, all, off[W],
                             static_cast<const double*>(nullptr), 0, int64_t(1), d_w0, d_w, d_V, n, k,
                             lr, true);
        }
      }
      RFM_HIP_CHECK(hipGetLastError());
      const int64_t slot = it - run_first;
      if (d_out_train_loss && batch > 0) {
        // the shard's part of the train loss: same batch, new parameters (src/fm.py:90-96)
        FwdArgs f{};
        f.ent = plan->ent.as<Entry>();
        f.rows = plan->rows.as<RowRec>();
        f.ell = plan->ell.as<char>();
        f.ell_stride = plan->ell_stride;
        f.ell_yp = plan->ell_yp.as<double2>();
        f.row_ids = ids;
        f.n_rows = batch;
        f.w0 = d_w0;
        f.w = d_w;
        f.V = d_V;
        f.k = k;
        f.eps = eps;
        train_parts = forward_loss_deferred(ctx, f, train_rows + slot * kMaxFwdGrid);
      }
      if (want_val && n_my_val > 0) {
        FwdArgs f = forward_args(d_val_indptr + vlo, d_val_indices, d_val_values, nullptr, n_my_val,
                                 d_w0, d_w, d_V, k);
        f.y = d_val_y + vlo;
        f.pscore = d_val_pscore + vlo;
        f.eps = eps;
        val_parts = forward_loss_deferred(ctx, f, val_rows + slot * kMaxFwdGrid);
      }
      if (slot + 1 == kRun) {
        finish(run_first, kRun);
        run_first = it + 1;
      }
    }
    finish(run_first, n_iters - run_first);
    // the ranks' sums -> the losses, the same on every rank
    if (exchange_on && (d_out_train_loss || want_val)) ex.all_reduce_sum(plan->dp_sums.as<double>(), 2 * n_iters);
    const int sgrid = int((n_iters + kBlock - 1) / kBlock);
    if (d_out_train_loss)
      hipLaunchKernelGGL(loss_scale_kernel, dim3(sgrid), dim3(kBlock), 0, st, sums_train, n_iters,
                         double(global_batch), d_out_train_loss);
    if (want_val)
      hipLaunchKernelGGL(loss_scale_kernel, dim3(sgrid), dim3(kBlock), 0, st, sums_val, n_iters,
                         double(n_val), d_out_val_loss);
    RFM_HIP_CHECK(hipGetLastError());
    int32_t flag[2] = {0, 0};
    RFM_HIP_CHECK(hipMemcpyAsync(flag, d_flag, 8, hipMemcpyDeviceToHost, st));
    RFM_HIP_CHECK(hipStreamSynchronize(st));
    if (flag[0] != 0)
      fail(RFM_ERR_INTERNAL,
           "iteration %d: the gradient records do not match the transfer plan derived from the row ids",
           flag[0] - 1);
  });
}
