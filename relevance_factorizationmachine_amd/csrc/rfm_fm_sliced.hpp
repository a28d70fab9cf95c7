// Scoring forwards for factor counts of several chunks per lane (even k > 128: every published
// run of the reference is k = 400 or 300), sliced by FACTORS (gfx950): the loss forwards of
// rfm_fm_train and the scores of rfm_fm_plan_forward.
//
// Reference arithmetic: src/fm.py:114-133 (predict), src/base.py:37-66 (loss, clipped sigmoid).
//
// The pair term of the logit is a sum over factors, 0.5 * sum_f [(sum_e v_ef x_e)^2 -
// sum_e (v_ef x_e)^2], so a slice of the factors contributes an independent partial logit.  A
// workgroup owns ONE slice (one slice per 256 factors -- a lane holds four; the slices are dealt
// to the XCDs so that an XCD's L2 only ever holds its slice of V) and a block of rows, and keeps
// the slice of the log's most frequent columns (side features, dense reals, popular items: 12 of
// a row's 15 entries on the KuaiRec shape) in LDS: at k = 400 the plain forward is bound by
// gathering 15 x 3 200 B per row from L2 (14 308 validation rows: 687 MB, 41 us = the guide's L2
// gather rate); here only the rare columns are gathered.  For a cached column the per-entry sum
// of squares is x^2 |v_slice|^2 with the norm computed once per workgroup, so a cached entry
// costs one FMA per factor.
//
// The logs are read in a TRANSLATED form (sl_translate_kernel: the training log once per plan,
// any other log once per call or once per registration, rfm_fm_plan_register_log): rows of
// 2^ml_log2 records {LDS offset of the
// column's cached slice, column, value} at a fixed stride, the cached columns' entries FIRST.  A
// wavefront works on one row at a time, entry e in lane e: "entry j" is then a v_readlane with a
// CONSTANT lane, the cached entries are lanes 0 .. n - 1, and an entry that is not cached points
// at a row of zeros in LDS -- no per-entry scalar work (bit scans, selects) is left in the loop.
// (The first form did that work per entry: 50 us, bound by the CU's one scalar unit.)  The rows of
// a wavefront are a three-stage pipeline: records of row j + 2 requested, gathers of row j + 1's
// uncached entries (and its w values and norms) requested, row j summed.  Rows longer than the
// stride are marked and read straight from the caller's CSR arrays (slow, correct).
//
// Output: zpart[slice][row]; slice 0 carries w0 + <w, x>.  The scores, the logarithms and the
// sums over rows are left to loss_from_slices_kernel, once per RUN of iterations (or to
// scores_from_slices_kernel).  What bounds it, measured: VALU issue.
#pragma once

#include "rfm_fm_kernels.hpp"

namespace rfm {

constexpr int kSlBlock = 1024;
constexpr int kSlWaves = kSlBlock / kWave;
constexpr int kSlPairs = 2;    // pairs of factors per lane: a wavefront covers 256 factors of a row
constexpr int kSlFill = 8;     // cached columns a wavefront requests together when it fills the LDS copy
constexpr int kSlGlobals = 2;  // gathers of a row's uncached entries requested a row ahead
#ifndef RFM_SL_BATCH
#define RFM_SL_BATCH 4
#endif
constexpr int kSlBatch = RFM_SL_BATCH;  // cached entries whose LDS reads are in flight together

struct SlicedArgs {
  // log A: rows row_ids[t] (or t) for t < n_a; log B: rows t - n_a for n_a <= t < n_a + n_b;
  // translated records (tr_*), and the caller's arrays for the rows marked kSlLong
  const SlEnt* tr_a;
  const int64_t* indptr_a;
  const int32_t* indices_a;
  const double* values_a;
  const int32_t* row_ids;
  int64_t n_a;
  const SlEnt* tr_b;
  const SlEnt* pad;            // one record {zero row, kSlPad, 0}
  const int64_t* indptr_b;
  const int32_t* indices_b;
  const double* values_b;
  int64_t n_b;
  const double* w0;
  const double* w;
  const double* V;
  int32_t k;
  int32_t ns, sw;              // slices, factors per slice (even; the last slice may be narrower)
  int32_t n_cached;            // columns whose slice a workgroup keeps in LDS
  const int32_t* cached_cols;  // [n_cached]
  const int32_t* cached_rank;  // [n_features]: rank in cached_cols, or -1
  int32_t ml_log2;             // records per translated row: 2^ml_log2 (16, 32 or 64)
  int32_t rows_per_wg;
  double* zpart;               // [ns][n_a + n_b]
  long long* stamps;           // -DRFM_SLICED_STAMPS builds only: [workgroup][wave][8] clock readings
};

inline size_t sliced_lds_bytes(int n_cached, int sw) {
  return size_t(n_cached + 1) * size_t(sliced_row_bytes(sw)) + kSlicedLdsSlack;
}

__device__ inline double readlane_f64(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, valid in LANE 63 only, without a trip through LDS: row sums by DPP
// mirrors, then the rows chained by row_bcast15 / row_bcast31 (GFX9 DPP).
template <int CTRL, int ROWS>
__device__ inline double dpp_move_rows(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum_lane63(double v) {
  v = group_sum<16>(v);
  v += dpp_move_rows<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_move_rows<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return v;
}

// what is requested a row ahead: the first kSlGlobals uncached entries' slices of V (+ their x),
// the counts of cached and uncached entries, and -- lane e, for entry e -- w[col] and x^2 |v|^2
struct SlAhead {
  double2 gv[kSlGlobals][kSlPairs];
  double gx[kSlGlobals];
  int nh, ng;
  double wl, hn;
  bool lng;
};

__global__ __launch_bounds__(kSlBlock) void fm_logit_slices_kernel(SlicedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char sl_lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // slice s = the XCDs [8 s / ns, 8 (s + 1) / ns): workgroups go to the XCDs round robin by id
  const int xs = 8 / a.ns;
  const int xcd = blockIdx.x & 7;
  const int slice = xcd / xs;
  const int64_t wi = int64_t(blockIdx.x >> 3) * xs + (xcd % xs);
  const int64_t NR = a.n_a + a.n_b;
  const int64_t row_begin = wi * a.rows_per_wg;
  const int64_t row_end = row_begin + a.rows_per_wg < NR ? row_begin + a.rows_per_wg : NR;
  if (row_begin >= row_end) return;  // (the whole workgroup)
  const int k = a.k, SW = a.sw, H = a.n_cached, ML = 1 << a.ml_log2;
  const int RS = sliced_row_bytes(SW);
  const int ZOFF = H * RS;  // the row of zeros: what an entry that is not cached reads
  const int f0 = slice * SW;
  const int width = SW < k - f0 ? SW : k - f0;
  // lane l holds the pairs of factors l and 64 + l of the slice (16-byte loads: an instruction
  // reads 1 KiB of consecutive bytes -- with the pairs 2 l and 2 l + 1 instead, the lanes of a
  // ds_read_b128 sat 32 bytes apart and half of the LDS cycles were bank conflicts); an idle pair
  // re-reads the slice's first pair and is dropped from the sums
  bool pair_ok[kSlPairs];
  int loff[kSlPairs];  // the pair's first factor
#pragma unroll
  for (int p = 0; p < kSlPairs; ++p) {
    pair_ok[p] = 2 * (kWave * p + lane) < width;
    loff[p] = pair_ok[p] ? 2 * (kWave * p + lane) : 0;
  }
  const double* Vs = a.V + f0;  // the slice of row 0
  const double w0 = slice == 0 ? a.w0[0] : 0.0;
#ifdef RFM_SLICED_STAMPS
#define RFM_STAMP(i) do { if (a.stamps && lane == 0) a.stamps[(int64_t(blockIdx.x) * kSlWaves + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define RFM_STAMP(i) do { } while (0)
#endif
  RFM_STAMP(0);

  // the rows of this wavefront: row_begin + wave + 16 j
  const int64_t rows_wg = row_end - row_begin;
  const int my_rows = wave < rows_wg ? int((rows_wg - wave + kSlWaves - 1) / kSlWaves) : 0;

  // A block of up to 64 of them: lane j holds the row of its log that the wavefront's j-th row of
  // the block is.  The first block's ids and its first two rows' records are requested BEFORE the
  // workgroup fills its LDS copy (they do not depend on it).
  int jb = 0, nj = 0, idv = 0;
  const auto block = [&]() {
    nj = my_rows - jb < kWave ? my_rows - jb : kWave;
    const int64_t tl = row_begin + wave + int64_t(kSlWaves) * (jb + lane);
    idv = 0;
    if (lane < nj) idv = tl < a.n_a ? (a.row_ids ? a.row_ids[tl] : int(tl)) : int(tl - a.n_a);
  };
  // (lanes past the row's records, and rows past the block, read ONE padding record instead: a
  // select of the ADDRESS -- a select of the loaded value would let the compiler put the load
  // under a branch, and a load under a branch ends the pipeline)
  const auto records = [&](int j, int& off, int& col, double& x) {
    const int jj = j < nj ? j : nj - 1;
    const int64_t t = row_begin + wave + int64_t(kSlWaves) * (jb + jj);
    const int r = __builtin_amdgcn_readlane(idv, jj);
    const SlEnt* row = (t < a.n_a ? a.tr_a : a.tr_b) + (int64_t(r) << a.ml_log2);
    const SlEnt e = *(j < nj && lane < ML ? row + lane : a.pad);
    off = e.off;
    col = e.col;
    x = e.x;
  };
  int o0 = 0, k0 = kSlPad, o1 = 0, k1 = kSlPad;
  double x0 = 0.0, x1 = 0.0;
  if (my_rows > 0) {
    block();
    records(0, o0, k0, x0);
    records(1, o1, k1, x1);
  }

  // the cached columns' slices [H + 1][SW | norm | -]: a wavefront per column, kSlFill columns'
  // loads in flight (no load under a branch, see `request` below: past the last column,
  // column 0 again, not stored); row H is zero
  for (int i = tid; i < RS / 8; i += kSlBlock) reinterpret_cast<double*>(sl_lds + ZOFF)[i] = 0.0;
  for (int hb = 0; hb < H; hb += kSlWaves * kSlFill) {
    int col[kSlFill];
    double2 v[kSlFill][kSlPairs];
#pragma unroll
    for (int u = 0; u < kSlFill; ++u) {
      const int h = hb + u * kSlWaves + wave;
      col[u] = a.cached_cols[h < H ? h : 0];
    }
#pragma unroll
    for (int u = 0; u < kSlFill; ++u)
#pragma unroll
      for (int p = 0; p < kSlPairs; ++p)
        v[u][p] = *reinterpret_cast<const double2*>(Vs + int64_t(col[u]) * k + loff[p]);
#pragma unroll
    for (int u = 0; u < kSlFill; ++u) {
      const int h = hb + u * kSlWaves + wave;
      double n = 0.0;
#pragma unroll
      for (int p = 0; p < kSlPairs; ++p) {
        if (!pair_ok[p]) v[u][p] = make_double2(0.0, 0.0);
        n += v[u][p].x * v[u][p].x + v[u][p].y * v[u][p].y;
      }
      n = wave_sum_lane63(n);
      if (h < H) {  // (uniform in the wavefront)
#pragma unroll
        for (int p = 0; p < kSlPairs; ++p)
          if (2 * (kWave * p + lane) < SW)
            *reinterpret_cast<double2*>(sl_lds + h * RS + (kWave * p + lane) * 16) = v[u][p];
        if (lane == kWave - 1) *reinterpret_cast<double*>(sl_lds + h * RS + SW * 8) = n;
      }
    }
  }
  RFM_STAMP(1);
  __syncthreads();
  RFM_STAMP(2);

  double q[2 * kSlPairs], s2[kSlPairs];  // (s2 per pair: an idle pair's sums are dropped whole)
  // Every load of the pipeline is UNCONDITIONAL and its value is always used: with a load under
  // a branch the compiler cannot count how many newer requests may be outstanding when an older
  // one is consumed, and waits for all of them -- no pipeline.  An absent uncached entry reads the
  // row's FIRST entry's column again with a multiplier of zero (a column the row holds: a
  // non-finite row of V still reaches only the rows that hold its column); a row without entries
  // reads column 0 and has its sums dropped at the end.
  const auto request = [&](int off, int col, double x, SlAhead& g) {
    g.lng = __builtin_amdgcn_readfirstlane(col) == kSlLong;
    g.nh = __popcll(__ballot(off != ZOFF));
    g.ng = __popcll(__ballot(off == ZOFF && col >= 0));
    const bool any = g.nh + g.ng > 0;
    const int cf = any ? __builtin_amdgcn_readlane(col, 0) : 0;  // (records are packed from lane 0)
    g.wl = a.w[col >= 0 ? col : cf];
    g.hn = x * x * *reinterpret_cast<const double*>(sl_lds + off + SW * 8);  // (zero row: 0)
#pragma unroll
    for (int n = 0; n < kSlGlobals; ++n) {
      const bool has = n < g.ng;
      const int j = has ? g.nh + n : 0;
      const int c = any ? __builtin_amdgcn_readlane(col, j) : 0;
      const double xj = readlane_f64(x, j);
      g.gx[n] = has ? xj : 0.0;
#pragma unroll
      for (int p = 0; p < kSlPairs; ++p)
        g.gv[n][p] = *reinterpret_cast<const double2*>(Vs + int64_t(c) * k + loff[p]);
    }
  };
  // the sums of one row (entry e in lane e) whose requests `g` were made earlier
  const auto add_gathered = [&](const double2 (&v)[kSlPairs], double xj) {
#pragma unroll
    for (int p = 0; p < kSlPairs; ++p) {
      const double t0 = v[p].x * xj, t1 = v[p].y * xj;
      q[2 * p] += t0;
      q[2 * p + 1] += t1;
      s2[p] = fma(t0, t0, s2[p]);
      s2[p] = fma(t1, t1, s2[p]);
    }
  };
  const auto consume = [&](int off, int col, double x, const SlAhead& g) {
    // cached entries: lanes 0 .. nh - 1, whole batches and then the last one to three.  (The LDS
    // reads are NOT clamped for idle pairs -- pair p sits 1 KiB after pair 0, one address register
    // and an immediate offset; what an idle pair reads is dropped by pair_ok, and the LDS
    // allocation ends 2 KiB after the row of zeros)
#pragma unroll
    for (int b = 0; b < kWave; b += kSlBatch) {
      if (b < g.nh) {  // (uniform)
        const int live = g.nh - b;  // entries of this batch
        double2 hv[kSlBatch][kSlPairs];
#pragma unroll
        for (int u = 0; u < kSlBatch; ++u) {
          if (u == 0 || u < live) {
            const char* row = sl_lds + __builtin_amdgcn_readlane(off, b + u);
#pragma unroll
            for (int p = 0; p < kSlPairs; ++p) hv[u][p] = *reinterpret_cast<const double2*>(row + lane * 16 + p * (kWave * 16));
          }
        }
#pragma unroll
        for (int u = 0; u < kSlBatch; ++u) {
          if (u == 0 || u < live) {
            const double xj = readlane_f64(x, b + u);
#pragma unroll
            for (int p = 0; p < kSlPairs; ++p) {
              q[2 * p] = fma(hv[u][p].x, xj, q[2 * p]);
              q[2 * p + 1] = fma(hv[u][p].y, xj, q[2 * p + 1]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < kSlGlobals; ++n) add_gathered(g.gv[n], g.gx[n]);  // (absent: x = 0)
    for (int n = kSlGlobals; n < g.ng; ++n) {  // (more uncached entries than gathers requested ahead)
      const int c = __builtin_amdgcn_readlane(col, g.nh + n);
      const double xj = readlane_f64(x, g.nh + n);
      double2 v[kSlPairs];
#pragma unroll
      for (int p = 0; p < kSlPairs; ++p) v[p] = *reinterpret_cast<const double2*>(Vs + int64_t(c) * k + loff[p]);
      add_gathered(v, xj);
    }
  };
  // pair term of the lane's factors, idle pairs dropped
  const auto pair_term = [&]() {
    double pl = 0.0;
#pragma unroll
    for (int p = 0; p < kSlPairs; ++p)
      if (pair_ok[p]) pl += q[2 * p] * q[2 * p] + q[2 * p + 1] * q[2 * p + 1] - s2[p];
    return pl;
  };

  while (jb < my_rows) {
    int o2, k2;
    double x2;
    SlAhead g0, g1;
    request(o0, k0, x0, g0);
    RFM_STAMP(3);
    for (int j = 0; j < nj; ++j) {
      if (j == 1) RFM_STAMP(4);
      if (j == 8) RFM_STAMP(5);
      records(j + 2, o2, k2, x2);
      request(o1, k1, x1, g1);
      const int64_t t = row_begin + wave + int64_t(kSlWaves) * (jb + j);
#pragma unroll
      for (int i = 0; i < 2 * kSlPairs; ++i) q[i] = 0.0;
#pragma unroll
      for (int p = 0; p < kSlPairs; ++p) s2[p] = 0.0;
      double val;
      if (!g0.lng) {
        consume(o0, k0, x0, g0);
        val = 0.5 * (pair_term() - g0.hn) + (slice == 0 ? g0.wl * x0 : 0.0);
        if (g0.nh + g0.ng == 0) val = 0.0;
      } else {  // longer than the stride: straight from the log, an entry at a time
        const bool second = t >= a.n_a;
        const int64_t r = __builtin_amdgcn_readlane(idv, j);
        const int64_t* ip = second ? a.indptr_b : a.indptr_a;
        const int32_t* idx = second ? a.indices_b : a.indices_a;
        const double* vals = second ? a.values_b : a.values_a;
        const int64_t b0 = ip[r], len = ip[r + 1] - b0;
        double hn = 0.0, linacc = 0.0;
        for (int64_t pb = 0; pb < len; pb += kWave) {
          int col = 0, rk = -1;
          double x = 0.0;
          if (pb + lane < len) {
            col = idx[b0 + pb + lane];
            x = vals[b0 + pb + lane];
            rk = a.cached_rank[col];
            if (slice == 0) linacc += a.w[col] * x;
            if (rk >= 0) hn += x * x * *reinterpret_cast<const double*>(sl_lds + rk * RS + SW * 8);
          }
          const int cnt = int(len - pb < kWave ? len - pb : kWave);
          for (int e = 0; e < cnt; ++e) {
            const int re = __builtin_amdgcn_readlane(rk, e);
            const double xe = readlane_f64(x, e);
            double2 v[kSlPairs];
            if (re >= 0) {
#pragma unroll
              for (int p = 0; p < kSlPairs; ++p) {
                v[p] = *reinterpret_cast<const double2*>(sl_lds + re * RS + loff[p] * 8);
                q[2 * p] = fma(v[p].x, xe, q[2 * p]);
                q[2 * p + 1] = fma(v[p].y, xe, q[2 * p + 1]);
              }
            } else {
              const int64_t c = __builtin_amdgcn_readlane(col, e);
#pragma unroll
              for (int p = 0; p < kSlPairs; ++p) v[p] = *reinterpret_cast<const double2*>(Vs + c * k + loff[p]);
              add_gathered(v, xe);
            }
          }
        }
        val = 0.5 * (pair_term() - hn) + linacc;
      }
      const double z = wave_sum_lane63(val) + w0;
      if (lane == kWave - 1) a.zpart[int64_t(slice) * NR + t] = z;
      o0 = o1; k0 = k1; x0 = x1;
      o1 = o2; k1 = k2; x1 = x2;
      g0 = g1;
    }
    RFM_STAMP(6);
    jb += kWave;
    if (jb < my_rows) {
      block();
      records(0, o0, k0, x0);
      records(1, o1, k1, x1);
    }
  }
}

// The loss terms of a RUN of iterations from their partial logits (ns >= 1 slices) or from their
// scores (ns = 0: the plain forwards' out_pred): block (x, y) sums a share
// of the rows [seg_first, seg_first + seg_rows) of iteration y in a fixed order;
// loss_finish_many_kernel adds the shares.  ids (per iteration, ids_stride apart): the rows'
// positions in y / p, or null for row t.
__global__ __launch_bounds__(kBlock) void loss_from_slices_kernel(
    const double* zbuf, int64_t it_stride, int ns, int64_t nr, int64_t seg_first, int64_t seg_rows,
    const int32_t* ids, int64_t ids_stride, const double* y, const double* p, double eps,
    double* partial, int64_t partial_stride) {
  __shared__ double lds[kBlock];
  const double* z = zbuf + int64_t(blockIdx.y) * it_stride + seg_first;
  const int32_t* id = ids ? ids + int64_t(blockIdx.y) * ids_stride : nullptr;
  double acc = 0.0;
  for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < seg_rows;
       t += int64_t(gridDim.x) * kBlock) {
    double zz = z[t];
    for (int s = 1; s < ns; ++s) zz += z[int64_t(s) * nr + t];
    const int64_t r = id ? int64_t(id[t]) : t;
    acc += logloss_term(y[r], p[r], ns > 0 ? sigmoid_clipped(zz) : zz, eps);  // (ns = 0: z holds scores)
  }
  const double s = block_sum<kBlock>(acc, lds);
  if (threadIdx.x == 0) partial[int64_t(blockIdx.y) * partial_stride + blockIdx.x] = s;
}

// scores of rows from their partial logits (rfm_fm_plan_forward)
__global__ __launch_bounds__(kBlock) void scores_from_slices_kernel(const double* z, int ns, int64_t nr,
                                                                  double* out) {
  for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < nr; t += int64_t(gridDim.x) * kBlock) {
    double zz = z[t];
    for (int s = 1; s < ns; ++s) zz += z[int64_t(s) * nr + t];
    out[t] = sigmoid_clipped(zz);
  }
}

}  // namespace rfm
