// FM kernels for gfx950 (MI355X) and their C-ABI launchers.
//
// Layout in HBM (all float64 unless noted):
//   CSR of the log:   indptr int64[N+1], indices int32[nnz], values f64[nnz]
//   parameters:       w0[1], w[n], V[n][k] row-major (the reference's NumPy layout)
//   per-step scratch: Q[B][k] (= X_b V_old), err[B] (= y/p - sigmoid(logit))
//   training plan:    columns are split in two classes when the plan is built
//     HOT columns     (expected entries per batch >= hot_min_count, at most what
//                     fits the LDS budget; side features, dense reals, top items):
//                     slot_of[p] = -1 - hot_rank.  Their gradient is accumulated
//                     on chip: every forward workgroup keeps [H][k+2] sums in LDS
//                     (ds_add_f64) and stores ONE slab at its end.
//     SPARSE columns  (one-hot users / items ...): a column-major ("slot") view
//                     of the whole training CSR restricted to these columns
//                     slot_of int32[nnz]  CSR entry -> slot (column-major rank)
//                     csc_x f64, csc_col int32   value / column of a slot
//                     slot_t int32        batch position of the slot's row, or -1
//
// One training step (rfm_fm_step) is three launches on one stream:
//   1. fm_forward_kernel   rows of the batch in parallel: q_t = V^T x_t, logit,
//                          residual; writes Q, err, marks slot_t for the sparse
//                          entries (plain stores) and adds the hot entries'
//                          err_t x_tj [q_t, 1, x_tj] into the LDS sums.
//   2. fm_consume_kernel   waves own disjoint slot windows (whole columns, or a
//                          chunk of a long column); a wave scans its window,
//                          accumulates sum_t err_t x_tj Q[t,:] over the marked
//                          slots IN SLOT ORDER and updates V[j,:], w[j] in
//                          place (or writes a chunk partial).  Resets slot_t.
//   3. fm_finalize_kernel  hot columns: slabs summed in block order and applied;
//                          long sparse columns: partials summed in chunk order;
//                          w0 from a fixed-order sum of err.
// No global float atomics.  Sparse-class sums have a fixed order (bitwise
// reproducible); hot-class sums inside one workgroup are LDS atomics, so their
// last bits may vary from run to run (hot_min_count < 0 turns the class off).
// Reference arithmetic: src/fm.py:80-88,114-187 (see rfm_hip.h).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <numeric>

#include "rfm_common.h"

namespace rfm {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr double kLogitClip = 700.0;      // src/base.py:65
constexpr size_t kHotLdsBudget = 64 << 10;  // bytes of LDS a forward workgroup spends on hot sums
constexpr int kMaxHot = 1024;

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
template <int LPR>
__device__ inline double group_sum(double v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LPR);
  return v;
}

// fixed-order block sum (tree over LDS); every thread gets the total
template <int BLOCK>
__device__ inline double block_sum(double v, double* lds) {
  const int tid = threadIdx.x;
  __syncthreads();
  lds[tid] = v;
  __syncthreads();
#pragma unroll
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) lds[tid] += lds[tid + s];
    __syncthreads();
  }
  return lds[0];
}

__device__ inline double sigmoid_clipped(double z) {
  z = fmin(fmax(z, -kLogitClip), kLogitClip);
  return 1.0 / (1.0 + exp(-z));
}

__device__ inline double logloss_term(double y, double p, double pred, double eps) {
  const double r = y / p;
  return r * log(pred + eps) + (1.0 - r) * log(1.0 - pred + eps);
}

template <int VEC>
struct Pack;
template <>
struct Pack<1> {
  double v[1];
  __device__ inline void load(const double* p) { v[0] = *p; }
  __device__ inline void store(double* p) const { *p = v[0]; }
};
template <>
struct Pack<2> {
  double v[2];
  __device__ inline void load(const double* p) {
    const double2 t = *reinterpret_cast<const double2*>(p);
    v[0] = t.x;
    v[1] = t.y;
  }
  __device__ inline void store(double* p) const {
    *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]);
  }
};

// ---------------------------------------------------------------------------
// 1. forward (+ residual, Q, slot marks, hot sums, loss partials)
// ---------------------------------------------------------------------------
struct FwdArgs {
  const int64_t* indptr;
  const int32_t* indices;
  const double* values;
  const int32_t* row_ids;  // may be null: row t
  int64_t n_rows;
  const double* w0;
  const double* w;
  const double* V;
  int32_t k;
  const double* y;       // needed for err / loss
  const double* pscore;  // needed for err / loss
  double* out_pred;      // nullable
  double* out_err;       // nullable
  double* out_Q;         // nullable [n_rows][k]
  const int32_t* slot_of;  // nullable: >=0 mark that slot, <0 hot column -1-slot_of
  int32_t* slot_t;
  int32_t n_hot;         // hot columns (training step only)
  double* hot_slab;      // [gridDim.x][n_hot][k+2]
  double* loss_partial;  // nullable: [gridDim.x]
  double eps;
  int32_t ablate;  // -DRFM_ABLATE builds only: bit mask of parts to skip (timing experiments)
};

#ifdef RFM_ABLATE
#define RFM_KEEP(a, bit) (((a).ablate & (bit)) == 0)
#else
#define RFM_KEEP(a, bit) true
#endif
// bits: 1 slot marks, 2 Q store, 4 V gathers, 8 hot LDS adds, 16 slab store, 32 hot pass entirely

// A row is handled by LPR consecutive lanes; lane l holds factors
// (c*LPR + l)*VEC .. +VEC-1 for c < NC.  k=32 -> LPR=16, VEC=2: one 16-byte
// load per lane covers a 256-byte row of V, four rows per wave.
template <int LPR, int VEC, int NC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fm_forward_kernel(FwdArgs a) {
  constexpr int GPB = BLOCK / LPR;  // row groups per block
  extern __shared__ double dyn_lds[];  // [BLOCK] reduction scratch, then [n_hot][k+2] hot sums
  double* red = dyn_lds;
  double* hot = dyn_lds + BLOCK;
  const int tid = threadIdx.x;
  const int l = tid % LPR;
  const int g = tid / LPR;
  const int k = a.k;
  const int hot_w = k + 2;
  const double w0 = a.w0[0];
  double loss_acc = 0.0;

  if (a.n_hot > 0) {
    for (int i = tid; i < a.n_hot * hot_w; i += BLOCK) hot[i] = 0.0;
    __syncthreads();
  }

  for (int64_t base = int64_t(blockIdx.x) * GPB; base < a.n_rows;
       base += int64_t(gridDim.x) * GPB) {
    const int64_t t = base + g;
    const bool valid = t < a.n_rows;
    int64_t r = 0, p0 = 0, p1 = 0;
    if (valid) {
      r = a.row_ids ? int64_t(a.row_ids[t]) : t;
      p0 = a.indptr[r];
      p1 = a.indptr[r + 1];
    }
    double q[NC][VEC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) q[c][v] = 0.0;
    double s2 = 0.0, lin = 0.0;

    for (int64_t pb = p0; pb < p1; pb += LPR) {
      const int64_t my = pb + l;
      int32_t col = 0;
      double x = 0.0;
      if (my < p1) {
        col = a.indices[my];
        x = a.values[my];
        lin += a.w[col] * x;
        if (a.slot_of) {
          const int32_t so = a.slot_of[my];
          if (so >= 0 && RFM_KEEP(a, 1)) a.slot_t[so] = int32_t(t);
        }
      }
      const int cnt = (p1 - pb) < int64_t(LPR) ? int(p1 - pb) : LPR;
      for (int j = 0; j < cnt; ++j) {
        const int32_t cj = __shfl(col, j, LPR);
        const double xj = __shfl(x, j, LPR);
        const double* vrow = a.V + int64_t(cj) * k;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int f = (c * LPR + l) * VEC;
          if (f < k) {
            Pack<VEC> pv;
            if (RFM_KEEP(a, 4)) pv.load(vrow + f);
            else pv.load(a.V + f);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              const double vx = pv.v[v] * xj;
              q[c][v] += vx;
              s2 += vx * vx;
            }
          }
        }
      }
    }

    double pair = -s2;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) pair += q[c][v] * q[c][v];
    pair = group_sum<LPR>(pair);
    lin = group_sum<LPR>(lin);

    if (valid) {
      const double pred = sigmoid_clipped(w0 + lin + 0.5 * pair);
      if (a.out_Q && RFM_KEEP(a, 2)) {
        double* qrow = a.out_Q + t * k;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int f = (c * LPR + l) * VEC;
          if (f < k) {
            Pack<VEC> pq;
#pragma unroll
            for (int v = 0; v < VEC; ++v) pq.v[v] = q[c][v];
            pq.store(qrow + f);
          }
        }
      }
      double err = 0.0;
      if (a.out_err || a.loss_partial) {
        const double yy = a.y[r], pp = a.pscore[r];
        err = yy / pp - pred;
        if (l == 0 && a.loss_partial) loss_acc += logloss_term(yy, pp, pred, a.eps);
      }
      if (l == 0) {
        if (a.out_pred) a.out_pred[t] = pred;
        if (a.out_err) a.out_err[t] = err;
      }
      if (a.n_hot > 0 && RFM_KEEP(a, 32)) {
        // hot entries of this row: err * x * [q, 1, x] into the workgroup's LDS sums
        for (int64_t pb = p0; pb < p1; pb += LPR) {
          const int64_t my = pb + l;
          int32_t so = 0;
          double x = 0.0;
          if (my < p1) {
            so = a.slot_of[my];
            x = a.values[my];
          }
          const int cnt = (p1 - pb) < int64_t(LPR) ? int(p1 - pb) : LPR;
          for (int j = 0; j < cnt; ++j) {
            const int32_t sj = __shfl(so, j, LPR);
            const double xj = __shfl(x, j, LPR);
            if (sj >= 0) continue;
            const double coef = err * xj;
            double* hrow = hot + (-1 - sj) * hot_w;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int f = (c * LPR + l) * VEC;
              if (f < k) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                  if (RFM_KEEP(a, 8)) unsafeAtomicAdd(hrow + f + v, coef * q[c][v]);
              }
            }
            if (l == 0 && RFM_KEEP(a, 8)) {
              unsafeAtomicAdd(hrow + k, coef);
              unsafeAtomicAdd(hrow + k + 1, coef * xj);
            }
          }
        }
      }
    }
  }

  if (a.n_hot > 0 && RFM_KEEP(a, 16)) {
    __syncthreads();
    double* slab = a.hot_slab + int64_t(blockIdx.x) * a.n_hot * hot_w;
    for (int i = tid; i < a.n_hot * hot_w; i += BLOCK) slab[i] = hot[i];
  }
  if (a.loss_partial) {
    const double s = block_sum<BLOCK>(loss_acc, red);
    if (tid == 0) a.loss_partial[blockIdx.x] = s;
  }
}

// loss = -(sum of partials)/n, fixed order
__global__ __launch_bounds__(kBlock) void loss_finish_kernel(const double* partial, int n_partial,
                                                            int64_t n_rows, double* out) {
  __shared__ double lds[kBlock];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += kBlock) acc += partial[i];
  const double s = block_sum<kBlock>(acc, lds);
  if (threadIdx.x == 0) out[0] = -s / double(n_rows);
}

// standalone IPS log-loss of given scores (src/base.py:37-61)
__global__ __launch_bounds__(kBlock) void logloss_kernel(const double* y, const double* pred,
                                                        const double* pscore,
                                                        const int32_t* row_ids, int64_t n_rows,
                                                        double eps, double* partial) {
  __shared__ double lds[kBlock];
  double acc = 0.0;
  for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < n_rows;
       t += int64_t(gridDim.x) * kBlock) {
    const int64_t r = row_ids ? int64_t(row_ids[t]) : t;
    acc += logloss_term(y[r], pscore[r], pred[t], eps);
  }
  const double s = block_sum<kBlock>(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------
// 2. column-owner gradient + update (sparse class)
// ---------------------------------------------------------------------------
struct WorkItem {
  int32_t slot_begin;
  int32_t slot_end;
  int32_t part;  // >=0: chunk of a long column -> write partial[part]; -1: whole columns
  int32_t pad;
};

struct ConsArgs {
  const WorkItem* items;
  int32_t n_items;
  int32_t* slot_t;
  const double* csc_x;
  const int32_t* csc_col;
  const double* err;
  const double* Q;
  int32_t k;
  int64_t n;     // features
  double* V;     // apply mode: updated in place; grad mode: read only
  double* w;
  double lr;
  double* partials;  // [n_parts][k+2]: M[0..k), sum coef, sum coef*x
  double* grad;      // nullable: grad mode -> [G_V | g_w | g_w0]
};

template <int LPR, int VEC, int NC>
struct ColAcc {
  double m[NC][VEC];
  double gw, d;
  __device__ inline void clear() {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) m[c][v] = 0.0;
    gw = 0.0;
    d = 0.0;
  }
  // sum the 64/LPR lane groups in group order; every lane gets the total
  __device__ inline void combine(int l) {
    constexpr int RPW = kWave / LPR;
    if (RPW == 1) return;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        double tot = 0.0;
#pragma unroll
        for (int gg = 0; gg < RPW; ++gg) tot += __shfl(m[c][v], l + gg * LPR, kWave);
        m[c][v] = tot;
      }
    double tg = 0.0, td = 0.0;
#pragma unroll
    for (int gg = 0; gg < RPW; ++gg) {
      tg += __shfl(gw, l + gg * LPR, kWave);
      td += __shfl(d, l + gg * LPR, kWave);
    }
    gw = tg;
    d = td;
  }
};

template <int LPR, int VEC, int NC>
__device__ inline void flush_column(ColAcc<LPR, VEC, NC>& acc, int32_t col, int32_t part,
                                    const ConsArgs& a, int l, int g) {
  acc.combine(l);
  if (g != 0) return;
  const int k = a.k;
  if (part >= 0) {
    double* prow = a.partials + int64_t(part) * (k + 2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int f = (c * LPR + l) * VEC;
      if (f < k) {
        // partial rows are (k+2)-strided: 16-byte alignment is not guaranteed
#pragma unroll
        for (int v = 0; v < VEC; ++v) prow[f + v] = acc.m[c][v];
      }
    }
    if (l == 0) {
      prow[k] = acc.gw;
      prow[k + 1] = acc.d;
    }
    return;
  }
  double* vrow = a.V + int64_t(col) * k;
  if (a.grad) {
    double* grow = a.grad + int64_t(col) * k;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int f = (c * LPR + l) * VEC;
      if (f < k) {
        Pack<VEC> pv, pg;
        pv.load(vrow + f);
#pragma unroll
        for (int v = 0; v < VEC; ++v) pg.v[v] = acc.d * pv.v[v] - acc.m[c][v];
        pg.store(grow + f);
      }
    }
    if (l == 0) a.grad[a.n * k + col] = -acc.gw;
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int f = (c * LPR + l) * VEC;
      if (f < k) {
        Pack<VEC> pv;
        pv.load(vrow + f);
#pragma unroll
        for (int v = 0; v < VEC; ++v) pv.v[v] += a.lr * (acc.m[c][v] - acc.d * pv.v[v]);
        pv.store(vrow + f);
      }
    }
    if (l == 0) a.w[col] += a.lr * acc.gw;
  }
}

template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kBlock) void fm_consume_kernel(ConsArgs a) {
  constexpr int RPW = kWave / LPR;
  const int lane = threadIdx.x % kWave;
  const int l = lane % LPR;
  const int g = lane / LPR;
  const int item_id = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (item_id >= a.n_items) return;
  const WorkItem it = a.items[item_id];
  const int k = a.k;

  ColAcc<LPR, VEC, NC> acc;
  acc.clear();
  int32_t cur = -1;  // column being accumulated (wave-uniform)

  for (int32_t base = it.slot_begin; base < it.slot_end; base += kWave) {
    const int32_t s = base + lane;
    int32_t t = -1;
    if (s < it.slot_end) t = a.slot_t[s];
    const bool active = t >= 0;
    unsigned long long mask = __ballot(active);
    if (mask == 0ull) continue;
    int32_t col = -1;
    double x = 0.0, coef = 0.0;
    if (active) {
      a.slot_t[s] = -1;
      col = a.csc_col[s];
      x = a.csc_x[s];
      coef = a.err[t] * x;
    }
    while (mask) {
      const int first = __ffsll((long long)mask) - 1;
      const int32_t c = __builtin_amdgcn_readfirstlane(__shfl(col, first, kWave));
      if (c != cur) {
        if (cur >= 0) {
          flush_column<LPR, VEC, NC>(acc, cur, it.part, a, l, g);
          acc.clear();
        }
        cur = c;
      }
      unsigned long long cmask = __ballot(active && col == c);
      mask &= ~cmask;
      // the column's marked slots of this window, RPW at a time, in slot order
      while (cmask) {
        int src = -1;
#pragma unroll
        for (int gi = 0; gi < RPW; ++gi) {
          if (cmask) {
            const int b = __ffsll((long long)cmask) - 1;
            cmask &= cmask - 1;
            if (gi == g) src = b;
          }
        }
        const int from = src >= 0 ? src : lane;
        const int32_t tt = __shfl(t, from, kWave);
        const double cc = __shfl(coef, from, kWave);
        const double xx = __shfl(x, from, kWave);
        if (src >= 0) {
          const double* qrow = a.Q + int64_t(tt) * k;
#pragma unroll
          for (int ch = 0; ch < NC; ++ch) {
            const int f = (ch * LPR + l) * VEC;
            if (f < k) {
              Pack<VEC> pq;
              pq.load(qrow + f);
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc.m[ch][v] += cc * pq.v[v];
            }
          }
          acc.gw += cc;
          acc.d += cc * xx;
        }
      }
    }
  }
  if (cur >= 0) {
    flush_column<LPR, VEC, NC>(acc, cur, it.part, a, l, g);
  } else if (it.part >= 0) {
    // an untouched chunk still owes its (zero) partial
    acc.clear();
    flush_column<LPR, VEC, NC>(acc, 0, it.part, a, l, g);
  }
}

// ---------------------------------------------------------------------------
// 3. hot columns (slabs in block order), long sparse columns (partials in chunk
//    order) and w0
// ---------------------------------------------------------------------------
struct SplitCol {
  int32_t col;
  int32_t part_begin;
  int32_t part_count;
  int32_t pad;
};

struct FinArgs {
  const SplitCol* split;
  int32_t n_split;
  const double* partials;
  const int32_t* hot_cols;
  int32_t n_hot;
  const double* hot_slab;
  int32_t n_slabs;
  const double* err;
  int64_t batch;
  int32_t k;
  int64_t n;
  double* w0;
  double* w;
  double* V;
  double lr;
  double* grad;  // nullable
};

// tot[f] = sum_{r<rows} base[r*stride + f], f < width, in a fixed order: the
// block's threads split into row groups x factor lanes, each group sums its
// rows in ascending order, the groups are then added in group order.
__device__ inline void ordered_rows_sum(const double* base, int64_t stride, int rows, int width,
                                        double* scratch /*[kBlock]*/, double* tot /*[width]*/) {
  int fw = 1;
  while (fw < width && fw < kBlock) fw <<= 1;
  const int nsg = kBlock / fw;
  const int sg = threadIdx.x / fw, fl = threadIdx.x % fw;
  for (int f0 = 0; f0 < width; f0 += fw) {
    const int f = f0 + fl;
    double acc = 0.0;
    if (f < width)
      for (int r = sg; r < rows; r += nsg) acc += base[int64_t(r) * stride + f];
    __syncthreads();
    scratch[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && f < width) {
      double s = 0.0;
      for (int j = 0; j < nsg; ++j) s += scratch[j * fw + fl];
      tot[f] = s;
    }
  }
  __syncthreads();
}

// blocks [0, n_split): one long sparse column each; [n_split, n_split+n_hot):
// one hot column each; last block: w0 from the fixed-order sum of the residuals.
__global__ __launch_bounds__(kBlock) void fm_finalize_kernel(FinArgs a) {
  __shared__ double scratch[kBlock];
  __shared__ double tot[RFM_MAX_FACTORS + 2];
  const int k = a.k;
  const int b = blockIdx.x;
  if (b < a.n_split + a.n_hot) {
    int32_t col;
    if (b < a.n_split) {
      const SplitCol sc = a.split[b];
      col = sc.col;
      ordered_rows_sum(a.partials + int64_t(sc.part_begin) * (k + 2), k + 2, sc.part_count, k + 2,
                       scratch, tot);
    } else {
      const int h = b - a.n_split;
      col = a.hot_cols[h];
      ordered_rows_sum(a.hot_slab + int64_t(h) * (k + 2), int64_t(a.n_hot) * (k + 2), a.n_slabs,
                       k + 2, scratch, tot);
    }
    const double gw = tot[k], d = tot[k + 1];
    for (int f = threadIdx.x; f < k; f += kBlock) {
      const int64_t at = int64_t(col) * k + f;
      if (a.grad)
        a.grad[at] = d * a.V[at] - tot[f];
      else
        a.V[at] += a.lr * (tot[f] - d * a.V[at]);
    }
    if (threadIdx.x == 0) {
      if (a.grad)
        a.grad[a.n * k + col] = -gw;
      else
        a.w[col] += a.lr * gw;
    }
    return;
  }
  double acc = 0.0;
  for (int64_t t = threadIdx.x; t < a.batch; t += kBlock) acc += a.err[t];
  const double s = block_sum<kBlock>(acc, scratch);
  if (threadIdx.x == 0) {
    if (a.grad)
      a.grad[a.n * k + a.n] = -s;
    else
      a.w0[0] += a.lr * s;
  }
}

// theta -= lr * grad over [V | w | w0]
__global__ __launch_bounds__(kBlock) void fm_apply_kernel(double* V, double* w, double* w0,
                                                         const double* grad, int64_t nk,
                                                         int64_t n, double lr) {
  const int64_t total = nk + n + 1;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * kBlock) {
    double* dst = i < nk ? V + i : (i < nk + n ? w + (i - nk) : w0);
    *dst -= lr * grad[i];
  }
}

// ---------------------------------------------------------------------------
// dispatch on the factor count
// ---------------------------------------------------------------------------
struct Shape {
  int lpr, vec, nc;
};

inline Shape shape_for(int k) {
  RFM_REQUIRE(k >= 1 && k <= RFM_MAX_FACTORS, "n_factors=%d unsupported (1..%d)", k,
              RFM_MAX_FACTORS);
  Shape s;
  s.vec = (k % 2 == 0) ? 2 : 1;
  const int units = (k + s.vec - 1) / s.vec;
  int lpr = 4;
  while (lpr < units && lpr < 64) lpr *= 2;
  s.lpr = lpr;
  int nc = 1;
  while (lpr * nc < units) nc *= 2;
  s.nc = nc;
  return s;
}

#define RFM_FOR_SHAPE(S, CALL)                                                         \
  do {                                                                                 \
    const ::rfm::Shape _s = (S);                                                       \
    if (_s.vec == 2) {                                                                 \
      if (_s.nc == 1) {                                                                \
        switch (_s.lpr) {                                                              \
          case 4: CALL(4, 2, 1); break;                                                \
          case 8: CALL(8, 2, 1); break;                                                \
          case 16: CALL(16, 2, 1); break;                                              \
          case 32: CALL(32, 2, 1); break;                                              \
          default: CALL(64, 2, 1); break;                                              \
        }                                                                              \
      } else if (_s.nc == 2) { CALL(64, 2, 2); }                                       \
      else if (_s.nc == 4) { CALL(64, 2, 4); }                                         \
      else { CALL(64, 2, 8); }                                                         \
    } else {                                                                           \
      if (_s.nc == 1) {                                                                \
        switch (_s.lpr) {                                                              \
          case 4: CALL(4, 1, 1); break;                                                \
          case 8: CALL(8, 1, 1); break;                                                \
          case 16: CALL(16, 1, 1); break;                                              \
          case 32: CALL(32, 1, 1); break;                                              \
          default: CALL(64, 1, 1); break;                                              \
        }                                                                              \
      } else if (_s.nc == 2) { CALL(64, 1, 2); }                                       \
      else if (_s.nc == 4) { CALL(64, 1, 4); }                                         \
      else if (_s.nc == 8) { CALL(64, 1, 8); }                                         \
      else { CALL(64, 1, 16); }                                                        \
    }                                                                                  \
  } while (0)

// forward launch geometry: large workgroups (few hot-sum slabs) once the batch
// fills the chip with them, 256-thread ones otherwise.  RFM_FWD_BLOCK /
// RFM_FWD_PER_CU override the choice (tuning experiments only).
struct FwdGeom {
  int block, grid;
};

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

inline FwdGeom forward_geom(const rfm_ctx* ctx, int64_t n_rows, int lpr) {
  static const int big_block = env_int("RFM_FWD_BLOCK", 512);
  static const int per_cu = env_int("RFM_FWD_PER_CU", 3);
  FwdGeom g;
  const int big = (big_block == 1024 || big_block == 512) ? big_block : 256;
  const int64_t rows_big = big / lpr;
  if (big > 256 && (n_rows + rows_big - 1) / rows_big >= int64_t(ctx->n_cu) * per_cu) {
    g.block = big;
    g.grid = ctx->n_cu * per_cu;
  } else {
    g.block = 256;
    const int gpb = 256 / lpr;
    const int64_t want = (n_rows + gpb - 1) / gpb;
    g.grid = int(std::max<int64_t>(1, std::min<int64_t>(want, int64_t(ctx->n_cu) * 8)));
  }
  g.grid = std::min(g.grid, 2048);
  return g;
}

constexpr int kMaxFwdGrid = 2048;  // upper bound of forward_geom().grid, sizes scratch

void launch_forward(rfm_ctx* ctx, FwdArgs a, FwdGeom geom) {
  if (a.n_rows <= 0) return;
  const Shape s = shape_for(a.k);
  const size_t lds = size_t(geom.block) * 8 + size_t(a.n_hot) * size_t(a.k + 2) * 8;
#define RFM_CALL_FWD(L, Vv, N)                                                                   \
  do {                                                                                           \
    if (geom.block == 1024)                                                                      \
      hipLaunchKernelGGL((fm_forward_kernel<L, Vv, N, 1024>), dim3(geom.grid), dim3(1024), lds,  \
                         ctx->stream, a);                                                        \
    else if (geom.block == 512)                                                                  \
      hipLaunchKernelGGL((fm_forward_kernel<L, Vv, N, 512>), dim3(geom.grid), dim3(512), lds,    \
                         ctx->stream, a);                                                        \
    else                                                                                         \
      hipLaunchKernelGGL((fm_forward_kernel<L, Vv, N, 256>), dim3(geom.grid), dim3(256), lds,    \
                         ctx->stream, a);                                                        \
  } while (0)
  RFM_FOR_SHAPE(s, RFM_CALL_FWD);
#undef RFM_CALL_FWD
  RFM_HIP_CHECK(hipGetLastError());
}

void launch_forward(rfm_ctx* ctx, FwdArgs a) {
  if (a.n_rows <= 0) return;
  launch_forward(ctx, a, forward_geom(ctx, a.n_rows, shape_for(a.k).lpr));
}

// forward with loss: partials in ctx scratch, finished into d_out_loss
void forward_loss(rfm_ctx* ctx, FwdArgs a, double* d_out_loss) {
  RFM_REQUIRE(a.n_rows > 0, "loss of zero rows");
  const FwdGeom geom = forward_geom(ctx, a.n_rows, shape_for(a.k).lpr);
  ctx->loss_partials.ensure(size_t(std::max(kMaxFwdGrid, ctx->n_cu * 8)) * sizeof(double));
  a.loss_partial = ctx->loss_partials.as<double>();
  launch_forward(ctx, a, geom);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, ctx->stream,
                     ctx->loss_partials.as<double>(), geom.grid, a.n_rows, d_out_loss);
  RFM_HIP_CHECK(hipGetLastError());
}

}  // namespace rfm

// ---------------------------------------------------------------------------
// training plan
// ---------------------------------------------------------------------------
struct rfm_fm_plan {
  int32_t device = 0;
  int64_t n_rows = 0, n_features = 0, nnz = 0, n_slots = 0, max_batch = 0;
  int32_t k = 0;
  int32_t n_items = 0, n_split = 0, n_parts = 0, n_hot = 0;
  rfm::DevBuf slot_of, slot_t, csc_x, csc_col, items, split, partials, Q, err, hot_cols, hot_slab;
  size_t device_bytes() const {
    return slot_of.bytes + slot_t.bytes + csc_x.bytes + csc_col.bytes + items.bytes +
           split.bytes + partials.bytes + Q.bytes + err.bytes + hot_cols.bytes + hot_slab.bytes;
  }
};

using namespace rfm;

namespace {

constexpr int32_t kPackSlots = 256;    // whole short columns packed per wave up to this
constexpr int32_t kChunkSlots = 4096;  // a longer column is cut into chunks of this
constexpr int32_t kDefaultHotMinCount = 32;

void upload(DevBuf& dst, const void* src, size_t bytes, hipStream_t stream) {
  dst.alloc(bytes);
  if (bytes) RFM_HIP_CHECK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, stream));
}

void check_step_args(const rfm_fm_plan* plan, const void* indptr, const void* indices,
                     const void* values, const void* y, const void* p, const void* ids,
                     int64_t batch) {
  RFM_REQUIRE(plan, "null plan");
  RFM_REQUIRE(indptr && indices && values && y && p && ids, "null pointer");
  RFM_REQUIRE(batch >= 1 && batch <= plan->max_batch, "batch=%lld outside 1..max_batch=%lld",
              (long long)batch, (long long)plan->max_batch);
}

// the three launches of one step; grad == nullptr -> update in place
void enqueue_step(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                  const int32_t* d_indices, const double* d_values, const double* d_y,
                  const double* d_pscore, const int32_t* d_row_ids, int64_t batch, double* d_w0,
                  double* d_w, double* d_V, double lr, double* d_grad) {
  const int k = plan->k;
  const Shape s = shape_for(k);
  FwdArgs f{};
  f.indptr = d_indptr;
  f.indices = d_indices;
  f.values = d_values;
  f.row_ids = d_row_ids;
  f.n_rows = batch;
  f.w0 = d_w0;
  f.w = d_w;
  f.V = d_V;
  f.k = k;
  f.y = d_y;
  f.pscore = d_pscore;
  f.out_err = plan->err.as<double>();
  f.out_Q = plan->Q.as<double>();
  f.slot_of = plan->slot_of.as<int32_t>();
  f.slot_t = plan->slot_t.as<int32_t>();
  f.n_hot = plan->n_hot;
  f.hot_slab = plan->hot_slab.as<double>();
#ifdef RFM_ABLATE
  f.ablate = env_int("RFM_ABLATE_MASK", 0);
#endif
  const FwdGeom geom = forward_geom(ctx, batch, s.lpr);
  ctx->prof_mark();
  launch_forward(ctx, f, geom);
  ctx->prof_mark();

  if (d_grad) {
    const size_t bytes = (size_t(plan->n_features) * (k + 1) + 1) * sizeof(double);
    RFM_HIP_CHECK(hipMemsetAsync(d_grad, 0, bytes, ctx->stream));
  }
  if (plan->n_items > 0) {
    ConsArgs c{};
    c.items = plan->items.as<WorkItem>();
    c.n_items = plan->n_items;
    c.slot_t = plan->slot_t.as<int32_t>();
    c.csc_x = plan->csc_x.as<double>();
    c.csc_col = plan->csc_col.as<int32_t>();
    c.err = plan->err.as<double>();
    c.Q = plan->Q.as<double>();
    c.k = k;
    c.n = plan->n_features;
    c.V = d_V;
    c.w = d_w;
    c.lr = lr;
    c.partials = plan->partials.as<double>();
    c.grad = d_grad;
    const int wpb = kBlock / kWave;
    const int grid = (plan->n_items + wpb - 1) / wpb;
#define RFM_CALL_CONS(L, Vv, N) \
  hipLaunchKernelGGL((fm_consume_kernel<L, Vv, N>), dim3(grid), dim3(kBlock), 0, ctx->stream, c)
    RFM_FOR_SHAPE(s, RFM_CALL_CONS);
#undef RFM_CALL_CONS
    RFM_HIP_CHECK(hipGetLastError());
  }
  ctx->prof_mark();
  FinArgs fa{};
  fa.split = plan->split.as<SplitCol>();
  fa.n_split = plan->n_split;
  fa.partials = plan->partials.as<double>();
  fa.hot_cols = plan->hot_cols.as<int32_t>();
  fa.n_hot = plan->n_hot;
  fa.hot_slab = plan->hot_slab.as<double>();
  fa.n_slabs = geom.grid;
  fa.err = plan->err.as<double>();
  fa.batch = batch;
  fa.k = k;
  fa.n = plan->n_features;
  fa.w0 = d_w0;
  fa.w = d_w;
  fa.V = d_V;
  fa.lr = lr;
  fa.grad = d_grad;
  hipLaunchKernelGGL(fm_finalize_kernel, dim3(plan->n_split + plan->n_hot + 1), dim3(kBlock), 0,
                     ctx->stream, fa);
  RFM_HIP_CHECK(hipGetLastError());
  ctx->prof_mark();
}

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
  return f;
}

}  // namespace

extern "C" {

int32_t rfm_fm_forward(rfm_ctx* ctx, const int64_t* d_indptr, const int32_t* d_indices,
                       const double* d_values, const int32_t* d_row_ids, int64_t n_rows,
                       const double* d_w0, const double* d_w, const double* d_V,
                       int64_t n_features, int32_t n_factors, double* d_out_pred) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    RFM_REQUIRE(n_rows >= 0 && n_features >= 1, "bad shape");
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
    RFM_REQUIRE(n_rows >= 1 && n_features >= 1, "bad shape");
    FwdArgs f = forward_args(d_indptr, d_indices, d_values, d_row_ids, n_rows, d_w0, d_w, d_V,
                             n_factors);
    f.y = d_y;
    f.pscore = d_pscore;
    f.eps = eps;
    f.out_pred = d_out_pred;
    forward_loss(ctx, f, d_out_loss);
  });
}

int32_t rfm_fm_plan_create(rfm_ctx* ctx, const int64_t* h_indptr, const int32_t* h_indices,
                           const double* h_values, int64_t n_rows, int64_t n_features,
                           int32_t n_factors, int64_t max_batch, int32_t hot_min_count,
                           rfm_fm_plan** out) {
  return guarded([&] {
    RFM_REQUIRE(ctx && h_indptr && out, "null pointer");
    RFM_REQUIRE(n_rows >= 1 && n_features >= 1 && max_batch >= 1, "bad shape");
    (void)shape_for(n_factors);
    const int64_t nnz = h_indptr[n_rows];
    RFM_REQUIRE(nnz >= 0 && nnz < (int64_t(1) << 31) - kWave, "nnz=%lld unsupported",
                (long long)nnz);
    RFM_REQUIRE(nnz == 0 || (h_indices && h_values), "null CSR arrays");
    RFM_REQUIRE(n_features < (int64_t(1) << 31), "n_features too large");
    const size_t nz = static_cast<size_t>(nnz);
    const size_t nf = static_cast<size_t>(n_features);

    // column lengths
    std::vector<int64_t> len(nf, 0);
    for (int64_t p = 0; p < nnz; ++p) {
      const int32_t c = h_indices[p];
      RFM_REQUIRE(c >= 0 && c < n_features, "column index %d out of range", c);
      len[size_t(c)]++;
    }
    // hot class: expected entries per batch >= hot_min, most frequent first, LDS budget
    std::vector<int32_t> hot_cols;
    std::vector<int32_t> hot_rank(nf, -1);
    if (hot_min_count >= 0) {
      const int64_t hot_min = hot_min_count > 0 ? hot_min_count : kDefaultHotMinCount;
      for (int64_t c = 0; c < n_features; ++c)
        if (len[size_t(c)] * max_batch >= hot_min * n_rows) hot_cols.push_back(int32_t(c));
      std::stable_sort(hot_cols.begin(), hot_cols.end(),
                       [&](int32_t x, int32_t y) { return len[size_t(x)] > len[size_t(y)]; });
      const size_t cap = std::min<size_t>(kMaxHot, kHotLdsBudget / (size_t(n_factors + 2) * 8));
      if (hot_cols.size() > cap) hot_cols.resize(cap);
      std::sort(hot_cols.begin(), hot_cols.end());
      for (size_t h = 0; h < hot_cols.size(); ++h) hot_rank[size_t(hot_cols[h])] = int32_t(h);
    }
    // column-major rank of every sparse-class entry (stable: row order inside a column)
    std::vector<int64_t> cptr(nf + 1, 0);
    for (size_t c = 0; c < nf; ++c) cptr[c + 1] = cptr[c] + (hot_rank[c] >= 0 ? 0 : len[c]);
    const int64_t n_slots = cptr[nf];
    const size_t ns = static_cast<size_t>(n_slots);
    std::vector<int32_t> slot_of(nz), csc_col(ns);
    std::vector<double> csc_x(ns);
    {
      std::vector<int64_t> cursor(cptr.begin(), cptr.end() - 1);
      for (int64_t p = 0; p < nnz; ++p) {
        const int32_t c = h_indices[p];
        if (hot_rank[size_t(c)] >= 0) {
          slot_of[size_t(p)] = -1 - hot_rank[size_t(c)];
          continue;
        }
        const int64_t s = cursor[size_t(c)]++;
        slot_of[size_t(p)] = int32_t(s);
        csc_col[size_t(s)] = c;
        csc_x[size_t(s)] = h_values[p];
      }
    }
    // work items: whole short columns packed up to kPackSlots, long columns chunked
    std::vector<WorkItem> items;
    std::vector<SplitCol> split;
    int32_t n_parts = 0;
    int64_t open_begin = -1;
    auto close_open = [&](int64_t end) {
      if (open_begin >= 0 && end > open_begin)
        items.push_back({int32_t(open_begin), int32_t(end), -1, 0});
      open_begin = -1;
    };
    for (size_t c = 0; c < nf; ++c) {
      const int64_t b = cptr[c], e = cptr[c + 1], clen = e - b;
      if (clen == 0) continue;
      if (clen > kPackSlots) {
        close_open(b);
        if (clen <= kChunkSlots) {
          items.push_back({int32_t(b), int32_t(e), -1, 0});
        } else {
          SplitCol sc{int32_t(c), n_parts, 0, 0};
          for (int64_t s = b; s < e; s += kChunkSlots) {
            items.push_back({int32_t(s), int32_t(std::min(e, s + kChunkSlots)), n_parts++, 0});
            sc.part_count++;
          }
          split.push_back(sc);
        }
        continue;
      }
      if (open_begin >= 0 && e - open_begin > kPackSlots) close_open(b);
      if (open_begin < 0) open_begin = b;
    }
    close_open(n_slots);
    // longest items first: the tail of the launch is then made of short ones
    std::stable_sort(items.begin(), items.end(), [](const WorkItem& x, const WorkItem& y) {
      return (x.slot_end - x.slot_begin) > (y.slot_end - y.slot_begin);
    });

    RFM_HIP_CHECK(hipSetDevice(ctx->device));
    auto plan = std::make_unique<rfm_fm_plan>();
    plan->device = ctx->device;
    plan->n_rows = n_rows;
    plan->n_features = n_features;
    plan->nnz = nnz;
    plan->n_slots = n_slots;
    plan->max_batch = max_batch;
    plan->k = n_factors;
    plan->n_items = int32_t(items.size());
    plan->n_split = int32_t(split.size());
    plan->n_parts = n_parts;
    plan->n_hot = int32_t(hot_cols.size());
    upload(plan->slot_of, slot_of.data(), nz * 4, ctx->stream);
    upload(plan->csc_col, csc_col.data(), ns * 4, ctx->stream);
    upload(plan->csc_x, csc_x.data(), ns * 8, ctx->stream);
    upload(plan->items, items.data(), items.size() * sizeof(WorkItem), ctx->stream);
    upload(plan->split, split.data(), split.size() * sizeof(SplitCol), ctx->stream);
    upload(plan->hot_cols, hot_cols.data(), hot_cols.size() * 4, ctx->stream);
    plan->slot_t.alloc((ns + kWave) * 4);
    RFM_HIP_CHECK(hipMemsetAsync(plan->slot_t.p, 0xFF, plan->slot_t.bytes, ctx->stream));
    plan->partials.alloc(size_t(std::max(n_parts, 1)) * size_t(n_factors + 2) * 8);
    plan->hot_slab.alloc(size_t(kMaxFwdGrid) * std::max<size_t>(hot_cols.size(), 1) *
                         size_t(n_factors + 2) * 8);
    plan->Q.alloc(size_t(max_batch) * size_t(n_factors) * 8);
    plan->err.alloc(size_t(max_batch) * 8);
    // host vectors die at scope exit: wait for the copies
    RFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = plan.release();
  });
}

int32_t rfm_fm_plan_destroy(rfm_fm_plan* plan) {
  return guarded([&] {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    delete plan;
  });
}

int32_t rfm_fm_plan_info(const rfm_fm_plan* plan, int64_t* h_out5) {
  return guarded([&] {
    RFM_REQUIRE(plan && h_out5, "null pointer");
    h_out5[0] = plan->n_items;
    h_out5[1] = plan->n_split;
    h_out5[2] = plan->n_hot;
    h_out5[3] = plan->nnz;
    h_out5[4] = int64_t(plan->device_bytes());
  });
}

int32_t rfm_fm_step(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                    const int32_t* d_indices, const double* d_values, const double* d_y,
                    const double* d_pscore, const int32_t* d_row_ids, int64_t batch,
                    double* d_w0, double* d_w, double* d_V, double lr) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V, "null pointer");
    check_step_args(plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_row_ids, batch);
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

int32_t rfm_fm_train(rfm_ctx* ctx, rfm_fm_plan* plan, const int64_t* d_indptr,
                     const int32_t* d_indices, const double* d_values, const double* d_y,
                     const double* d_pscore, const int32_t* d_ids, int64_t batch,
                     int64_t n_iters, double* d_w0, double* d_w, double* d_V, double lr,
                     const int64_t* d_val_indptr, const int32_t* d_val_indices,
                     const double* d_val_values, const double* d_val_y,
                     const double* d_val_pscore, int64_t n_val, double eps,
                     double* d_out_train_loss, double* d_out_val_loss) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_w0 && d_w && d_V, "null pointer");
    RFM_REQUIRE(n_iters >= 0, "negative n_iters");
    if (n_iters == 0) return;
    check_step_args(plan, d_indptr, d_indices, d_values, d_y, d_pscore, d_ids, batch);
    if (d_out_val_loss)
      RFM_REQUIRE(d_val_indptr && d_val_indices && d_val_values && d_val_y && d_val_pscore &&
                      n_val >= 1,
                  "validation arrays missing");
    for (int64_t it = 0; it < n_iters; ++it) {
      const int32_t* ids = d_ids + it * batch;
      enqueue_step(ctx, plan, d_indptr, d_indices, d_values, d_y, d_pscore, ids, batch, d_w0,
                   d_w, d_V, lr, nullptr);
      if (d_out_train_loss) {
        FwdArgs f = forward_args(d_indptr, d_indices, d_values, ids, batch, d_w0, d_w, d_V,
                                 plan->k);
        f.y = d_y;
        f.pscore = d_pscore;
        f.eps = eps;
        forward_loss(ctx, f, d_out_train_loss + it);
      }
      if (d_out_val_loss) {
        FwdArgs f = forward_args(d_val_indptr, d_val_indices, d_val_values, nullptr, n_val,
                                 d_w0, d_w, d_V, plan->k);
        f.y = d_val_y;
        f.pscore = d_val_pscore;
        f.eps = eps;
        forward_loss(ctx, f, d_out_val_loss + it);
      }
    }
  });
}

}  // extern "C"
