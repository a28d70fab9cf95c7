// Device code of the FM path (gfx950).  See rfm_fm.hip for the launch side and
// DESIGN.md section 4 for the algorithm.  Reference arithmetic: src/fm.py:80-88,
// 114-187, src/base.py:37-66.
#pragma once

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rfm {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr double kLogitClip = 700.0;  // src/base.py:65

// ---------------------------------------------------------------------------
// records of the training plan (built once per fit on the host)
// ---------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------
// helpers
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

// waves per SIMD the big forward shape is compiled for (register budget 512/N)
#ifndef RFM_FWD_BIG_WAVES
#define RFM_FWD_BIG_WAVES 4
#endif
// rows per lane group of the big forward shape (single-chunk factor counts)
#ifndef RFM_FWD_ROWS
#define RFM_FWD_ROWS 2
#endif

// rows a lane group works on concurrently (independent load chains in flight)
constexpr int rows_in_flight(int nc) { return nc == 1 ? RFM_FWD_ROWS : 1; }

// ---------------------------------------------------------------------------
// 1. forward (+ residual, Q, slot marks, hot sums, loss partials)
// ---------------------------------------------------------------------------
struct FwdArgs {
  // the log: either the plan's records (training) ...
  const Entry* ent;
  const RowRec* rows;
  // ... or the caller's CSR arrays (+ labels / propensities when a loss is asked for)
  const int64_t* indptr;
  const int32_t* indices;
  const double* values;
  const double* y;
  const double* pscore;
  const int32_t* row_ids;  // may be null: row t
  int64_t n_rows;
  const double* w0;
  const double* w;
  const double* V;
  int32_t k;
  double* out_pred;      // nullable
  double* out_err;       // nullable
  double* out_Q;         // nullable [n_rows][k]
  int32_t* slot_t;       // nullable: mark slot_t[slot] = t for sparse-class entries
  int32_t n_hot;         // hot columns (training step only)
  double* hot_slab;      // [n_hot][gridDim.x][k+2]
  double* err_partial;   // nullable: [gridDim.x] per-workgroup sums of the residual (for w0)
  double* loss_partial;  // nullable: [gridDim.x]
  double eps;
  int32_t ablate;  // -DRFM_ABLATE builds only: bit mask of parts to skip (timing experiments)
};

#ifdef RFM_ABLATE
#define RFM_KEEP(a, bit) (((a).ablate & (bit)) == 0)
#else
#define RFM_KEEP(a, bit) true
#endif
// bits: 1 slot marks, 2 Q store, 4 V gathers, 8 hot LDS adds, 16 slab store, 32 hot pass

// A row is handled by LPR consecutive lanes; lane l holds factors
// (c*LPR + l)*VEC .. +VEC-1 for c < NC.  k=32 -> LPR=16, VEC=2: one 16-byte
// load per lane covers a 256-byte row of V, four rows per wave.
//
// One batch is only a few hundred rows per CU, so the kernel is bound by the
// latency of the chain row id -> row record -> entries -> V rows, not by bytes.
// Hence: a lane group keeps R rows in flight; the R x LPR entry records of a
// round are loaded once (one per lane), parked in LDS and re-read as 16-byte
// broadcasts (no cross-lane shuffles); and NO load sits under a divergent
// branch -- indices are clamped and results masked instead -- so that the
// loads of the R rows and of consecutive entries are issued back to back
// rather than each waiting for the previous one.
//
// REC: read the training plan's records (RowRec / Entry); otherwise the
// caller's CSR arrays.
// LDS (dynamic): red[BLOCK] f64 | entry buffer [BLOCK/LPR][R][LPR] Entry | hot sums [H][k+2] f64
template <int LPR, int VEC, int NC, int BLOCK, int R, bool REC>
__global__ __launch_bounds__(BLOCK, (BLOCK == 512 ? RFM_FWD_BIG_WAVES : 1)) void fm_forward_kernel(
    FwdArgs a) {
  constexpr int GPB = BLOCK / LPR;  // lane groups per block
  extern __shared__ double dyn_lds[];
  const int tid = threadIdx.x;
  const int l = tid % LPR;
  const int g = tid / LPR;
  const int k = a.k;
  const int hot_w = k + 2;
  const int H = a.n_hot;
  double* red = dyn_lds;
  Entry* ebuf = reinterpret_cast<Entry*>(dyn_lds + BLOCK) + g * (R * LPR);  // this group's [R][LPR]
  double* hot = dyn_lds + BLOCK + 2 * BLOCK * R;
  const double w0 = a.w0[0];
  const int64_t last_row = a.n_rows - 1;
  double loss_acc = 0.0, err_acc = 0.0;
  // factor offsets of this lane; lanes past k read offset 0 and are masked
  int fo[NC];
  bool fok[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = (c * LPR + l) * VEC;
    fok[c] = f < k;
    fo[c] = fok[c] ? f : 0;
  }

  if (H > 0) {
    for (int i = tid; i < H * hot_w; i += BLOCK) hot[i] = 0.0;
    __syncthreads();
  }

  for (int64_t base = int64_t(blockIdx.x) * (GPB * R); base < a.n_rows;
       base += int64_t(gridDim.x) * (GPB * R)) {
    int64_t t[R], r[R];
    int32_t p0[R];  // entry offsets fit 31 bits (checked when the plan / call is set up)
    int len[R];
    double yy[R], pp[R];
    bool valid[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      t[i] = base + i * GPB + g;
      valid[i] = t[i] <= last_row;
      r[i] = valid[i] ? t[i] : last_row;
    }
    if (a.row_ids) {  // uniform: the R loads stay in one block and overlap
#pragma unroll
      for (int i = 0; i < R; ++i) r[i] = a.row_ids[r[i]];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (REC) {
        const RowRec rec = a.rows[r[i]];
        p0[i] = int32_t(rec.begin);
        len[i] = valid[i] ? int(rec.len) : 0;
        yy[i] = rec.y;
        pp[i] = rec.p;
      } else {
        const int64_t b0 = a.indptr[r[i]], b1 = a.indptr[r[i] + 1];
        p0[i] = int32_t(b0);
        len[i] = valid[i] ? int(b1 - b0) : 0;
        yy[i] = 0.0;
        pp[i] = 1.0;
      }
    }
    if (!REC && a.y) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        yy[i] = a.y[r[i]];
        pp[i] = a.pscore[r[i]];
      }
    }
    int maxlen = len[0];
#pragma unroll
    for (int i = 1; i < R; ++i) maxlen = max(maxlen, len[i]);

    double q[R][NC][VEC];
    double s2[R], lin[R], err[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      s2[i] = 0.0;
      lin[i] = 0.0;
      err[i] = 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) q[i][c][v] = 0.0;
    }

    for (int pb = 0; pb < maxlen; pb += LPR) {
      // stage this round's entries: one per lane and row (padding: column 0, x = 0)
      Entry e[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int my = pb + l;
        // clamp into the row's own entries (entry 0 of the log for an empty row)
        const int32_t at = len[i] > 0 ? p0[i] + min(my, len[i] - 1) : 0;
        if (REC) {
          e[i] = a.ent[at];
        } else {
          e[i].col = a.indices[at];
          e[i].slot = 0;
          e[i].x = a.values[at];
        }
        if (my >= len[i]) e[i] = Entry{0, 0, 0.0};
      }
#pragma unroll
      for (int i = 0; i < R; ++i) {
        lin[i] += a.w[e[i].col] * e[i].x;
        if (REC && a.slot_t && e[i].slot >= 0 && pb + l < len[i] && RFM_KEEP(a, 1))
          a.slot_t[e[i].slot] = int32_t(t[i]);
        ebuf[i * LPR + l] = e[i];
      }
      const int cnt = (maxlen - pb) < LPR ? (maxlen - pb) : LPR;
#pragma unroll 2
      for (int j = 0; j < cnt; ++j) {
        Entry ej[R];
        Pack<VEC> pv[R][NC];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          ej[i] = ebuf[i * LPR + j];  // same address in the lane group: broadcast
          const double* vrow = a.V + (RFM_KEEP(a, 4) ? int64_t(ej[i].col) * k : 0);
#pragma unroll
          for (int c = 0; c < NC; ++c) pv[i][c].load(vrow + fo[c]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const bool act = pb + j < len[i];
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              const double vx = (act && fok[c]) ? pv[i][c].v[v] * ej[i].x : 0.0;
              q[i][c][v] += vx;
              s2[i] += vx * vx;
            }
        }
      }
    }

#pragma unroll
    for (int i = 0; i < R; ++i) {
      double pair = -s2[i];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) pair += q[i][c][v] * q[i][c][v];
      pair = group_sum<LPR>(pair);
      const double linsum = group_sum<LPR>(lin[i]);
      const double pred = sigmoid_clipped(w0 + linsum + 0.5 * pair);
      err[i] = valid[i] ? yy[i] / pp[i] - pred : 0.0;
      if (valid[i]) {
        if (a.out_Q && RFM_KEEP(a, 2)) {
          double* qrow = a.out_Q + t[i] * k;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            if (fok[c]) {
              Pack<VEC> pq;
#pragma unroll
              for (int v = 0; v < VEC; ++v) pq.v[v] = q[i][c][v];
              pq.store(qrow + fo[c]);
            }
          }
        }
        if (l == 0) {
          if (a.out_pred) a.out_pred[t[i]] = pred;
          if (a.out_err) a.out_err[t[i]] = err[i];
          if (a.loss_partial) loss_acc += logloss_term(yy[i], pp[i], pred, a.eps);
          err_acc += err[i];
        }
      }
    }

    if (REC && H > 0 && RFM_KEEP(a, 32)) {
      // hot entries: err * x * [q, 1, x] into the workgroup's LDS sums.  A single
      // round (rows of at most LPR entries) still has its entries parked in LDS.
      for (int pb = 0; pb < maxlen; pb += LPR) {
        if (maxlen > LPR) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            Entry e = a.ent[len[i] > 0 ? p0[i] + min(pb + l, len[i] - 1) : 0];
            if (pb + l >= len[i]) e = Entry{0, 0, 0.0};
            ebuf[i * LPR + l] = e;
          }
        }
        const int cnt = (maxlen - pb) < LPR ? (maxlen - pb) : LPR;
        for (int j = 0; j < cnt; ++j) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const Entry e = ebuf[i * LPR + j];
            if (e.slot >= 0 || pb + j >= len[i]) continue;
            const double coef = err[i] * e.x;
            double* hrow = hot + (-1 - e.slot) * hot_w;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              if (fok[c]) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                  if (RFM_KEEP(a, 8)) unsafeAtomicAdd(hrow + fo[c] + v, coef * q[i][c][v]);
              }
            }
            if (l == 0 && RFM_KEEP(a, 8)) {
              unsafeAtomicAdd(hrow + k, coef);
              unsafeAtomicAdd(hrow + k + 1, coef * e.x);
            }
          }
        }
      }
    }
  }

  if (H > 0 && RFM_KEEP(a, 16)) {
    __syncthreads();
    // slab layout [H][gridDim.x][k+2]: the slabs of one column are contiguous
    for (int i = tid; i < H * hot_w; i += BLOCK) {
      const int h = i / hot_w, f = i % hot_w;
      a.hot_slab[(int64_t(h) * gridDim.x + blockIdx.x) * hot_w + f] = hot[i];
    }
  }
  if (a.err_partial) {
    const double s = block_sum<BLOCK>(err_acc, red);
    if (tid == 0) a.err_partial[blockIdx.x] = s;
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

struct SlotRec {  // one slot of the column-major view, 16 B
  double x;       // feature value
  int32_t col;    // feature column
  int32_t pad;
};

struct ConsArgs {
  const WorkItem* items;
  int32_t n_items;
  int32_t* slot_t;
  const SlotRec* slots;
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

template <int VEC, int NC>
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
};

// the lane group owns column `col` (or chunk `part` of it): apply / emit
template <int LPR, int VEC, int NC>
__device__ inline void flush_column(const ColAcc<VEC, NC>& acc, int32_t col, int32_t part,
                                    const ConsArgs& a, int l) {
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

// One work item per LANE GROUP (64/LPR items per wave).  The group streams its
// slot window in pieces of PLANES x LPR slots: the marks of the NEXT piece are
// prefetched while the current one is processed, the per-slot data of all marked
// slots of a piece are fetched together, and the marked slots' contributions
// err*x*[Q[t,:], 1, x] are added strictly in slot order, four Q-row gathers in
// flight at a time.
template <int PLANES, typename T>
__device__ inline T plane_select(const T (&arr)[PLANES], int pl) {
  T v = arr[0];
#pragma unroll
  for (int i = 1; i < PLANES; ++i) v = (pl == i) ? arr[i] : v;
  return v;
}

template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kBlock) void fm_consume_kernel(ConsArgs a) {
  constexpr int GPW = kWave / LPR;
  constexpr int PLANES = LPR >= 64 ? 1 : (LPR == 32 ? 2 : 4);
  constexpr int WIN = PLANES * LPR;
  constexpr int BATCH = 4;
  constexpr unsigned long long GMASK = LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull);
  const int lane = threadIdx.x % kWave;
  const int l = lane % LPR;
  const int g = lane / LPR;
  const int item_id = (blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * GPW + g;
  WorkItem it{0, 0, -1, 0};
  if (item_id < a.n_items) it = a.items[item_id];
  const int k = a.k;

  ColAcc<VEC, NC> acc;
  acc.clear();
  int32_t cur = -1;  // column being accumulated (uniform in the lane group)

  int32_t tk[PLANES], tn[PLANES];
#pragma unroll
  for (int pl = 0; pl < PLANES; ++pl) {
    const int32_t s = it.slot_begin + pl * LPR + l;
    tk[pl] = (s < it.slot_end) ? a.slot_t[s] : -1;
  }
  for (int32_t w0 = it.slot_begin; w0 < it.slot_end; w0 += WIN) {
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) {
      const int32_t s = w0 + WIN + pl * LPR + l;
      tn[pl] = (s < it.slot_end) ? a.slot_t[s] : -1;
    }
    // marked slots of this piece: bit (pl*LPR + lane) of M, i.e. slot order
    unsigned long long M = 0ull;
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) {
      const unsigned long long m = (__ballot(tk[pl] >= 0) >> (g * LPR)) & GMASK;
      M |= m << ((pl * LPR) & 63);
    }
    int32_t col[PLANES];
    double coef[PLANES], cx[PLANES];
    if (M) {
      // per-slot data of the whole piece, loaded unconditionally (clamped) so
      // that the loads of the PLANES planes overlap; unmarked slots are masked
      SlotRec sr[PLANES];
      double ee[PLANES];
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const int32_t s = min(w0 + pl * LPR + l, it.slot_end - 1);
        sr[pl] = a.slots[s];
        ee[pl] = a.err[max(tk[pl], 0)];
      }
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const bool active = tk[pl] >= 0;
        if (active) a.slot_t[w0 + pl * LPR + l] = -1;
        col[pl] = active ? sr[pl].col : -1;
        coef[pl] = active ? ee[pl] * sr[pl].x : 0.0;
        cx[pl] = coef[pl] * sr[pl].x;
      }
    }
    while (M) {
      int bsel[BATCH];
      int nb = 0;
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        bsel[u] = u ? bsel[0] : 0;
        if (M) {
          bsel[u] = __ffsll((long long)M) - 1;
          M &= M - 1;
          ++nb;
        }
      }
      int32_t tt[BATCH], cc[BATCH];
      double ff[BATCH], xx[BATCH];
      Pack<VEC> qq[BATCH][NC];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int pl = bsel[u] / LPR, ln = bsel[u] % LPR;
        tt[u] = __shfl(plane_select<PLANES>(tk, pl), ln, LPR);
        cc[u] = __shfl(plane_select<PLANES>(col, pl), ln, LPR);
        ff[u] = __shfl(plane_select<PLANES>(coef, pl), ln, LPR);
        xx[u] = __shfl(plane_select<PLANES>(cx, pl), ln, LPR);
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
          const int f = (ch * LPR + l) * VEC;
          if (f < k) qq[u][ch].load(a.Q + int64_t(tt[u]) * k + f);
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (u < nb) {
          if (cc[u] != cur) {
            if (cur >= 0) flush_column<LPR, VEC, NC>(acc, cur, it.part, a, l);
            acc.clear();
            cur = cc[u];
          }
#pragma unroll
          for (int ch = 0; ch < NC; ++ch) {
            const int f = (ch * LPR + l) * VEC;
            if (f < k) {
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc.m[ch][v] += ff[u] * qq[u][ch].v[v];
            }
          }
          acc.gw += ff[u];
          acc.d += xx[u];
        }
      }
    }
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) tk[pl] = tn[pl];
  }
  if (cur >= 0) {
    flush_column<LPR, VEC, NC>(acc, cur, it.part, a, l);
  } else if (it.part >= 0) {
    // an untouched chunk still owes its (zero) partial
    flush_column<LPR, VEC, NC>(acc, 0, it.part, a, l);
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
  const double* err_partial;  // [n_slabs] per-workgroup sums of the residual
  int32_t k;
  int64_t n;
  double* w0;
  double* w;
  double* V;
  double lr;
  double* grad;  // nullable
};

// tot[f] = sum_{r<rows} base[r*width + f], f < width, rows contiguous, in a
// fixed order: the block's threads split into row groups x factor lanes, each
// group sums its rows in ascending order (8 loads in flight), the groups are
// then added in group order.
__device__ inline void ordered_rows_sum(const double* base, int rows, int width,
                                        double* scratch /*[kBlock]*/, double* tot /*[width]*/) {
  int fw = 1;
  while (fw < width && fw < kBlock) fw <<= 1;
  const int nsg = kBlock / fw;
  const int sg = threadIdx.x / fw, fl = threadIdx.x % fw;
  for (int f0 = 0; f0 < width; f0 += fw) {
    const int f = f0 + fl;
    double acc = 0.0;
    if (f < width) {
      for (int r = sg; r < rows; r += nsg * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r + u * nsg;
          v[u] = rr < rows ? base[int64_t(rr) * width + f] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
    }
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
// one hot column each; last block: w0 from the forward workgroups' residual sums.
__global__ __launch_bounds__(kBlock) void fm_finalize_kernel(FinArgs a) {
  __shared__ double scratch[kBlock];
  __shared__ double tot[1024 + 2];
  const int k = a.k;
  const int b = blockIdx.x;
  if (b < a.n_split + a.n_hot) {
    int32_t col;
    if (b < a.n_split) {
      const SplitCol sc = a.split[b];
      col = sc.col;
      ordered_rows_sum(a.partials + int64_t(sc.part_begin) * (k + 2), sc.part_count, k + 2,
                       scratch, tot);
    } else {
      const int h = b - a.n_split;
      col = a.hot_cols[h];
      ordered_rows_sum(a.hot_slab + int64_t(h) * a.n_slabs * (k + 2), a.n_slabs, k + 2, scratch,
                       tot);
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
  for (int i = threadIdx.x; i < a.n_slabs; i += kBlock) acc += a.err_partial[i];
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

}  // namespace rfm
