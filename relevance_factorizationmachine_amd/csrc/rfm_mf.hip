// Logistic-MF kernels for gfx950 and their C-ABI launchers.
//
// Layout in HBM: P[n_users][k], Q[n_items][k] row-major f64 (the reference's
// NumPy layout), b_u[n_users], b_i[n_items]; the (user, item) pairs of the log
// are two int32 arrays.  A pair is handled by LPR consecutive lanes (same
// factor-to-lane map as the FM kernels): k=128 -> 64 lanes x 16 bytes.
//
// Training keeps the reference's strictly sequential per-example semantics
// (src/mf.py:97-108) through the level schedule of rfm_mf_schedule: examples
// of one level touch disjoint rows of P, Q, b_u, b_i.  Levels with many
// examples get a grid-wide launch each; runs of small levels are executed by
// ONE workgroup that walks the levels with a barrier in between, so a batch
// whose popular item forms a long chain costs one launch, not one per level.
#include <algorithm>
#include <atomic>
#include <cmath>

#include "rfm_common.h"
#include "rfm_device_utils.hpp"

namespace rfm {

constexpr int kMfBlock = 256;
#ifndef RFM_SEQ_BLOCK
#define RFM_SEQ_BLOCK 1024
#endif
constexpr int kSeqBlock = RFM_SEQ_BLOCK;  // threads of the sequential workgroup
struct MfPredArgs {
  const int32_t* users;
  const int32_t* items;
  const int32_t* row_ids;  // nullable
  int64_t n_rows;
  const double* P;
  const double* Q;
  const double* bu;
  const double* bi;
  double b;
  int32_t k;
  const double* y;
  const double* pscore;
  double* out_pred;      // nullable
  double* loss_partial;  // nullable
  double eps;
};

template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kMfBlock) void mf_predict_kernel(MfPredArgs a) {
  constexpr int GPB = kMfBlock / LPR;
  __shared__ double lds[kMfBlock];
  const int tid = threadIdx.x;
  const int l = tid % LPR;
  const int g = tid / LPR;
  const int k = a.k;
  double loss_acc = 0.0;
  for (int64_t base = int64_t(blockIdx.x) * GPB; base < a.n_rows;
       base += int64_t(gridDim.x) * GPB) {
    const int64_t t = base + g;
    const bool valid = t < a.n_rows;
    int64_t r = 0;
    int32_t u = 0, i = 0;
    if (valid) {
      r = a.row_ids ? int64_t(a.row_ids[t]) : t;
      u = a.users[r];
      i = a.items[r];
    }
    double dot = 0.0;
    if (valid) {
      const double* pu = a.P + int64_t(u) * k;
      const double* qi = a.Q + int64_t(i) * k;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int f = (c * LPR + l) * VEC;
        if (f < k) {
          Pack<VEC> pp, pq;
          pp.load(pu + f);
          pq.load(qi + f);
#pragma unroll
          for (int v = 0; v < VEC; ++v) dot += pp.v[v] * pq.v[v];
        }
      }
    }
    dot = group_sum<LPR>(dot);
    if (valid && l == 0) {
      const double pred = sigmoid_clipped(dot + a.bu[u] + a.bi[i] + a.b);
      if (a.out_pred) a.out_pred[t] = pred;
      if (a.loss_partial) {
        const double rr = a.y[r] / a.pscore[r];
        loss_acc += rr * log(pred + a.eps) + (1.0 - rr) * log(1.0 - pred + a.eps);
      }
    }
  }
  if (a.loss_partial) {
    lds[tid] = loss_acc;
    __syncthreads();
#pragma unroll
    for (int s = kMfBlock / 2; s > 0; s >>= 1) {
      if (tid < s) lds[tid] += lds[tid + s];
      __syncthreads();
    }
    if (tid == 0) a.loss_partial[blockIdx.x] = lds[0];
  }
}

__global__ __launch_bounds__(kMfBlock) void mf_loss_finish_kernel(const double* partial,
                                                                 int n_partial, int64_t n_rows,
                                                                 double* out) {
  __shared__ double lds[kMfBlock];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += kMfBlock) acc += partial[i];
  lds[threadIdx.x] = acc;
  __syncthreads();
#pragma unroll
  for (int s = kMfBlock / 2; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) lds[threadIdx.x] += lds[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = -lds[0] / double(n_rows);
}

struct MfSgdArgs {
  const int32_t* users;
  const int32_t* items;
  const double* y;
  const double* pscore;
  const int32_t* pos_rows;   // batch position -> training row
  const int32_t* order;      // batch positions grouped by level (null: identity)
  const int32_t* level_ptr;  // device copy (sequential kernel only)
  int32_t lo, hi;            // wide kernel: order[lo, hi); seq kernel: levels [lo, hi)
  double* P;
  double* Q;
  double* bu;
  double* bi;
  double b;
  int32_t k;
  double lr, reg;
};

// one example, LPR lanes: src/mf.py:99-108 with :172-216
template <int LPR, int VEC, int NC>
__device__ inline void mf_example(const MfSgdArgs& a, int32_t s, int l) {
  const int k = a.k;
  const int64_t r = a.pos_rows[s];
  const int32_t u = a.users[r], i = a.items[r];
  double* pu = a.P + int64_t(u) * k;
  double* qi = a.Q + int64_t(i) * k;
  Pack<VEC> pp[NC], pq[NC];
  double dot = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = (c * LPR + l) * VEC;
    if (f < k) {
      pp[c].load(pu + f);
      pq[c].load(qi + f);
#pragma unroll
      for (int v = 0; v < VEC; ++v) dot += pp[c].v[v] * pq[c].v[v];
    }
  }
  dot = group_sum<LPR>(dot);
  const double bu = a.bu[u], bi = a.bi[i];
  const double err = a.y[r] / a.pscore[r] - sigmoid_clipped(dot + bu + bi + a.b);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = (c * LPR + l) * VEC;
    if (f < k) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const double p_new = pp[c].v[v] - a.lr * (-err * pq[c].v[v] + a.reg * pp[c].v[v]);
        // the item row sees the user row this example has just updated (src/mf.py:193)
        const double q_new = pq[c].v[v] - a.lr * (-err * p_new + a.reg * pq[c].v[v]);
        pp[c].v[v] = p_new;
        pq[c].v[v] = q_new;
      }
      pp[c].store(pu + f);
      pq[c].store(qi + f);
    }
  }
  if (l == 0) {
    a.bu[u] = bu - a.lr * (-err + a.reg * bu);
    a.bi[i] = bi - a.lr * (-err + a.reg * bi);
  }
}

// one level, many workgroups
template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kMfBlock) void mf_sgd_wide_kernel(MfSgdArgs a) {
  constexpr int GPB = kMfBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  for (int64_t idx = int64_t(a.lo) + int64_t(blockIdx.x) * GPB + g; idx < a.hi;
       idx += int64_t(gridDim.x) * GPB)
    mf_example<LPR, VEC, NC>(a, a.order ? a.order[idx] : int32_t(idx), l);
}

// levels [lo, hi) by one workgroup, barrier between levels
template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kSeqBlock) void mf_sgd_seq_kernel(MfSgdArgs a) {
  constexpr int GPB = kSeqBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  for (int lev = a.lo; lev < a.hi; ++lev) {
    const int32_t b0 = a.level_ptr[lev], b1 = a.level_ptr[lev + 1];
    for (int32_t idx = b0 + g; idx < b1; idx += GPB) mf_example<LPR, VEC, NC>(a, a.order[idx], l);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// level-ordered example records: the fast form of the same schedule
// ---------------------------------------------------------------------------
struct MfEx {       // one example of a batch, stored in level order, 24 B
  int32_t u, i;     // user, item
  int32_t cslot;    // >= 0: LDS cache slot of the item row in the sequential kernel;
                    // -1: the item occurs once in the batch; -2: repeated but not cached
  int32_t gap;      // levels between this example and the previous writer of its user row
                    // (RFM_MF_NO_WRITER if none in the batch): the row is final that far ahead
  double ry;        // label / propensity
};

struct MfExArgs {
  const MfEx* ex;            // [batch], grouped by level
  const int32_t* level_ptr;  // device copy (sequential kernel only)
  int32_t lo, hi;            // wide kernel: ex[lo, hi); seq kernel: levels [lo, hi)
  int32_t rec_lo, rec_hi;    // seq kernel: level_ptr[lo], level_ptr[hi]
  const int32_t* cache_items;  // items whose rows the sequential kernel keeps in LDS
  int32_t n_cached;
  double* P;
  double* Q;
  double* bu;
  double* bi;
  double b;
  int32_t k;
  double lr, reg;
};

// the arithmetic of one example on rows already in registers (src/mf.py:99-108,
// 172-216): returns the residual, rows and biases are updated in place
template <int LPR, int VEC, int NC>
__device__ inline void mf_update(Pack<VEC> (&pp)[NC], Pack<VEC> (&pq)[NC], double& bu,
                                 double& bi, double ry, double b, double lr, double reg, int k,
                                 int l) {
  double dot = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = (c * LPR + l) * VEC;
    if (f < k) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) dot += pp[c].v[v] * pq[c].v[v];
    }
  }
  dot = group_sum<LPR>(dot);
  const double err = ry - sigmoid_clipped(dot + bu + bi + b);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double p_new = pp[c].v[v] - lr * (-err * pq[c].v[v] + reg * pp[c].v[v]);
      // the item row sees the user row this example has just updated (src/mf.py:193)
      const double q_new = pq[c].v[v] - lr * (-err * p_new + reg * pq[c].v[v]);
      pp[c].v[v] = p_new;
      pq[c].v[v] = q_new;
    }
  }
  bu = bu - lr * (-err + reg * bu);
  bi = bi - lr * (-err + reg * bi);
}

template <int LPR, int VEC, int NC>
__device__ inline void mf_load_row(Pack<VEC> (&dst)[NC], const double* row, int k, int l) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = (c * LPR + l) * VEC;
    dst[c].load(row + (f < k ? f : 0));
  }
}

template <int LPR, int VEC, int NC>
__device__ inline void mf_store_row(const Pack<VEC> (&src)[NC], double* row, int k, int l) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = (c * LPR + l) * VEC;
    if (f < k) src[c].store(row + f);
  }
}

// one level, many workgroups (examples of a level touch disjoint rows)
template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kMfBlock) void mf_sgd_wide_ex_kernel(MfExArgs a) {
  constexpr int GPB = kMfBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  const int k = a.k;
  for (int64_t idx = int64_t(a.lo) + int64_t(blockIdx.x) * GPB + g; idx < a.hi;
       idx += int64_t(gridDim.x) * GPB) {
    const MfEx e = a.ex[idx];
    Pack<VEC> pp[NC], pq[NC];
    mf_load_row<LPR, VEC, NC>(pp, a.P + int64_t(e.u) * k, k, l);
    mf_load_row<LPR, VEC, NC>(pq, a.Q + int64_t(e.i) * k, k, l);
    double bu = a.bu[e.u], bi = a.bi[e.i];
    mf_update<LPR, VEC, NC>(pp, pq, bu, bi, e.ry, a.b, a.lr, a.reg, k, l);
    mf_store_row<LPR, VEC, NC>(pp, a.P + int64_t(e.u) * k, k, l);
    mf_store_row<LPR, VEC, NC>(pq, a.Q + int64_t(e.i) * k, k, l);
    if (l == 0) {
      a.bu[e.u] = bu;
      a.bi[e.i] = bi;
    }
  }
}

// ---------------------------------------------------------------------------
// The chain of small levels (the path through a popular item is hundreds of levels
// long), by ONE workgroup with a barrier between levels.  Nothing on the
// level-to-level path waits for an index load:
//  * the chunk's level pointers and example records are copied to LDS first;
//  * the rows (and biases) of the items that occur more than once in the batch
//    live in LDS for the whole launch;
//  * a lane group knows the example it runs kMfAhead levels ahead and loads the
//    user row / bias (when final by then: MfEx.gap) and the row of an item that
//    occurs only once into a register slot of a ring of kMfAhead slots.  The ring
//    is unrolled with fixed slots and its loads are unconditional (unused ones read
//    row 0).
// What remains per level (about 0.7 us) is the dependent arithmetic of one example --
// dot product, lane-group reduction, exp, division, the two row updates -- issued by a
// single wavefront; a one-wavefront variant without barriers, with an LDS ring that
// forwards rows between levels, measured no faster and was dropped.
// ---------------------------------------------------------------------------
constexpr int kMfAhead = RFM_MF_READ_AHEAD;
static_assert(kMfAhead == 4, "the level loop below is unrolled for four slots");
constexpr int kSeqMaxLevels = 1024;  // per launch: level pointers, 4 KiB of LDS
constexpr int kSeqMaxRecs = 1024;    // per launch: example records, 24 KiB of LDS
constexpr int kSeqMaxCacheBytes = 32 << 10;  // (96 KiB measured no faster at k = 16 ... 400: more rows to copy in and write back per launch)

template <int VEC, int NC>
struct MfSlot {
  MfEx e;
  bool has;       // this lane group has an example at the slot's level
  bool got_user;  // pp / bu hold the user row (final)
  bool got_item;  // pq / bi hold the item row (an item that occurs once in the batch)
  Pack<VEC> pp[NC], pq[NC];
  double bu, bi;
};

struct MfSeqLds {
  const int32_t* lptr;  // [n_lev + 1], relative to the chunk's first record
  const MfEx* exs;      // [n_rec]
  double* qcache;       // [n_cached][k+2]: item row, item bias, pad
  int32_t n_lev;
};

// what lane group g runs at chunk level t, and its rows if they may be read now: the
// user row is final when its previous writer lies more than `ahead` levels back
template <int LPR, int VEC, int NC, bool PF>
__device__ __forceinline__ void mf_slot_fetch(const MfExArgs& a, const MfSeqLds& m, int t, int g,
                                              int l, int ahead, MfSlot<VEC, NC>& s) {
  int32_t idx = 0;
  bool has = false;
  if (t < m.n_lev) {
    idx = m.lptr[t] + g;
    has = idx < m.lptr[t + 1];
  }
  s.e = m.exs[has ? idx : 0];
  s.has = has;
  s.got_user = PF && has && s.e.gap > ahead;
  s.got_item = PF && has && s.e.cslot == -1;
  // a wavefront none of whose lane groups has an example at that level issues nothing
  // (the condition is uniform in the wavefront, so the loads of an active one stay
  // unconditional and together)
  if (PF && __ballot(has) != 0ull) {
    const int k = a.k;
    const int32_t u = s.got_user ? s.e.u : 0;
    const int32_t it = s.got_item ? s.e.i : 0;
    mf_load_row<LPR, VEC, NC>(s.pp, a.P + int64_t(u) * k, k, l);
    s.bu = a.bu[u];
    mf_load_row<LPR, VEC, NC>(s.pq, a.Q + int64_t(it) * k, k, l);
    s.bi = a.bi[it];
  }
}

template <int LPR, int VEC, int NC>
__device__ __forceinline__ void mf_slot_run(const MfExArgs& a, const MfSeqLds& m, int l,
                                            MfSlot<VEC, NC>& s) {
  if (!s.has) return;
  const int k = a.k;
  const MfEx e = s.e;
  if (!s.got_user) {  // written within the last kMfAhead levels: read it now
    mf_load_row<LPR, VEC, NC>(s.pp, a.P + int64_t(e.u) * k, k, l);
    s.bu = a.bu[e.u];
  }
  if (e.cslot < 0 && !s.got_item) {  // item row through memory
    mf_load_row<LPR, VEC, NC>(s.pq, a.Q + int64_t(e.i) * k, k, l);
    s.bi = a.bi[e.i];
  }
  double* crow = m.qcache + (e.cslot >= 0 ? e.cslot : 0) * (k + 2);
  if (e.cslot >= 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int f = (c * LPR + l) * VEC;
      s.pq[c].load(crow + (f < k ? f : 0));
    }
    s.bi = crow[k];
  }
  mf_update<LPR, VEC, NC>(s.pp, s.pq, s.bu, s.bi, e.ry, a.b, a.lr, a.reg, k, l);
  mf_store_row<LPR, VEC, NC>(s.pp, a.P + int64_t(e.u) * k, k, l);
  if (l == 0) a.bu[e.u] = s.bu;
  if (e.cslot >= 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int f = (c * LPR + l) * VEC;
      if (f < k) s.pq[c].store(crow + f);
    }
    if (l == 0) crow[k] = s.bi;
  } else {
    mf_store_row<LPR, VEC, NC>(s.pq, a.Q + int64_t(e.i) * k, k, l);
    if (l == 0) a.bi[e.i] = s.bi;
  }
}

// levels [lo, hi) (at most kSeqMaxLevels, kSeqMaxRecs examples, each level of at most
// kSeqBlock/LPR examples); a.rec_lo / a.rec_hi = level_ptr[lo] / level_ptr[hi]
// (BLOCK: 1024 threads for one chunk of factors per lane; 512 for two to four chunks -- the
// reference's published k = 300 / 400 -- whose ring of four slots needs 128 + VGPRs: eight lane
// groups then, and more than four chunks run without the ring)
constexpr int seq_block(int nc) { return nc > 1 ? 512 : kSeqBlock; }

template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(seq_block(NC)) void mf_sgd_seq_ex_kernel(MfExArgs a) {
  extern __shared__ double seq_lds[];
  constexpr int kSeqBlock = seq_block(NC);  // (shadows the one-chunk constant inside this kernel)
  // rows wider than four chunks per lane leave no registers for a ring of them
  constexpr bool PF = NC <= 4;
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  const int k = a.k;
  const int cw = k + 2;
  const int n_lev = a.hi - a.lo;
  const int n_rec = a.rec_hi - a.rec_lo;
  int32_t* lptr = reinterpret_cast<int32_t*>(seq_lds);
  MfEx* exs = reinterpret_cast<MfEx*>(seq_lds + (n_lev + 2) / 2);
  double* qcache = reinterpret_cast<double*>(exs + n_rec);
  for (int i = threadIdx.x; i <= n_lev; i += kSeqBlock) lptr[i] = a.level_ptr[a.lo + i] - a.rec_lo;
  {
    const double* src = reinterpret_cast<const double*>(a.ex + a.rec_lo);
    double* dst = reinterpret_cast<double*>(exs);
    for (int i = threadIdx.x; i < n_rec * int(sizeof(MfEx) / 8); i += kSeqBlock) dst[i] = src[i];
  }
  for (int i = threadIdx.x; i < a.n_cached * cw; i += kSeqBlock) {
    const int c = i / cw, f = i % cw;
    const int32_t item = a.cache_items[c];
    qcache[i] = f < k ? a.Q[int64_t(item) * k + f] : (f == k ? a.bi[item] : 0.0);
  }
  __syncthreads();
  const MfSeqLds m{lptr, exs, qcache, n_lev};

  MfSlot<VEC, NC> s0, s1, s2, s3;
  // every level before lo has run, so the first level's rows are final; a row of one
  // of the next levels is final if its previous writer lies before this launch
  mf_slot_fetch<LPR, VEC, NC, PF>(a, m, 0, g, l, 0, s0);
  mf_slot_fetch<LPR, VEC, NC, PF>(a, m, 1, g, l, 1, s1);
  mf_slot_fetch<LPR, VEC, NC, PF>(a, m, 2, g, l, 2, s2);
  mf_slot_fetch<LPR, VEC, NC, PF>(a, m, 3, g, l, 3, s3);
#ifdef RFM_MF_PROBE_NOBARRIER  // timing probe only (results are NOT valid): no barrier between two one-example levels
#define RFM_MF_SYNC(tt) do { const int t_ = (tt); if (!(t_ + 2 <= n_lev && lptr[t_ + 1] - lptr[t_] == 1 && lptr[t_ + 2] - lptr[t_ + 1] == 1)) __syncthreads(); } while (0)
#else
#define RFM_MF_SYNC(tt) __syncthreads()
#endif
  for (int t = 0; t < n_lev; t += kMfAhead) {
    // the rows of level t+4 are read while level t is still running: final if the
    // previous writer lies before level t, i.e. more than four levels back
    mf_slot_run<LPR, VEC, NC>(a, m, l, s0);
    mf_slot_fetch<LPR, VEC, NC, PF>(a, m, t + 4, g, l, kMfAhead, s0);
    RFM_MF_SYNC(t);
    if (t + 1 >= n_lev) break;
    mf_slot_run<LPR, VEC, NC>(a, m, l, s1);
    mf_slot_fetch<LPR, VEC, NC, PF>(a, m, t + 5, g, l, kMfAhead, s1);
    RFM_MF_SYNC(t + 1);
    if (t + 2 >= n_lev) break;
    mf_slot_run<LPR, VEC, NC>(a, m, l, s2);
    mf_slot_fetch<LPR, VEC, NC, PF>(a, m, t + 6, g, l, kMfAhead, s2);
    RFM_MF_SYNC(t + 2);
    if (t + 3 >= n_lev) break;
    mf_slot_run<LPR, VEC, NC>(a, m, l, s3);
    mf_slot_fetch<LPR, VEC, NC, PF>(a, m, t + 7, g, l, kMfAhead, s3);
    RFM_MF_SYNC(t + 3);
  }
  __syncthreads();
  // write the cached item rows back
  for (int i = threadIdx.x; i < a.n_cached * cw; i += kSeqBlock) {
    const int c = i / cw, f = i % cw;
    const int32_t item = a.cache_items[c];
    if (f < k)
      a.Q[int64_t(item) * k + f] = qcache[i];
    else if (f == k)
      a.bi[item] = qcache[i];
  }
}

// the sequential kernel with its dynamic-LDS limit raised where the item cache needs it
// (remembered per device and instantiation)
template <int L, int Vv, int N>
static void launch_seq_ex(rfm_ctx* ctx, const MfExArgs& a, int threads, size_t lds) {
  const auto kern = &mf_sgd_seq_ex_kernel<L, Vv, N>;
  static std::atomic<bool> raised[64];
  const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (lds > (64u << 10) && !raised[dev].load(std::memory_order_relaxed)) {
    RFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    raised[dev].store(true, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, ctx->stream, a);
}

static void mf_predict_launch(rfm_ctx* ctx, MfPredArgs a, double* d_out_loss) {
  if (a.n_rows <= 0) return;
  const Shape s = shape_for(a.k);
  const int gpb = kMfBlock / s.lpr;
  const int grid = int(std::max<int64_t>(
      1, std::min<int64_t>((a.n_rows + gpb - 1) / gpb, int64_t(ctx->n_cu) * 8)));
  if (d_out_loss) {
    ctx->loss_partials.ensure(size_t(ctx->n_cu) * 8 * sizeof(double));
    a.loss_partial = ctx->loss_partials.as<double>();
  }
#define RFM_CALL_MFP(L, Vv, N) \
  hipLaunchKernelGGL((mf_predict_kernel<L, Vv, N>), dim3(grid), dim3(kMfBlock), 0, ctx->stream, a)
  RFM_FOR_SHAPE(s, RFM_CALL_MFP);
#undef RFM_CALL_MFP
  if (d_out_loss)
    hipLaunchKernelGGL(mf_loss_finish_kernel, dim3(1), dim3(kMfBlock), 0, ctx->stream,
                       ctx->loss_partials.as<double>(), grid, a.n_rows, d_out_loss);
  RFM_HIP_CHECK(hipGetLastError());
}

}  // namespace rfm

using namespace rfm;

// Replica merge of the user-partitioned MF mode: every rank runs the
// exact sequential SGD on the examples of ITS users against its own replica of Q / b_i;
// after the batch the replicas' changes are added up.
//   delta: out = cur - sync                 (what this rank's examples changed)
//   merge: sync += total; cur = sync        (total = all ranks' deltas, all-reduced)
__global__ __launch_bounds__(256) void mf_delta_kernel(const double* cur, const double* sync,
                                                      double* out, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
    out[i] = cur[i] - sync[i];
}
__global__ __launch_bounds__(256) void mf_merge_kernel(double* cur, double* sync,
                                                      const double* total, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const double v = sync[i] + total[i];
    sync[i] = v;
    cur[i] = v;
  }
}

extern "C" {

int32_t rfm_mf_predict(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                       const int32_t* d_row_ids, int64_t n_rows, const double* d_P,
                       const double* d_Q, const double* d_bu, const double* d_bi, double b,
                       int32_t n_factors, double* d_out_pred) {
  return guarded([&] {
    RFM_REQUIRE(ctx, "null ctx");
    RFM_REQUIRE(n_rows >= 0, "negative n_rows");
    if (n_rows == 0) return;
    RFM_REQUIRE(d_P && d_Q && d_bu && d_bi && d_out_pred, "null pointer");
    RFM_REQUIRE(d_users && d_items, "null pair arrays");
    MfPredArgs a{};
    a.users = d_users;
    a.items = d_items;
    a.row_ids = d_row_ids;
    a.n_rows = n_rows;
    a.P = d_P;
    a.Q = d_Q;
    a.bu = d_bu;
    a.bi = d_bi;
    a.b = b;
    a.k = n_factors;
    a.out_pred = d_out_pred;
    mf_predict_launch(ctx, a, nullptr);
  });
}

int32_t rfm_mf_predict_loss(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                            const double* d_y, const double* d_pscore,
                            const int32_t* d_row_ids, int64_t n_rows, const double* d_P,
                            const double* d_Q, const double* d_bu, const double* d_bi,
                            double b, int32_t n_factors, double eps, double* d_out_pred,
                            double* d_out_loss) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_users && d_items && d_y && d_pscore && d_P && d_Q && d_bu && d_bi &&
                    d_out_loss,
                "null pointer");
    RFM_REQUIRE(n_rows >= 1, "loss of zero rows");
    MfPredArgs a{};
    a.users = d_users;
    a.items = d_items;
    a.row_ids = d_row_ids;
    a.n_rows = n_rows;
    a.P = d_P;
    a.Q = d_Q;
    a.bu = d_bu;
    a.bi = d_bi;
    a.b = b;
    a.k = n_factors;
    a.y = d_y;
    a.pscore = d_pscore;
    a.eps = eps;
    a.out_pred = d_out_pred;
    mf_predict_launch(ctx, a, d_out_loss);
  });
}

int32_t rfm_mf_sgd_levels(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                          const double* d_y, const double* d_pscore,
                          const int32_t* d_pos_rows, const int32_t* d_order,
                          const int32_t* h_level_ptr, const int32_t* d_level_ptr,
                          int32_t n_levels, double* d_P, double* d_Q, double* d_bu,
                          double* d_bi, double b, int32_t n_factors, double lr, double reg) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_users && d_items && d_y && d_pscore && d_pos_rows && d_order &&
                    h_level_ptr && d_level_ptr && d_P && d_Q && d_bu && d_bi,
                "null pointer");
    RFM_REQUIRE(n_levels >= 0, "negative n_levels");
    const Shape s = shape_for(n_factors);
    MfSgdArgs a{};
    a.users = d_users;
    a.items = d_items;
    a.y = d_y;
    a.pscore = d_pscore;
    a.pos_rows = d_pos_rows;
    a.order = d_order;
    a.level_ptr = d_level_ptr;
    a.P = d_P;
    a.Q = d_Q;
    a.bu = d_bu;
    a.bi = d_bi;
    a.b = b;
    a.k = n_factors;
    a.lr = lr;
    a.reg = reg;
    // a level is "small" when one pass of the sequential workgroup covers it
    const int seq_cap = 2 * (kSeqBlock / s.lpr);
    int lev = 0;
    while (lev < n_levels) {
      const int cnt = h_level_ptr[lev + 1] - h_level_ptr[lev];
      RFM_REQUIRE(cnt >= 0, "level_ptr not monotone");
      if (cnt > seq_cap) {
        a.lo = h_level_ptr[lev];
        a.hi = h_level_ptr[lev + 1];
        const int gpb = kMfBlock / s.lpr;
        const int grid = std::min((cnt + gpb - 1) / gpb, ctx->n_cu * 8);
#define RFM_CALL_WIDE(L, Vv, N)                                                               \
  hipLaunchKernelGGL((mf_sgd_wide_kernel<L, Vv, N>), dim3(grid), dim3(kMfBlock), 0, ctx->stream, \
                     a)
        RFM_FOR_SHAPE(s, RFM_CALL_WIDE);
#undef RFM_CALL_WIDE
        ++lev;
      } else {
        int end = lev;
        while (end < n_levels && h_level_ptr[end + 1] - h_level_ptr[end] <= seq_cap) ++end;
        a.lo = lev;
        a.hi = end;
#define RFM_CALL_SEQ(L, Vv, N) \
  hipLaunchKernelGGL((mf_sgd_seq_kernel<L, Vv, N>), dim3(1), dim3(kSeqBlock), 0, ctx->stream, a)
        RFM_FOR_SHAPE(s, RFM_CALL_SEQ);
#undef RFM_CALL_SEQ
        lev = end;
      }
    }
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_mf_cache_capacity(int32_t n_factors, int32_t* h_out) {
  return guarded([&] {
    RFM_REQUIRE(h_out && n_factors >= 1, "bad arguments");
    *h_out = int32_t(std::min<int64_t>(1024, int64_t(kSeqMaxCacheBytes) / (int64_t(n_factors + 2) * 8)));
  });
}

int32_t rfm_mf_sgd_levels_ex(rfm_ctx* ctx, const void* d_ex, const int32_t* h_level_ptr,
                             const int32_t* d_level_ptr, int32_t n_levels,
                             const int32_t* d_cache_items, int32_t n_cached, double* d_P,
                             double* d_Q, double* d_bu, double* d_bi, double b,
                             int32_t n_factors, double lr, double reg) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_ex && h_level_ptr && d_level_ptr && d_P && d_Q && d_bu && d_bi,
                "null pointer");
    RFM_REQUIRE(n_levels >= 0 && n_cached >= 0 && (n_cached == 0 || d_cache_items),
                "bad schedule");
    const Shape s = shape_for(n_factors);
    const size_t cache_bytes = size_t(n_cached) * size_t(n_factors + 2) * sizeof(double);
    RFM_REQUIRE(cache_bytes <= size_t(kSeqMaxCacheBytes),
                "item cache of %d rows exceeds %d bytes of LDS", n_cached, kSeqMaxCacheBytes);
    MfExArgs a{};
    a.ex = static_cast<const MfEx*>(d_ex);
    a.level_ptr = d_level_ptr;
    a.cache_items = d_cache_items;
    a.n_cached = n_cached;
    a.P = d_P;
    a.Q = d_Q;
    a.bu = d_bu;
    a.bi = d_bi;
    a.b = b;
    a.k = n_factors;
    a.lr = lr;
    a.reg = reg;
    // a level is "small" when the sequential workgroup covers it in one pass
    const int seq_threads = seq_block(s.nc);
    const int seq_cap = seq_threads / s.lpr;
    int lev = 0;
    while (lev < n_levels) {
      const int cnt = h_level_ptr[lev + 1] - h_level_ptr[lev];
      RFM_REQUIRE(cnt >= 0, "level_ptr not monotone");
      if (cnt > seq_cap) {
        a.lo = h_level_ptr[lev];
        a.hi = h_level_ptr[lev + 1];
        const int gpb = kMfBlock / s.lpr;
        const int grid = std::min((cnt + gpb - 1) / gpb, ctx->n_cu * 8);
#define RFM_CALL_WIDE_EX(L, Vv, N)                                                            \
  hipLaunchKernelGGL((mf_sgd_wide_ex_kernel<L, Vv, N>), dim3(grid), dim3(kMfBlock), 0,        \
                     ctx->stream, a)
        RFM_FOR_SHAPE(s, RFM_CALL_WIDE_EX);
#undef RFM_CALL_WIDE_EX
        ++lev;
      } else {
        // a chunk of consecutive small levels that fits the kernel's LDS tables
        int end = lev;
        while (end < n_levels && end - lev < kSeqMaxLevels) {
          const int c = h_level_ptr[end + 1] - h_level_ptr[end];
          RFM_REQUIRE(c >= 0, "level_ptr not monotone");
          if (c > seq_cap || h_level_ptr[end + 1] - h_level_ptr[lev] > kSeqMaxRecs) break;
          ++end;
        }
        a.lo = lev;
        a.hi = end;
        a.rec_lo = h_level_ptr[lev];
        a.rec_hi = h_level_ptr[end];
        const size_t lds = size_t((end - lev + 2) / 2) * 8 + size_t(a.rec_hi - a.rec_lo) * sizeof(MfEx) +
                           cache_bytes;
#define RFM_CALL_SEQ_EX(L, Vv, N) launch_seq_ex<L, Vv, N>(ctx, a, seq_threads, lds)
        RFM_FOR_SHAPE(s, RFM_CALL_SEQ_EX);
#undef RFM_CALL_SEQ_EX
        lev = end;
      }
    }
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_mf_sgd_hogwild(rfm_ctx* ctx, const int32_t* d_users, const int32_t* d_items,
                           const double* d_y, const double* d_pscore,
                           const int32_t* d_pos_rows, int64_t batch, double* d_P, double* d_Q,
                           double* d_bu, double* d_bi, double b, int32_t n_factors, double lr,
                           double reg) {
  return guarded([&] {
    RFM_REQUIRE(ctx && d_users && d_items && d_y && d_pscore && d_pos_rows && d_P && d_Q &&
                    d_bu && d_bi,
                "null pointer");
    RFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31), "batch out of range");
    if (batch == 0) return;
    const Shape s = shape_for(n_factors);
    MfSgdArgs a{};
    a.users = d_users;
    a.items = d_items;
    a.y = d_y;
    a.pscore = d_pscore;
    a.pos_rows = d_pos_rows;
    a.order = nullptr;  // batch order, all examples at once
    a.lo = 0;
    a.hi = int32_t(batch);
    a.P = d_P;
    a.Q = d_Q;
    a.bu = d_bu;
    a.bi = d_bi;
    a.b = b;
    a.k = n_factors;
    a.lr = lr;
    a.reg = reg;
    const int gpb = kMfBlock / s.lpr;
    const int grid = int(std::min<int64_t>((batch + gpb - 1) / gpb, int64_t(ctx->n_cu) * 8));
#define RFM_CALL_HOG(L, Vv, N)                                                                \
  hipLaunchKernelGGL((mf_sgd_wide_kernel<L, Vv, N>), dim3(grid), dim3(kMfBlock), 0, ctx->stream, \
                     a)
    RFM_FOR_SHAPE(s, RFM_CALL_HOG);
#undef RFM_CALL_HOG
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_mf_delta(rfm_ctx* ctx, const double* d_cur, const double* d_sync, double* d_out,
                     int64_t count) {
  return guarded([&] {
    RFM_REQUIRE(ctx && count >= 0, "bad argument");
    if (count == 0) return;
    RFM_REQUIRE(d_cur && d_sync && d_out, "null pointer");
    const int grid = int(std::min<int64_t>((count + 255) / 256, int64_t(ctx->n_cu) * 16));
    hipLaunchKernelGGL(mf_delta_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_cur, d_sync, d_out,
                       count);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

int32_t rfm_mf_merge(rfm_ctx* ctx, double* d_cur, double* d_sync, const double* d_total,
                     int64_t count) {
  return guarded([&] {
    RFM_REQUIRE(ctx && count >= 0, "bad argument");
    if (count == 0) return;
    RFM_REQUIRE(d_cur && d_sync && d_total, "null pointer");
    const int grid = int(std::min<int64_t>((count + 255) / 256, int64_t(ctx->n_cu) * 16));
    hipLaunchKernelGGL(mf_merge_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_cur, d_sync,
                       d_total, count);
    RFM_HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
