// Device code of the FM path (gfx950).  See rfm_fm.hip for the launch side and
// DESIGN.md section 4 for the algorithm.  Reference arithmetic: src/fm.py:80-88,
// 114-187, src/base.py:37-66.
#pragma once

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "rfm_device_utils.hpp"
#include "rfm_fm_records.h"

namespace rfm {

// ---------------------------------------------------------------------------
// helpers (lane-group sums, packs, sigmoid: rfm_device_utils.hpp)
// ---------------------------------------------------------------------------
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

__device__ inline double logloss_term(double y, double p, double pred, double eps) {
  const double r = y / p;
  return r * log(pred + eps) + (1.0 - r) * log(1.0 - pred + eps);
}

// waves per SIMD the big forward shape is compiled for (register budget 512/N)
#ifndef RFM_FWD_BIG_BLOCK
#define RFM_FWD_BIG_BLOCK 1024
#endif
constexpr int kBigBlock = RFM_FWD_BIG_BLOCK;  // threads of the forward's many-rows-in-flight shape
#ifndef RFM_FWD_BIG_WAVES
#define RFM_FWD_BIG_WAVES 4
#endif
// rows per lane group of the big forward shape (single-chunk factor counts)
#ifndef RFM_FWD_ROWS
#define RFM_FWD_ROWS 2
#endif

// entries of a row whose V gathers the one-row forward shape keeps in flight
#ifndef RFM_FWD_BIG_UNROLL
#define RFM_FWD_BIG_UNROLL 3
#endif
#ifndef RFM_FWD_SMALL_UNROLL
#define RFM_FWD_SMALL_UNROLL 8
#endif
// ... of the one-row shape at two to four chunks of factors per lane (k = 300 / 400: an entry is
// NC gathers per lane; the shape runs about two waves per SIMD, so registers are not the limit)
#ifndef RFM_FWD_SMALL_WIDE_UNROLL
#define RFM_FWD_SMALL_WIDE_UNROLL 2
#endif

// rows a lane group works on concurrently (independent load chains in flight)
constexpr int rows_in_flight(int nc) { return nc == 1 ? RFM_FWD_ROWS : 1; }
// ... of the PLAIN forward (the caller's CSR arrays: predict, validation loss), which keeps no Q
// row, leaves no marks and sums no hot class, so its registers hold more rows
#ifndef RFM_FWD_ROWS_PLAIN
#define RFM_FWD_ROWS_PLAIN RFM_FWD_ROWS
#endif
#ifndef RFM_FWD_PLAIN_UNROLL
#define RFM_FWD_PLAIN_UNROLL RFM_FWD_BIG_UNROLL
#endif
constexpr int rows_in_flight_plain(int nc) { return nc == 1 ? RFM_FWD_ROWS_PLAIN : 1; }

// Hot-class sums in a fixed order (training forward): the rows a workgroup holds in one trip
// leave their Q rows and per-entry coefficients in LDS, every hot column gets the set of those
// rows that hold it (a bitmap) and one lane group -- its owner -- adds them up in row order.
// Possible when a lane holds one chunk of factors and a trip is at most 128 rows; the other
// shapes add with LDS float atomics (arrival order).
constexpr bool hot_fixed_order(int lpr, int nc, int block, int rows) {
  return nc == 1 && (block / lpr) * rows <= 128;
}
#ifndef RFM_HOT_U
#define RFM_HOT_U 2
#endif
constexpr int kHotPosPad = 4;  // bytes between the position rows of two columns (bank spread)
// LDS bytes of the forward kernel (what a launch asks for)
inline size_t forward_lds_bytes(int block, int lpr, int vec, int nc, int rows, int n_hot, int k,
                                bool fixed_order) {
  size_t bytes = size_t(block) * 8 + (size_t(block) * rows + size_t(block / lpr)) * sizeof(Entry) +
                 size_t(n_hot) * size_t(k + 2) * 8;
  // (many-rows shape without a hot class -- the loss forwards: two stages of a trip's
  // {score, label, propensity}, see the kernel)
  if (block == kBigBlock && n_hot == 0) bytes += 2 * size_t(block / lpr) * rows * 24;
  if (fixed_order && n_hot > 0 && hot_fixed_order(lpr, nc, block, rows)) {
    const size_t rt = size_t(block / lpr) * rows;
    bytes += rt * size_t(lpr * vec) * 8;                                 // Q rows of the trip
    if (rt / 32 > 1) bytes += size_t(block / lpr) * size_t(k + 2) * 8;   // cell sums
    bytes += ((size_t(n_hot) * ((rt + 31) / 32) + 1) & ~size_t(1)) * 4;  // row sets
    bytes += (size_t(n_hot) * (rt + kHotPosPad) + 7) / 8 * 8;            // entry positions
  }
  return bytes;
}

// ---------------------------------------------------------------------------
// 1. forward (+ residual, Q, slot marks, hot sums, loss partials)
// ---------------------------------------------------------------------------
struct FwdArgs {
  // the log: either the plan's records (training) ...
  const Entry* ent;
  const RowRec* rows;
  const char* ell;       // ... in their padded form when every row fits one round: row blocks
  int64_t ell_stride;    //     Entry[LPR] of this many bytes (then ent / rows are null), with
  const double2* ell_yp; //     the rows' {label, propensity} pairs (one line per row)
  // ... or the caller's CSR arrays (+ labels / propensities when a loss is asked for)
  const int64_t* indptr;
  const int32_t* indices;
  const double* values;
  const double* y;
  const double* pscore;
  const int32_t* row_ids;  // may be null: row t
  int64_t n_rows;
  // SEG form (loss forwards of fit(), caller's CSR arrays): rows [0, n_rows_a) are rows row_ids[t]
  // of the log above, rows [n_rows_a, n_rows) are rows t - n_rows_a of a SECOND log; each part
  // has its own loss partials -- the train-loss and validation-loss forwards in one launch
  int64_t n_rows_a;
  const int64_t* indptr2;
  const int32_t* indices2;
  const double* values2;
  const double* y2;
  const double* pscore2;
  double* loss_partial2;
  const double* w0;
  const double* w;
  const double* V;
  int32_t k;
  double* out_pred;      // nullable
  double* out_err;       // nullable
  double* out_Q;         // nullable [n_rows][k]
  SlotMark* slot_mark;   // nullable: leave {t, residual} at the slot of every sparse-class entry ...
  unsigned long long* slot_bits;  // ... and set the slot's bit (what fm_consume_kernel scans)
  int32_t n_hot;         // hot columns (training step only)
  int32_t hot_rounds;    // rounds of LPR entries that cover the longest row of the plan
  int32_t hot_fixed;     // 1: the plan asks for fixed-order hot sums (the DET instantiations)
  double* hot_slab;      // [n_hot][gridDim.x][k+2]
  double* err_partial;   // nullable: [gridDim.x] per-workgroup sums of the residual (for w0)
  double* loss_partial;  // nullable: [gridDim.x]
  double eps;
  int32_t ablate;  // -DRFM_ABLATE builds only: bit mask of parts to skip (timing experiments)
  // XTRA form (with REC, one-row shape): workgroups grid_main .. gridDim.x - 1 only SCORE the rows
  // row_ids_x[0 .. n_rows_x) of the same plan (-> out_pred_x) -- no Q rows, marks, hot sums or
  // residuals: the train-loss forward of the previous batch, which reads the same parameters as
  // this step's forward, rides in its launch (rfm_fm_train)
  int32_t grid_main;
  const int32_t* row_ids_x;
  int64_t n_rows_x;
  double* out_pred_x;
  // ... and, after grid_x of those, workgroups that score EVERY row of another log given as row and
  // entry records (the validation log: rfm_fm_plan_register_log) -> out_pred_y
  int32_t grid_x;
  const RowRec* rows_y;
  const Entry* ent_y;
  int64_t n_rows_y;
  double* out_pred_y;
#ifdef RFM_FWD_STAMPS
  long long* stamps;  // -DRFM_FWD_STAMPS builds only: [workgroup][wave][trip 0/1][8] clock readings
#endif
};

#ifdef RFM_ABLATE
#define RFM_KEEP(a, bit) (((a).ablate & (bit)) == 0)
#else
#define RFM_KEEP(a, bit) true
#endif
// bits: 1 slot marks, 2 Q store, 4 V gathers, 8 hot LDS adds, 16 slab store, 32 hot pass,
// 512 no adds of the five most frequent columns after a workgroup's first trip

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
// LDS (dynamic): red[BLOCK] f64 | entry buffer [BLOCK/LPR][R*LPR+1] Entry | hot sums [H][k+2] f64
// ELL (with REC): the records come as padded row blocks (a.ell).
// DET (with REC): the hot-class sums in a fixed order (hot_fixed_order) instead of LDS atomics.
// SEG (without REC, many-rows shape): two logs in one launch, see FwdArgs.
template <int LPR, int VEC, int NC, int BLOCK, int R, bool REC, bool ELL = false, bool DET = false,
          bool SEG = false, bool XTRA = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == kBigBlock ? RFM_FWD_BIG_WAVES : 1)) void fm_forward_kernel(
    FwdArgs a) {
  constexpr int GPB = BLOCK / LPR;  // lane groups per block
  // workgroup index and count among the workgroups of its kind (XTRA: the step's, or the ones
  // that only score the extra rows -- a workgroup-uniform switch of what the arguments mean)
  unsigned bx = blockIdx.x, gx = gridDim.x;
  if constexpr (XTRA && REC) {
    if (int(blockIdx.x) >= a.grid_main) {
      const bool second = int(blockIdx.x) >= a.grid_main + a.grid_x;
      bx = blockIdx.x - a.grid_main - (second ? a.grid_x : 0);
      gx = second ? gridDim.x - a.grid_main - a.grid_x : a.grid_x;
      a.row_ids = second ? nullptr : a.row_ids_x;
      a.n_rows = second ? a.n_rows_y : a.n_rows_x;
      if (second) {
        a.rows = a.rows_y;
        a.ent = a.ent_y;
      }
      a.n_hot = 0;
      a.slot_mark = nullptr;
      a.slot_bits = nullptr;
      a.out_Q = nullptr;
      a.out_err = nullptr;
      a.err_partial = nullptr;
      a.loss_partial = nullptr;
      a.out_pred = second ? a.out_pred_y : a.out_pred_x;
    } else {
      gx = a.grid_main;
    }
  }
  constexpr bool seg = SEG && !REC;
  extern __shared__ double dyn_lds[];
  const int tid = threadIdx.x;
  const int l = tid % LPR;
  const int g = tid / LPR;
  const int k = a.k;
  const int hot_w = k + 2;
  const int H = a.n_hot;
  double* red = dyn_lds;
  // this group's [R][LPR] entry records; groups are one record apart in bank space so that
  // the broadcast reads of different groups do not collide
  Entry* ebuf = reinterpret_cast<Entry*>(dyn_lds + BLOCK) + g * (R * LPR + 1);
  double* hot = dyn_lds + BLOCK + 2 * (BLOCK * R + GPB);
  // fixed-order hot sums (see hot_fixed_order): Q rows | row sets | entry positions
  constexpr bool FIX = DET && REC && hot_fixed_order(LPR, NC, BLOCK, R);
  constexpr int RT = GPB * R;            // rows of a trip
  constexpr int KS = LPR * VEC;          // doubles of a parked Q row
  constexpr int NW = (RT + 31) / 32;     // 32-bit words of a column's row set
  constexpr int NCELL = RT / 32 > 1 ? RT / 32 : 1;  // cells (32 rows) a frequent column is cut into
  constexpr int NHV = NCELL > 1 ? GPB / NCELL : 0;  // columns cut into cells: one cell per group
  constexpr int PS = RT + kHotPosPad;    // bytes of a column's position row
  double* Qs = hot + H * hot_w;
  double* part = Qs + RT * KS;           // [GPB][k+2] cell sums (NCELL > 1)
  unsigned int* hbits = reinterpret_cast<unsigned int*>(part + (NCELL > 1 ? GPB * hot_w : 0));
  uint8_t* hpos = reinterpret_cast<uint8_t*>(hbits + ((H * NW + 1) & ~1));
  const double w0 = a.w0[0];
  const int64_t last_row = a.n_rows - 1;
  double loss_acc = 0.0, loss_acc2 = 0.0, err_acc = 0.0;
  // many-rows shape, loss asked for, no hot class (whose sums would sit where the stages do)
  const bool stage_loss = BLOCK == kBigBlock && a.loss_partial != nullptr && H == 0;
  double* lstage = hot;  // [2][RT][3]
  int trip = 0;
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
    if (FIX)
      for (int i = tid; i < H * NW; i += BLOCK) hbits[i] = 0u;
    __syncthreads();
  }
  // The wavefronts of a workgroup run their trips in lockstep: all sixteen gather (memory latency,
  // LDS idle), then all sixteen add their hot entries (LDS atomics saturated, nobody gathers).  Half
  // of them starting one gather phase late puts the two kinds of work side by side
  // (RFM_FWD_STAGGER = the delay in units of 64 clocks; 0: none).
#ifndef RFM_FWD_STAGGER
#define RFM_FWD_STAGGER 0
#endif
  if constexpr (RFM_FWD_STAGGER > 0 && REC && BLOCK == kBigBlock) {
    if (H > 0 && tid / kWave >= BLOCK / kWave / 2) {
#pragma unroll
      for (int i = 0; i < (RFM_FWD_STAGGER + 126) / 127; ++i)
        __builtin_amdgcn_s_sleep(RFM_FWD_STAGGER < 127 ? RFM_FWD_STAGGER : 127);
    }
  }

#ifdef RFM_FWD_STAMPS
  int trip_no = 0;
#define RFM_FSTAMP(i) do { if (a.stamps && (tid & (kWave - 1)) == 0 && trip_no < 2) a.stamps[((int64_t(blockIdx.x) * (BLOCK / kWave) + tid / kWave) * 2 + trip_no) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define RFM_FSTAMP(i) do { } while (0)
#endif
  for (int64_t base = int64_t(bx) * (GPB * R); base < a.n_rows;
       base += int64_t(gx) * (GPB * R)) {
    RFM_FSTAMP(0);
    int64_t t[R];
    int32_t r[R];  // rows of a log (or of a batch) fit 31 bits
    // entry offsets: the plan checks that they fit 31 bits; the caller's CSR is taken as it is
    using EntryOff = typename std::conditional<REC, int32_t, int64_t>::type;
    EntryOff p0[R];
    int len[R];
    double yy[R], pp[R];
    bool valid[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      t[i] = base + i * GPB + g;
      valid[i] = t[i] <= last_row;
      r[i] = int32_t(valid[i] ? t[i] : last_row);
    }
    bool sb[R];  // SEG: the row belongs to the second log
#pragma unroll
    for (int i = 0; i < R; ++i) sb[i] = seg && (valid[i] ? t[i] : last_row) >= a.n_rows_a;
    if (a.row_ids) {  // uniform: the R loads stay in one block and overlap
#pragma unroll
      for (int i = 0; i < R; ++i) r[i] = a.row_ids[sb[i] ? 0 : r[i]];
    }
    if constexpr (seg) {
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (sb[i]) r[i] = int32_t((valid[i] ? t[i] : last_row) - a.n_rows_a);
    }
    // the workgroup's NEXT trip: its row ids now, and (below, under the hot pass) a touch of
    // its row blocks, so that the next trip's first two dependent loads find their lines
    // on chip instead of waiting for memory
    int32_t nxt[R];
    const int64_t nbase = base + int64_t(gx) * (GPB * R);
    constexpr bool ell = REC && ELL;
    const bool warm = ell && BLOCK == kBigBlock && a.row_ids && nbase < a.n_rows;
    int32_t* parked = reinterpret_cast<int32_t*>(red);  // (red is only used after the trips)
    if (warm) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int64_t nt = nbase + i * GPB + g;
        nxt[i] = a.row_ids[nt <= last_row ? nt : last_row];
      }
    }
    Entry e0[R];  // ell: the row's entry of this lane, loaded next to the row's head
    if (ell) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        e0[i] = reinterpret_cast<const Entry*>(a.ell + int64_t(r[i]) * a.ell_stride)[l];
        const double2 yp = a.ell_yp[r[i]];
        yy[i] = yp.x;
        pp[i] = yp.y;
        p0[i] = 0;
      }
      // the row's length: its entries that are not padding (they come first)
      constexpr int GPW = kWave / LPR;  // lane groups of a wave
#pragma unroll
      for (int i = 0; i < R; ++i) {
        unsigned long long live = __ballot(e0[i].slot != kNilSlot);
        if (GPW > 1) live = (live >> ((g % GPW) * LPR)) & ((1ull << (LPR % 64)) - 1ull);
        len[i] = valid[i] ? __popcll(live) : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (ell) {
      } else if (REC) {
        const RowRec rec = a.rows[r[i]];
        p0[i] = EntryOff(rec.begin);
        len[i] = valid[i] ? int(rec.len) : 0;
        yy[i] = rec.y;
        pp[i] = rec.p;
      } else {
        const int64_t* ip = sb[i] ? a.indptr2 : a.indptr;
        const int64_t b0 = ip[r[i]], b1 = ip[r[i] + 1];
        p0[i] = EntryOff(b0);
        len[i] = valid[i] ? int(b1 - b0) : 0;
        yy[i] = 0.0;
        pp[i] = 1.0;
      }
    }
    if (!REC && a.y) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        yy[i] = (sb[i] ? a.y2 : a.y)[r[i]];
        pp[i] = (sb[i] ? a.pscore2 : a.pscore)[r[i]];
      }
    }
    int maxlen = len[0];
#pragma unroll
    for (int i = 1; i < R; ++i) maxlen = max(maxlen, len[i]);
    if (warm) {
      // the next trip's ids arrived with this trip's heads; they wait in LDS while the
      // gathers need every register
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (l == 0) parked[i * GPB + g] = nxt[i];
    }

    RFM_FSTAMP(1);  // (row blocks have arrived: maxlen is known)
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
        const EntryOff at = len[i] > 0 ? p0[i] + min(my, len[i] - 1) : 0;
        if (ell) {
          e[i] = e0[i];  // one round; the block is padded with the row's last entry
        } else if (REC) {
          e[i] = a.ent[at];
        } else {
          e[i].col = (sb[i] ? a.indices2 : a.indices)[at];
          e[i].slot = 0;
          e[i].x = (sb[i] ? a.values2 : a.values)[at];
        }
        // padding: the row's own last entry again with x = 0 (not some fixed column: a
        // non-finite row of V must reach only the rows that hold its column; an empty
        // row, whose padding is entry 0 of the log, has its sums cleared below)
        if (my >= len[i]) {
          e[i].slot = 0;
          e[i].x = 0.0;
        }
      }
#pragma unroll
      for (int i = 0; i < R; ++i) {
        lin[i] += a.w[e[i].col] * e[i].x;
        ebuf[i * LPR + l] = e[i];
      }
      const int cnt = (maxlen - pb) < LPR ? (maxlen - pb) : LPR;
      // entries whose gathers are in flight together: the many-rows shape is at its register
      // budget with two (x R rows); the one-row shape has registers to spare
#pragma unroll(BLOCK == kBigBlock || NC > 1 ? (NC == 1 && !DET && (ELL || !REC) ? (REC ? RFM_FWD_BIG_UNROLL : RFM_FWD_PLAIN_UNROLL) : (BLOCK == kBigBlock || NC > 4 ? 2 : RFM_FWD_SMALL_WIDE_UNROLL)) : RFM_FWD_SMALL_UNROLL)
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
        // padding entries carry x = 0 and lanes past k get a zero multiplier (both re-read
        // values of a column the row holds anyway): the products vanish without a select
        // or branch, so the gathers above stay unconditional and in flight together
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const double xm = fok[c] ? ej[i].x : 0.0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              const double vx = pv[i][c].v[v] * xm;
              q[i][c][v] += vx;
              s2[i] += vx * vx;
            }
          }
        }
      }
    }

    RFM_FSTAMP(2);  // (gathers summed)
    if (warm) {  // back from LDS (same wave wrote them): the gathers' registers are free again
#pragma unroll
      for (int i = 0; i < R; ++i) nxt[i] = parked[i * GPB + g];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (len[i] == 0) {  // an empty row scores sigmoid(w0) whatever its padding gathered
        s2[i] = 0.0;
        lin[i] = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int v = 0; v < VEC; ++v) q[i][c][v] = 0.0;
      }
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
          // (Q belongs to a plan: max_batch * k fits 31 bits, checked where it is made)
          const uint32_t qoff = uint32_t(t[i]) * uint32_t(k);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            if (fok[c]) {
              Pack<VEC> pq;
#pragma unroll
              for (int v = 0; v < VEC; ++v) pq.v[v] = q[i][c][v];
              pq.store(a.out_Q + (qoff + uint32_t(fo[c])));
            }
          }
        }
        if (l == 0) {
          if (a.out_pred) a.out_pred[t[i]] = pred;
          if (a.out_err) a.out_err[t[i]] = err[i];
          if (a.loss_partial) {
            if (stage_loss) {  // the logarithms wait for the trip's barrier below
              double* st = lstage + ((trip & 1) * RT + i * GPB + g) * 3;
              st[0] = pred;
              st[1] = yy[i];
              st[2] = pp[i];
            } else if (sb[i]) {
              loss_acc2 += logloss_term(yy[i], pp[i], pred, a.eps);
            } else {
              loss_acc += logloss_term(yy[i], pp[i], pred, a.eps);
            }
          }
          err_acc += err[i];
        }
      } else if (stage_loss && l == 0) {
        lstage[((trip & 1) * RT + i * GPB + g) * 3] = -1.0;  // no row here (a score is in [0, 1] or NaN)
      }
    }
    if (stage_loss) {
      // The two logarithms of a row's loss term are ~200 dependent f64 instructions; computed by
      // lane 0 of every lane group they ran once per row and WAVE (validation forward, 100 k
      // rows: 38 of 27 + us).  Staged, the trip's rows are spread over the workgroup's threads.
      __syncthreads();
      if (tid < RT) {
        const double* st = lstage + ((trip & 1) * RT + tid) * 3;
        if (!(st[0] < 0.0)) {  // (slot tid is row base + tid of the launch)
          const double term = logloss_term(st[1], st[2], st[0], a.eps);
          if (seg && base + tid >= a.n_rows_a)
            loss_acc2 += term;
          else
            loss_acc += term;
        }
      }
    }
    ++trip;
    RFM_FSTAMP(3);  // (scores, residuals, Q rows out)

    int32_t touched[R];
    if (warm) {
#pragma unroll
      for (int i = 0; i < R; ++i)  // lane l touches the 16 bytes of its own entry
        touched[i] = *reinterpret_cast<const int32_t*>(a.ell + int64_t(nxt[i]) * a.ell_stride +
                                                      16 * l);
    }
    if (REC && (a.slot_mark || (H > 0 && RFM_KEEP(a, 32)))) {
      // after the residual is known: marks of the sparse-class entries, and the hot
      // entries' err * x * [q, 1, x] into the workgroup's LDS sums.  A single round (rows
      // of at most LPR entries) still has its entries parked in LDS.
      // (fixed-order hot sums meet at workgroup barriers: every group then makes the rounds
      // of the plan's longest row, whatever its own rows need)
      const bool fix_hot = FIX && H > 0 && RFM_KEEP(a, 32);
      const int rounds_len = fix_hot ? a.hot_rounds * LPR : maxlen;
      for (int pb = 0; pb < rounds_len; pb += LPR) {
        Entry em[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          if (rounds_len > LPR) {  // (never in the padded form)
            em[i] = a.ent[len[i] > 0 ? p0[i] + min(pb + l, len[i] - 1) : 0];
            if (pb + l >= len[i]) em[i] = Entry{0, 0, 0.0};
            ebuf[i * LPR + l] = em[i];
          } else {
            em[i] = ebuf[i * LPR + l];
          }
        }
        if (a.slot_mark && RFM_KEEP(a, 1)) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            if (pb + l < len[i] && em[i].slot >= 0) {
              // the row's batch position and residual at the entry's slot (plain store: the
              // row ids of a step are distinct), and the slot's bit in the map the gradient
              // launch scans (no-return atomic)
              a.slot_mark[em[i].slot] = SlotMark{int32_t(t[i]), 0, err[i]};
              atomicOr(a.slot_bits + (em[i].slot >> 6), 1ull << (em[i].slot & 63));
            }
          }
        }
        if (!(H > 0 && RFM_KEEP(a, 32))) continue;
        if constexpr (FIX) {
          // ---- park: Q rows (once), err * x * [1, x] in place of the entries, row sets ----
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const int row = i * GPB + g;
            if (pb == 0 && fok[0]) {
              Pack<VEC> pq;
#pragma unroll
              for (int v = 0; v < VEC; ++v) pq.v[v] = q[i][0][v];
              pq.store(Qs + row * KS + fo[0]);
            }
            const bool live = pb + l < len[i];
            if (live && em[i].slot < 0) {
              const int h = -1 - em[i].slot;
              atomicOr(&hbits[h * NW + (row >> 5)], 1u << (row & 31));
              hpos[h * PS + row] = uint8_t(l);
            }
            const double coef = live ? err[i] * em[i].x : 0.0;
            Pack<2> pc;
            pc.v[0] = coef;
            pc.v[1] = coef * em[i].x;
            pc.store(reinterpret_cast<double*>(ebuf + i * LPR + l));
          }
          __syncthreads();
          if (RFM_KEEP(a, 8)) {
            const Entry* ebase = reinterpret_cast<const Entry*>(dyn_lds + BLOCK);
            // sum of err * x * [q, 1, x] over the rows of a set (lo: rows base.., hi: rows
            // base + 64..), ascending; U rows' LDS reads are in flight together
            double acc[VEC], accw, accx;
            const auto add_rows = [&](int h, unsigned long long lo, unsigned long long hi, int base) {
              while (lo | hi) {
                constexpr int U = RFM_HOT_U;
                int rowu[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  ok[u] = (lo | hi) != 0;
                  int bit = 0;
                  if (lo) {
                    bit = __builtin_ctzll(lo);
                    lo &= lo - 1;
                  } else if (hi) {
                    bit = 64 + __builtin_ctzll(hi);
                    hi &= hi - 1;
                  }
                  rowu[u] = base + bit;
                }
                int ju[U];
#pragma unroll
                for (int u = 0; u < U; ++u) ju[u] = hpos[h * PS + rowu[u]] & (LPR - 1);
                Pack<2> cf[U];
                Pack<VEC> qv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  const int gg = rowu[u] % GPB, ii = rowu[u] / GPB;
                  cf[u].load(reinterpret_cast<const double*>(ebase + gg * (R * LPR + 1) + ii * LPR + ju[u]));
                  qv[u].load(Qs + rowu[u] * KS + fo[0]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  if (ok[u]) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += cf[u].v[0] * qv[u].v[v];
                    accw += cf[u].v[0];
                    accx += cf[u].v[1];
                  }
                }
              }
            };
            const auto clear_acc = [&] {
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
              accw = 0.0;
              accx = 0.0;
            };
            const auto add_to = [&](double* row_out, bool overwrite) {
              if (fok[0]) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                  row_out[fo[0] + v] = overwrite ? acc[v] : row_out[fo[0] + v] + acc[v];
              }
              if (l == 0) {
                row_out[k] = overwrite ? accw : row_out[k] + accw;
                row_out[k + 1] = overwrite ? accx : row_out[k + 1] + accx;
              }
            };
            // the NHV most frequent columns are cut into cells of 32 rows, one per lane group:
            // a cell's sum goes to its own row of `part`, combined in cell order below
            if (NCELL > 1 && g / NCELL < H) {
              const int h = g / NCELL, cell = g % NCELL;
              const unsigned int word = hbits[h * NW + cell];
              if (l == 0) hbits[h * NW + cell] = 0u;  // (the group has read it: one wave)
              clear_acc();
              add_rows(h, word, 0ull, cell * 32);
              add_to(part + g * hot_w, true);
            }
            // the other columns: one lane group each, dealt boustrophedon over the ranks (which
            // descend by frequency), starting opposite to the cells' order
            for (int s2 = 0; NHV + s2 * GPB < H; ++s2) {
              const int h = NHV + s2 * GPB + ((s2 & 1) ? g : GPB - 1 - g);
              if (h >= H) continue;
              unsigned long long lo = 0, hi = 0;
#pragma unroll
              for (int wi = 0; wi < NW; ++wi) {
                const unsigned long long word = hbits[h * NW + wi];
                if (wi < 2)
                  lo |= word << (32 * wi);
                else
                  hi |= word << (32 * (wi - 2));
              }
              if ((lo | hi) == 0) continue;
              if (l < NW) hbits[h * NW + l] = 0u;
              clear_acc();
              add_rows(h, lo, hi, 0);
              add_to(hot + h * hot_w, false);
            }
          }
          __syncthreads();  // the next round / trip rewrites the entries
          if (NCELL > 1 && RFM_KEEP(a, 8)) {
            // cells of a column, in cell order (lane group h: nobody else touches hot[h])
            if (g < NHV && g < H) {
              double* hrow = hot + g * hot_w;
              for (int f = l; f < hot_w; f += LPR) {
                double sum = hrow[f];
#pragma unroll
                for (int c = 0; c < NCELL; ++c) sum += part[(g * NCELL + c) * hot_w + f];
                hrow[f] = sum;
              }
            }
          }
          continue;
        }
        const int cnt = (maxlen - pb) < LPR ? (maxlen - pb) : LPR;
        // the lane groups of a wave start at different entries: rows of one log
        // tend to hold the same hot column at the same position, and adds to one
        // LDS address from several groups in one instruction serialise
        const int rot = ((g % (kWave / LPR)) * cnt) / (kWave / LPR);
        for (int jj = 0; jj < cnt; ++jj) {
          const int j = jj + rot < cnt ? jj + rot : jj + rot - cnt;
          Entry eh[R];
          bool on[R];
#pragma unroll
          for (int i = 0; i < R; ++i) {
            eh[i] = ebuf[i * LPR + j];
            on[i] = eh[i].slot < 0 && pb + j < len[i];
          }
          // two rows of the group holding the same hot column: one add for both
          const bool pair01 = R == 2 && on[0] && on[R - 1] && eh[0].slot == eh[R - 1].slot;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            if (!on[i] || (pair01 && i == R - 1)) continue;
            const double coef = err[i] * eh[i].x;
            const double coef2 = pair01 ? err[R - 1] * eh[R - 1].x : 0.0;
            double* hrow = hot + (-1 - eh[i].slot) * hot_w;
#ifdef RFM_ABLATE
            // bit 512: what carrying the five most frequent columns (present in every row of the
            // KuaiRec-shaped log) in registers across a workgroup's trips would save at best --
            // their adds vanish on every trip after a workgroup's first
            if ((a.ablate & 512) && -1 - eh[i].slot < 5 && base >= int64_t(gx) * (GPB * R)) continue;
#endif
            if (RFM_KEEP(a, 8)) {
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                if (fok[c]) {
#pragma unroll
                  for (int v = 0; v < VEC; ++v)
                    unsafeAtomicAdd(hrow + fo[c] + v, coef * q[i][c][v] + coef2 * q[R - 1][c][v]);
                }
              }
              // sum coef (lane 0) and sum coef*x (lane 1) in one instruction
              if (l < 2)
                unsafeAtomicAdd(hrow + k + l, l == 0 ? coef + coef2
                                                     : coef * eh[i].x + coef2 * eh[R - 1].x);
            }
          }
        }
      }
    }
    if (warm) {
#pragma unroll
      for (int i = 0; i < R; ++i) asm volatile("" ::"v"(touched[i]));  // keep the touches alive
    }
    RFM_FSTAMP(4);  // (marks left, hot entries added)
#ifdef RFM_FWD_STAMPS
    ++trip_no;
#endif
  }
#ifdef RFM_FWD_STAMPS
  trip_no = 1;
#endif
  RFM_FSTAMP(5);  // (all trips done)

  if (H > 0 && RFM_KEEP(a, 16)) {
    __syncthreads();
    // slab layout [H][gridDim.x][k+2]: the slabs of one column are contiguous
    for (int i = tid; i < H * hot_w; i += BLOCK) {
      const int h = i / hot_w, f = i % hot_w;
      a.hot_slab[(int64_t(h) * gx + bx) * hot_w + f] = hot[i];
    }
  }
  if (a.err_partial) {
    const double s = block_sum<BLOCK>(err_acc, red);
    if (tid == 0) a.err_partial[bx] = s;
  }
  if (a.loss_partial) {
    const double s = block_sum<BLOCK>(loss_acc, red);
    if (tid == 0) a.loss_partial[bx] = s;
  }
  if (seg && a.loss_partial2) {
    const double s = block_sum<BLOCK>(loss_acc2, red);
    if (tid == 0) a.loss_partial2[bx] = s;
  }
  RFM_FSTAMP(6);  // (slab and partial sums stored)
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

// the same for a run of launches at once: block b finishes the partials of launch b
__global__ __launch_bounds__(kBlock) void loss_finish_many_kernel(const double* partial,
                                                                 int64_t stride, int n_partial,
                                                                 int64_t n_rows, double* out) {
  __shared__ double lds[kBlock];
  const double* row = partial + int64_t(blockIdx.x) * stride;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += kBlock) acc += row[i];
  const double s = block_sum<kBlock>(acc, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = -s / double(n_rows);
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
// fixed-order reductions over rows of a table (used by launches 2 and 3)
// ---------------------------------------------------------------------------
// tot[f] = sum_{r<rows} base[r*width + f], f < width, rows contiguous, in a
// fixed order: the block's threads split into row groups x factor lanes, each
// group sums its rows in ascending order (16 loads in flight), the groups are
// then added in group order.
__device__ inline void ordered_rows_sum(const double* base, int rows, int width,
                                        double* scratch /*[kBlock]*/, double* tot /*[width]*/) {
  // (loads in flight per thread: one round covers the slabs of a small batch's forward --
  // 125 workgroups at B = 2 000 -- with seven row groups at k = 32)
  constexpr int U = 32;
  const int fw = width < kBlock ? width : kBlock;  // factor lanes per row group
  const int nsg = kBlock / fw;
  const int sg = threadIdx.x / fw, fl = threadIdx.x % fw;
  const bool live = sg < nsg;
  for (int f0 = 0; f0 < width; f0 += fw) {
    const int f = f0 + fl;
    double acc = 0.0;
    if (live && f < width) {
      for (int r = sg; r < rows; r += nsg * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = r + u * nsg;
          v[u] = base[int64_t(rr < rows ? rr : r) * width + f];
          if (rr >= rows) v[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
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

// Same for the stamped partial rows of one split column: rows first .. first+rows-1 of
// `parts`, each valid iff its stamp is this step's.
__device__ inline void ordered_part_sum(const double* parts, int first, int rows, int k,
                                        double stamp, double* scratch, double* tot) {
  constexpr int U = 8;
  const int width = k + 2;
  const int fw = width < kBlock ? width : kBlock;
  const int nsg = kBlock / fw;
  const int sg = threadIdx.x / fw, fl = threadIdx.x % fw;
  const bool live = sg < nsg;
  for (int f0 = 0; f0 < width; f0 += fw) {
    const int f = f0 + fl;
    double acc = 0.0;
    if (live && f < width) {
      for (int r = sg; r < rows; r += nsg * U) {
        const double* row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = r + u * nsg;
          row[u] = parts + int64_t(first + (rr < rows ? rr : r)) * (k + 3);
        }
        double v[U], st[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v[u] = row[u][f];
          st[u] = row[u][k + 2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (r + u * nsg < rows && st[u] == stamp) ? v[u] : 0.0;
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

// ---------------------------------------------------------------------------
// 2. sparse-class gradient + update (tasks of whole columns), hot columns, w0
// ---------------------------------------------------------------------------
struct WinRec {  // a marked slot of the word being processed, parked in LDS, 24 B
  int32_t t;     // batch position of the slot's row
  int32_t col;   // feature column
  double coef;   // err_t * x
  double cx;     // err_t * x * x
};

struct ConsArgs {
  const TaskRec* tasks;  // [nb_tasks * tasks per workgroup]
  int32_t task_words;    // 64-slot words of the slot bitmap per task
  unsigned long long* slot_bits;  // one bit per slot: set by the forward, cleared here
  unsigned long long* slot_bits_other;  // CH form: the bitmap of the step before, cleared here
  const SlotMark* slot_mark;      // {batch position, residual} of a marked slot's row
  const SlotRec* slots;
  const double* Q;
  int32_t k;
  int64_t n;        // features
  double* w0;
  double* V;        // apply mode: updated in place; grad mode: read only
  double* w;
  double lr;
  double* parts;    // [n_parts][k+3]: M[0..k), sum coef, sum coef*x, step stamp (split columns)
  double stamp;     // id of this step (a partial row is valid iff its stamp matches)
  double* grad;     // nullable: grad mode -> [G_V | g_w | g_w0]
  int32_t* touch;   // nullable (grad mode): touch[col] = touch_id for every column written
  int32_t touch_id;
  // the hot columns and w0 ride in the same launch: the workgroups after the tasks reduce
  // the forward's slabs / residual sums (independent of the sparse class, so they overlap)
  int32_t nb_tasks;  // workgroups that process tasks
  int32_t n_hot;
  const int32_t* hot_cols;
  const double* hot_slab;
  int32_t n_slabs;
  const double* err_partial;  // [n_slabs] per-workgroup sums of the residual
  int32_t n_chunks;           // CH form: chunks of 64 lanes x VEC factors
  int32_t xcd_chunks;         // CH form: 1 = chunks dealt to XCDs (1-D grid), 0 = blockIdx.y
  // PREP form: the task's records and their number, the rows' residuals by batch position
  const PrepRec* prep_rec;    // [tasks][kPrepCap]
  const int32_t* prep_cnt;    // [tasks]
  const double* err;          // [batch]
#ifdef RFM_CONS_STAMPS
  long long* stamps;  // -DRFM_CONS_STAMPS builds only: [workgroup][wave][8] clock readings of the task workgroups
#endif
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

// V[col,:] and w[col] from the sums of one column: update in place, or write the
// gradient row (grad mode).  vold = the row as it was read before.
// (fb: first factor of the chunk this lane group works on -- 0 unless the chunks of a row are
// dealt to different workgroups; the scalar part, w / touch, is then written by chunk 0)
// *p += x by its only writer of the step.  As `*p += x` the wavefront stands still for the load of *p
// (nothing else waits for it) -- once per finished column in the gradient launch, eight times per task
// where columns are short, and one dependent level in the finalize; the no-return atomic add gives the
// same sum without the wait (-DRFM_W_RMW=1: the read-modify-write, for timing).
__device__ inline void add_by_only_writer(double* p, double x) {
#if defined(RFM_W_RMW) && RFM_W_RMW
  *p += x;
#else
  unsafeAtomicAdd(p, x);
#endif
}

template <int LPR, int VEC, int NC>
__device__ inline void apply_column(const ColAcc<VEC, NC>& acc, const Pack<VEC> (&vold)[NC],
                                    int32_t col, double* V, double* w, double* grad, int64_t n,
                                    int k, double lr, int l, int32_t* touch = nullptr,
                                    int32_t touch_id = 0, int fb = 0) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int f = fb + (c * LPR + l) * VEC;
    if (f < k) {
      Pack<VEC> out;
      if (grad) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out.v[v] = acc.d * vold[c].v[v] - acc.m[c][v];
        out.store(grad + int64_t(col) * k + f);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          out.v[v] = vold[c].v[v] + lr * (acc.m[c][v] - acc.d * vold[c].v[v]);
        out.store(V + int64_t(col) * k + f);
      }
    }
  }
  if (l == 0 && fb == 0) {
    if (grad) {
      grad[n * k + col] = -acc.gw;
      if (touch) touch[col] = touch_id;  // touched-row mode: the row is valid for this step
    } else {
      add_by_only_writer(w + col, lr * acc.gw);
    }
  }
}

// The same from a whole workgroup, sums in LDS: tot[0..k) = M, tot[k] = sum coef,
// tot[k+1] = sum coef*x.
// (vpre: the row as the caller read it before the sums were formed -- threadIdx.x + j * kBlock
// -- so that the load is in flight beside the sums' own loads; null: read here)
constexpr int kRowPre = (RFM_MAX_FACTORS + kBlock - 1) / kBlock;
__device__ inline void apply_column_block(const double* tot, int32_t col, double* V, double* w,
                                          double* grad, int64_t n, int k, double lr,
                                          int32_t* touch, int32_t touch_id,
                                          const double* vpre = nullptr) {
  const double gw = tot[k], d = tot[k + 1];
  int j = 0;
  for (int f = threadIdx.x; f < k; f += kBlock, ++j) {
    const int64_t at = int64_t(col) * k + f;
    const double vold = vpre ? vpre[j] : V[at];
    if (grad)
      grad[at] = d * vold - tot[f];
    else
      V[at] = vold + lr * (tot[f] - d * vold);
  }
  if (threadIdx.x == 0) {
    if (grad) {
      grad[n * k + col] = -gw;
      if (touch) touch[col] = touch_id;
    } else {
      add_by_only_writer(w + col, lr * gw);
    }
  }
}

// slots a lane group handles per pass over a 64-slot word of the bitmap: PLANES pieces of
// LPR consecutive slots
template <int LPR>
struct WinShape {
  static constexpr int PLANES = LPR >= 64 ? 1 : (LPR == 32 ? 2 : 4);
  static constexpr int WIN = PLANES * LPR;  // 64 for LPR >= 16; 16 / 32 for LPR = 4 / 8
};

// inclusive prefix sum over the LPR lanes of a lane group
template <int LPR>
__device__ inline int group_scan(int v, int l) {
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) {
    const int up = __shfl_up(v, o, LPR);
    if (l >= o) v += up;
  }
  return v;
}

// One TASK per lane group: `task_words` 64-slot words of the slot view, the same number for
// every task of a plan (chosen so that a task expects a handful of marked slots), so the
// task's bitmap words are the kernel's FIRST load -- nothing has to be looked up before
// them.  Whole columns are packed into the tasks in column order; a column may run over
// several consecutive tasks, but never over the edge of a workgroup's tasks unless it is
// longer than all of them together.  The group lists the task's marked slots, in slot
// order, in LDS, then runs the chain ONCE: the marked slots' records and marks (parked in
// LDS) -> Q rows and V rows (eight gathers in flight) -> err*x*[Q[t,:], 1, x] accumulated
// strictly in slot order.  A column that lies inside the task is updated in place -- this
// group is its only writer.  For a column that continues from the previous task the sums
// go to LDS (`head` row of the group); the group in whose task such a column STARTS owns
// it: after a workgroup barrier it adds the followers' head rows in task (= slot) order to
// its own and updates the column.  Only a column longer than a whole workgroup's tasks
// leaves partial rows in memory (one per workgroup) for fm_finalize_kernel.  The work of a
// launch follows the batch (marked slots): an untouched task reads its bitmap words and
// waits at the barrier.
// (The lane groups of a wave run in lock step: ballots and shuffles are wave-wide.)
#ifndef RFM_CONS_CH_BATCH
#define RFM_CONS_CH_BATCH 4
#endif
// CH (factor counts of more than one chunk per lane): the chunks of a row are dealt to
// DIFFERENT workgroups -- blockIdx.y = chunk, LPR x VEC factors each, NC = 1 in registers -- so
// that a task's chain is one 16-byte load per lane and entry instead of NC of them, NC times as
// many lane groups share the step's entries (measured at k = 400, B = 2 000: equal to the
// all-chunks-in-registers form within noise -- both wait on the same chain of dependent levels --
// at half the registers and with one instantiation for every chunk count).  Every
// chunk lists the task's marks itself (same bitmap words: they hit in L2), so the words cannot be
// cleared while a sibling may still read them: the bitmap is double-buffered by step parity and
// chunk 0 clears the task's words of the OTHER buffer (the step before, long consumed).
// PREP: the task's marked entries come as records in slot order at a fixed place (prepared
// ahead of the loop from the row ids, see PrepRec) instead of being discovered from the bitmap:
// first load = the records, second level = residuals + Q / V rows, third = the stores.  Sums
// and their order are those of the bitmap form, bit for bit.
// (waves per SIMD the chunk-per-workgroup form is compiled for: its grid is task workgroups x
// chunks -- 1 400 workgroups at the published point -- and how many of them are resident sets
// the launch's length; RFM_CONS_CH_WAVES=0: no cap)
#ifndef RFM_CONS_CH_WAVES
#define RFM_CONS_CH_WAVES 0
#endif
template <int LPR, int VEC, int NC, bool CH = false, bool PREP = false>
__global__ __launch_bounds__(kBlock, (CH && RFM_CONS_CH_WAVES > 0 ? RFM_CONS_CH_WAVES : 1)) void fm_consume_kernel(ConsArgs a) {
  static_assert(!CH || NC == 1, "the chunked form holds one chunk per lane group");
  constexpr int GPB = kBlock / LPR;  // tasks of a workgroup
  constexpr int PLANES = WinShape<LPR>::PLANES;
  constexpr int WIN = WinShape<LPR>::WIN;  // marked slots a group lists before it runs the chain
  constexpr int BATCH = CH ? RFM_CONS_CH_BATCH : 4;  // entries whose Q and V rows are in flight together
  // CH: which chunk and which group of tasks this workgroup takes.  Workgroups go to the eight
  // XCDs round robin by their linear id, and every XCD has an L2 of its own: with the chunks
  // dealt to XCDs (chunk c = the XCDs [8c / n, 8(c+1) / n)) an XCD only ever touches ITS slice of
  // the Q rows and of V -- a quarter of them at k = 400 -- instead of all of both.
  int bx = int(blockIdx.x), chunk = 0;
  if (CH) {
    if (a.xcd_chunks) {
      const int xcd = bx & 7, slot = bx >> 3, nch = a.n_chunks;
      chunk = xcd * nch / 8;
      const int first = (chunk * 8 + nch - 1) / nch, next = ((chunk + 1) * 8 + nch - 1) / nch;
      bx = slot * (next - first) + (xcd - first);
      if (bx >= a.nb_tasks + a.n_hot + 1) return;  // (the grid is rounded up)
    } else {
      chunk = int(blockIdx.y);
    }
  }
  const int fb = chunk * (LPR * VEC);  // first factor of this workgroup's chunk
  constexpr int TRIPS = kTaskTrips;        // bitmap words a lane loads (task_words <= TRIPS*LPR)
  // LDS: per group the list of marked slots and their parked records, then the head rows; or
  // (hot-column / w0 workgroups) reduction scratch
  constexpr int kGroupBytes = WIN * (int(sizeof(WinRec)) + 8);
  constexpr int kHotBytes = (kBlock + 1024 + 2) * int(sizeof(double));
  extern __shared__ double lds_raw[];  // max(GPB * kGroupBytes + GPB * (k+3) * 8, kHotBytes)
  const int k = a.k;
  // the hot-column workgroups come last in the grid (measured: first, they delay the
  // tasks and the launch takes longer)
  if (bx >= a.nb_tasks) {
    if (CH && chunk != 0) return;
    const int hb = bx - a.nb_tasks;
    double* tot = lds_raw + kBlock;
    if (hb < a.n_hot) {
      // a hot column: the forward workgroups' slabs, in block order; the column's row of V is
      // requested first, so that it arrives with the slabs and not after them
      const int32_t col = a.hot_cols[hb];
      double vpre[kRowPre];
#pragma unroll
      for (int j = 0; j < kRowPre; ++j) {
        const int f = int(threadIdx.x) + j * kBlock;
        vpre[j] = a.V[int64_t(col) * k + (f < k ? f : 0)];
      }
      ordered_rows_sum(a.hot_slab + int64_t(hb) * a.n_slabs * (k + 2), a.n_slabs, k + 2, lds_raw,
                       tot);
      apply_column_block(tot, col, a.V, a.w, a.grad, a.n, k, a.lr, a.touch, a.touch_id, vpre);
    } else {
      // w0 from the forward workgroups' residual sums
      double acc = 0.0;
      for (int i = threadIdx.x; i < a.n_slabs; i += kBlock) acc += a.err_partial[i];
      const double s = block_sum<kBlock>(acc, lds_raw);
      if (threadIdx.x == 0) {
        if (a.grad)
          a.grad[a.n * k + a.n] = -s;
        else
          add_by_only_writer(a.w0, a.lr * s);
      }
    }
    return;
  }
  (void)kHotBytes;
  const int lane = threadIdx.x % kWave;
  const int l = lane % LPR;
#ifdef RFM_CONS_STAMPS
#define RFM_CSTAMP(i) do { if (a.stamps && (threadIdx.x & (kWave - 1)) == 0) a.stamps[(int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define RFM_CSTAMP(i) do { } while (0)
#endif
  RFM_CSTAMP(0);
  const int gb = threadIdx.x / LPR;  // group in the workgroup
  const int task = bx * GPB + gb;
  const int W = a.task_words;
  const int32_t slot0 = task * W * 64;  // first slot of the task
  // first loads: the task's bitmap words (one per lane and trip) -- or, PREP, its records --
  // and its description
  unsigned long long wd[TRIPS];
  PrepRec pr[PLANES];
  int prep_n = 0;
  if constexpr (PREP) {
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) pr[pl] = a.prep_rec[int64_t(task) * kPrepCap + pl * LPR + l];
    prep_n = a.prep_cnt[task];
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) wd[u] = 0ull;
  } else {
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) {
      const int idx = u * LPR + l;
      wd[u] = idx < W ? a.slot_bits[int64_t(task) * W + idx] : 0ull;
    }
  }
  const TaskRec tk = a.tasks[task];
  if constexpr (!PREP) {
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) {
      if (CH) {
        if (fb == 0 && u * LPR + l < W) a.slot_bits_other[int64_t(task) * W + u * LPR + l] = 0ull;
      } else if (wd[u]) {
        a.slot_bits[int64_t(task) * W + u * LPR + l] = 0ull;  // consumed: cleared at once
      }
    }
  }

  char* gmem = reinterpret_cast<char*>(lds_raw) + gb * kGroupBytes;
  WinRec* wrec = reinterpret_cast<WinRec*>(gmem);
  int32_t* list = reinterpret_cast<int32_t*>(gmem + WIN * sizeof(WinRec));
  double* heads = reinterpret_cast<double*>(reinterpret_cast<char*>(lds_raw) + GPB * kGroupBytes);
  double* head = heads + gb * (k + 3);  // [0..k) M, [k] sum coef, [k+1] sum coef*x, [k+2] flags

  const bool head_open = tk.flags & 1;  // first_col continues from the previous task
  const bool tail_open = tk.flags & 2;  // last_col continues into the next task
  ColAcc<VEC, NC> acc;
  acc.clear();
  Pack<VEC> vold[NC];   // V row of the column being accumulated
  int32_t cur = -1;     // that column (uniform in the lane group)
  int fill = 0;         // marked slots listed and not yet consumed (uniform in the lane group)
  bool head_done = false;

  // a finished column: into the head row (continues from the previous task), or updated in
  // place; the last column of a tail-open task stays in `acc` for the combine below
  const auto finish = [&]() {
    if (head_open && cur == tk.first_col) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int f = fb + (c * LPR + l) * VEC;
        if (f < k) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) head[f + v] = acc.m[c][v];
        }
      }
      if (l == 0) {
        head[k] = acc.gw;
        head[k + 1] = acc.d;
      }
      head_done = true;
    } else {
      apply_column<LPR, VEC, NC>(acc, vold, cur, a.V, a.w, a.grad, a.n, k, a.lr, l, a.touch,
                                 a.touch_id, fb);
    }
  };

  // the chain over the group's list of `fill` marked slots
  const auto run_list = [&]() {
    // records and marks of the listed slots, parked in LDS by list position
    if constexpr (!PREP) {
      SlotRec sr[PLANES];
      SlotMark mk[PLANES];
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const int e = pl * LPR + l;
        const int32_t s = e < fill ? list[e] : slot0;
        mk[pl] = a.slot_mark[s];
        sr[pl] = a.slots[s];
      }
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const int e = pl * LPR + l;
        if (e < fill) {
          const double coef = mk[pl].err * sr[pl].x;
          wrec[e] = WinRec{mk[pl].t, sr[pl].col, coef, coef * sr[pl].x};
        }
      }
    }
    // every group of the wave loops as long as any of them has entries left
    int at = 0;
    while (__ballot(at < fill)) {
      const int nb = max(min(BATCH, fill - at), 0);
      WinRec rec[BATCH];
      Pack<VEC> qq[BATCH][NC], vv[BATCH][NC];
      double ee[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        rec[u] = wrec[u < nb ? at + u : 0];
        if (u >= nb) {  // nothing there: any row of Q / V will do
          rec[u].t = 0;
          rec[u].col = 0;
        }
        if constexpr (PREP) ee[u] = a.err[rec[u].t];  // the row's residual, with its Q row
        // the entry's Q row, and the V row of its column in case the entry starts a new
        // column: fetched together, so that a run of one-entry columns (one-hot users and
        // items in a small batch) costs one round trip, not one per column
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
          const int f = fb + (ch * LPR + l) * VEC;
          qq[u][ch].load(a.Q + int64_t(rec[u].t) * k + (f < k ? f : 0));
        }
        // (an entry of the column the one before it belongs to needs no V row)
        if (rec[u].col != (u == 0 ? cur : rec[u > 0 ? u - 1 : 0].col)) {
#pragma unroll
          for (int ch = 0; ch < NC; ++ch) {
            const int f = fb + (ch * LPR + l) * VEC;
            vv[u][ch].load(a.V + int64_t(rec[u].col) * k + (f < k ? f : 0));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (u < nb) {
          if (rec[u].col != cur) {
            if (cur >= 0) finish();
            acc.clear();
            cur = rec[u].col;
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) vold[ch] = vv[u][ch];
          }
          // (PREP: the record holds x; coef = err * x and cx = coef * x as the bitmap form parks them)
          // (the two products are rounded on their own -- no contraction into the sums below --
          // so that both forms add the same numbers)
          double coef = rec[u].coef, cx = rec[u].cx;
          if constexpr (PREP) {
#pragma clang fp contract(off)
            coef = ee[u] * rec[u].coef;
            cx = coef * rec[u].coef;
          }
#pragma unroll
          for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc.m[ch][v] += coef * qq[u][ch].v[v];
          acc.gw += coef;
          acc.d += cx;
        }
      }
      at += nb;
    }
    fill = 0;
  };

  if constexpr (PREP) {
    // passes of WIN records (one, but for a task with unusually many marks): park, run the chain
    for (int base = 0; __ballot(base < prep_n); base += WIN) {
      if (base > 0) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
          const int e = base + pl * LPR + l;
          pr[pl] = a.prep_rec[int64_t(task) * kPrepCap + (e < kPrepCap ? e : 0)];
        }
      }
      fill = max(min(prep_n - base, WIN), 0);
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const int e = pl * LPR + l;
        if (e < fill) wrec[e] = WinRec{pr[pl].t, pr[pl].col, pr[pl].x, 0.0};
      }
      if (__ballot(fill > 0)) run_list();
    }
  }
#pragma unroll
  for (int u = 0; u < (PREP ? 0 : TRIPS); ++u) {
    unsigned long long word = wd[u];
    const int32_t s0 = slot0 + (u * LPR + l) * 64;  // first slot of this lane's word
    // move the trip's marked slots into the list in lane (= slot) order, as many as the list
    // has room for; when a group's list is full and it has marks left, every group of the
    // wave runs the chain on what it has listed so far
    while (__ballot(word != 0ull)) {
      const int cnt = __popcll(word);
      const int before = group_scan<LPR>(cnt, l) - cnt;  // marks of the group's earlier lanes
      const int take = min(cnt, max(WIN - fill - before, 0));
      int pos = fill + before;
      for (int i = 0; i < take; ++i) {
        list[pos++] = s0 + (__ffsll((long long)word) - 1);
        word &= word - 1ull;
      }
      int added = take;
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) added += __shfl_xor(added, o, LPR);
      fill += added;
      if (__ballot(word != 0ull)) run_list();
    }
  }
  RFM_CSTAMP(1);  // (bitmap words in, marked slots listed)
  if (!PREP && __ballot(fill > 0)) run_list();
  RFM_CSTAMP(2);  // (chain run: records, Q / V rows, sums, columns inside the task updated)

  // ---- columns that run over several tasks of this workgroup -----------------------------
  // the last column of a tail-open task is still in `acc` (if it got any entry here); a
  // column that both comes in and goes on (the task lies inside it) counts as a head row
  const bool through = head_open && tail_open && tk.first_col == tk.last_col;
  // this group owns the combine of tk.last_col: the column starts (or, for a piece of a very
  // long column, this workgroup's share of it starts) in this task
  const bool own = (tail_open && !through) || tk.part >= 0;
  bool own_acc = false;  // ... and holds entries of it in `acc`
  if (cur >= 0) {
    if (own && cur == tk.last_col) {
      own_acc = true;
    } else {
      finish();
    }
  }
  if (head_open && l == 0) head[k + 2] = (head_done ? 1.0 : 0.0) + (through ? 2.0 : 0.0);
  if (!head_open && l == 0) head[k + 2] = 0.0;
  __syncthreads();
  RFM_CSTAMP(3);
  if (own) {
    if (!own_acc) {
      acc.clear();
#pragma unroll
      for (int ch = 0; ch < NC; ++ch) {
        const int f = fb + (ch * LPR + l) * VEC;
        vold[ch].load(a.V + int64_t(tk.last_col) * k + (f < k ? f : 0));
      }
    }
    bool any = own_acc;
    for (int g2 = gb + 1; g2 < GPB; ++g2) {
      const double* h2 = heads + g2 * (k + 3);
      const int fl = int(h2[k + 2]);
      if (fl & 1) {
        any = true;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
          const int f = fb + (ch * LPR + l) * VEC;
          if (f < k) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc.m[ch][v] += h2[f + v];
          }
        }
        acc.gw += h2[k];
        acc.d += h2[k + 1];
      }
      if (!(fl & 2)) break;  // the column ends in that task
    }
    if (tk.part >= 0) {
      // a piece of a column longer than the workgroup's tasks: its sums for fm_finalize_kernel
      double* row = a.parts + int64_t(tk.part) * (k + 3);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int f = fb + (c * LPR + l) * VEC;
        if (f < k) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) row[f + v] = acc.m[c][v];
        }
      }
      if (l == 0 && fb == 0) {
        row[k] = acc.gw;
        row[k + 1] = acc.d;
        row[k + 2] = any ? a.stamp : 0.0;
      }
    } else if (any) {
      apply_column<LPR, VEC, NC>(acc, vold, tk.last_col, a.V, a.w, a.grad, a.n, k, a.lr, l,
                                 a.touch, a.touch_id, fb);
    }
  }
  RFM_CSTAMP(4);  // (columns that run over several tasks combined)
}

// ---------------------------------------------------------------------------
// 3. columns split over several tasks: their partial rows, in slot order
// ---------------------------------------------------------------------------
struct FinArgs {
  const SplitCol* split;  // [n_split_short | n_split_long]
  int32_t n_split_short;  // few partial rows: one lane group per column
  int32_t n_split_long;   // many partial rows: one workgroup per column
  const double* parts;
  double stamp;
  int32_t k;
  int64_t n;
  double* w;
  double* V;
  double lr;
  double* grad;  // nullable
  int32_t* touch;  // nullable (grad mode): see ConsArgs
  int32_t touch_id;
};

// blocks [0, nb_short): short split columns, one per lane group; then one block per long
// split column.  Launched only when the plan has split columns.
template <int LPR, int VEC, int NC>
__global__ __launch_bounds__(kBlock) void fm_finalize_kernel(FinArgs a, int nb_short) {
  __shared__ double scratch[kBlock];
  __shared__ double tot[1024 + 2];
  __shared__ int touched;
  const int k = a.k;
  const int b = blockIdx.x;
  if (b < nb_short) {
    constexpr int GPB = kBlock / LPR;
    const int l = threadIdx.x % LPR;
    const int ci = b * GPB + threadIdx.x / LPR;
    if (ci >= a.n_split_short) return;
    const SplitCol cc = a.split[ci];

    ColAcc<VEC, NC> acc;
    acc.clear();
    Pack<VEC> vold[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
      const int f = (ch * LPR + l) * VEC;
      vold[ch].load(a.V + int64_t(cc.col) * k + (f < k ? f : 0));
    }
    bool any = false;
    constexpr int CB = 4;  // partial rows per trip, loaded unconditionally; stale rows are masked
    for (int i = 0; i < cc.part_count; i += CB) {
      const double* row[CB];
      bool in[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        in[u] = i + u < cc.part_count;
        row[u] = a.parts + int64_t(cc.part_begin + (in[u] ? i + u : i)) * (k + 3);
      }
      double mm[CB][NC][VEC], gg[CB], dd[CB], ss[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
          const int f = (ch * LPR + l) * VEC;
#pragma unroll
          for (int v = 0; v < VEC; ++v) mm[u][ch][v] = row[u][(f < k ? f : 0) + v];
        }
        gg[u] = row[u][k];
        dd[u] = row[u][k + 1];
        ss[u] = row[u][k + 2];
      }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const bool ok = in[u] && ss[u] == a.stamp;
        any = any || ok;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc.m[ch][v] += ok ? mm[u][ch][v] : 0.0;
        acc.gw += ok ? gg[u] : 0.0;
        acc.d += ok ? dd[u] : 0.0;
      }
    }
    if (any)
      apply_column<LPR, VEC, NC>(acc, vold, cc.col, a.V, a.w, a.grad, a.n, k, a.lr, l, a.touch,
                                 a.touch_id);
    return;
  }
  const SplitCol cc = a.split[a.n_split_short + (b - nb_short)];
  // is any partial row of this step's?  (an untouched column is left alone)
  if (threadIdx.x == 0) touched = 0;
  __syncthreads();
  for (int r = threadIdx.x; r < cc.part_count; r += kBlock)
    if (a.parts[int64_t(cc.part_begin + r) * (k + 3) + k + 2] == a.stamp) touched = 1;
  __syncthreads();
  if (!touched) return;
  ordered_part_sum(a.parts, cc.part_begin, cc.part_count, k, a.stamp, scratch, tot);
  apply_column_block(tot, cc.col, a.V, a.w, a.grad, a.n, k, a.lr, a.touch, a.touch_id);
}

// The same for factor counts of several chunks per lane, one chunk of 64 x VEC factors per
// workgroup (blockIdx.y), as fm_consume_kernel's CH form: a short column's partial rows (at
// most kShortSplit = 8) are ONE round of loads, a long column's are summed by one thread per
// factor of the chunk with 32 rows in flight; stamps ride with the rows and the column's row of
// V is requested before the sums, so a workgroup's chain is two dependent levels instead of
// eight (k = 400, B = 2 000: 8.1 -> 6.8 us per launch, profiles/r3j vs r3q).
template <int VEC>
__global__ __launch_bounds__(kBlock) void fm_finalize_chunk_kernel(FinArgs a, int nb_short) {
  constexpr int LPR = kWave, CW = kWave * VEC;
  __shared__ double tot[CW + 2];
  const int k = a.k;
  const int b = blockIdx.x;
  const int fb = int(blockIdx.y) * CW;
  if (b < nb_short) {
    constexpr int GPB = kBlock / LPR;
    constexpr int CB = 8;  // = kShortSplit: every partial row of a short column in one round
    const int l = threadIdx.x % LPR;
    const int ci = b * GPB + threadIdx.x / LPR;
    if (ci >= a.n_split_short) return;
    const SplitCol cc = a.split[ci];
    const int f = fb + l * VEC;
    const int fc = f < k ? f : 0;
    ColAcc<VEC, 1> acc;
    acc.clear();
    Pack<VEC> vold[1];
    vold[0].load(a.V + int64_t(cc.col) * k + fc);
    bool any = false;
    for (int i = 0; i < cc.part_count; i += CB) {
      Pack<VEC> mm[CB];
      double gg[CB], dd[CB], ss[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const double* row = a.parts + int64_t(cc.part_begin + (i + u < cc.part_count ? i + u : i)) * (k + 3);
        mm[u].load(row + fc);
        gg[u] = row[k];
        dd[u] = row[k + 1];
        ss[u] = row[k + 2];
      }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const bool ok = i + u < cc.part_count && ss[u] == a.stamp;
        any = any || ok;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc.m[0][v] += ok ? mm[u].v[v] : 0.0;
        acc.gw += ok ? gg[u] : 0.0;
        acc.d += ok ? dd[u] : 0.0;
      }
    }
    if (any)
      apply_column<LPR, VEC, 1>(acc, vold, cc.col, a.V, a.w, a.grad, a.n, k, a.lr, l, a.touch, a.touch_id, fb);
    return;
  }
  const SplitCol cc = a.split[a.n_split_short + (b - nb_short)];
  const int fcnt = k - fb < CW ? k - fb : CW;  // factors of this chunk
  const int j = threadIdx.x;                   // < fcnt: factor fb + j; fcnt, fcnt + 1: sum coef, sum coef * x
  const bool live = j < fcnt + 2;
  const int cidx = j < fcnt ? fb + j : k + (j - fcnt);
  const double vold = a.V[int64_t(cc.col) * k + (j < fcnt ? fb + j : 0)];
  double acc = 0.0;
  bool any = false;
  constexpr int U = 32;
  for (int r0 = 0; r0 < cc.part_count; r0 += U) {
    double v[U], st[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* row = a.parts + int64_t(cc.part_begin + (r0 + u < cc.part_count ? r0 + u : r0)) * (k + 3);
      v[u] = row[live ? cidx : 0];
      st[u] = row[k + 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = r0 + u < cc.part_count && st[u] == a.stamp;
      any = any || ok;  // (the stamps are the same for every thread: uniform)
      acc += ok ? v[u] : 0.0;
    }
  }
  if (!any) return;  // an untouched column is left alone
  if (live) tot[j] = acc;
  __syncthreads();
  const double gw = tot[fcnt], d = tot[fcnt + 1];
  if (j < fcnt) {
    const int64_t at = int64_t(cc.col) * k + fb + j;
    if (a.grad)
      a.grad[at] = d * vold - acc;
    else
      a.V[at] = vold + a.lr * (acc - d * vold);
  }
  if (j == 0 && fb == 0) {
    if (a.grad) {
      a.grad[a.n * k + cc.col] = -gw;
      if (a.touch) a.touch[cc.col] = a.touch_id;
    } else {
      add_by_only_writer(a.w + cc.col, a.lr * gw);
    }
  }
}

// RFM_CHECK_IDS=1: the row ids of one step must lie in the log and be distinct (a row's
// batch position is recorded with a plain store, so a repeated id would lose one of its
// contributions).  seen[r] holds the stamp of the last iteration that named row r.
__global__ __launch_bounds__(kBlock) void ids_check_kernel(const int32_t* ids, int64_t batch,
                                                          int64_t n_iters, int64_t n_rows,
                                                          int32_t* seen, int32_t stamp0,
                                                          int32_t* flags) {
  const int64_t total = batch * n_iters;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * kBlock) {
    const int32_t r = ids[i];
    const int32_t stamp = stamp0 + int32_t(i / batch);
    if (r < 0 || r >= n_rows) {
      atomicOr(&flags[0], 1);
    } else if (atomicExch(&seen[r], stamp) == stamp) {
      atomicOr(&flags[1], 1);
    }
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
