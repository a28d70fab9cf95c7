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
struct SlotRec {  // one slot of the column-major view, 16 B
  double x;       // feature value
  int32_t col;    // feature column
  int32_t pad;
};
struct WinInfo {      // static description of one slot window, 16 B
  int32_t first_col;  // column of the window's first slot
  int32_t last_col;   // column of its last slot
  int32_t flags;      // bit0: first column is not wholly inside the window
                      // bit1: last column (!= first) continues after the window
  int32_t pad;
};
struct CrossCol {      // a sparse-class column spanning more than one window
  int32_t col;
  int32_t idx_begin;   // its carry rows: carry_idx[idx_begin .. +idx_count)
  int32_t idx_count;
  int32_t pad;
};

// slab ranges a hot column's reduction is cut into (one workgroup each)
constexpr int kHotParts = 4;

}  // namespace rfm
