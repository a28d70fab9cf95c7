// Host-side pieces of librfm_hip.so: error state, the exact mini-batch sampler
// and the order-preserving MF schedule.  No device code here.
#include <algorithm>
#include <atomic>
#include <thread>

#include "rfm_common.h"

namespace rfm {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

// --------------------------------------------------------------------------
// MT19937 as NumPy's legacy RandomState runs it (numpy/random/src/mt19937).
// Restated from the published algorithm (Matsumoto & Nishimura 1998) and
// checked bit-for-bit against numpy.random.RandomState in the tests.
// --------------------------------------------------------------------------
struct Mt19937 {
  static constexpr int N = 624, M = 397;
  uint32_t key[N];
  int pos;

  explicit Mt19937(uint32_t seed) {
    for (int i = 0; i < N; ++i) {
      key[i] = seed;
      seed = 1812433253u * (seed ^ (seed >> 30)) + uint32_t(i) + 1u;
    }
    pos = N;
  }

  void refill() {
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int i = 0;
    for (; i < N - M; ++i) {
      uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
    }
    for (; i < N - 1; ++i) {
      uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + (M - N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
    }
    uint32_t y = (key[N - 1] & UPPER) | (key[0] & LOWER);
    key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
    pos = 0;
  }

  // refill + temper a whole block at once (two plain loops the compiler vectorises);
  // `out` then holds the next N outputs of the generator
  uint32_t out[N];
  void refill_tempered() {
    refill();
    for (int i = 0; i < N; ++i) {
      uint32_t y = key[i];
      y ^= y >> 11;
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= y >> 18;
      out[i] = y;
    }
  }

  inline uint32_t next32() {
    if (pos == N) refill_tempered();
    return out[pos++];
  }

  // NumPy's random_interval for max <= 0xffffffff: mask to the next power of
  // two minus one, redraw while above max.
  static inline uint32_t mask_for(uint32_t max) {
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    return mask;
  }
  inline uint32_t interval(uint32_t max) {
    if (max == 0) return 0;
    const uint32_t mask = mask_for(max);
    uint32_t v;
    while ((v = (next32() & mask)) > max) {
    }
    return v;
  }

  // interval(max), interval(max-1), ..., `want` of them (want <= max), into dst[0..want)
  // (dst has room for want+1).  The redraw-while-above-max loop is a coin flip per output
  // for the branch predictor; here every masked output is stored and the write position
  // advances by the comparison's result instead, so the only branches are loop bounds.
  void intervals_down(uint32_t max, int want, uint32_t* dst) {
    int got = 0;
    uint32_t cur = max;
    while (got < want) {
      if (pos == N) refill_tempered();
      const uint32_t mask = mask_for(cur);
      const uint32_t half = mask >> 1;  // the mask holds while cur > half
      int p = pos;
      while (p < N && got < want && cur > half) {
        const uint32_t v = out[p++] & mask;
        dst[got] = v;
        const uint32_t ok = v <= cur ? 1u : 0u;
        got += int(ok);
        cur -= ok;
      }
      pos = p;
    }
  }
};

// RandomState(epoch).shuffle(arange(n)) -> first batch entries.
static void sample_one(int64_t n_rows, int64_t batch, uint32_t seed, int32_t* scratch,
                       int32_t* out) {
  for (int64_t i = 0; i < n_rows; ++i) scratch[i] = int32_t(i);
  Mt19937 rng(seed);
  // the draws do not depend on the array, so a block of swap partners is drawn (and
  // their cache lines requested) before the block's swaps are done, in the same order
  constexpr int kBlock = 128;
  uint32_t js[kBlock + 1];
  for (int64_t i = n_rows - 1; i >= 1;) {
    const int cnt = int(std::min<int64_t>(kBlock, i));
    rng.intervals_down(uint32_t(i), cnt, js);
    for (int c = 0; c < cnt; ++c) __builtin_prefetch(scratch + js[c], 1);
    for (int c = 0; c < cnt; ++c) {
      const int32_t tmp = scratch[i - c];
      scratch[i - c] = scratch[js[c]];
      scratch[js[c]] = tmp;
    }
    i -= cnt;
  }
  std::memcpy(out, scratch, size_t(batch) * sizeof(int32_t));
}

}  // namespace rfm

using namespace rfm;

extern "C" {

int32_t rfm_version(void) { return RFM_VERSION; }

int32_t rfm_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return RFM_ERR_BAD_ARG;
  std::snprintf(buf, n, "%s", g_last_error.c_str());
  return RFM_OK;
}

int32_t rfm_sample_batches(int64_t n_rows, int64_t batch_size, int64_t epoch_begin,
                           int64_t n_epochs, int32_t* h_out_ids, int32_t n_threads) {
  return guarded([&] {
    RFM_REQUIRE(n_rows > 0 && n_rows < (int64_t(1) << 31), "n_rows=%lld out of range",
                (long long)n_rows);
    RFM_REQUIRE(batch_size > 0, "batch_size must be positive");
    RFM_REQUIRE(batch_size <= n_rows,
                "Cannot sample %lld out of arrays with dim %lld when replace is False",
                (long long)batch_size, (long long)n_rows);
    RFM_REQUIRE(epoch_begin >= 0 && epoch_begin + n_epochs <= (int64_t(1) << 32),
                "epoch seeds must fit 32 bits");
    RFM_REQUIRE(n_epochs >= 0 && (n_epochs == 0 || h_out_ids), "null output");
    if (n_epochs == 0) return;
    int nt = n_threads > 0 ? n_threads : int(std::thread::hardware_concurrency());
    nt = std::max(1, std::min<int>(nt, int(std::min<int64_t>(n_epochs, 256))));
    std::atomic<int64_t> next{0};
    auto worker = [&] {
      std::vector<int32_t> scratch(static_cast<size_t>(n_rows), 0);
      for (;;) {
        int64_t e = next.fetch_add(1);
        if (e >= n_epochs) break;
        sample_one(n_rows, batch_size, uint32_t(epoch_begin + e), scratch.data(),
                   h_out_ids + e * batch_size);
      }
    };
    if (nt == 1) {
      worker();
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < nt; ++t) pool.emplace_back(worker);
      for (auto& th : pool) th.join();
    }
  });
}

int32_t rfm_mf_schedule(const int32_t* h_users, const int32_t* h_items, int64_t batch,
                        int32_t n_users, int32_t n_items, int32_t* h_order,
                        int32_t* h_level_ptr, int32_t* h_n_levels) {
  return guarded([&] {
    RFM_REQUIRE(h_users && h_items && h_order && h_level_ptr && h_n_levels, "null pointer");
    RFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31), "batch out of range");
    // last level seen per user / item; only touched entries are reset afterwards
    static thread_local std::vector<int32_t> last_u, last_i;
    if (int64_t(last_u.size()) < n_users) last_u.assign(size_t(n_users), -1);
    if (int64_t(last_i.size()) < n_items) last_i.assign(size_t(n_items), -1);
    std::vector<int32_t> level(static_cast<size_t>(batch), 0);
    int32_t n_levels = 0;
    bool bad = false;
    for (int64_t s = 0; s < batch; ++s) {
      int32_t u = h_users[s], i = h_items[s];
      if (u < 0 || u >= n_users || i < 0 || i >= n_items) {
        bad = true;
        break;
      }
      int32_t lv = std::max(last_u[u], last_i[i]) + 1;
      level[s] = lv;
      last_u[u] = lv;
      last_i[i] = lv;
      n_levels = std::max(n_levels, lv + 1);
    }
    for (int64_t s = 0; s < batch; ++s) {
      int32_t u = h_users[s], i = h_items[s];
      if (u >= 0 && u < n_users) last_u[u] = -1;
      if (i >= 0 && i < n_items) last_i[i] = -1;
    }
    RFM_REQUIRE(!bad, "user/item id out of range");
    // stable counting sort of batch positions by level
    std::vector<int32_t> cnt(static_cast<size_t>(n_levels) + 1, 0);
    for (int64_t s = 0; s < batch; ++s) cnt[size_t(level[s]) + 1]++;
    for (int32_t l = 0; l < n_levels; ++l) cnt[size_t(l) + 1] += cnt[size_t(l)];
    for (int32_t l = 0; l <= n_levels; ++l) h_level_ptr[l] = cnt[size_t(l)];
    for (int64_t s = 0; s < batch; ++s) h_order[cnt[size_t(level[s])]++] = int32_t(s);
    *h_n_levels = n_levels;
  });
}

// the level-ordered record of rfm_mf.hip (MfEx)
struct HostMfEx {
  int32_t u, i, cslot, gap;
  double ry;
};

int32_t rfm_mf_schedule_ex(const int32_t* h_users, const int32_t* h_items, const double* h_y,
                           const double* h_pscore, int64_t batch, int32_t n_users,
                           int32_t n_items, int32_t cache_cap, void* h_ex,
                           int32_t* h_level_ptr, int32_t* h_n_levels, int32_t* h_cache_items,
                           int32_t* h_n_cached) {
  return guarded([&] {
    RFM_REQUIRE(h_users && h_items && h_y && h_pscore && h_ex && h_level_ptr && h_n_levels &&
                    h_cache_items && h_n_cached,
                "null pointer");
    RFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31) && cache_cap >= 0, "bad shape");
    static thread_local std::vector<int32_t> last_u, last_i, cnt_i, slot_i;
    if (int64_t(last_u.size()) < n_users) last_u.assign(size_t(n_users), -1);
    if (int64_t(last_i.size()) < n_items) {
      last_i.assign(size_t(n_items), -1);
      cnt_i.assign(size_t(n_items), 0);
      slot_i.assign(size_t(n_items), -1);
    }
    std::vector<int32_t> level(static_cast<size_t>(batch), 0), gap(static_cast<size_t>(batch), 0);
    int32_t n_levels = 0;
    bool bad = false;
    for (int64_t s = 0; s < batch; ++s) {
      const int32_t u = h_users[s], i = h_items[s];
      if (u < 0 || u >= n_users || i < 0 || i >= n_items) {
        bad = true;
        break;
      }
      const int32_t lv = std::max(last_u[u], last_i[i]) + 1;
      level[size_t(s)] = lv;
      // levels back to the previous writer of the user row: the row is final, and may be
      // read, that far ahead of lv
      gap[size_t(s)] = last_u[u] < 0 ? RFM_MF_NO_WRITER : lv - last_u[u];
      last_u[u] = lv;
      last_i[i] = lv;
      cnt_i[i]++;
      n_levels = std::max(n_levels, lv + 1);
    }
    // items that occur more than once, most frequent first, get the LDS slots
    std::vector<int32_t> repeated;
    if (!bad)
      for (int64_t s = 0; s < batch; ++s) {
        const int32_t i = h_items[s];
        if (cnt_i[i] >= 2 && slot_i[i] == -1) {
          slot_i[i] = -2;  // seen
          repeated.push_back(i);
        }
      }
    std::stable_sort(repeated.begin(), repeated.end(),
                     [&](int32_t x, int32_t y) { return cnt_i[x] > cnt_i[y]; });
    const int32_t n_cached = int32_t(std::min<size_t>(repeated.size(), size_t(cache_cap)));
    for (int32_t c = 0; c < n_cached; ++c) {
      slot_i[repeated[size_t(c)]] = c;
      h_cache_items[c] = repeated[size_t(c)];
    }
    // level-ordered records (stable: ascending batch position inside a level)
    std::vector<int32_t> cnt(static_cast<size_t>(n_levels) + 1, 0);
    if (!bad) {
      for (int64_t s = 0; s < batch; ++s) cnt[size_t(level[size_t(s)]) + 1]++;
      for (int32_t l = 0; l < n_levels; ++l) cnt[size_t(l) + 1] += cnt[size_t(l)];
      for (int32_t l = 0; l <= n_levels; ++l) h_level_ptr[l] = cnt[size_t(l)];
      HostMfEx* ex = static_cast<HostMfEx*>(h_ex);
      for (int64_t s = 0; s < batch; ++s) {
        const int32_t i = h_items[s];
        const int32_t cs = cnt_i[i] >= 2 ? slot_i[i] : -1;  // -2: repeated, no slot
        ex[cnt[size_t(level[size_t(s)])]++] =
            HostMfEx{h_users[s], i, cs, gap[size_t(s)], h_y[s] / h_pscore[s]};
      }
    }
    // reset only what this batch touched
    for (int64_t s = 0; s < batch; ++s) {
      const int32_t u = h_users[s], i = h_items[s];
      if (u >= 0 && u < n_users) last_u[u] = -1;
      if (i >= 0 && i < n_items) {
        last_i[i] = -1;
        cnt_i[i] = 0;
        slot_i[i] = -1;
      }
    }
    RFM_REQUIRE(!bad, "user/item id out of range");
    *h_n_levels = n_levels;
    *h_n_cached = n_cached;
  });
}

}  // extern "C"

// --------------------------------------------------------------------------
// 64-bit content fingerprint of a host buffer (what the upload caches compare before they
// trust a device copy: EVERY byte is read).  The buffer is cut into 1 MiB pieces hashed by
// the host threads -- four multiply-rotate lanes of 8 bytes each, the construction of the
// published xxHash64 round -- and the pieces' hashes are folded in piece order, so the result
// does not depend on the thread count.
// --------------------------------------------------------------------------
namespace rfm {
namespace {
constexpr uint64_t kP1 = 0x9E3779B185EBCA87ull, kP2 = 0xC2B2AE3D27D4EB4Full,
                   kP3 = 0x165667B19E3779F9ull, kP4 = 0x85EBCA77C2B2AE63ull,
                   kP5 = 0x27D4EB2F165667C5ull;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t lane_round(uint64_t acc, uint64_t in) { return rotl(acc + in * kP2, 31) * kP1; }
inline uint64_t avalanche(uint64_t h) {
  h ^= h >> 33;
  h *= kP2;
  h ^= h >> 29;
  h *= kP3;
  h ^= h >> 32;
  return h;
}
uint64_t hash_piece(const unsigned char* p, size_t n, uint64_t seed) {
  uint64_t a = seed + kP1 + kP2, b = seed + kP2, c = seed, d = seed - kP1;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
    a = lane_round(a, w[0]);
    b = lane_round(b, w[1]);
    c = lane_round(c, w[2]);
    d = lane_round(d, w[3]);
  }
  uint64_t h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18) + uint64_t(n);
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = rotl(h ^ lane_round(0, w), 27) * kP1 + kP4;
  }
  for (; i < n; ++i) h = rotl(h ^ (uint64_t(p[i]) * kP5), 11) * kP1;
  return avalanche(h);
}
}  // namespace
}  // namespace rfm

extern "C" int32_t rfm_hash_bytes(const void* h_data, int64_t n_bytes, int32_t n_threads,
                                  uint64_t* h_out) {
  return rfm::guarded([&] {
    RFM_REQUIRE(h_out && n_bytes >= 0 && (h_data || n_bytes == 0), "bad arguments");
    constexpr int64_t kPiece = 1 << 20;
    const int64_t n_pieces = (n_bytes + kPiece - 1) / kPiece;
    std::vector<uint64_t> piece(size_t(std::max<int64_t>(n_pieces, 1)), 0);
    const unsigned char* base = static_cast<const unsigned char*>(h_data);
    const int threads = int(std::max<int64_t>(1, std::min<int64_t>(std::max(n_threads, 1), n_pieces)));
    std::atomic<int64_t> next{0};
    const auto work = [&] {
      for (int64_t i = next.fetch_add(1); i < n_pieces; i = next.fetch_add(1)) {
        const int64_t lo = i * kPiece, len = std::min<int64_t>(kPiece, n_bytes - lo);
        piece[size_t(i)] = rfm::hash_piece(base + lo, size_t(len), uint64_t(i));
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    uint64_t h = rfm::kP5 + uint64_t(n_bytes);
    for (int64_t i = 0; i < n_pieces; ++i) h = rfm::rotl(h ^ piece[size_t(i)], 27) * rfm::kP1 + rfm::kP4;
    *h_out = rfm::avalanche(h);
  });
}
