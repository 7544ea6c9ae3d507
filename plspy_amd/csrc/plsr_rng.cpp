// Bit-exact native restatement of the NumPy *legacy* RandomState draws the
// reference makes per resample (host code, no GPU):
//
//   np.random.permutation(x)          resample.py:66,71,77; split_half_resampling.py:136,271,282,316
//   np.random.choice(n, n, True)      resample.py:141
//
// NumPy's algorithms (numpy/random/_mt19937, mtrand.pyx, distributions.c;
// pinned by tests against numpy 2.2.6 in tests/test_native_rng.py):
//   * MT19937, 624-word key + position; next_uint32 with the standard tempering.
//   * permutation / shuffle of a 1-d array: for i = n-1 .. 1: j = interval(i);
//     swap(x[i], x[j]), where interval(max) draws next_uint32 & mask (mask =
//     smallest 2^b - 1 >= max) until the value is <= max.
//   * choice(n, size, replace=True) = randint(0, n, size): the same masked
//     rejection with max = n - 1, one accepted value per output; n == 1
//     consumes nothing.
//
// The caller hands over np.random.get_state()'s key/pos, gets them back
// advanced, and puts them into np.random.set_state(), so the global stream
// continues exactly as if NumPy had made the draws.  Host index generation was
// 5x the GPU time of a phase; this removes the Python loop.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/plsr.h"

namespace {

constexpr int N = 624, M = 397;

struct MT {
  uint32_t *key;
  int32_t pos;
  void gen() {
    constexpr uint32_t MATRIX_A = 0x9908b0dfU, UPPER = 0x80000000U, LOWER = 0x7fffffffU;
    uint32_t y;
    int kk;
    for (kk = 0; kk < N - M; kk++) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; kk < N - 1; kk++) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (key[N - 1] & UPPER) | (key[0] & LOWER);
    key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    pos = 0;
  }
  inline uint32_t next() {
    if (pos == N) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
  }
  // legacy random_interval / buffered_bounded_masked_uint32 for max < 2^32
  inline uint32_t interval(uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    while ((v = (next() & mask)) > max) {
    }
    return v;
  }
  inline void shuffle(int32_t *x, int n) {
    for (int i = n - 1; i >= 1; --i) {
      const uint32_t j = interval((uint32_t)i);
      const int32_t t = x[i];
      x[i] = x[j];
      x[j] = t;
    }
  }
};

}  // namespace

extern "C" int plsr_rng_permutation(uint32_t *key, int32_t *pos, int32_t n, int32_t count,
                                    int32_t *out) {
  if (!key || !pos || !out || n <= 0 || count < 0 || *pos < 0 || *pos > N) return PLSR_EINVAL;
  MT mt{key, *pos};
  for (int c = 0; c < count; ++c) {
    int32_t *x = out + (int64_t)c * n;
    for (int i = 0; i < n; ++i) x[i] = i;
    mt.shuffle(x, n);
  }
  *pos = mt.pos;
  return PLSR_OK;
}

extern "C" int plsr_rng_permutation_seq(uint32_t *key, int32_t *pos, const int32_t *sizes, int32_t nsizes,
                                        int32_t count, int32_t *out) {
  if (!key || !pos || !sizes || !out || nsizes <= 0 || count < 0 || *pos < 0 || *pos > N) return PLSR_EINVAL;
  for (int s = 0; s < nsizes; ++s)
    if (sizes[s] <= 0) return PLSR_EINVAL;
  MT mt{key, *pos};
  int32_t *x = out;
  for (int c = 0; c < count; ++c)
    for (int s = 0; s < nsizes; ++s) {
      const int n = sizes[s];
      for (int i = 0; i < n; ++i) x[i] = i;
      mt.shuffle(x, n);
      x += n;
    }
  *pos = mt.pos;
  return PLSR_OK;
}

extern "C" int plsr_rng_task_permutations(uint32_t *key, int32_t *pos, const int32_t *table,
                                          int32_t nsub, int32_t nc, int32_t count, int32_t *out) {
  if (!key || !pos || !table || !out || nsub <= 0 || nc <= 0 || count < 0 || *pos < 0 || *pos > N)
    return PLSR_EINVAL;
  MT mt{key, *pos};
  std::vector<int32_t> within((size_t)nsub * nc), colbuf(nsub);
  for (int r = 0; r < count; ++r) {
    // resample.py:66 -- every subject's conditions
    std::memcpy(within.data(), table, sizeof(int32_t) * (size_t)nsub * nc);
    for (int s = 0; s < nsub; ++s) mt.shuffle(within.data() + (size_t)s * nc, nc);
    // :69-73 -- every condition slot over all subjects; slot-major output
    int32_t *o = out + (int64_t)r * nsub * nc;
    for (int c = 0; c < nc; ++c) {
      for (int s = 0; s < nsub; ++s) colbuf[s] = within[(size_t)s * nc + c];
      mt.shuffle(colbuf.data(), nsub);
      std::memcpy(o + (size_t)c * nsub, colbuf.data(), sizeof(int32_t) * nsub);
    }
  }
  *pos = mt.pos;
  return PLSR_OK;
}

extern "C" int plsr_rng_bootstraps(uint32_t *key, int32_t *pos, const int32_t *table,
                                   const int32_t *group_subjects, int32_t ngroups, int32_t nc,
                                   int32_t count, int32_t *out) {
  if (!key || !pos || !table || !group_subjects || !out || ngroups <= 0 || nc <= 0 || count < 0 ||
      *pos < 0 || *pos > N)
    return PLSR_EINVAL;
  MT mt{key, *pos};
  int64_t nsub = 0;
  int maxg = 0;
  for (int g = 0; g < ngroups; ++g) {
    if (group_subjects[g] <= 0) return PLSR_EINVAL;
    nsub += group_subjects[g];
    if (group_subjects[g] > maxg) maxg = group_subjects[g];
  }
  std::vector<int32_t> pick(maxg);
  const int64_t n = nsub * nc;
  for (int r = 0; r < count; ++r) {
    int32_t *o = out + (int64_t)r * n;
    int64_t sub0 = 0;      // first subject row of the group in `table`
    for (int g = 0; g < ngroups; ++g) {
      const int ns = group_subjects[g];
      for (int s = 0; s < ns; ++s) pick[s] = (int32_t)mt.interval((uint32_t)(ns - 1));   // resample.py:141
      for (int c = 0; c < nc; ++c)                                                      // :143-151
        for (int s = 0; s < ns; ++s) *o++ = table[(sub0 + pick[s]) * nc + c];
      sub0 += ns;
    }
  }
  *pos = mt.pos;
  return PLSR_OK;
}

// ---------------------------------------------------------------------------
// multiblock candidates: two draws per try, interleaved on the one stream
// ---------------------------------------------------------------------------
namespace {
void one_task_permutation(MT &mt, const int32_t *table, int nsub, int nc, std::vector<int32_t> &within,
                          std::vector<int32_t> &colbuf, int32_t *o) {
  std::memcpy(within.data(), table, sizeof(int32_t) * (size_t)nsub * nc);
  for (int s = 0; s < nsub; ++s) mt.shuffle(within.data() + (size_t)s * nc, nc);
  for (int c = 0; c < nc; ++c) {
    for (int s = 0; s < nsub; ++s) colbuf[s] = within[(size_t)s * nc + c];
    mt.shuffle(colbuf.data(), nsub);
    std::memcpy(o + (size_t)c * nsub, colbuf.data(), sizeof(int32_t) * nsub);
  }
}

void one_bootstrap(MT &mt, const int32_t *table, const int32_t *group_subjects, int ngroups, int nc,
                   std::vector<int32_t> &pick, int32_t *o) {
  int64_t sub0 = 0;
  for (int g = 0; g < ngroups; ++g) {
    const int ns = group_subjects[g];
    for (int s = 0; s < ns; ++s) pick[s] = (int32_t)mt.interval((uint32_t)(ns - 1));
    for (int c = 0; c < nc; ++c)
      for (int s = 0; s < ns; ++s) *o++ = table[(sub0 + pick[s]) * nc + c];
    sub0 += ns;
  }
}
}  // namespace

extern "C" int plsr_rng_mb_permutations(uint32_t *key, int32_t *pos, const int32_t *table, int32_t nsub,
                                        int32_t nc, int32_t nrows, int32_t count, int32_t *out_task,
                                        int32_t *out_rows) {
  if (!key || !pos || !table || !out_task || !out_rows || nsub <= 0 || nc <= 0 || nrows <= 0 || count < 0 ||
      *pos < 0 || *pos > N)
    return PLSR_EINVAL;
  MT mt{key, *pos};
  std::vector<int32_t> within((size_t)nsub * nc), colbuf(nsub);
  for (int r = 0; r < count; ++r) {
    one_task_permutation(mt, table, nsub, nc, within, colbuf, out_task + (int64_t)r * nsub * nc);   // :343
    int32_t *x = out_rows + (int64_t)r * nrows;                                                      // :347
    for (int i = 0; i < nrows; ++i) x[i] = i;
    mt.shuffle(x, nrows);
  }
  *pos = mt.pos;
  return PLSR_OK;
}

extern "C" int plsr_rng_mb_bootstraps(uint32_t *key, int32_t *pos, const int32_t *table,
                                      const int32_t *group_subjects, int32_t ngroups, int32_t nc,
                                      const int32_t *btable, int32_t bnc, int32_t count, int32_t *out_task,
                                      int32_t *out_beh) {
  if (!key || !pos || !table || !btable || !group_subjects || !out_task || !out_beh || ngroups <= 0 ||
      nc <= 0 || bnc <= 0 || count < 0 || *pos < 0 || *pos > N)
    return PLSR_EINVAL;
  MT mt{key, *pos};
  int64_t nsub = 0;
  int maxg = 0;
  for (int g = 0; g < ngroups; ++g) {
    if (group_subjects[g] <= 0) return PLSR_EINVAL;
    nsub += group_subjects[g];
    maxg = std::max(maxg, (int)group_subjects[g]);
  }
  std::vector<int32_t> pick(maxg);
  for (int r = 0; r < count; ++r) {
    one_bootstrap(mt, table, group_subjects, ngroups, nc, pick, out_task + (int64_t)r * nsub * nc);   // :547
    one_bootstrap(mt, btable, group_subjects, ngroups, bnc, pick, out_beh + (int64_t)r * nsub * bnc); // :551
  }
  *pos = mt.pos;
  return PLSR_OK;
}
