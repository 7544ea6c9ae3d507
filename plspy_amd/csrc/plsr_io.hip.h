// F4: the upstream feed -- X built on the device.
//
// plspy/io/io.py:427-460 (apply_mask_matrices: m[broadcast(mask)], i.e. for every time point
// the voxels the mask selects, in C order) and :680-698 (concat_flatten_all_groups: all
// subjects stacked, one row each) as a stream compaction of the mask followed by a row gather
// that writes straight into the rows of X.  HBM-bound byte work: the mask is read twice
// (1 B per voxel), every selected value once (8-B sectors of the source volume), X written once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plsr {

constexpr int IO_THREADS = 256;
constexpr int IO_PER_THREAD = 8;
constexpr int IO_BLOCK = IO_THREADS * IO_PER_THREAD;      // mask bytes per workgroup

// exclusive prefix of the per-thread counts of a workgroup (returns the block total)
__device__ __forceinline__ int block_exclusive_scan(int mine, int *scratch, int &total) {
  const int tid = threadIdx.x;
  scratch[tid] = mine;
  __syncthreads();
  for (int off = 1; off < IO_THREADS; off <<= 1) {
    const int v = tid >= off ? scratch[tid - off] : 0;
    __syncthreads();
    scratch[tid] += v;
    __syncthreads();
  }
  total = scratch[IO_THREADS - 1];
  return scratch[tid] - mine;
}

// selected voxels per block of IO_BLOCK mask bytes
__global__ __launch_bounds__(IO_THREADS) void mask_count_kernel(const uint8_t *mask, int64_t n, int64_t *block_cnt) {
  __shared__ int scratch[IO_THREADS];
  const int64_t base = (int64_t)blockIdx.x * IO_BLOCK + (int64_t)threadIdx.x * IO_PER_THREAD;
  int mine = 0;
#pragma unroll
  for (int u = 0; u < IO_PER_THREAD; ++u) mine += base + u < n && mask[base + u] != 0;
  int total;
  block_exclusive_scan(mine, scratch, total);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

// exclusive prefix over the blocks (one workgroup; in place) and the grand total
__global__ __launch_bounds__(IO_THREADS) void mask_scan_kernel(int64_t *block_cnt, int64_t nblocks, int64_t *total) {
  __shared__ int scratch[IO_THREADS];
  int64_t carry = 0;
  for (int64_t b0 = 0; b0 < nblocks; b0 += IO_THREADS) {
    const int64_t b = b0 + threadIdx.x;
    const int mine = b < nblocks ? (int)block_cnt[b] : 0;
    int sum;
    const int ex = block_exclusive_scan(mine, scratch, sum);
    if (b < nblocks) block_cnt[b] = carry + ex;
    carry += sum;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// ascending flat indices of the selected voxels
__global__ __launch_bounds__(IO_THREADS) void mask_scatter_kernel(const uint8_t *mask, int64_t n, const int64_t *block_off,
                                                                 int64_t *idx) {
  __shared__ int scratch[IO_THREADS];
  const int64_t base = (int64_t)blockIdx.x * IO_BLOCK + (int64_t)threadIdx.x * IO_PER_THREAD;
  bool sel[IO_PER_THREAD];
  int mine = 0;
#pragma unroll
  for (int u = 0; u < IO_PER_THREAD; ++u) {
    sel[u] = base + u < n && mask[base + u] != 0;
    mine += sel[u];
  }
  int total;
  int64_t o = block_off[blockIdx.x] + block_exclusive_scan(mine, scratch, total);
#pragma unroll
  for (int u = 0; u < IO_PER_THREAD; ++u)
    if (sel[u]) idx[o++] = base + u;
}

// out[r][j] = in[r][idx[j]]  (rows = time points / subjects; out row stride ld_out: the rows of X)
template <class T>
__global__ __launch_bounds__(IO_THREADS) void mask_apply_kernel(const T *in, int64_t ld_in, const int64_t *idx,
                                                               int64_t nsel, double *out, int64_t ld_out) {
  const int64_t j = (int64_t)blockIdx.x * IO_THREADS + threadIdx.x;
  if (j >= nsel) return;
  const int64_t r = blockIdx.y;
  out[r * ld_out + j] = (double)in[r * ld_in + idx[j]];
}

}  // namespace plsr
