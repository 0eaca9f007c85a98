// bpr_refresh.hip — AdaptiveSampler.update_stats on device
// (reference: revisit_bpr/modules/neg_samplers.py:126-132; experiments/bpr/exp.py:344-354).
//
// The reference snapshots Qᵀ [d, I] and, at every sample, argsorts one masked row of it.  The only
// thing those argsorts ever use of the snapshot is each factor's ORDER of the items, so the
// snapshot kept here is order[f][:] = argsort_desc(Q[:, f]) (stable, ties by item id) plus
// sigma_f = unbiased std of Q[1:, f].  Sort = rocPRIM segmented radix sort, d segments of I keys.
#include <hipcub/hipcub.hpp>

#include "bpr_ctx.h"

namespace bpr {

// Q [I, d] → T [d, I] through a padded 32x32 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ Q,
                                                   float* __restrict__ T, int64_t I, int d) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  const int f0 = blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int64_t i = i0 + ty + r;
    const int f = f0 + tx;
    tile[ty + r][tx] = (i < I && f < d) ? Q[i * d + f] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int f = f0 + ty + r;
    const int64_t i = i0 + tx;
    if (f < d && i < I) T[(int64_t)f * I + i] = tile[tx][ty + r];
  }
}

// sigma_f = std(Q[1:, f], unbiased): one block per factor over the contiguous transposed row
__global__ __launch_bounds__(256) void k_sigma(const float* __restrict__ T, int64_t I,
                                               float* __restrict__ sigma) {
  __shared__ double red[256];
  const float* row = T + (int64_t)blockIdx.x * I;
  double s = 0.0;
  for (int64_t i = 1 + threadIdx.x; i < I; i += 256) s += (double)row[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double mean = red[0] / (double)(I - 1);
  __syncthreads();
  double ss = 0.0;
  for (int64_t i = 1 + threadIdx.x; i < I; i += 256) {
    const double c = (double)row[i] - mean;
    ss += c * c;
  }
  red[threadIdx.x] = ss;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sigma[blockIdx.x] = (float)sqrt(red[0] / (double)(I - 2));
}

__global__ void k_iota(int32_t* ids, int32_t* offs, int64_t I, int d) {
  const int64_t n = (int64_t)d * I;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x)
    ids[k] = (int32_t)(k % I);
  if (blockIdx.x == 0)
    for (int f = threadIdx.x; f <= d; f += blockDim.x) offs[f] = (int32_t)((int64_t)f * I);
}

void refresh_free(bpr_ctx* c) {
  hipFree(c->order);
  hipFree(c->sigma);
  hipFree(c->keysT);
  hipFree(c->keys_sorted);
  hipFree(c->ids_in);
  hipFree(c->seg_offsets);
  hipFree(c->sort_tmp);
  c->order = nullptr;
  c->sigma = nullptr;
  c->keysT = c->keys_sorted = nullptr;
  c->ids_in = c->seg_offsets = nullptr;
  c->sort_tmp = nullptr;
  c->sort_tmp_bytes = 0;
  c->have_snapshot = false;
}

int refresh_impl(bpr_ctx* c) {
  const int64_t I = c->I;
  const int d = c->d;
  const int64_t n = (int64_t)d * I;
  if (n >= (int64_t)1 << 31) {
    set_error("bpr_adaptive_refresh: d*I must be < 2^31");
    return BPR_ERR_UNSUPPORTED;
  }
  if (I < 3) {
    set_error("bpr_adaptive_refresh: need at least 3 item rows");
    return BPR_ERR_INVALID;
  }
  if (c->order == nullptr) {
    BPR_HIP_CHECK(hipMalloc(&c->order, sizeof(int32_t) * n));
    BPR_HIP_CHECK(hipMalloc(&c->sigma, sizeof(float) * d));
    BPR_HIP_CHECK(hipMalloc(&c->keysT, sizeof(float) * n));
    BPR_HIP_CHECK(hipMalloc(&c->keys_sorted, sizeof(float) * n));
    BPR_HIP_CHECK(hipMalloc(&c->ids_in, sizeof(int32_t) * n));
    BPR_HIP_CHECK(hipMalloc(&c->seg_offsets, sizeof(int32_t) * (d + 1)));
    hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, c->stream, c->ids_in, c->seg_offsets, I,
                       d);
    size_t bytes = 0;
    BPR_HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(
        nullptr, bytes, c->keysT, c->keys_sorted, c->ids_in, c->order, (int)n, d, c->seg_offsets,
        c->seg_offsets + 1, 0, 32, c->stream));
    BPR_HIP_CHECK(hipMalloc(&c->sort_tmp, bytes > 0 ? bytes : 16));
    c->sort_tmp_bytes = bytes;
  }
  dim3 tgrid((unsigned)((I + 31) / 32), (unsigned)((d + 31) / 32));
  hipLaunchKernelGGL(k_transpose, tgrid, dim3(256), 0, c->stream, c->Q, c->keysT, I, d);
  hipLaunchKernelGGL(k_sigma, dim3(d), dim3(256), 0, c->stream, c->keysT, I, c->sigma);
  size_t bytes = c->sort_tmp_bytes;
  BPR_HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(
      c->sort_tmp, bytes, c->keysT, c->keys_sorted, c->ids_in, c->order, (int)n, d,
      c->seg_offsets, c->seg_offsets + 1, 0, 32, c->stream));
  BPR_HIP_CHECK(hipGetLastError());
  c->have_snapshot = true;
  return BPR_OK;
}

}  // namespace bpr
