// bpr_eval.hip — the one evaluation kernel of libbprcore: ROC-AUC of a block of score rows (r6).
//
// The reference's RocAucMany / RocAucManySlow (revisit_bpr/metrics/auc.py:70-130 of the reference: every
// (positive, negative) pair of a row; the ML-20M / MSD configs list it among their 14 metrics) compares
// [B, I, I] scores or loops over users in Python; the PyTorch-ROCm restatement here sorts every row
// (revisit_bpr/metrics/auc.py: one [B, I] sort per block, ~3/4 of an evaluation's time).  The quantity needs
// far less: per row, for each of its T positives the number of negatives scored strictly below it.  One workgroup
// per row: the T positive scores are ranked in LDS (counting, T^2 / 256 per thread: T is tens), every score of the
// row finds the number of positives <= it by binary search in LDS and bumps one bin of a (T + 1)-bin histogram; a
// prefix sum of the bins gives "scores strictly below positive p", minus the positives among them.  One pass over
// the scores, bit-for-bit the counts of the sort-based form.
#include <hip/hip_runtime.h>
#include <math.h>

#include "bpr_host.h"

namespace bpr {

constexpr int AUC_T_MAX = 4096;  // positives per row held in LDS (more: the caller's sort-based path)

__global__ __launch_bounds__(256) void k_auc_rows(const float* __restrict__ scores, int64_t I,
                                                  const int64_t* __restrict__ pos_indptr,
                                                  const int32_t* __restrict__ pos_items, float* __restrict__ auc) {
  __shared__ float s_sorted[AUC_T_MAX];
  __shared__ uint32_t s_hist[AUC_T_MAX + 1];
  const int64_t row = blockIdx.x;
  const int64_t p0 = pos_indptr[row];
  const int T = (int)(pos_indptr[row + 1] - p0);
  const float* __restrict__ s = scores + row * I;
  if (T <= 0 || T > AUC_T_MAX) {  // no positives: 0 / 0 as the metric classes; too many: NaN marks "not computed"
    if (threadIdx.x == 0) auc[row] = nanf("");
    return;
  }
  // rank the positives' scores (ascending, ties by index): sorted[rank] = score  (staged through the bins' LDS)
  float* const s_pos = reinterpret_cast<float*>(s_hist);
  for (int p = threadIdx.x; p < T; p += blockDim.x) s_pos[p] = s[pos_items[p0 + p]];
  __syncthreads();
  for (int p = threadIdx.x; p < T; p += blockDim.x) {
    const float sp = s_pos[p];
    int rank = 0;
    for (int q = 0; q < T; ++q) {
      const float sq = s_pos[q];
      rank += (sq < sp || (sq == sp && q < p)) ? 1 : 0;
    }
    s_sorted[rank] = sp;
  }
  __syncthreads();
  for (int b = threadIdx.x; b <= T; b += blockDim.x) s_hist[b] = 0u;
  __syncthreads();
  // every score of the row: bin = #{positives <= it}; a thread counts runs of one bin privately
  uint32_t run_bin = 0xFFFFFFFFu, run_cnt = 0u;
  for (int64_t j = threadIdx.x; j < I; j += blockDim.x) {
    const float sj = s[j];
    int lo = 0, hi = T;  // upper bound: first index with sorted > sj
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_sorted[mid] <= sj) lo = mid + 1; else hi = mid;
    }
    if ((uint32_t)lo == run_bin) {
      ++run_cnt;
    } else {
      if (run_cnt) atomicAdd(&s_hist[run_bin], run_cnt);
      run_bin = (uint32_t)lo;
      run_cnt = 1u;
    }
  }
  if (run_cnt) atomicAdd(&s_hist[run_bin], run_cnt);
  __syncthreads();
  // below[p] = scores strictly below sorted[p] = sum of bins 0..p; the positives among them = first index of
  // sorted[p]'s value (ties are not "strictly below")
  if (threadIdx.x == 0) {
    double wins = 0.0;
    uint64_t below = 0;
    int first_equal = 0;
    for (int p = 0; p < T; ++p) {
      below += s_hist[p];
      if (p > 0 && s_sorted[p] != s_sorted[p - 1]) first_equal = p;
      wins += (double)(below - (uint64_t)first_equal);
    }
    auc[row] = (float)(wins / ((double)T * (double)(I - T)));
  }
}

}  // namespace bpr

extern "C" int bpr_auc_rows(const float* scores, int64_t n, int64_t I, const int64_t* pos_indptr,
                            const int32_t* pos_items, float* auc_out, void* hip_stream) {
  using namespace bpr;
  if (n < 0 || I <= 0 || (n > 0 && (!scores || !pos_indptr || !pos_items || !auc_out)))
    return fail(BPR_ERR_INVALID, "bpr_auc_rows: bad argument");
  if (n == 0) return BPR_OK;
  hipLaunchKernelGGL(k_auc_rows, dim3((unsigned)n), dim3(256), 0, (hipStream_t)hip_stream, scores, I, pos_indptr,
                     pos_items, auc_out);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}
