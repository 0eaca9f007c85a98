// bpr_refresh.hip — AdaptiveSampler.update_stats on device
// (reference: revisit_bpr/modules/neg_samplers.py:126-132; experiments/bpr/exp.py:344-354).
//
// The reference snapshots Qᵀ [d, I] and, at every sample, argsorts one masked row of it.  The only
// thing those argsorts ever use of the snapshot is each factor's ORDER of the items, so the
// snapshot kept here is order[f][:] = argsort_desc(Q[:, f]) (stable, ties by item id) plus
// sigma_f = unbiased std of Q[1:, f].  Sort = ONE device-wide rocPRIM radix sort over composite
// 64-bit keys (factor << 32 | descending-orderable float bits), restricted to the 32 + log2(d)
// significant bits — measured 2.1x faster than DeviceSegmentedRadixSort / DeviceSegmentedSort for
// d = 128 segments of 20 k keys (tools/ubench/sort_bench.hip: 0.23 ms vs 0.49 ms).
#include <string.h>  // (rocprim's texture iterator calls the host memset)
#include <rocprim/rocprim.hpp>

// bits per pass of the in-LDS block radix sort (0 = rocPRIM's default, 8)
#ifndef BPR_SORT_RADIX_BITS
#define BPR_SORT_RADIX_BITS 0
#endif

#include <stdlib.h>

#include <algorithm>
#include <vector>
#include <stdio.h>

#include "bpr_ctx.h"

namespace bpr {

// Q [I, d] → T [d, I] through a padded 32x32 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ Q,
                                                   float* __restrict__ T, int64_t I, int d,
                                                   double* __restrict__ sig_acc) {
  __shared__ float tile[32][33];
  // also clears the per-factor sigma accumulators of a split sort (saves a memset launch)
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (int k = threadIdx.x; k < 2 * d; k += 256) sig_acc[k] = 0.0;
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

// ---------------------------------------------------------------------------------------------
// Fast path (I <= 36,864 items): ONE 1024-thread workgroup per factor sorts the whole column in
// registers + LDS (rocprim::block_radix_sort — LSD radix, stable, so ties keep ascending item id) and
// computes sigma_f on the way: d independent workgroups, no inter-block traffic, no memsets.
// The column sits in 1024 x ITEMS registers; the ~100 KiB of LDS is the radix exchange buffer.
// ---------------------------------------------------------------------------------------------
// Block (s, f) sorts sub-column s of factor f: items [s*len, min((s+1)*len, I)).  With SUB == 1 the
// result is the final order; otherwise sorted (key, id) runs go to scratch for k_merge_runs, so
// that 2 (or 4) workgroups per factor share the work and all 256 CUs are busy.
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_sort_sub(const float* __restrict__ T, int64_t I,
                                                   int64_t len, int32_t* __restrict__ order,
                                                   float* __restrict__ keys_out,
                                                   int32_t* __restrict__ ids_out,
                                                   float* __restrict__ sigma,
                                                   double* __restrict__ sig_acc,
                                                   const int32_t* __restrict__ only_flagged = nullptr) {
  using Sort = rocprim::block_radix_sort<float, 1024, ITEMS, uint16_t, 1, 1, BPR_SORT_RADIX_BITS>;
  __shared__ union {
    typename Sort::storage_type sort;
    double red[2][16];
  } sm;
  const int f = blockIdx.y;
  if (only_flagged != nullptr && only_flagged[2 * f] >= 0) return;  // (the fallback of k_sort_binned_split)
  const int64_t base = (int64_t)blockIdx.x * len;
  const int64_t cnt = min(len, I - base);
  const bool single = gridDim.x == 1;
  const float* row = T + (int64_t)f * I;
  const int t = threadIdx.x;
  float keys[ITEMS];
  uint16_t vals[ITEMS];
  double s1 = 0.0, s2 = 0.0;
  const float first = row[1];  // shift: removes the mean's magnitude from the sums
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t l = (int64_t)t * ITEMS + k;  // blocked arrangement: sort stability = item order
    const bool valid = l < cnt;
    const float v = valid ? row[base + l] : -__builtin_huge_valf();
    keys[k] = v;
    vals[k] = (uint16_t)l;
    if (valid && base + l >= 1) {
      const double c = (double)v - (double)first;
      s1 += c;
      s2 += c * c;
    }
  }
  // sigma_f = unbiased std over rows 1..I-1 (neg_samplers.py:132)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if ((t & 63) == 0) {
    sm.red[0][t >> 6] = s1;
    sm.red[1][t >> 6] = s2;
  }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) {
      a += sm.red[0][w];
      b += sm.red[1][w];
    }
    if (single) {
      const double n = (double)(I - 1);
      sigma[f] = (float)sqrt(fmax(b - a * a / n, 0.0) / (n - 1.0));
    } else {  // finalised by k_merge_runs (last level)
      atomicAdd(&sig_acc[2 * f + 0], a);
      atomicAdd(&sig_acc[2 * f + 1], b);
    }
  }
  __syncthreads();
  Sort().sort_desc_to_striped(keys, vals, sm.sort);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t pos = (int64_t)k * 1024 + t;
    if (pos < cnt) {
      const int64_t o = (int64_t)f * I + base + pos;
      if (single) {
        order[o] = (int32_t)(base + vals[k]);
      } else {
        keys_out[o] = keys[k];
        ids_out[o] = (int32_t)(base + vals[k]);
      }
    }
  }
}

// Merge neighbouring sorted runs of length `run` (descending keys; on equal keys the left run —
// lower item ids — goes first, which keeps the order identical to a stable full sort).
// Two-level merge path.  A workgroup owns MERGE_TILE consecutive outputs of one pair of runs: two
// lanes find where the tile starts and ends in both runs (binary search along the cross diagonals,
// in HBM), the block copies those two slices — at most MERGE_TILE elements together — into LDS with
// coalesced loads, every thread then finds its own MERGE_PER_THREAD outputs by a second diagonal
// search in LDS and merges serially out of LDS; results go back through LDS so the stores are
// coalesced as well.  LDS indices are padded by one word per 16 so the threads' serial walks spread
// over the banks.  HBM traffic: keys + ids read once, written once (ids only on the last level).
#ifndef BPR_MERGE_PER_THREAD
#define BPR_MERGE_PER_THREAD 16
#endif
constexpr int MERGE_PER_THREAD = BPR_MERGE_PER_THREAD;
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_TILE = MERGE_THREADS * MERGE_PER_THREAD;
__device__ __forceinline__ int merge_pad(int k) { return k + k / MERGE_PER_THREAD; }

// number of elements the first `k` merged outputs take from run A (lenA) — B (lenB) gets k - that
template <typename KeyA, typename KeyB>
__device__ __forceinline__ int64_t merge_split(int64_t k, int64_t lenA, int64_t lenB,
                                               const KeyA& A, const KeyB& B) {
  int64_t lo = max((int64_t)0, k - lenB), hi = min(k, lenA);
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (A(mid) >= B(k - mid - 1)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(MERGE_THREADS) void k_merge_runs(
    const float* __restrict__ keys_in, const int32_t* __restrict__ ids_in, int64_t I, int64_t run,
    int tiles_per_pair, float* __restrict__ keys_out, int32_t* __restrict__ ids_out, int last,
    float* __restrict__ sigma, const double* __restrict__ sig_acc,
    const int32_t* __restrict__ only_flagged = nullptr) {
  __shared__ float lk[MERGE_TILE + MERGE_THREADS + 1];
  __shared__ int32_t lv[MERGE_TILE + MERGE_THREADS + 1];
  __shared__ int64_t cut[2];
  const int f = blockIdx.y;
  const int t = threadIdx.x;
  if (only_flagged != nullptr && only_flagged[2 * f] >= 0) return;
  if (last && blockIdx.x == 0 && t == 0) {
    const double a = sig_acc[2 * f], b = sig_acc[2 * f + 1], n = (double)(I - 1);
    sigma[f] = (float)sqrt(fmax(b - a * a / n, 0.0) / (n - 1.0));
  }
  const int64_t pair = blockIdx.x / tiles_per_pair;
  const int64_t tile = blockIdx.x % tiles_per_pair;
  const int64_t a0 = pair * 2 * run;
  if (a0 >= I) return;
  const int64_t lenA = min(run, I - a0);
  const int64_t b0 = a0 + lenA;
  const int64_t lenB = max((int64_t)0, min(run, I - b0));
  const int64_t k0 = tile * MERGE_TILE;
  if (k0 >= lenA + lenB) return;
  const int64_t k1 = min(k0 + MERGE_TILE, lenA + lenB);
  const float* K = keys_in + (int64_t)f * I;
  const int32_t* V = ids_in + (int64_t)f * I;
  if (t < 128) {
    // 64-ary diagonal search in HBM: wave 0 finds the start of the tile, wave 1 its end.  The
    // predicate "A(x) >= B(k-x-1)" is true on a prefix of [lo, hi); every lane probes one point per
    // step, so the interval shrinks 65-fold per round trip (3 instead of 17 dependent loads).
    const int l = t & 63;
    const int64_t k = t < 64 ? k0 : k1;
    int64_t lo = max((int64_t)0, k - lenB), hi = min(k, lenA);
    while (lo < hi) {
      const int64_t span = hi - lo;
      const bool fine = span <= 64;  // last step: one lane per remaining position
      const int64_t x = fine ? lo + l : lo + (int64_t)(l + 1) * span / 65;
      const bool in = x < hi;
      const bool pred = in && K[a0 + x] >= K[b0 + (k - x - 1)];
      const int c = __popcll(__ballot(pred));  // trues form a prefix of the probes
      if (fine) {
        lo += c;
        hi = lo;
      } else {
        const int64_t below = c == 0 ? lo : lo + (int64_t)c * span / 65 + 1;
        const int64_t above = c == 64 ? hi : lo + (int64_t)(c + 1) * span / 65;
        lo = below;
        hi = above;
      }
    }
    if (l == 0) cut[t >> 6] = lo;
  }
  __syncthreads();
  const int64_t a_lo = cut[0], a_hi = cut[1];
  const int64_t b_lo = k0 - a_lo, b_hi = k1 - a_hi;
  const int nA = (int)(a_hi - a_lo), nB = (int)(b_hi - b_lo);
  for (int x = t; x < nA + nB; x += MERGE_THREADS) {
    const int64_t src = x < nA ? a0 + a_lo + x : b0 + b_lo + (x - nA);
    lk[merge_pad(x)] = K[src];
    lv[merge_pad(x)] = V[src];
  }
  __syncthreads();
  const int n_tile = (int)(k1 - k0);
  const int kk = min(t * MERGE_PER_THREAD, n_tile);
  const int n_out = min(MERGE_PER_THREAD, n_tile - kk);
  int a = (int)merge_split(kk, nA, nB, [&](int64_t x) { return lk[merge_pad((int)x)]; },
                           [&](int64_t x) { return lk[merge_pad(nA + (int)x)]; });
  int b = kk - a;
  float ok[MERGE_PER_THREAD];
  int32_t ov[MERGE_PER_THREAD];
  float ka = a < nA ? lk[merge_pad(a)] : 0.f, kb = b < nB ? lk[merge_pad(nA + b)] : 0.f;
#pragma unroll
  for (int q = 0; q < MERGE_PER_THREAD; ++q) {
    const bool take_a = (a < nA) && (b >= nB || ka >= kb);
    if (q < n_out) {
      ov[q] = lv[merge_pad(take_a ? a : nA + b)];
      ok[q] = take_a ? ka : kb;
      if (take_a) {
        ++a;
        ka = a < nA ? lk[merge_pad(a)] : 0.f;
      } else {
        ++b;
        kb = b < nB ? lk[merge_pad(nA + b)] : 0.f;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MERGE_PER_THREAD; ++q) {
    if (q < n_out) {
      lk[merge_pad(kk + q)] = ok[q];
      lv[merge_pad(kk + q)] = ov[q];
    }
  }
  __syncthreads();
  const int64_t o_base = (int64_t)f * I + a0 + k0;
  for (int x = t; x < n_tile; x += MERGE_THREADS) {
    ids_out[o_base + x] = lv[merge_pad(x)];
    if (!last) keys_out[o_base + x] = lk[merge_pad(x)];
  }
}

template <int ITEMS>
static void launch_sort_sub(bpr_ctx* c, hipStream_t st, int nf, const float* keysT, double* sig_acc,
                            int32_t* order, float* sigma, int sub, int64_t len, float* keysA,
                            int32_t* idsA, const int32_t* only_flagged = nullptr) {
  hipLaunchKernelGGL((k_sort_sub<ITEMS>), dim3(sub, nf), dim3(1024), 0, st, keysT, c->I, len,
                     order, keysA, idsA, sigma, sig_acc, only_flagged);
}


// ---------------------------------------------------------------------------------------------
// PARTIAL snapshot (r5): what the adaptive sampler reads of a column is its two ends
// — rank Geometric(p) + seen-skips from the top or from the bottom (neg_samplers.py:90-121) — so only
// the ends are sorted exactly and the middle is BUCKETED:
//   order[0 .. Kt)        the Kt largest keys, exact descending order (ties by ascending id)
//   order[Kt .. I - Kb)   the middle in MID_BINS value-linear bins, bins in descending key order,
//                         any order inside a bin; the first entry of a bin carries MID_FLAG
//   order[I - Kb .. I)    the Kb smallest keys, exact (the first of them carries MID_FLAG too)
// Kt / Kb come from two cuts read off a coarse histogram of the column (any counts are legal;
// meta[f] = {Kt, Kb}).  A walk that leaves an exact end keeps counting unseen entries — counting
// does not care about the order inside a bin — and finishes INSIDE one bin by ranking its <= MID_BIN_MAX
// keys on the fly (bpr_device.h adaptive_finish_in_bin).  A column the scheme does not fit (a threshold
// that does not separate, an end that overflows its compaction buffer, a bin with more than
// MID_BIN_MAX keys: many equal keys) reports meta[f] = {-1, -1} and is sorted whole by the kernel
// launched behind this one (k_sort_sub over the flagged columns).
// One 1024-thread workgroup per column, I <= 1024 * ITEMS.
// ---------------------------------------------------------------------------------------------
constexpr int MID_BINS = 4096;
constexpr int MID_BIN_MAX = 64;
constexpr int PART_CI = 2;                    // compacted keys per thread in the sort of the two ends
constexpr int PART_CAP = 1024 * PART_CI / 2;  // ... i.e. at most 1,024 keys per end
constexpr uint32_t MID_FLAG = 0x80000000u;    // == bpr::ORDER_FLAG (bpr_device.h)

template <int ITEMS>
__global__ __launch_bounds__(1024) void k_sort_partial(const float* __restrict__ T, int64_t I,
                                                       int32_t* __restrict__ order,
                                                       float* __restrict__ sigma,
                                                       int32_t* __restrict__ meta, int target) {
  using EndSort = rocprim::block_radix_sort<float, 1024, PART_CI, uint16_t>;
  __shared__ union {
    typename EndSort::storage_type ends;
    double red[2][16];
  } sm;
  __shared__ float s_ck[2 * PART_CAP];     // compacted keys: [0, CAP) the top end, [CAP, 2 CAP) the bottom end
  __shared__ uint16_t s_ci[2 * PART_CAP];  // ... and their item ids
  __shared__ uint32_t s_hist[MID_BINS];    // keys per middle bin, then the bins' first positions
  __shared__ uint32_t s_coarse[1024];      // keys per coarse bin of the whole column
  __shared__ uint32_t s_cum[1024];         // ... and before it, from the top
  __shared__ uint32_t s_scan[1024];
  __shared__ int32_t s_cnt[4];             // the cuts' coarse bins, largest middle bin
  __shared__ int32_t s_mid[1024 * ITEMS];  // the middle, staged: written back in whole lines (a scattered 4-byte
                                           // store is a 64-B write request at the memory side — the very
                                           // resource k_stream, running beside this kernel, is bound by)
  const int f = blockIdx.x;
  const float* row = T + (int64_t)f * I;
  const int t = threadIdx.x;
  const int n = (int)I;
  float keys[ITEMS];
  double s1 = 0.0, s2 = 0.0;
  const float first = row[1];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int l = t * ITEMS + k;  // blocked: a thread's ids ascend, and so do the threads'
    const bool valid = l < n;
    const float v = valid ? row[l] : 0.f;
    keys[k] = v;
    if (valid && l >= 1) {
      const double c = (double)v - (double)first;
      s1 += c;
      s2 += c * c;
    }
  }
  for (int k = t; k < MID_BINS; k += 1024) s_hist[k] = 0u;
  s_coarse[t] = 0u;
  if (t < 4) s_cnt[t] = t == 1 ? 1023 : 0;  // [0] / [1]: the cuts' coarse bins (defaults: nothing in the ends)
  // sigma_f = unbiased std over rows 1..I-1 (neg_samplers.py:132), as k_sort_sub
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if ((t & 63) == 0) {
    sm.red[0][t >> 6] = s1;
    sm.red[1][t >> 6] = s2;
  }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) {
      a += sm.red[0][w];
      b += sm.red[1][w];
    }
    const double nn = (double)(I - 1);
    sigma[f] = (float)sqrt(fmax(b - a * a / nn, 0.0) / (nn - 1.0));
    sm.red[0][0] = a;  // the totals, for everybody (the coarse bins' range)
    sm.red[1][0] = b;
  }
  __syncthreads();
  // ---- a coarse histogram of the whole column: 1,024 value-linear bins over mean +- 5 sigma (out-of-range
  // keys in the end bins).  Everything below is decided by a key's coarse bin and its place inside it — a
  // monotone function of the key — so classes and bins agree with the order whatever the rounding, equal keys
  // stay together, and any distribution works: a skewed column gets unequal ends, a column with a spike
  // (the cold items of a trained model: thousands of keys within +-0.004 of zero) gets as many fine bins
  // there as it has keys there (the fine bins are cut along the coarse CDF, not along the value axis).
  float cmax, cscale;
  {
    const double nn = (double)(I - 1);
    const double a = sm.red[0][0], b = sm.red[1][0];
    const double mean = (double)first + a / nn;
    const double sd = sqrt(fmax(b - a * a / nn, 0.0) / (nn - 1.0));
    cmax = (float)(mean + 5.0 * sd);
    cscale = sd > 0.0 ? (float)(1024.0 / (10.0 * sd)) : 0.f;
  }
  auto coarse = [&](float v) { return min(1023, max(0, (int)((cmax - v) * cscale))); };
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (t * ITEMS + k < n) atomicAdd(&s_coarse[coarse(keys[k])], 1u);
  __syncthreads();
  {
    const int mine = (int)s_coarse[t];  // one coarse bin per thread
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += u;
    }
    if ((t & 63) == 63) s_scan[t >> 6] = (uint32_t)incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += (int)s_scan[w];
    incl += base;                        // keys in coarse bins 0..t (from the top)
    s_cum[t] = (uint32_t)(incl - mine);  // keys above coarse bin t
  }
  __syncthreads();
  // ---- classify by a key's interpolated RANK r = (keys above its coarse bin) + (its place inside the bin) x
  // (keys in the bin): top r < target, bottom r >= n - target, middle between; the middle's fine bin is r
  // scaled to MID_BINS.  r is a monotone function of the key (equal keys: equal r), so classes and bins
  // agree with the order; inside a coarse bin the density is taken as uniform, which is what makes the
  // cuts and the bins equi-DEPTH rather than equi-width.
  const float rt = (float)min(target, n / 4), rb = (float)n - rt;
  const bool separates = cscale > 0.f;
  const float fscale = (float)MID_BINS / fmaxf(rb - rt, 1.f);
  int my_top = 0, my_bot = 0;
  uint32_t packed[ITEMS];  // bit 31: not a middle key (bit 0: top); else bin << 8 | ordinal inside the bin
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int l = t * ITEMS + k;
    packed[k] = 0x80000000u;
    if (l >= n || !separates) continue;
    const float x = (cmax - keys[k]) * cscale;
    const int cb = min(1023, max(0, (int)x));
    const float frac = fminf(fmaxf(x - (float)cb, 0.f), 0.999f);
    const float r = (float)s_cum[cb] + frac * (float)s_coarse[cb];
    if (r < rt) {
      packed[k] = 0x80000001u;
      ++my_top;
    } else if (r >= rb) {
      ++my_bot;
    } else {
      const int bin = min(MID_BINS - 1, max(0, (int)((r - rt) * fscale)));
      const uint32_t ord = atomicAdd(&s_hist[bin], 1u);
      packed[k] = ((uint32_t)bin << 8) | min(ord, 255u);
    }
  }
  // ---- the ends' keys in the compaction buffers: block-wide exclusive scans of the threads' counts
  auto block_excl = [&](int v, int* total) {
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += u;
    }
    __syncthreads();
    if ((t & 63) == 63) s_scan[t >> 6] = (uint32_t)incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = (int)s_scan[w];
      if (w < (t >> 6)) base += c;
      tot += c;
    }
    *total = tot;
    return base + incl - v;
  };
  int Kt = 0, Kb = 0, n_mid = 0;
  const int top_at = block_excl(my_top, &Kt);
  const int bot_at = block_excl(my_bot, &Kb);
  n_mid = n - Kt - Kb;
  // ---- the fine bins' sizes -> first positions (exclusive scan over MID_BINS = 4 per thread)
  int biggest = 0;
  {
    uint32_t c4[MID_BINS / 1024];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < MID_BINS / 1024; ++q) {
      c4[q] = s_hist[t * (MID_BINS / 1024) + q];
      mine += (int)c4[q];
      biggest = max(biggest, (int)c4[q]);
    }
    int tot_mid = 0;
    int at = block_excl(mine, &tot_mid);
#pragma unroll
    for (int q = 0; q < MID_BINS / 1024; ++q) {
      s_hist[t * (MID_BINS / 1024) + q] = (uint32_t)at;
      at += (int)c4[q];
    }
  }
  if (biggest > MID_BIN_MAX) atomicMax(&s_cnt[2], biggest);
  __syncthreads();
  const bool ok = separates && Kt <= PART_CAP && Kb <= PART_CAP && Kt >= 1 && Kb >= 1 && n_mid >= 0 && s_cnt[2] == 0;
  if (!ok) {  // (uniform over the block) the column is sorted whole by the kernel behind this one
    if (t == 0) {
      meta[2 * f] = -1;
      meta[2 * f + 1] = -1;
    }
    return;
  }
  if (t == 0) {
    meta[2 * f] = Kt;
    meta[2 * f + 1] = Kb;
  }
  // ---- scatter (all in LDS): ends to the compaction buffers (ids ascending among equal keys: the sort is
  // stable), middle keys to their place in the staged middle
  // a key between the two ends for the pads: the smallest top key and the largest bottom key bracket it
  // (block minimum / maximum through s_scan)
  float tmin = __builtin_huge_valf(), bmax = -__builtin_huge_valf();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    if (t * ITEMS + k >= n || (packed[k] & 0x80000000u) == 0u) continue;
    if (packed[k] & 1u) tmin = fminf(tmin, keys[k]); else bmax = fmaxf(bmax, keys[k]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    tmin = fminf(tmin, __shfl_xor(tmin, off, 64));
    bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64));
  }
  __syncthreads();
  if ((t & 63) == 0) {
    s_scan[t >> 6] = __float_as_uint(tmin);
    s_scan[16 + (t >> 6)] = __float_as_uint(bmax);
  }
  __syncthreads();
  for (int w = 0; w < 16; ++w) {
    tmin = fminf(tmin, __uint_as_float(s_scan[w]));
    bmax = fmaxf(bmax, __uint_as_float(s_scan[16 + w]));
  }
  const float pad_key = 0.5f * tmin + 0.5f * bmax;
  __syncthreads();
  for (int k = t; k < 2 * PART_CAP; k += 1024) {
    // pads sort between the two ends: a key of the middle's range; should it tie with an end's key the pads
    // still sit on the right side of the tie — top keys precede them in the buffer, bottom keys follow them
    s_ck[k] = pad_key;
    s_ci[k] = 0;
  }
  __syncthreads();
  {
    int ta = top_at, ba = bot_at;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int l = t * ITEMS + k;
      if (l >= n) continue;
      if ((packed[k] & 0x80000000u) == 0u) {
        const int bin = (int)(packed[k] >> 8), ord = (int)(packed[k] & 255u);
        s_mid[(int)s_hist[bin] + ord] = (int32_t)((uint32_t)l | (ord == 0 ? MID_FLAG : 0u));
      } else if (packed[k] & 1u) {
        s_ck[ta] = keys[k];
        s_ci[ta] = (uint16_t)l;
        ++ta;
      } else {
        // the bottom end sits at the END of its half: pads before it, so that pads win ties with it
        const int at = 2 * PART_CAP - Kb + ba;
        s_ck[at] = keys[k];
        s_ci[at] = (uint16_t)l;
        ++ba;
      }
    }
  }
  __syncthreads();
  int32_t* col = order + (int64_t)f * I;
  for (int k = t; k < n_mid; k += 1024) col[Kt + k] = s_mid[k];  // whole lines
  // ---- the two ends in ONE stable descending sort of 2 x PART_CAP (key, id) pairs
  float ek[PART_CI];
  uint16_t ev[PART_CI];
#pragma unroll
  for (int k = 0; k < PART_CI; ++k) {
    ek[k] = s_ck[t * PART_CI + k];
    ev[k] = s_ci[t * PART_CI + k];
  }
  __syncthreads();
  EndSort().sort_desc_to_striped(ek, ev, sm.ends);
#pragma unroll
  for (int k = 0; k < PART_CI; ++k) {
    const int pos = k * 1024 + t;  // rank in the sorted sequence: top end, pads, bottom end
    if (pos < Kt) col[pos] = (int32_t)ev[k];
    else if (pos >= 2 * PART_CAP - Kb) {
      const int b = pos - (2 * PART_CAP - Kb);  // 0 .. Kb-1
      col[n - Kb + b] = (int32_t)((uint32_t)ev[k] | (b == 0 ? MID_FLAG : 0u));
    }
  }
}

// the columns k_sort_partial gave up on (meta[f] < 0), sorted whole: k_sort_sub's single-workgroup form
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_sort_flagged(const float* __restrict__ T, int64_t I,
                                                       int32_t* __restrict__ order,
                                                       int32_t* __restrict__ meta) {
  using Sort = rocprim::block_radix_sort<float, 1024, ITEMS, uint16_t, 1, 1, BPR_SORT_RADIX_BITS>;
  __shared__ typename Sort::storage_type sm;
  const int f = blockIdx.x;
  if (meta[2 * f] >= 0) return;
  const float* row = T + (int64_t)f * I;
  const int t = threadIdx.x;
  float keys[ITEMS];
  uint16_t vals[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t l = (int64_t)t * ITEMS + k;
    keys[k] = l < I ? row[l] : -__builtin_huge_valf();
    vals[k] = (uint16_t)l;
  }
  Sort().sort_desc_to_striped(keys, vals, sm);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t pos = (int64_t)k * 1024 + t;
    if (pos < I) order[(int64_t)f * I + pos] = (int32_t)vals[k];
  }
  __syncthreads();
  if (t == 0) {
    meta[2 * f] = (int32_t)I;  // every position exact
    meta[2 * f + 1] = 0;
  }
}

template <int ITEMS>
static void launch_sort_partial(bpr_ctx* c, hipStream_t st, const float* keysT, int32_t* order, float* sigma,
                                int32_t* meta, int target) {
  hipLaunchKernelGGL((k_sort_partial<ITEMS>), dim3(c->d), dim3(1024), 0, st, keysT, c->I, order, sigma, meta,
                     target);
  hipLaunchKernelGGL((k_sort_flagged<ITEMS>), dim3(c->d), dim3(1024), 0, st, keysT, c->I, order, meta);
}

// ---------------------------------------------------------------------------------------------
// BINNED snapshot sort (r5): the WHOLE column in exact descending order (ties by
// ascending item id, -0 == +0: the order of the stable radix sort above, bit for bit) without a radix sort.
// An interpolated rank — a monotone function of the key read off a 1,024-bin histogram of the column's
// [min, max] and a second 1,024-bin level over its crowded stretch — drops every key into one of BINS equi-DEPTH bins (n / BINS ~ 2.5 keys each);
// a one-pass counting sort stages (orderable key, id) by bin in LDS; then, POSITION by position (a
// wave takes 64 consecutive staged entries: its lanes read the same few words — broadcasts, no bank
// conflicts — and find their bin's bounds from three ballots of first-of-bin flags), every key counts
// the members of its bin that precede it; the ids move to their final places in LDS and leave in
// whole lines.  Work per key: two LDS atomics, two LDS writes, ~bin-size LDS reads — against eight
// (radix 4) or four (radix 8) ranked LDS exchanges of the 32-bit radix sort.  A column the scheme does
// not fit (a bin over BIN_MAX keys: a spike narrower than a coarse bin, thousands of equal keys; no
// spread at all) reports meta[2f] = -1 and is sorted whole by k_sort_flagged behind this kernel.
// One 1024-thread workgroup per column, I <= 1024 * ITEMS <= 32,768 (the id shares 16 bits with the
// first-of-bin flag); LDS = 6 B per key + 4 B per bin.
// ---------------------------------------------------------------------------------------------
constexpr int BIN_MAX = 64;  // (the ballots below look one 64-entry window back and one ahead)
constexpr int BIN_CROWD = 32;
constexpr uint32_t BIN_FIRST = 0x8000u;

__device__ __forceinline__ uint32_t orderable_desc(float v) {  // larger float <=> larger uint; -0 == +0, as a
  uint32_t b = __float_as_uint(v);                             // comparison and rocPRIM's radix digits have it
  if (b == 0x80000000u) b = 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <int ITEMS>
__global__ __launch_bounds__(1024) void k_sort_binned(const float* __restrict__ T, int64_t I,
                                                      int32_t* __restrict__ order,
                                                      float* __restrict__ sigma,
                                                      int32_t* __restrict__ meta) {
  static_assert(1024 * ITEMS <= 32768 && ITEMS >= 4, "ids share 16 bits with the first-of-bin flag");
  constexpr int BINS = ITEMS <= 6 ? 2048 : ITEMS <= 10 ? 4096 : 8192;
  constexpr int BPT = BINS / 1024;
  __shared__ uint32_t s_key[1024 * ITEMS + 4];  // orderable keys, staged by bin (before that: the histograms)
  __shared__ uint16_t s_id[1024 * ITEMS];   // their item ids | BIN_FIRST; then the ids in final order
  __shared__ uint32_t s_hist[BINS + 1];     // keys per bin, then the bins' first positions ([BINS]: the pads' bin)
  __shared__ uint32_t s_scan[16];
  __shared__ double s_red[2][16];
  __shared__ float s_mm[2][16];
  __shared__ int32_t s_big;
  uint32_t* const s_coarse = s_key;         // keys per coarse bin of the column
  uint32_t* const s_cum = s_key + 1024;     // ... and before it, from the top
  uint32_t* const s_fine = s_key + 2048;    // second level: keys per bin of the crowded stretch
  uint32_t* const s_fcum = s_key + 3072;    // ... and before it, inside the stretch
  __shared__ int32_t s_hull[2];             // the crowded stretch: first / last coarse bin over BIN_CROWD keys
  const int f = blockIdx.x;
  const float* row = T + (int64_t)f * I;
  const int t = threadIdx.x;
  const int n = (int)I;
  // (array elements are assigned outside any branch: a conditional store into a register array makes the
  // compiler carry the whole array through the branch as one vector value — 5,600 spilled VGPRs at ITEMS = 20)
  float keys[ITEMS];
  double s1 = 0.0, s2 = 0.0;
  float vmin = __builtin_huge_valf(), vmax = -__builtin_huge_valf();
  const float first = row[1];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int l = k * 1024 + t;  // striped: coalesced loads (the order below does not lean on the arrangement)
    const bool valid = l < n;
    const float v = valid ? row[l] : 0.f;
    keys[k] = v;
    const double c = (double)v - (double)first;
    s1 += (valid && l >= 1) ? c : 0.0;
    s2 += (valid && l >= 1) ? c * c : 0.0;
    vmin = valid ? fminf(vmin, v) : vmin;
    vmax = valid ? fmaxf(vmax, v) : vmax;
  }
  for (int k = t; k <= BINS; k += 1024) s_hist[k] = 0u;
  s_coarse[t] = 0u;
  s_fine[t] = 0u;
  if (t == 0) {
    s_big = 0;
    s_hull[0] = 1024;
    s_hull[1] = -1;
  }
  // sigma_f = unbiased std over rows 1..I-1 (neg_samplers.py:132), as k_sort_sub
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
    vmin = fminf(vmin, __shfl_xor(vmin, off, 64));
    vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
  }
  if ((t & 63) == 0) {
    s_red[0][t >> 6] = s1;
    s_red[1][t >> 6] = s2;
    s_mm[0][t >> 6] = vmin;
    s_mm[1][t >> 6] = vmax;
  }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) {
      a += s_red[0][w];
      b += s_red[1][w];
    }
    const double nn = (double)(I - 1);
    sigma[f] = (float)sqrt(fmax(b - a * a / nn, 0.0) / (nn - 1.0));
  }
  // ---- coarse histogram: 1,024 value-linear bins over the column's [min, max] — not mean +- 5 sigma: the
  // columns of a trained table have tails out to 11 sigma, and everything past a clipped range lands in ONE bin
  for (int w = 0; w < 16; ++w) {
    vmin = fminf(vmin, s_mm[0][w]);
    vmax = fmaxf(vmax, s_mm[1][w]);
  }
  const float cmax = vmax;
  const float cscale = vmax > vmin ? 1024.0f / (vmax - vmin) : 0.f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (k * 1024 + t < n) atomicAdd(&s_coarse[min(1023, max(0, (int)((cmax - keys[k]) * cscale)))], 1u);
  __syncthreads();
  auto block_excl = [&](int v) {  // exclusive prefix of one value per thread over the block
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += u;
    }
    __syncthreads();
    if ((t & 63) == 63) s_scan[t >> 6] = (uint32_t)incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += (int)s_scan[w];
    return base + incl - v;
  };
  s_cum[t] = (uint32_t)block_excl((int)s_coarse[t]);  // keys above coarse bin t
  // ---- second level.  A fine bin never holds more than the coarse bins it touches, so only CROWDED coarse bins
  // (over BIN_CROWD keys) can overflow one — and they do when the column is a spike plus a few far outliers: the
  // take-off of training, when popular items have grown a hundred times past the untouched rest and mean +- 5
  // sigma puts ten thousand keys into a handful of coarse bins.  The stretch from the first to the last crowded
  // coarse bin gets 1,024 value-linear bins of its own; a key inside it takes its rank from those.
  {
    const unsigned long long crowded = __ballot(s_coarse[t] > (uint32_t)BIN_CROWD);
    if ((t & 63) == 0 && crowded != 0ull) {
      atomicMin(&s_hull[0], (t & ~63) + __ffsll(crowded) - 1);
      atomicMax(&s_hull[1], (t & ~63) + 63 - __clzll(crowded));
    }
  }
  __syncthreads();
  const int h_lo = s_hull[0], h_hi = s_hull[1];  // (h_lo > h_hi: no crowded bin)
  const float ftop = cmax - (float)h_lo / cscale;  // the stretch's upper edge (any value near it does: membership
  const float fscale = cscale * (1024.0f / (float)max(h_hi - h_lo + 1, 1));  // is decided by the COARSE bin)
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int cb = min(1023, max(0, (int)((cmax - keys[k]) * cscale)));
    if (k * 1024 + t < n && cb >= h_lo && cb <= h_hi)
      atomicAdd(&s_fine[min(1023, max(0, (int)((ftop - keys[k]) * fscale)))], 1u);
  }
  __syncthreads();
  s_fcum[t] = (uint32_t)block_excl((int)s_fine[t]);
  __syncthreads();
  const float hull_above = h_lo <= h_hi ? (float)s_cum[h_lo] : 0.f;  // keys above the stretch
  // ---- a key's bin from its interpolated rank r = (keys above its coarse bin) + (its place inside the bin) x
  // (keys in the bin): monotone in the key, equal keys equal r — bins agree with the order whatever the rounding
  const float bscale = (float)BINS / (float)n;
  uint32_t packed[ITEMS];  // bin << 8 | ordinal inside the bin
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const float x = (cmax - keys[k]) * cscale;
    const int cb = min(1023, max(0, (int)x));
    const bool inside = cb >= h_lo && cb <= h_hi;
    const float x2 = (ftop - keys[k]) * fscale;
    const int fb = min(1023, max(0, (int)x2));
    const float frac = fminf(fmaxf(inside ? x2 - (float)fb : x - (float)cb, 0.f), 0.999f);
    const float r = inside ? hull_above + ((float)s_fcum[fb] + frac * (float)s_fine[fb])
                           : (float)s_cum[cb] + frac * (float)s_coarse[cb];
    const int bin = k * 1024 + t < n ? min(BINS - 1, max(0, (int)(r * bscale))) : BINS;
    const uint32_t ord = atomicAdd(&s_hist[bin], 1u);
    packed[k] = ((uint32_t)bin << 8) | min(ord, 255u);
  }
  __syncthreads();
  // ---- the bins' sizes -> first positions
  {
    uint32_t c4[BPT];
    int mine = 0, biggest = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      c4[q] = s_hist[t * BPT + q];
      mine += (int)c4[q];
      biggest = max(biggest, (int)c4[q]);
    }
    if (biggest > BIN_MAX) atomicMax(&s_big, biggest);
    int at = block_excl(mine);
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      s_hist[t * BPT + q] = (uint32_t)at;
      at += (int)c4[q];
    }
  }
  __syncthreads();  // (the coarse histogram is dead from here: the staged keys take its place)
  if (s_big != 0 || cscale <= 0.f) {  // (uniform over the block) sorted whole by k_sort_flagged
    if (t == 0) {
      meta[2 * f] = -1;
      meta[2 * f + 1] = -1;
    }
    return;
  }
  if (t == 0) {
    meta[2 * f] = n;
    meta[2 * f + 1] = 0;
  }
  // ---- counting sort into LDS; the first entry of a bin carries BIN_FIRST
  if (t < 4) s_key[n + t] = 0u;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int l = k * 1024 + t;
    const uint32_t ord = packed[k] & 255u;
    const int at = (int)s_hist[packed[k] >> 8] + (int)ord;
    if (l < n) {
      s_key[at] = orderable_desc(keys[k]);
      s_id[at] = (uint16_t)((uint32_t)l | (ord == 0u ? BIN_FIRST : 0u));
    }
  }
  __syncthreads();
  // ---- a key's place inside its bin, position by position: the members that precede it (larger key, or equal
  // key and lower id).  A wave walks ITEMS consecutive 64-entry windows; the bin's bounds come from the
  // first-of-bin flags of its window, the one before and the one after (a bin holds <= BIN_MAX = 64 entries;
  // position n counts as flagged), each window's flags read once and handed on.
  const int lane = t & 63;
  uint32_t out[ITEMS];  // final position << 16 | id
  {
    int base = (t >> 6) * ITEMS * 64;
    uint32_t me = base + lane < n ? (uint32_t)s_id[base + lane] : 0u;
    const uint32_t before = base >= 64 && base + lane - 64 < n ? (uint32_t)s_id[base + lane - 64] : 0u;
    unsigned long long bc = __ballot((me & BIN_FIRST) != 0u || base + lane == n);
    unsigned long long bp = __ballot((before & BIN_FIRST) != 0u);
    const unsigned long long upto = (2ull << lane) - 1ull;  // bits 0 .. lane
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int p = base + lane;
      const bool valid = p < n;
      const uint32_t next = p + 64 < n ? (uint32_t)s_id[p + 64] : 0u;
      const unsigned long long bn = __ballot((next & BIN_FIRST) != 0u || p + 64 == n);
      const unsigned long long at_or_before = bc & upto, after = bc & ~upto;
      int lo = at_or_before ? base + 63 - __clzll(at_or_before) : base - 1 - __clzll(bp);
      int hi = after ? base + __ffsll(after) - 1 : bn ? base + 63 + __ffsll(bn) : n;
      if (!valid) lo = hi = 0;
      const uint32_t u = valid ? s_key[p] : 0u;
      const int id = (int)(me & (BIN_FIRST - 1u));
      // four members in flight, no bounds tests: an entry past the bin's end belongs to a LATER bin — its key is
      // strictly smaller (equal keys share a bin), so it counts neither as larger nor as equal; the four words
      // past position n hold 0, the smallest orderable key
      int rank = 0, equal = 0;
#pragma unroll 1  // (bins hold ~2.5 keys: one or two trips — unrolled further, the remainder tests cost more than the loop)
      for (int j = lo; j < hi; j += 4) {
        const uint32_t o0 = s_key[j], o1 = s_key[j + 1], o2 = s_key[j + 2], o3 = s_key[j + 3];
        rank += (o0 > u ? 1 : 0) + (o1 > u ? 1 : 0) + (o2 > u ? 1 : 0) + (o3 > u ? 1 : 0);
        equal += (o0 == u ? 1 : 0) + (o1 == u ? 1 : 0) + (o2 == u ? 1 : 0) + (o3 == u ? 1 : 0);
      }
      if (equal > 1) {  // equal keys (rare; `equal` counts the key itself once): the lower id goes first
#pragma unroll 1
        for (int j = lo; j < hi; ++j)
          rank += s_key[j] == u && (int)((uint32_t)s_id[j] & (BIN_FIRST - 1u)) < id ? 1 : 0;
      }
      out[k] = ((uint32_t)(lo + rank) << 16) | (uint32_t)id;
      bp = bc;
      bc = bn;
      me = next;
      base += 64;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (((t >> 6) * ITEMS + k) * 64 + lane < n) s_id[out[k] >> 16] = (uint16_t)(out[k] & 0xffffu);
  __syncthreads();
  int32_t* col = order + (int64_t)f * I;
  for (int k = t; k < n; k += 1024) col[k] = (int32_t)s_id[k];  // whole lines
}

// ---------------------------------------------------------------------------------------------
// The binned sort with G workgroups per column: workgroup g orders the g-th stretch of RANKS.  For columns
// that do not fit one workgroup's LDS (20,480 < I <= 65,535: MSD's 41,141).  Every workgroup reads the whole column
// (L2-resident) and builds the same two-level histogram; nothing is kept in registers between passes — a
// count pass and a fill pass over the keys replace the remembered ordinals — and only the keys whose bin falls
// into its stretch are staged: about I / G, at most 1,024 x SITEMS.  The first-of-bin flags live in a bit
// array (past 32,767 items the ids need all 16 bits): a 64-entry window's flags are one 64-bit word.  Its part
// of the order starts at (keys above its stretch).  A workgroup that cannot (a bin over BIN_MAX keys, a
// stretch over its capacity, no spread) flags the column — meta[2f] = -1, cleared to 0 before the launch — and
// the radix path behind it redoes exactly the flagged columns.
// ---------------------------------------------------------------------------------------------
constexpr int SPLIT_BINS = 4096;  // bins per workgroup

template <int SITEMS>
__global__ __launch_bounds__(1024) void k_sort_binned_split(const float* __restrict__ T, int64_t I,
                                                            int32_t* __restrict__ order,
                                                            float* __restrict__ sigma,
                                                            int32_t* __restrict__ meta) {
  constexpr int CAP = 1024 * SITEMS;
  constexpr int BPT = SPLIT_BINS / 1024;
  __shared__ uint32_t s_key[CAP + 4];
  __shared__ uint16_t s_id[CAP];
  __shared__ uint32_t s_flag[CAP / 32 + 4];  // first-of-bin bits
  __shared__ uint32_t s_hist[SPLIT_BINS];
  __shared__ uint32_t s_coarse[1024], s_cum[1024], s_fine[1024], s_fcum[1024];
  __shared__ uint32_t s_scan[16];
  __shared__ double s_red[2][16];
  __shared__ float s_mm[2][16];
  __shared__ int32_t s_big;
  __shared__ int32_t s_hull[2];
  __shared__ int32_t s_tot[2];
  const int g = blockIdx.x, G = gridDim.x;
  const int f = blockIdx.y;
  const float* row = T + (int64_t)f * I;
  const int t = threadIdx.x;
  const int n = (int)I;
  const int items = (n + 1023) / 1024;
  // a pass over the column: eight loads in flight, then the work on them (a load per trip would leave every
  // trip waiting for the L2: 5 passes x 41 trips x ~0.6 us on MSD)
  auto for_keys = [&](auto&& fn) {
    for (int k0 = 0; k0 < items; k0 += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int l = (k0 + q) * 1024 + t;
        v[q] = l < n ? row[l] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int l = (k0 + q) * 1024 + t;
        if (l < n) fn(l, v[q]);
      }
    }
  };
  // ---- pass 1: sigma, min, max (the sums in k_sort_binned's order)
  double s1 = 0.0, s2 = 0.0;
  float vmin = __builtin_huge_valf(), vmax = -__builtin_huge_valf();
  const float first = row[1];
  for_keys([&](int l, float v) {
    if (l >= 1) {
      const double c = (double)v - (double)first;
      s1 += c;
      s2 += c * c;
    }
    vmin = fminf(vmin, v);
    vmax = fmaxf(vmax, v);
  });
  for (int k = t; k < SPLIT_BINS; k += 1024) s_hist[k] = 0u;
  for (int k = t; k < CAP / 32 + 4; k += 1024) s_flag[k] = 0u;
  s_coarse[t] = 0u;
  s_fine[t] = 0u;
  if (t == 0) {
    s_big = 0;
    s_hull[0] = 1024;
    s_hull[1] = -1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
    vmin = fminf(vmin, __shfl_xor(vmin, off, 64));
    vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
  }
  if ((t & 63) == 0) {
    s_red[0][t >> 6] = s1;
    s_red[1][t >> 6] = s2;
    s_mm[0][t >> 6] = vmin;
    s_mm[1][t >> 6] = vmax;
  }
  __syncthreads();
  if (t == 0 && g == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) {
      a += s_red[0][w];
      b += s_red[1][w];
    }
    const double nn = (double)(I - 1);
    sigma[f] = (float)sqrt(fmax(b - a * a / nn, 0.0) / (nn - 1.0));
  }
  for (int w = 0; w < 16; ++w) {
    vmin = fminf(vmin, s_mm[0][w]);
    vmax = fmaxf(vmax, s_mm[1][w]);
  }
  const float cmax = vmax;
  const float cscale = vmax > vmin ? 1024.0f / (vmax - vmin) : 0.f;
  // ---- pass 2: the coarse histogram
  for_keys([&](int, float v) { atomicAdd(&s_coarse[min(1023, max(0, (int)((cmax - v) * cscale)))], 1u); });
  __syncthreads();
  auto block_excl = [&](int v) {
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += u;
    }
    __syncthreads();
    if ((t & 63) == 63) s_scan[t >> 6] = (uint32_t)incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += (int)s_scan[w];
    return base + incl - v;
  };
  s_cum[t] = (uint32_t)block_excl((int)s_coarse[t]);
  {
    const unsigned long long crowded = __ballot(s_coarse[t] > (uint32_t)BIN_CROWD);
    if ((t & 63) == 0 && crowded != 0ull) {
      atomicMin(&s_hull[0], (t & ~63) + __ffsll(crowded) - 1);
      atomicMax(&s_hull[1], (t & ~63) + 63 - __clzll(crowded));
    }
  }
  __syncthreads();
  const int h_lo = s_hull[0], h_hi = s_hull[1];
  const float ftop = cmax - (float)h_lo / cscale;
  const float fscale = cscale * (1024.0f / (float)max(h_hi - h_lo + 1, 1));
  // ---- pass 3: the second level over the crowded stretch
  for_keys([&](int, float v) {
    const int cb = min(1023, max(0, (int)((cmax - v) * cscale)));
    if (cb >= h_lo && cb <= h_hi) atomicAdd(&s_fine[min(1023, max(0, (int)((ftop - v) * fscale)))], 1u);
  });
  __syncthreads();
  s_fcum[t] = (uint32_t)block_excl((int)s_fine[t]);
  __syncthreads();
  const float hull_above = h_lo <= h_hi ? (float)s_cum[h_lo] : 0.f;
  const int all_bins = G * SPLIT_BINS;
  const float bscale = (float)all_bins / (float)n;
  const int my_lo = g * SPLIT_BINS;
  auto bin_of = [&](float v) {  // the key's bin among all G x SPLIT_BINS: k_sort_binned's arithmetic
    const float x = (cmax - v) * cscale;
    const int cb = min(1023, max(0, (int)x));
    const bool inside = cb >= h_lo && cb <= h_hi;
    const float x2 = (ftop - v) * fscale;
    const int fb = min(1023, max(0, (int)x2));
    const float frac = fminf(fmaxf(inside ? x2 - (float)fb : x - (float)cb, 0.f), 0.999f);
    const float r = inside ? hull_above + ((float)s_fcum[fb] + frac * (float)s_fine[fb])
                           : (float)s_cum[cb] + frac * (float)s_coarse[cb];
    return min(all_bins - 1, max(0, (int)(r * bscale)));
  };
  // ---- pass 4: this stretch's bins counted, and the keys above the stretch
  int above = 0;
  for_keys([&](int, float v) {
    const int b = bin_of(v) - my_lo;
    above += b < 0 ? 1 : 0;
    if (b >= 0 && b < SPLIT_BINS) atomicAdd(&s_hist[b], 1u);
  });
  __syncthreads();
  {
    const int before = block_excl(above);
    if (t == 1023) s_tot[0] = before + above;
    uint32_t c4[BPT];
    int mine = 0, biggest = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      c4[q] = s_hist[t * BPT + q];
      mine += (int)c4[q];
      biggest = max(biggest, (int)c4[q]);
    }
    if (biggest > BIN_MAX) atomicMax(&s_big, biggest);
    int at = block_excl(mine);
    if (t == 1023) s_tot[1] = at + mine;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      s_hist[t * BPT + q] = (uint32_t)at;  // the bin's first position: the fill pass counts it up
      if (c4[q] != 0u && at < CAP) atomicOr(&s_flag[at >> 5], 1u << (at & 31));
      at += (int)c4[q];
    }
  }
  __syncthreads();
  const int rank0 = s_tot[0], n_mine = s_tot[1];
  if (s_big != 0 || cscale <= 0.f || n_mine > CAP) {  // (uniform over the block)
    if (t == 0) meta[2 * f] = -1;
    return;
  }
  if (t == 0) atomicOr(&s_flag[n_mine >> 5], 1u << (n_mine & 31));  // the end counts as a bin's first entry
  if (t < 4) s_key[n_mine + t] = 0u;                                 // ... and past it the smallest orderable key
  // ---- pass 5: fill
  for_keys([&](int l, float v) {
    const int b = bin_of(v) - my_lo;
    if (b >= 0 && b < SPLIT_BINS) {
      const int at = (int)atomicAdd(&s_hist[b], 1u);
      s_key[at] = orderable_desc(v);
      s_id[at] = (uint16_t)l;
    }
  });
  __syncthreads();
  // ---- ranking inside the bins, position by position (k_sort_binned's; a window's flags are one 64-bit word)
  const int lane = t & 63;
  uint32_t out[SITEMS];  // final position << 16 | id ... two words past 32,767 items: position and id apart
  uint32_t oid[SITEMS];
  {
    int base = (t >> 6) * SITEMS * 64;
    const unsigned long long upto = (2ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < SITEMS; ++k) {
      const int p = base + lane;
      const bool valid = p < n_mine;
      const int w = base >> 5;  // (base is a multiple of 64)
      const unsigned long long bc = (unsigned long long)s_flag[w] | ((unsigned long long)s_flag[w + 1] << 32);
      const unsigned long long bp = base >= 64 ? (unsigned long long)s_flag[w - 2] | ((unsigned long long)s_flag[w - 1] << 32) : 0ull;
      const unsigned long long bn = base + 64 <= CAP ? (unsigned long long)s_flag[w + 2] | ((unsigned long long)s_flag[w + 3] << 32) : 0ull;
      const unsigned long long at_or_before = bc & upto, after = bc & ~upto;
      int lo = at_or_before ? base + 63 - __clzll(at_or_before) : base - 1 - __clzll(bp);
      int hi = after ? base + __ffsll(after) - 1 : bn ? base + 63 + __ffsll(bn) : n_mine;
      if (!valid) lo = hi = 0;
      const uint32_t u = valid ? s_key[p] : 0u;
      const int id = valid ? (int)s_id[p] : 0;
      int rank = 0, equal = 0;
#pragma unroll 1
      for (int j = lo; j < hi; j += 4) {
        const uint32_t o0 = s_key[j], o1 = s_key[j + 1], o2 = s_key[j + 2], o3 = s_key[j + 3];
        rank += (o0 > u ? 1 : 0) + (o1 > u ? 1 : 0) + (o2 > u ? 1 : 0) + (o3 > u ? 1 : 0);
        equal += (o0 == u ? 1 : 0) + (o1 == u ? 1 : 0) + (o2 == u ? 1 : 0) + (o3 == u ? 1 : 0);
      }
      if (equal > 1) {
#pragma unroll 1
        for (int j = lo; j < hi; ++j) rank += s_key[j] == u && (int)s_id[j] < id ? 1 : 0;
      }
      out[k] = (uint32_t)(lo + rank);
      oid[k] = (uint32_t)id;
      base += 64;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SITEMS; ++k)
    if (((t >> 6) * SITEMS + k) * 64 + lane < n_mine) s_id[out[k]] = (uint16_t)oid[k];
  __syncthreads();
  int32_t* col = order + (int64_t)f * I + rank0;
  for (int k = t; k < n_mine; k += 1024) col[k] = (int32_t)s_id[k];
}

template <int SITEMS>
static int launch_sort_binned_split(bpr_ctx* c, hipStream_t st, int G, int nf, const float* keysT, int32_t* order,
                                    float* sigma, int32_t* meta) {
  BPR_HIP_CHECK(hipMemsetAsync(meta, 0, sizeof(int32_t) * 2 * nf, st));  // a workgroup that gives up writes -1
  hipLaunchKernelGGL((k_sort_binned_split<SITEMS>), dim3(G, nf), dim3(1024), 0, st, keysT, c->I, order, sigma, meta);
  return BPR_OK;
}

template <int ITEMS>
static void launch_sort_binned(bpr_ctx* c, hipStream_t st, int nf, const float* keysT, int32_t* order, float* sigma,
                               int32_t* meta) {
  hipLaunchKernelGGL((k_sort_binned<ITEMS>), dim3(nf), dim3(1024), 0, st, keysT, c->I, order, sigma, meta);
  hipLaunchKernelGGL((k_sort_flagged<ITEMS>), dim3(nf), dim3(1024), 0, st, keysT, c->I, order, meta);
}

// composite sort key: (factor << 32) | ~orderable(value)  → ascending sort = per-factor descending
__global__ void k_compose_keys(const float* __restrict__ T, uint64_t* __restrict__ keys, int64_t n,
                               int64_t I) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    uint32_t b = __float_as_uint(T[k]);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    keys[k] = ((uint64_t)(k / I) << 32) | (uint64_t)(~b);
  }
}

__global__ void k_iota(int32_t* ids, int32_t* offs, int64_t I, int d) {
  const int64_t n = (int64_t)d * I;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x)
    ids[k] = (int32_t)(k % I);
  if (blockIdx.x == 0)
    for (int f = threadIdx.x; f <= d; f += blockDim.x) offs[f] = (int32_t)((int64_t)f * I);
}

// ---------------------------------------------------------------------------------------------
// Epoch planner: DataLoader(shuffle=True) of the reference (example.py:307-321, exp.py:109-118)
// re-stated for the STREAM kernel.  A keyed Feistel network gives a pseudo-random permutation
// pi of [0, n) that every thread can evaluate on its own; triple t goes to chunk pi(t) / chunk,
// and one radix sort by (chunk, user) makes every chunk contiguous and grouped by user.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint64_t feistel_perm(uint64_t x, uint64_t n, int half_bits,
                                                 uint64_t seed) {
  const uint32_t mask = (half_bits >= 32) ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
  do {  // cycle-walk: the network permutes [0, 4^half_bits) ⊇ [0, n)
    uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
      const uint32_t k = (uint32_t)(seed >> (16 * (round & 1))) + 0x9E3779B9u * (uint32_t)(round + 1) +
                         (uint32_t)(seed >> 32);
      const uint32_t f = mix32(r ^ k) & mask;
      const uint32_t nl = r;
      r = l ^ f;
      l = nl;
    }
    x = ((uint64_t)l << half_bits) | r;
  } while (x >= n);
  return x;
}

// K: uint32_t when (chunk, user) fits 32 bits — the usual case (ML-20M: 6 + 18 bits): the radix
// sort is bound by the bytes it moves, and 8-byte (key, value) pairs instead of 12 make it a third
// faster (0.49 -> 0.3 ms per 9.55 M-triple epoch) — else uint64_t
template <typename K>
__global__ void k_plan_keys(const int32_t* __restrict__ users, int64_t n, int64_t chunk,
                            int half_bits, int ubits, uint64_t seed, K* __restrict__ keys) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t c = feistel_perm((uint64_t)t, (uint64_t)n, half_bits, seed) / (uint64_t)chunk;
    keys[t] = (K)((c << ubits) | (uint64_t)(uint32_t)users[t]);
  }
}

template <typename K>
__global__ void k_plan_users(const K* __restrict__ keys, int64_t n, int ubits,
                             int32_t* __restrict__ users_out) {
  const uint64_t mask = (1ull << ubits) - 1ull;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x)
    users_out[t] = (int32_t)((uint64_t)keys[t] & mask);
}

// the inverse permutation (same network run backwards, same cycle-walk): pi^-1(pi(t)) = t
__device__ __forceinline__ uint64_t feistel_inv(uint64_t y, uint64_t n, int half_bits, uint64_t seed) {
  const uint32_t mask = (half_bits >= 32) ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
  do {
    uint32_t l = (uint32_t)(y >> half_bits) & mask, r = (uint32_t)y & mask;
#pragma unroll
    for (int round = 3; round >= 0; --round) {
      const uint32_t k = (uint32_t)(seed >> (16 * (round & 1))) + 0x9E3779B9u * (uint32_t)(round + 1) +
                         (uint32_t)(seed >> 32);
      const uint32_t pr = l;                      // the forward round's input r
      const uint32_t pl = r ^ (mix32(pr ^ k) & mask);
      l = pl;
      r = pr;
    }
    y = ((uint64_t)l << half_bits) | r;
  } while (y >= n);
  return y;
}

// bpr_plan_chunk: the members of ONE chunk of the epoch plan, found through the inverse permutation
// (chunk c = pi^-1 of [c * chunk, (c + 1) * chunk)) instead of by sorting the whole epoch
__global__ void k_plan_chunk(const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
                             int64_t n, int64_t j0, int64_t m, int half_bits, uint64_t seed,
                             uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m;
       k += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t t = feistel_inv((uint64_t)(j0 + k), (uint64_t)n, half_bits, seed);
    keys[k] = (uint32_t)users[t];
    vals[k] = pos[t];
  }
}

// ---- grouping one chunk by user in three small kernels (the chunk is planned on the CU-masked side
// stream, where every kernel launch costs ~12 us: a device-wide radix sort of 199 k keys is NINE of
// them).  Users fall into nb <= 2048 buckets of 2^shift consecutive ids: (1) members + bucket
// histogram, (2) scatter into the buckets' ranges, (3) one workgroup per bucket orders its members by
// user with a counting sort over the bucket's 2^shift ids in LDS.  Output: users ascending, the
// order of one user's triples as the atomics fell.
constexpr int PC_MAX_BUCKETS = 2048, PC_MAX_LOCAL = 1024;

__global__ __launch_bounds__(256) void k_pc_members(const int32_t* __restrict__ users,
                                                    const int32_t* __restrict__ pos, int64_t n, int64_t j0,
                                                    int m, int half_bits, uint64_t seed, int shift,
                                                    uint32_t* __restrict__ mu, int32_t* __restrict__ mp,
                                                    uint32_t* __restrict__ cnt) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
    const uint64_t t = feistel_inv((uint64_t)(j0 + k), (uint64_t)n, half_bits, seed);
    const uint32_t u = (uint32_t)users[t];
    mu[k] = u;
    mp[k] = pos[t];
    atomicAdd(&cnt[u >> shift], 1u);
  }
}

__global__ __launch_bounds__(256) void k_pc_scatter(const uint32_t* __restrict__ mu,
                                                    const int32_t* __restrict__ mp, int m, int shift, int nb,
                                                    const uint32_t* __restrict__ cnt, uint32_t* __restrict__ cur,
                                                    uint32_t* __restrict__ base_out, uint32_t* __restrict__ bu,
                                                    int32_t* __restrict__ bp) {
  __shared__ uint32_t base[PC_MAX_BUCKETS];
  __shared__ uint32_t part[256];
  // exclusive scan of the bucket counts, redundantly in every workgroup (nb <= 2048: 8 per thread)
  const int per = (nb + 255) / 256;
  uint32_t acc = 0;
  for (int q = 0; q < per; ++q) {
    const int b = threadIdx.x * per + q;
    acc += b < nb ? cnt[b] : 0u;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int k = 0; k < 256; ++k) {
      const uint32_t v = part[k];
      part[k] = run;
      run += v;
    }
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (int q = 0; q < per; ++q) {
    const int b = threadIdx.x * per + q;
    if (b < nb) {
      base[b] = run;
      if (blockIdx.x == 0) base_out[b] = run;
      run += cnt[b];
    }
  }
  __syncthreads();
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
    const uint32_t u = mu[k];
    const uint32_t b = u >> shift;
    const uint32_t slot = base[b] + atomicAdd(&cur[b], 1u);
    bu[slot] = u;
    bp[slot] = mp[k];
  }
}

__global__ __launch_bounds__(128) void k_pc_group(const uint32_t* __restrict__ bu, const int32_t* __restrict__ bp,
                                                  int shift, uint32_t* __restrict__ cnt, uint32_t* __restrict__ cur,
                                                  const uint32_t* __restrict__ base, int32_t* __restrict__ users_out,
                                                  int32_t* __restrict__ pos_out) {
  __shared__ uint32_t c[PC_MAX_LOCAL], o[PC_MAX_LOCAL];
  const int b = blockIdx.x;
  const uint32_t lo = base[b], sz = cnt[b];
  const int L = 1 << shift;
  const uint32_t mask = (uint32_t)L - 1u;
  for (int k = threadIdx.x; k < L; k += blockDim.x) c[k] = 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < sz; k += blockDim.x) atomicAdd(&c[bu[lo + k] & mask], 1u);
  __syncthreads();
  // exclusive scan over the bucket's ids (<= 1024): a segment per thread, the 128 segment sums by one
  __shared__ uint32_t seg[128];
  const int per = (L + 127) / 128;
  {
    uint32_t acc = 0;
    for (int q = 0; q < per; ++q) {
      const int k = threadIdx.x * per + q;
      acc += k < L ? c[k] : 0u;
    }
    seg[threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int k = 0; k < 128; ++k) {
      const uint32_t v = seg[k];
      seg[k] = run;
      run += v;
    }
  }
  __syncthreads();
  {
    uint32_t run = seg[threadIdx.x];
    for (int q = 0; q < per; ++q) {
      const int k = threadIdx.x * per + q;
      if (k < L) {
        o[k] = run;
        run += c[k];
      }
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < sz; k += blockDim.x) {
    const uint32_t u = bu[lo + k];
    const uint32_t slot = lo + atomicAdd(&o[u & mask], 1u);
    users_out[slot] = (int32_t)u;
    pos_out[slot] = bp[lo + k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // the counters of this bucket are zero again for the next chunk
    cnt[b] = 0u;
    cur[b] = 0u;
  }
}

static int bits_for(uint64_t v) {  // bits needed to represent values 0..v
  int b = 1;
  while ((v >> b) != 0) ++b;
  return b;
}

int plan_epoch_impl(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n,
                    int64_t chunk, uint64_t seed, int32_t* users_out, int32_t* pos_out) {
  if (n == 0) return BPR_OK;
  if (n >= ((int64_t)1 << 31)) {
    set_error("bpr_plan_epoch: n must be < 2^31");
    return BPR_ERR_UNSUPPORTED;
  }
  if (c->hot_key_ptr != pos_in || c->hot_key_n != n) {  // new training set: measure popularity
    if (int rc = hot_build_impl(c, pos_in, n)) return rc;
  }
  const int ubits = bits_for((uint64_t)(c->U - 1));
  const int64_t n_chunks = (n + chunk - 1) / chunk;
  const int cbits = bits_for((uint64_t)(n_chunks - 1));
  int half_bits = (bits_for((uint64_t)(n - 1)) + 1) / 2;
  if (half_bits < 1) half_bits = 1;
  if (c->plan_cap < n) {
    hipFree(c->plan_keys); hipFree(c->plan_keys_sorted); hipFree(c->plan_tmp);
    c->plan_keys = c->plan_keys_sorted = nullptr;
    c->plan_tmp = nullptr;
    c->plan_cap = 0;
    BPR_HIP_CHECK(hipMalloc(&c->plan_keys, sizeof(uint64_t) * n));
    BPR_HIP_CHECK(hipMalloc(&c->plan_keys_sorted, sizeof(uint64_t) * n));
    size_t bytes = 0;
    BPR_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, c->plan_keys,
                                                     c->plan_keys_sorted, pos_in, pos_out, (int)n,
                                                     0, 64, c->stream));
    size_t bytes32 = 0;
    BPR_HIP_CHECK(rocprim::radix_sort_pairs(
        nullptr, bytes32, reinterpret_cast<uint32_t*>(c->plan_keys),
        reinterpret_cast<uint32_t*>(c->plan_keys_sorted), pos_in, pos_out, (int)n, 0, 32, c->stream));
    bytes = std::max(bytes, bytes32);
    BPR_HIP_CHECK(hipMalloc(&c->plan_tmp, bytes > 0 ? bytes : 16));
    c->plan_tmp_bytes = bytes;
    c->plan_cap = n;
  }
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  size_t bytes = c->plan_tmp_bytes;  // sized for 64-bit keys: enough for 32-bit ones
  if (ubits + cbits <= 32) {
    uint32_t* k32 = reinterpret_cast<uint32_t*>(c->plan_keys);
    uint32_t* k32s = reinterpret_cast<uint32_t*>(c->plan_keys_sorted);
    hipLaunchKernelGGL(k_plan_keys<uint32_t>, dim3(grid), dim3(256), 0, c->stream, users_in, n,
                       chunk, half_bits, ubits, seed, k32);
    // (input promised sorted by user — bpr_set_tuning "plan_input_sorted": a STABLE sort on the chunk bits alone
    // leaves every chunk grouped by user, the same output in one radix pass instead of three)
    BPR_HIP_CHECK(rocprim::radix_sort_pairs(c->plan_tmp, bytes, k32, k32s, pos_in, pos_out,
                                                     (int)n, c->tune_plan_sorted ? ubits : 0, ubits + cbits, c->stream));
    hipLaunchKernelGGL(k_plan_users<uint32_t>, dim3(grid), dim3(256), 0, c->stream, k32s, n, ubits,
                       users_out);
  } else {
    hipLaunchKernelGGL(k_plan_keys<uint64_t>, dim3(grid), dim3(256), 0, c->stream, users_in, n,
                       chunk, half_bits, ubits, seed, c->plan_keys);
    BPR_HIP_CHECK(rocprim::radix_sort_pairs(c->plan_tmp, bytes, c->plan_keys,
                                                     c->plan_keys_sorted, pos_in, pos_out, (int)n,
                                                     c->tune_plan_sorted ? ubits : 0, ubits + cbits, c->stream));
    hipLaunchKernelGGL(k_plan_users<uint64_t>, dim3(grid), dim3(256), 0, c->stream,
                       c->plan_keys_sorted, n, ubits, users_out);
  }
  BPR_HIP_CHECK(hipGetLastError());
  c->plan_users = users_out;
  c->plan_pos = pos_out;
  c->plan_n = n;
  c->plan_chunk = chunk;
  return BPR_OK;
}


// One chunk of the plan (the same member set as chunk `index` of bpr_plan_epoch with the same seed,
// grouped by user), on `st`.  The plan does not depend on the model, so it can be computed for the
// chunk after next on the split refresh's side stream, in the time the sort leaves idle.
int plan_chunk_impl(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n, int64_t chunk,
                    uint64_t seed, int64_t index, int32_t* users_out, int32_t* pos_out, hipStream_t st) {
  const int64_t j0 = index * chunk;
  if (j0 >= n) return BPR_OK;
  const int64_t m = std::min<int64_t>(chunk, n - j0);
  int half_bits = (bits_for((uint64_t)(n - 1)) + 1) / 2;
  if (half_bits < 1) half_bits = 1;
  const int ubits = bits_for((uint64_t)(c->U - 1));
  if (c->pc_cap < m) {
    hipFree(c->pc_keys); hipFree(c->pc_vals); hipFree(c->pc_tmp);
    hipFree(c->pc_keys2); hipFree(c->pc_vals2); hipFree(c->pc_cnt);
    c->pc_keys = c->pc_keys2 = nullptr; c->pc_vals = c->pc_vals2 = nullptr; c->pc_tmp = nullptr;
    c->pc_cnt = nullptr;
    c->pc_cap = 0;
    BPR_HIP_CHECK(hipMalloc(&c->pc_keys, sizeof(uint32_t) * m));
    BPR_HIP_CHECK(hipMalloc(&c->pc_vals, sizeof(int32_t) * m));
    BPR_HIP_CHECK(hipMalloc(&c->pc_keys2, sizeof(uint32_t) * m));
    BPR_HIP_CHECK(hipMalloc(&c->pc_vals2, sizeof(int32_t) * m));
    BPR_HIP_CHECK(hipMalloc(&c->pc_cnt, sizeof(uint32_t) * 3 * PC_MAX_BUCKETS));
    BPR_HIP_CHECK(hipMemsetAsync(c->pc_cnt, 0, sizeof(uint32_t) * 3 * PC_MAX_BUCKETS, st));
    size_t bytes = 0;
    BPR_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, c->pc_keys,
                                                     reinterpret_cast<uint32_t*>(users_out), c->pc_vals,
                                                     pos_out, (int)m, 0, 32, st));
    BPR_HIP_CHECK(hipMalloc(&c->pc_tmp, bytes > 0 ? bytes : 16));
    c->pc_tmp_bytes = bytes;
    c->pc_cap = m;
  }
  const unsigned grid = (unsigned)std::min<int64_t>((m + 255) / 256, 1024);
  // three kernels when the users fit 2048 buckets of <= 1024 ids (U <= 2 M: every BASELINE shape);
  // BPR_PLAN_CHUNK_SORT=1 forces the device-wide sort (tests)
  const int shift = std::max(0, ubits - 11);
  static const bool force_sort = getenv("BPR_PLAN_CHUNK_SORT") != nullptr;
  if ((1 << shift) <= PC_MAX_LOCAL && !force_sort) {
    const int nb = (int)(((c->U - 1) >> shift) + 1);
    uint32_t *cnt = c->pc_cnt, *cur = c->pc_cnt + PC_MAX_BUCKETS, *base = c->pc_cnt + 2 * PC_MAX_BUCKETS;
    hipLaunchKernelGGL(k_pc_members, dim3(grid), dim3(256), 0, st, users_in, pos_in, n, j0, (int)m, half_bits,
                       seed, shift, c->pc_keys, c->pc_vals, cnt);
    hipLaunchKernelGGL(k_pc_scatter, dim3(std::min(grid, 256u)), dim3(256), 0, st, c->pc_keys, c->pc_vals, (int)m,
                       shift, nb, cnt, cur, base, c->pc_keys2, c->pc_vals2);
    hipLaunchKernelGGL(k_pc_group, dim3(nb), dim3(128), 0, st, c->pc_keys2, c->pc_vals2, shift, cnt, cur, base,
                       users_out, pos_out);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  }
  hipLaunchKernelGGL(k_plan_chunk, dim3(grid), dim3(256), 0, st, users_in, pos_in, n, j0, m, half_bits,
                     seed, c->pc_keys, c->pc_vals);
  size_t bytes = c->pc_tmp_bytes;
  BPR_HIP_CHECK(rocprim::radix_sort_pairs(c->pc_tmp, bytes, c->pc_keys,
                                                   reinterpret_cast<uint32_t*>(users_out), c->pc_vals,
                                                   pos_out, (int)m, 0, ubits, st));
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Hot item rows: popularity of the training positives -> the H most popular rows get replica
// delta rows for their STREAM updates (same-line atomic contention on the few
// hundred hot lines is what sets the kernel's floor on popularity-skewed data).
// ---------------------------------------------------------------------------------------------
__global__ void k_item_hist(const int32_t* __restrict__ pos, int64_t n, int64_t I,
                            uint32_t* __restrict__ counts) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int32_t it = pos[k];
    if (it >= 0 && it < I) atomicAdd(&counts[it], 1u);
  }
}
void hot_free(bpr_ctx* c) {
  hipFree(c->hot_slot);
  hipFree(c->hot_items);
  hipFree(c->hot_delta_alloc);
  hipFree(c->hot_canon);
  hipFree(c->hot_code);
  hipFree(c->hot_by_rank);
  c->hot_canon = c->hot_code = c->hot_by_rank = nullptr;
  c->hot_explicit = false;
  c->hot_tier = false;
  c->hot_delta_alloc = nullptr;
  c->hot_slot = c->hot_items = nullptr;
  c->hot_delta = nullptr;
  c->hot_H = c->hot_R = 0;
  c->hot_key_ptr = nullptr;
  c->hot_key_n = 0;
}

// Channel model behind the slot assignment (tools/ubench/atomic_bench.hip): memory
// is interleaved over HOT_CHANNELS channels in HOT_GRANULE-byte units, a row update sends one
// atomic request per 128-byte line, and a STREAM launch lasts as long as its most loaded channel
// (uniform popularity 0.185 ms; max/mean channel load 1.12 -> 0.207 ms, 1.455 -> 0.27 ms).
constexpr int HOT_CHANNELS = 128;
constexpr int64_t HOT_GRANULE = 256;
static inline int channel_of(uint64_t byte_addr) {
  return (int)((byte_addr / HOT_GRANULE) % HOT_CHANNELS);
}

// The H most popular rows take their STREAM updates in the delta block.  Which SLOT a row gets
// decides which channels carry its load: the rows are placed greedily, heaviest first, each into
// the free slot whose channels end up least loaded — counting the load the rows left in Q put on
// every channel — so the block evens out the whole launch, not only itself.
// cnt: positives per item (the channel model's load; may be all zero for a given hot set);
// given != NULL: the hot set, in the caller's canonical order (bpr_set_hot_items).
static int hot_build_from(bpr_ctx* c, std::vector<uint32_t>& cnt, const int32_t* given, int H, int64_t n) {
  const int64_t I = c->I;
  const int R = c->hot_reps_opt > 0 ? c->hot_reps_opt : 1;
  std::vector<int32_t> by((size_t)I);
  if (given != nullptr) {
    for (int k = 0; k < H; ++k) by[k] = given[k];
  } else {
    // the H most popular rows (ties by ascending id; the pad row and rows nobody likes stay out)
    for (int64_t i = 0; i < I; ++i) by[i] = (int32_t)i;
    if (c->pad_item >= 0 && c->pad_item < I) cnt[c->pad_item] = 0;
    std::partial_sort(by.begin(), by.begin() + H, by.end(), [&](int32_t x, int32_t y) {
      return cnt[x] != cnt[y] ? cnt[x] > cnt[y] : x < y;
    });
    while (H > 0 && cnt[by[H - 1]] == 0) --H;
  }
  if (H == 0) return BPR_OK;
  // the block starts on a channel-round boundary so that slot -> channels is known
  const size_t round_bytes = (size_t)HOT_CHANNELS * HOT_GRANULE;
  const size_t block_bytes = sizeof(float) * (size_t)R * H * c->d;
  BPR_HIP_CHECK(hipMalloc(&c->hot_delta_alloc, block_bytes + round_bytes));
  c->hot_delta = reinterpret_cast<float*>(((uintptr_t)c->hot_delta_alloc + round_bytes - 1) /
                                          round_bytes * round_bytes);
  BPR_HIP_CHECK(hipMemsetAsync(c->hot_delta, 0, block_bytes, c->stream));
  const int64_t row_bytes = (int64_t)c->d * 4;
  const int lines = (int)((row_bytes + 127) / 128);
  // expected line requests per row and launch: its positives, plus the negatives — close to
  // uniform over the items under both samplers (profiles/r03_neg_hist.txt)
  const double neg_share = (double)n / (double)(I - 1);
  std::vector<char> is_hot((size_t)I, 0);
  for (int k = 0; k < H; ++k) is_hot[by[k]] = 1;
  double load[HOT_CHANNELS] = {0.0};
  const uint64_t qbase = (uint64_t)(uintptr_t)c->Q;
  for (int64_t i = 1; i < I; ++i) {
    if (is_hot[i]) continue;
    const double w = (double)cnt[i] + neg_share;
    for (int l = 0; l < lines; ++l) load[channel_of(qbase + (uint64_t)(i * row_bytes + l * 128))] += w;
  }
  // placement order: heaviest first (a given set need not be sorted by popularity)
  std::vector<int> place((size_t)H);
  for (int k = 0; k < H; ++k) place[k] = k;
  std::stable_sort(place.begin(), place.end(), [&](int x, int y) { return cnt[by[x]] > cnt[by[y]]; });
  std::vector<int32_t> slot_of((size_t)I, -1), item_of((size_t)H, -1), canon_of((size_t)H, -1);
  std::vector<int32_t> code_of((size_t)I, -1), by_rank((size_t)H, -1);  // LDS tier: rank = placement order, heaviest first
  std::vector<char> used((size_t)H, 0);
  const uint64_t hbase = (uint64_t)(uintptr_t)c->hot_delta;
  static const bool naive = getenv("BPR_HOT_NAIVE") != nullptr;  // measurement aid: slot = rank
  // (the greedy search is H^2 slot evaluations: beyond 4,096 rows the block is filled in order —
  // that many rows even out over the channels by themselves)
  const bool in_order = naive || H > 4096;
  for (int kk = 0; kk < H; ++kk) {
    const int k = place[kk];
    const int32_t it = by[k];
    const double w = (double)cnt[it] + neg_share;
    int best = -1;
    double best_cost = 0.0;
    for (int s = 0; s < H && !in_order; ++s) {
      if (used[s]) continue;
      double cost = 0.0;  // the most loaded channel among the slot's lines, after the row moved in
      for (int l = 0; l < lines; ++l)
        cost = std::max(cost, load[channel_of(hbase + (uint64_t)(s * row_bytes + l * 128))] + w);
      if (best < 0 || cost < best_cost) {
        best = s;
        best_cost = cost;
      }
    }
    if (in_order) best = kk;
    used[best] = 1;
    slot_of[it] = best;
    item_of[best] = it;
    canon_of[best] = k;
    if (H < 32768) code_of[it] = (int32_t)(((uint32_t)kk << 16) | (uint32_t)best);
    by_rank[kk] = best;
    for (int l = 0; l < lines; ++l)
      load[channel_of(hbase + (uint64_t)(best * row_bytes + l * 128))] += w;
  }
  {
    double mx = 0.0, sum = 0.0;
    for (double v : load) {
      mx = std::max(mx, v);
      sum += v;
    }
    c->hot_balance = sum > 0.0 ? mx / (sum / HOT_CHANNELS) : 1.0;
    if (getenv("BPR_HOT_VERBOSE"))
      fprintf(stderr, "[bprcore] hot block: %d rows, modelled channel load max/mean = %.3f\n", H,
              c->hot_balance);
  }
  BPR_HIP_CHECK(hipMalloc(&c->hot_slot, sizeof(int32_t) * I));
  BPR_HIP_CHECK(hipMalloc(&c->hot_items, sizeof(int32_t) * H));
  BPR_HIP_CHECK(hipMalloc(&c->hot_canon, sizeof(int32_t) * H));
  BPR_HIP_CHECK(hipMemcpyAsync(c->hot_slot, slot_of.data(), sizeof(int32_t) * I,
                               hipMemcpyHostToDevice, c->stream));
  BPR_HIP_CHECK(hipMemcpyAsync(c->hot_items, item_of.data(), sizeof(int32_t) * H,
                               hipMemcpyHostToDevice, c->stream));
  BPR_HIP_CHECK(hipMemcpyAsync(c->hot_canon, canon_of.data(), sizeof(int32_t) * H,
                               hipMemcpyHostToDevice, c->stream));
  if (H < 32768) {
    BPR_HIP_CHECK(hipMalloc(&c->hot_code, sizeof(int32_t) * I));
    BPR_HIP_CHECK(hipMalloc(&c->hot_by_rank, sizeof(int32_t) * H));
    BPR_HIP_CHECK(hipMemcpyAsync(c->hot_code, code_of.data(), sizeof(int32_t) * I, hipMemcpyHostToDevice, c->stream));
    BPR_HIP_CHECK(hipMemcpyAsync(c->hot_by_rank, by_rank.data(), sizeof(int32_t) * H, hipMemcpyHostToDevice,
                                 c->stream));
  }
  BPR_HIP_CHECK(hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  c->hot_H = H;
  c->hot_R = R;
  return BPR_OK;
}

int hot_build_impl(bpr_ctx* c, const int32_t* pos, int64_t n) {
  if (c->hot_explicit) {  // the caller's hot set stays (bpr_set_hot_items); only note the training set
    c->hot_key_ptr = pos;
    c->hot_key_n = n;
    return BPR_OK;
  }
  hot_free(c);
  const int64_t I = c->I;
  int H = c->hot_rows_opt;
  if (H > I - 1) H = (int)(I - 1);
  c->hot_key_ptr = pos;
  c->hot_key_n = n;
  if (H <= 0 || c->hot_reps_opt <= 0 || n <= 0) return BPR_OK;
  uint32_t* counts = nullptr;
  BPR_HIP_CHECK(hipMalloc(&counts, sizeof(uint32_t) * I));
  BPR_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(uint32_t) * I, c->stream));
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_item_hist, dim3(grid), dim3(256), 0, c->stream, pos, n, I, counts);
  std::vector<uint32_t> cnt((size_t)I);
  BPR_HIP_CHECK(hipMemcpyAsync(cnt.data(), counts, sizeof(uint32_t) * I, hipMemcpyDeviceToHost,
                               c->stream));
  BPR_HIP_CHECK(hipStreamSynchronize(c->stream));  // one-time setup per training set
  hipFree(counts);
  return hot_build_from(c, cnt, nullptr, H, n);
}

// bpr_set_hot_items: the hot set as the caller gives it (the ranks of a multi-GPU job agree on it);
// counts (per item, may be NULL) only steer the slot placement.
int hot_set_items_impl(bpr_ctx* c, const int32_t* items, int H, const uint32_t* counts) {
  const void* key_ptr = c->hot_key_ptr;
  const int64_t key_n = c->hot_key_n;
  hot_free(c);
  c->hot_key_ptr = (const int32_t*)key_ptr;
  c->hot_key_n = key_n;
  c->hot_explicit = H > 0;
  if (H <= 0) return BPR_OK;
  std::vector<uint32_t> cnt((size_t)c->I, 0u);
  int64_t n = 0;
  if (counts != nullptr)
    for (int64_t i = 0; i < c->I; ++i) {
      cnt[i] = counts[i];
      n += counts[i];
    }
  if (c->hot_reps_opt <= 0) c->hot_reps_opt = 1;
  return hot_build_from(c, cnt, items, H, n > 0 ? n : 1);
}

// ---------------------------------------------------------------------------------------------
// Heavy users' seen bitmaps (bpr_device.h SeenBitmap / SeenList): users with more than T seen
// items get an I-bit row in HBM, filled once per seen CSR.
// ---------------------------------------------------------------------------------------------
__global__ void k_heavy_mark(const int64_t* __restrict__ indptr, int64_t U, int T, uint32_t words,
                             uint32_t* __restrict__ off, uint32_t* __restrict__ counter) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cnt = indptr[u + 1] - indptr[u];
    off[u] = cnt > (int64_t)T ? atomicAdd(counter, 1u) * words : 0xFFFFFFFFu;
  }
}
__global__ void k_heavy_fill(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                             int64_t U, const uint32_t* __restrict__ off,
                             uint32_t* __restrict__ bits) {
  for (int64_t u = blockIdx.x; u < U; u += gridDim.x) {  // a block per user (the few heavy ones work)
    const uint32_t o = off[u];
    if (o == 0xFFFFFFFFu) continue;
    const int64_t lo = indptr[u], hi = indptr[u + 1];
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
      const int32_t it = indices[k];
      atomicOr(&bits[o + (uint32_t)(it >> 5)], 1u << (it & 31));
    }
  }
}

void heavy_free(bpr_ctx* c) {
  hipFree(c->heavy_off);
  hipFree(c->heavy_bits);
  c->heavy_off = c->heavy_bits = nullptr;
  c->heavy_n = 0;
  c->heavy_for = nullptr;
}

int heavy_build_impl(bpr_ctx* c) {
  if (c->heavy_for == c->indptr) return BPR_OK;
  heavy_free(c);
  c->heavy_for = c->indptr;
  int T = c->heavy_T_opt;  // bpr_set_heavy_users (-1 = no heavy table)
  if (T < 0 || c->indptr == nullptr) return BPR_OK;
  const uint32_t words = (uint32_t)(((c->I + 31) / 32 + 3) / 4 * 4);
  uint32_t* counter = nullptr;
  BPR_HIP_CHECK(hipMalloc(&counter, sizeof(uint32_t)));
  BPR_HIP_CHECK(hipMalloc(&c->heavy_off, sizeof(uint32_t) * c->U));
  const unsigned grid = (unsigned)std::min<int64_t>((c->U + 255) / 256, 2048);
  uint32_t n_heavy = 0;
  for (;;) {  // at most 2^31 words (8 GB) of bitmaps: raise the threshold until they fit
    BPR_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_heavy_mark, dim3(grid), dim3(256), 0, c->stream, c->indptr, c->U, T, words,
                       c->heavy_off, counter);
    BPR_HIP_CHECK(hipMemcpyAsync(&n_heavy, counter, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                 c->stream));
    BPR_HIP_CHECK(hipStreamSynchronize(c->stream));  // one-time setup per seen CSR
    // the bitmaps must fit the cap (bpr_set_heavy_users; at most 2^31 words): raise the threshold
    // until they do
    const uint64_t cap_words = std::min<uint64_t>((uint64_t)1 << 31, (uint64_t)c->heavy_max_bytes / 4);
    if ((uint64_t)n_heavy * words < cap_words) break;
    T = T > 0 ? T * 2 : 1;
  }
  hipFree(counter);
  c->heavy_T = T;
  c->heavy_n = n_heavy;
  if (n_heavy == 0) {
    hipFree(c->heavy_off);
    c->heavy_off = nullptr;
    return BPR_OK;
  }
  const size_t bytes = sizeof(uint32_t) * (size_t)n_heavy * words;
  BPR_HIP_CHECK(hipMalloc(&c->heavy_bits, bytes));
  BPR_HIP_CHECK(hipMemsetAsync(c->heavy_bits, 0, bytes, c->stream));
  hipLaunchKernelGGL(k_heavy_fill, dim3((unsigned)std::min<int64_t>(c->U, 65535)), dim3(256), 0,
                     c->stream, c->indptr, c->indices, c->U, c->heavy_off, c->heavy_bits);
  BPR_HIP_CHECK(hipGetLastError());
  if (getenv("BPR_HOT_VERBOSE"))
    fprintf(stderr, "[bprcore] heavy users (> %d seen items): %u, %.1f MB of bitmaps\n", T, n_heavy,
            bytes / 1e6);
  return BPR_OK;
}

void refresh_free(bpr_ctx* c) {
  hot_free(c);
  heavy_free(c);
  hipFree(c->plan_keys);
  hipFree(c->plan_keys_sorted);
  hipFree(c->plan_tmp);
  c->plan_keys = c->plan_keys_sorted = nullptr;
  c->plan_tmp = nullptr;
  c->plan_cap = 0;
  if (c->side != nullptr) hipStreamSynchronize(c->side);
  hipFree(c->pc_keys); hipFree(c->pc_vals); hipFree(c->pc_tmp);
  hipFree(c->pc_keys2); hipFree(c->pc_vals2); hipFree(c->pc_cnt);
  c->pc_keys = c->pc_keys2 = nullptr; c->pc_vals = c->pc_vals2 = nullptr; c->pc_tmp = nullptr;
  c->pc_cnt = nullptr;
  c->pc_cap = 0;
  for (int k = 0; k < 2; ++k) {
    hipFree(c->order_alloc[k]);
    hipFree(c->sigma_buf[k]);
    c->order_alloc[k] = nullptr;
    c->sigma_buf[k] = nullptr;
  }
  for (int k = 0; k < 2; ++k) {
    hipFree(c->keysT_buf[k]);
    hipFree(c->sig_acc_buf[k]);
    hipFree(c->snap_meta[k]);
    c->keysT_buf[k] = nullptr;
    c->sig_acc_buf[k] = nullptr;
    c->snap_meta[k] = nullptr;
    c->snap_partial[k] = false;
    c->snap_keys[k] = nullptr;
  }
  c->meta_front = nullptr;
  c->keys_front = nullptr;
  hipFree(c->keys_sorted);
  hipFree(c->ids_in);
  hipFree(c->seg_offsets);
  c->sig_acc = nullptr;
  c->keys_cut = false;
  hipFree(c->sort_tmp);
  c->order = nullptr;
  c->sigma = nullptr;
  c->keysT = c->keys_sorted = nullptr;
  c->ids_in = c->seg_offsets = nullptr;
  c->sort_tmp = nullptr;
  c->sort_tmp_bytes = 0;
  c->have_snapshot = false;
  c->refresh_pending = false;
  c->part_pending = false;
}

void side_free(bpr_ctx* c) {  // bpr_ctx_destroy: the split refresh's events and (if ours) stream
  if (c->ev_keys) hipEventDestroy(c->ev_keys);
  if (c->ev_sorted) hipEventDestroy(c->ev_sorted);
  c->ev_keys = c->ev_sorted = nullptr;
  if (c->side_owned && c->side) hipStreamDestroy(c->side);
  c->side = nullptr;
  c->side_owned = false;
}

// The snapshot is taken in two steps.  CUT: k_transpose copies Q into the key buffer on the
// caller's stream — that instant is the snapshot's point in time.  SORT: the per-factor orders and
// sigma of those keys go to the BACK snapshot.  split = false: sort on the caller's stream and swap
// (AdaptiveSampler.update_stats as the reference calls it).  split = true
// (bpr_adaptive_refresh_begin): the sort runs on c->side behind an event, the caller's stream goes
// on — typically with the next STREAM launch, which is bound by the L2 atomic units and leaves the
// CUs mostly idle — and refresh_commit_impl (bpr_adaptive_refresh_commit) orders the swap.
int refresh_alloc(bpr_ctx* c) {
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
  if (c->order_alloc[0] != nullptr) return BPR_OK;
  for (int k = 0; k < 2; ++k) {
    BPR_HIP_CHECK(hipMalloc(&c->order_alloc[k], sizeof(int32_t) * (n + 2 * BPR_ORDER_PAD)));
    BPR_HIP_CHECK(hipMemsetAsync(c->order_alloc[k], 0, sizeof(int32_t) * (n + 2 * BPR_ORDER_PAD),
                                 c->stream));
    BPR_HIP_CHECK(hipMalloc(&c->sigma_buf[k], sizeof(float) * d));
  }
  for (int k = 0; k < 2; ++k) BPR_HIP_CHECK(hipMalloc(&c->snap_meta[k], sizeof(int32_t) * 2 * d));
  c->snap_front = 0;
  c->order = c->order_alloc[0] + BPR_ORDER_PAD;
  c->sigma = c->sigma_buf[0];
  for (int k = 0; k < 2; ++k) {
    BPR_HIP_CHECK(hipMalloc(&c->keysT_buf[k], sizeof(float) * n));
    BPR_HIP_CHECK(hipMalloc(&c->sig_acc_buf[k], sizeof(double) * 2 * d));
  }
  c->keys_w = 0;
  c->keysT = c->keysT_buf[0];
  c->sig_acc = c->sig_acc_buf[0];
  BPR_HIP_CHECK(hipMalloc(&c->keys_sorted, sizeof(uint64_t) * 2 * n));  // composite keys in|out
  BPR_HIP_CHECK(hipMalloc(&c->ids_in, sizeof(int32_t) * n));
  BPR_HIP_CHECK(hipMalloc(&c->seg_offsets, sizeof(int32_t) * (d + 1)));
  hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, c->stream, c->ids_in, c->seg_offsets, I, d);
  size_t bytes = 0;
  uint64_t* k64 = reinterpret_cast<uint64_t*>(c->keys_sorted);
  BPR_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, k64, k64 + n, c->ids_in,
                                                   c->order, (int)n, 0, 64, c->stream));
  BPR_HIP_CHECK(hipMalloc(&c->sort_tmp, bytes > 0 ? bytes : 16));
  c->sort_tmp_bytes = bytes;
  return BPR_OK;
}

int refresh_impl(bpr_ctx* c, bool split, int f_lo, int f_hi) {
  if (c->refresh_pending && !split && f_lo == 0 && f_hi == c->d) {
    // a lagged schedule ends every epoch with a split refresh in flight (StreamTrainer(refresh_lag=1));
    // whoever asks for a synchronous snapshot next — StrictTrainer, bpr_train_strict's refresh_every,
    // a plain bpr_adaptive_refresh — gets it: the pending one is committed first (its sort is waited
    // for, its buffers become the front pair) and the new snapshot then replaces it
    if (int rc = refresh_commit_impl(c)) return rc;
  }
  if (c->refresh_pending || c->part_pending) {
    set_error("bpr_adaptive_refresh: a split or sharded refresh is pending (bpr_adaptive_refresh_commit / "
              "_publish first)");
    return BPR_ERR_INVALID;
  }
  if (int rc = refresh_alloc(c)) return rc;
  if (f_lo < 0 || f_hi > c->d || f_lo >= f_hi) {
    set_error("bpr_adaptive_refresh_part: need 0 <= f_lo < f_hi <= d");
    return BPR_ERR_INVALID;
  }
  const int64_t I = c->I;
  const int d = c->d;
  const int64_t n = (int64_t)d * I;
  // f_lo .. f_hi: the factors (columns) this call sorts — all of them, or this rank's share of a
  // refresh sharded over the ranks of a multi-GPU job (bpr_adaptive_refresh_part: the caller
  // gathers the other columns into the back snapshot and publishes it)
  const bool part = f_lo != 0 || f_hi != d;
  const int nf = f_hi - f_lo;
  const int64_t foff = (int64_t)f_lo * I;
  const int back = c->have_snapshot ? (c->snap_front ^ 1) : c->snap_front;
  int32_t* const order = c->order_alloc[back] + BPR_ORDER_PAD + foff;
  float* const sigma = c->sigma_buf[back] + f_lo;
  // ---- cut (unless the last STREAM launch's epilogue did it: bpr_train_stream_cut)
  const bool event_on_cut = c->keys_cut && c->keys_event;  // ev_keys rode on the cut kernel
  if (!c->keys_cut) {
    dim3 tgrid((unsigned)((I + 31) / 32), (unsigned)((d + 31) / 32));
    hipLaunchKernelGGL(k_transpose, tgrid, dim3(256), 0, c->stream, c->Q, c->keysT, I, d,
                       c->sig_acc);
  }
  if (c->keys_cut && c->keys_on_side && !split) {
    // the keys were cut on the side stream (bpr_train_stream_acut) and this sort runs on the launch
    // stream: order it behind that cut (the split sort below waits for ev_keys on the side stream anyway)
    BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_keys, 0));
    c->acut_pending = false;
  }
  c->keys_cut = false;
  c->keys_event = false;
  c->keys_on_side = false;
  // the sort reads the buffers just cut; the next cut goes to the other pair
  const float* const keysT = c->keysT + foff;
  double* const sig_acc = c->sig_acc + 2 * f_lo;
  c->keys_w ^= 1;
  c->keysT = c->keysT_buf[c->keys_w];
  c->sig_acc = c->sig_acc_buf[c->keys_w];
  hipStream_t st = c->stream;
  if (split) {
    if (c->side == nullptr) {
      BPR_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
      c->side_owned = true;
    }
    if (c->ev_keys == nullptr) {
      BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_keys, hipEventDisableTiming));
      BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming));
    }
    if (!event_on_cut) BPR_HIP_CHECK(hipEventRecord(c->ev_keys, c->stream));
    BPR_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_keys, 0));
    st = c->side;
  }
  // ---- sort
  static const bool no_fast = getenv("BPR_NO_FAST_REFRESH") != nullptr;  // (process-wide test aids, read once)
  const int force_sub = c->tune_refresh_sub;  // bpr_set_tuning("refresh_sub", ...): tests force the split / merge paths on small tables
  // One 1024-thread workgroup sorts a (sub-)column of <= 36 keys per thread in LDS.  Columns are
  // split over 2 or 4 workgroups — sorted runs merged pairwise by k_merge_runs — when they do not
  // fit, or when d workgroups would leave CUs idle and the pieces stay >= 5,000 keys (measured on
  // ML-20M, refresh + launch gaps per step: d=128 0.106 -> 0.095 ms with 2, d=64 0.100 -> 0.079 ms
  // with 4; d=256 and Netflix's 4.8 k-item columns are fastest unsplit).  A split refresh shares
  // the chip with the caller's kernels: it keeps the columns whole (fewer, longer workgroups and
  // no merge pass) whenever they fit.
  int sub = 1;
  while (sub < 4 && (I + sub - 1) / sub > 1024 * 36) sub *= 2;
  if (!split)
    while (sub < 4 && nf * sub < 256 && I / (2 * sub) >= 5000) sub *= 2;
  if (force_sub == 1 || force_sub == 2 || force_sub == 4) sub = force_sub;
  // BINNED sort (r5): a column of <= 20,480 keys is ordered exactly by one workgroup in about a third of the radix
  // sort's time (k_sort_binned) — whole columns then beat split-and-merge on the idle chip too
  const bool binned_ok = c->tune_binned != 0 && !no_fast && force_sub == 0 && I >= 2048 &&
                         !(split && !part && c->tune_partial != 0);
  // ... with G workgroups per column (k_sort_binned_split) when a column does not fit one workgroup's LDS
  // (I <= 65,535: 16-bit ids)
  int binned_g = 0, binned_sitems = 0;
  if (binned_ok && I <= 65535) {
    if (c->tune_binned_split > 0) binned_g = c->tune_binned_split;  // (tests)
    else if (I > 1024 * 20) binned_g = (int)((I * 106 / 100 + 20 * 1024 - 1) / (20 * 1024));
    else binned_g = 1;  // (two workgroups per column on the idle chip were measured: 54.6 against 50.7 us per
                        // ML-20M refresh — every workgroup repeats the histogram passes)
    if (binned_g > 1) {
      for (;; ++binned_g) {  // the staged stretch (I / G keys + 6 % + a window) in 8 / 12 / 16 / 20 k entries
        const int64_t need = (I / binned_g) * 106 / 100 + 64;
        binned_sitems = need <= 8 * 1024 ? 8 : need <= 12 * 1024 ? 12 : need <= 16 * 1024 ? 16 : need <= 20 * 1024 ? 20 : 0;
        if (binned_sitems != 0) break;
      }
    } else if (I > 1024 * 20) {
      binned_g = 0;  // (forced to one workgroup per column but the column does not fit: the radix sort)
    }
  }
  const bool binned = binned_g == 1;
  if (binned || (binned_g > 1 && I <= 1024 * 36)) sub = 1;  // (the fallback of a flagged column: k_sort_flagged)
  int64_t len = (I + sub - 1) / sub;
  len = (len + 15) / 16 * 16;
  // PARTIAL order (r5): the split refresh of a column that one workgroup holds — the exact ends + a
  // bucketed middle, ~half the sort's work (k_sort_partial); everybody but k_stream gets the snapshot
  // completed on demand (snapshot_complete_impl)
  const bool partial = split && !part && c->tune_partial != 0 && sub == 1 && len <= 1024 * 24 && I <= 65535 &&
                       I >= 2048 && !no_fast;
  c->snap_partial[back] = partial;
  c->snap_keys[back] = keysT;
  if (partial) {
    const int items = (int)((len + 1023) / 1024);
    int32_t* meta = c->snap_meta[back];
    const int target = c->partial_target;
    if (items <= 6) launch_sort_partial<6>(c, st, keysT, order, sigma, meta, target);
    else if (items <= 10) launch_sort_partial<10>(c, st, keysT, order, sigma, meta, target);
    else if (items <= 12) launch_sort_partial<12>(c, st, keysT, order, sigma, meta, target);
    else if (items <= 16) launch_sort_partial<16>(c, st, keysT, order, sigma, meta, target);
    else if (items <= 20) launch_sort_partial<20>(c, st, keysT, order, sigma, meta, target);
    else launch_sort_partial<24>(c, st, keysT, order, sigma, meta, target);
    BPR_HIP_CHECK(hipGetLastError());
  } else if (binned) {
    const int items = (int)((len + 1023) / 1024);
    int32_t* meta = c->snap_meta[back] + 2 * f_lo;
    if (items <= 6) launch_sort_binned<6>(c, st, nf, keysT, order, sigma, meta);
    else if (items <= 10) launch_sort_binned<10>(c, st, nf, keysT, order, sigma, meta);
    else if (items <= 16) launch_sort_binned<16>(c, st, nf, keysT, order, sigma, meta);
    else launch_sort_binned<20>(c, st, nf, keysT, order, sigma, meta);
    BPR_HIP_CHECK(hipGetLastError());
  } else {
  // the radix path: every column, or — behind the split binned sort — the columns it flagged
  const int32_t* only_flagged = nullptr;
  if (binned_g > 1) {
    int32_t* meta = c->snap_meta[back] + 2 * f_lo;
    int rc = binned_sitems == 8    ? launch_sort_binned_split<8>(c, st, binned_g, nf, keysT, order, sigma, meta)
             : binned_sitems == 12 ? launch_sort_binned_split<12>(c, st, binned_g, nf, keysT, order, sigma, meta)
             : binned_sitems == 16 ? launch_sort_binned_split<16>(c, st, binned_g, nf, keysT, order, sigma, meta)
                                   : launch_sort_binned_split<20>(c, st, binned_g, nf, keysT, order, sigma, meta);
    if (rc != BPR_OK) return rc;
    BPR_HIP_CHECK(hipGetLastError());
    only_flagged = meta;
  }
  if (only_flagged != nullptr && sub == 1) {  // a flagged column fits one workgroup: k_sort_flagged
    const int items = (int)((len + 1023) / 1024);
    int32_t* meta = c->snap_meta[back] + 2 * f_lo;
    if (items <= 10) hipLaunchKernelGGL((k_sort_flagged<10>), dim3(nf), dim3(1024), 0, st, keysT, c->I, order, meta);
    else if (items <= 20) hipLaunchKernelGGL((k_sort_flagged<20>), dim3(nf), dim3(1024), 0, st, keysT, c->I, order, meta);
    else if (items <= 28) hipLaunchKernelGGL((k_sort_flagged<28>), dim3(nf), dim3(1024), 0, st, keysT, c->I, order, meta);
    else hipLaunchKernelGGL((k_sort_flagged<36>), dim3(nf), dim3(1024), 0, st, keysT, c->I, order, meta);
    BPR_HIP_CHECK(hipGetLastError());
  } else
  if (len <= 1024 * 36 && !no_fast) {
    float* keysA = reinterpret_cast<float*>(c->keys_sorted);
    int32_t* idsA = reinterpret_cast<int32_t*>(keysA + n);
    float* keysB = reinterpret_cast<float*>(idsA + n);
    int32_t* idsB = reinterpret_cast<int32_t*>(keysB + n);
    keysA += foff; idsA += foff; keysB += foff; idsB += foff;  // (the kernels index columns from 0)
    const int items = (int)((len + 1023) / 1024);
    if (items <= 6) launch_sort_sub<6>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else if (items <= 10) launch_sort_sub<10>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else if (items <= 12) launch_sort_sub<12>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else if (items <= 16) launch_sort_sub<16>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else if (items <= 20) launch_sort_sub<20>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else if (items <= 24) launch_sort_sub<24>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else if (items <= 28) launch_sort_sub<28>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    else launch_sort_sub<36>(c, st, nf, keysT, sig_acc, order, sigma, sub, len, keysA, idsA, only_flagged);
    int64_t run = len;
    for (int level = sub; level > 1; level /= 2, run *= 2) {
      const int last = level == 2;
      const int tiles_per_pair = (int)((2 * run + MERGE_TILE - 1) / MERGE_TILE);
      const unsigned mgrid = (unsigned)(((I + 2 * run - 1) / (2 * run)) * tiles_per_pair);
      hipLaunchKernelGGL(k_merge_runs, dim3(mgrid, nf), dim3(MERGE_THREADS), 0, st, keysA, idsA, I,
                         run, tiles_per_pair, keysB, last ? order : idsB, last, sigma, sig_acc, only_flagged);
      std::swap(keysA, keysB);
      std::swap(idsA, idsB);
    }
    BPR_HIP_CHECK(hipGetLastError());
  } else if (only_flagged == nullptr) {  // device-wide sort: always every column (a sharded refresh is merely redundant here)
    hipLaunchKernelGGL(k_sigma, dim3(d), dim3(256), 0, st, keysT - foff, I, sigma - f_lo);
    uint64_t* k64 = reinterpret_cast<uint64_t*>(c->keys_sorted);
    hipLaunchKernelGGL(k_compose_keys, dim3(2048), dim3(256), 0, st, keysT - foff, k64, n, I);
    int key_bits = 32;
    while ((1 << (key_bits - 32)) < d) ++key_bits;
    size_t bytes = c->sort_tmp_bytes;
    BPR_HIP_CHECK(rocprim::radix_sort_pairs(c->sort_tmp, bytes, k64, k64 + n, c->ids_in,
                                                     order - foff, (int)n, 0, key_bits, st));
    BPR_HIP_CHECK(hipGetLastError());
  }
  }
  if (part) {  // the caller fills the other columns and publishes (refresh_publish_impl)
    c->part_pending = true;
    return BPR_OK;
  }
  if (split) {
    BPR_HIP_CHECK(hipEventRecord(c->ev_sorted, c->side));
    c->refresh_pending = true;
    return BPR_OK;
  }
  c->snap_front = back;
  c->order = order;
  c->sigma = sigma;
  c->meta_front = c->snap_partial[back] ? c->snap_meta[back] : nullptr;
  c->keys_front = c->snap_keys[back];
  c->keys_front_stale = false;
  c->have_snapshot = true;
  return BPR_OK;
}

// A partial front snapshot sorted whole, in place, from the keys it was cut from: for every reader but
// k_stream (the sampler kernels behind the Python API, the batched STREAM kernel, bpr_adaptive_get_snapshot).
int snapshot_complete_impl(bpr_ctx* c) {
  if (c->meta_front == nullptr || !c->have_snapshot) return BPR_OK;
  if (c->keys_front_stale) {
    // the lag-1 pipeline's next cut writes the buffer this partial front was sorted from (the launch that
    // read the front is over by then); completing it now would sort NEWER keys into an order k_stream never walked
    set_error("the partial snapshot in front can no longer be completed: a later cut (bpr_train_stream_cut / _acut) "
              "overwrote the keys it was sorted from — commit the pending refresh (bpr_adaptive_refresh_commit) or "
              "refresh again before sampling through the API");
    return BPR_ERR_INVALID;
  }
  const int front = c->snap_front;
  const int64_t I = c->I;
  int64_t len = (I + 15) / 16 * 16;
  const int items = (int)((len + 1023) / 1024);
  int32_t* order = c->order_alloc[front] + BPR_ORDER_PAD;
  float* sigma = c->sigma_buf[front];
  double* acc = c->sig_acc;  // (unused by the single-workgroup form)
  const float* keys = c->keys_front;
  hipStream_t st = c->stream;
  if (items <= 6) launch_sort_sub<6>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else if (items <= 10) launch_sort_sub<10>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else if (items <= 12) launch_sort_sub<12>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else if (items <= 16) launch_sort_sub<16>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else if (items <= 20) launch_sort_sub<20>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else if (items <= 24) launch_sort_sub<24>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else if (items <= 28) launch_sort_sub<28>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  else launch_sort_sub<36>(c, st, c->d, keys, acc, order, sigma, 1, len, nullptr, nullptr);
  BPR_HIP_CHECK(hipGetLastError());
  c->snap_partial[front] = false;
  c->meta_front = nullptr;
  return BPR_OK;
}

int refresh_publish_impl(bpr_ctx* c) {
  if (!c->part_pending) {
    set_error("bpr_adaptive_refresh_publish: no sharded refresh is pending");
    return BPR_ERR_INVALID;
  }
  const int back = c->have_snapshot ? (c->snap_front ^ 1) : c->snap_front;
  c->snap_front = back;
  c->order = c->order_alloc[back] + BPR_ORDER_PAD;
  c->sigma = c->sigma_buf[back];
  c->meta_front = nullptr;  // (a sharded refresh is always sorted whole)
  c->keys_front = c->snap_keys[back];
  c->keys_front_stale = false;
  c->have_snapshot = true;
  c->part_pending = false;
  return BPR_OK;
}

int refresh_commit_impl(bpr_ctx* c) {
  if (!c->refresh_pending) {
    set_error("bpr_adaptive_refresh_commit: no split refresh is pending");
    return BPR_ERR_INVALID;
  }
  BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_sorted, 0));
  const int back = c->have_snapshot ? (c->snap_front ^ 1) : c->snap_front;
  c->snap_front = back;
  c->order = c->order_alloc[back] + BPR_ORDER_PAD;
  c->sigma = c->sigma_buf[back];
  c->meta_front = c->snap_partial[back] ? c->snap_meta[back] : nullptr;
  c->keys_front = c->snap_keys[back];
  c->keys_front_stale = false;
  c->have_snapshot = true;
  c->refresh_pending = false;
  return BPR_OK;
}

}  // namespace bpr
