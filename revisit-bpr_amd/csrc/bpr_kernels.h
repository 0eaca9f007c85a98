// bpr_kernels.h — the kernels of libbprcore (gfx950).  See DESIGN.md for the roofline of each.
//
//   k_stream<G,E,SAMPLER>    THE hot path (STREAM mode): a group walks a run of consecutive triples of
//                            a user-grouped chunk with the user row in registers:
//                            [sample j] → gather q_i, q_j → x, σ(−x) → SGD; item rows updated with
//                            full-line fp32 atomics, the user row written back once per run.
//   k_triples<G,E,MODE>      STRICT mode, one triple per group:
//                            MODE_FORWARD: logits + loss scalars only
//                            MODE_GRAD   : + accumulate per-row gradients (phase A)
//   k_apply<G,E>             STRICT phase B: one optimizer step per touched row
//   k_discard<G,E>, k_flush_lazy<G,E>, k_sample<G,E,WHAT> (samplers behind the Python API).
#pragma once
#include <type_traits>

#include "bpr_device.h"
#include "bpr_opt.h"

namespace bpr {

enum { MODE_FORWARD = 0, MODE_GRAD = 1 };

// occupancy request for k_stream (waves per SIMD). Measured on MI355X, ML-20M shape, adaptive: 4 → 0.325 ms, 5 → 0.305 ms, 6 → 0.444 ms, 8 → 0.544 ms per chunk (above 5 the allocator spills to scratch)
#ifndef BPR_STREAM_WAVES_PER_EU
#define BPR_STREAM_WAVES_PER_EU 5
#endif
struct ApplyArgs {
  float* P;
  float* Q;
  float* bias;
  float* GP;
  float* GQ;
  float* Gb;
  float *mP, *vP, *mQ, *vQ, *mb, *vb;
  int32_t* lastP;
  int32_t* lastQ;
  int32_t* flagP;
  int32_t* flagQ;
  const uint32_t* touched;
  const uint32_t* touched_cnt;
  uint32_t* next_cnt;  // the other half of the double-buffered counter: cleared for the next step
  int64_t U, I;
  int d;
  int pad_user, pad_item;
  OptDev o;
};

struct TripleArgs {
  float* P;
  float* Q;
  float* bias;
  int64_t I;
  int d;
  int pad_user, pad_item;
  float au, ai, an;
  // the batch
  const int32_t* users;
  const int32_t* pos;
  const int32_t* neg;
  int64_t n;
  // outputs
  float* lpos;
  float* lneg;
  float* scalars;   // caller's 4 floats (non-NULL = statistics wanted)
  float* partials;  // [gridDim.x, 4] per-block partial sums (ctx scratch)
  int32_t acc_partials;  // 1: add to the block's slot (bpr_train_strict sums once per call)
  // STRICT accumulators
  float* GP;
  float* GQ;
  float* Gb;
  int32_t* flagP;
  int32_t* flagQ;
  uint32_t* touched;
  uint32_t* touched_cnt;
  // stateful optimizers (STRICT): state + last-touched step, to read rows "as of now"
  const float *mP, *vP, *mQ, *vQ, *mb, *vb;
  const int32_t* lastP;
  const int32_t* lastQ;
  OptDev o;  // o.t = number of the step this forward belongs to
};

__device__ __forceinline__ void mark_touched(int32_t* flag, uint32_t row, uint32_t table,
                                             uint32_t* touched, uint32_t* cnt) {
  if (atomicExch(&flag[row], 1) == 0) {
    const uint32_t slot = atomicAdd(cnt, 1u);
    touched[slot] = row * 2u + table;
  }
}

__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ partials,
                                                      int n_blocks, float* __restrict__ out) {
  __shared__ double red[256][4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < n_blocks; b += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += (double)partials[(int64_t)b * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = acc[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
    __syncthreads();
  }
  if (threadIdx.x < 4) out[threadIdx.x] += (float)red[0][threadIdx.x];
}

// After a STREAM launch: fold the hot rows' replica deltas into Q and clear them (blocks 1..fold),
// and add the launch's loss statistics to the caller's scalars (block 0, when requested).
struct EpilogueArgs {
  const float* partials;
  float* out;
  float* Q;
  float* delta;
  const int32_t* hot_items;
  int32_t n_blocks, H, R, d, fold_blocks;
  // the item_bias k_stream worked on (one item per 128-B line, StreamArgs::bias) back into the
  // caller's dense vector: the blocks past the hot fold (NULL: nothing to write back)
  const float* bias_w;
  float* bias;
  int32_t I, bias_blocks;
};

__global__ __launch_bounds__(256) void k_stream_epilogue(const EpilogueArgs a) {
  if (blockIdx.x == 0) {
    if (a.out == nullptr) return;
    __shared__ double red[256][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < a.n_blocks; b += 256)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += (double)a.partials[(int64_t)b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
      __syncthreads();
    }
    if (threadIdx.x < 4) a.out[threadIdx.x] += (float)red[0][threadIdx.x];
    return;
  }
  const int d = a.d;
  if ((int)blockIdx.x > a.fold_blocks) {  // the bias blocks
    for (int32_t i = (int32_t)(blockIdx.x - 1 - a.fold_blocks) * 256 + (int32_t)threadIdx.x; i < a.I;
         i += a.bias_blocks * 256)
      a.bias[i] = a.bias_w[(size_t)i * 32];
    return;
  }
  const int64_t n = (int64_t)a.H * d;
  for (int64_t k = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x; k < n;
       k += (int64_t)a.fold_blocks * 256) {
    const int32_t it = a.hot_items[k / d];
    float sum = 0.f;
    for (int r = 0; r < a.R; ++r) {
      sum += a.delta[(int64_t)r * n + k];
      a.delta[(int64_t)r * n + k] = 0.f;
    }
    if (it >= 0) a.Q[(int64_t)it * d + (k % d)] += sum;
  }
}

// The same epilogue with the CUT of the next adaptive snapshot in one pass (bpr_train_stream_cut):
// every element of the item table is read once anyway to be transposed into the snapshot's key
// buffer T [d, I]; rows of the hot block take their delta on the way (Q += delta, delta = 0).  Block
// (0, 0) of the extra row gridDim.y - 1 sums the loss statistics.  Replaces k_stream_epilogue +
// k_transpose (bpr_refresh.hip) and one kernel boundary between two STREAM launches.
struct EpilogueCutArgs {
  const float* partials;
  float* out;
  float* Q;
  float* delta;
  const int32_t* hot_slot;  // NULL = no hot block
  float* T;
  double* sig_acc;
  int32_t n_blocks, H, R, d, I;
  // 0: read-only cut (bpr_train_stream_acut): the keys are Q + delta, nothing is folded — this pass
  // then runs on the side stream WHILE the next launch already updates the table
  int32_t fold;
  // item_bias write-back (EpilogueArgs): the statistics row of the grid does it, 32 items per block
  const float* bias_w;
  float* bias;
};

__global__ __launch_bounds__(256) void k_stream_epilogue_cut(const EpilogueCutArgs a) {
  if (blockIdx.y == gridDim.y - 1) {  // the statistics row
    if (a.bias != nullptr && threadIdx.x < 32) {
      const int32_t i = (int32_t)blockIdx.x * 32 + (int32_t)threadIdx.x;
      if (i < a.I) a.bias[i] = a.bias_w[(size_t)i * 32];
    }
    if (blockIdx.x != 0) return;
    for (int k = threadIdx.x; k < 2 * a.d; k += 256) a.sig_acc[k] = 0.0;  // (as k_transpose does)
    if (a.out == nullptr) return;
    __shared__ double red[256][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < a.n_blocks; b += 256)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += (double)a.partials[(int64_t)b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
      __syncthreads();
    }
    if (threadIdx.x < 4) a.out[threadIdx.x] += (float)red[0][threadIdx.x];
    return;
  }
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  const int f0 = blockIdx.y * 32;
  const int d = a.d;
  const int64_t hd = (int64_t)a.H * d;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int64_t i = i0 + ty + r;
    const int f = f0 + tx;
    float v = 0.f;
    if (i < a.I && f < d) {
      v = a.Q[i * d + f];
      const int32_t s = a.hot_slot != nullptr ? a.hot_slot[i] : -1;
      if (s >= 0) {
        float sum = 0.f;
        for (int rep = 0; rep < a.R; ++rep) {
          float* p = a.delta + rep * hd + (int64_t)s * d + f;
          sum += *p;
          if (a.fold) *p = 0.f;
        }
        v += sum;
        if (a.fold) a.Q[i * d + f] = v;
      }
    }
    tile[ty + r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int f = f0 + ty + r;
    const int64_t i = i0 + tx;
    if (f < d && i < a.I) a.T[(int64_t)f * a.I + i] = tile[tx][ty + r];
  }
}

// The multi-GPU form of that pass (r4, bpr_sync_cut): between two launches a rank of an N-rank job
// folds the hot exchange in flight and cuts the next one (k_hot_step), folds the cold reconciliation
// in flight and cuts the next delta (k_item_fold_delta), and cuts the keys of its next snapshot
// (k_transpose) — four kernels and four boundaries, 53 us measured (bench.py --emulate-ranks 8).
// Every one of them is an elementwise pass over the item table: here they are ONE, tile by tile as
// the cut needs it.  Per element of a HOT row (slot s, canonical index c):
//     hb[c] += htot[c] (fold_prev);  dl = sum of delta replicas, delta = 0;  htot[c] = dl;
//     q = hb[c] + dl;  cold base = q, cold own = tot = 0   (hot rows travel in the hot tier only)
// of a COLD row:   cold_mode 2: st = scale*tot; base += st; q += st - own; own = tot = q - base
//                  cold_mode 1: own = tot = q - base          (0: no cold tier this step)
// and T[f, i] = q.  The statistics row sums the launch's loss partials (deferred epilogue).
// (Streaming / nontemporal accesses to the four reconciliation buffers were tried: 29 us instead
// of 21 for the pass, same gaps.  Plain.)
struct SyncCutArgs {
  EpilogueCutArgs e;
  const int32_t* canon;
  float* hb;
  float* htot;
  float* base;
  float* own;
  float* tot;
  float scale;
  int32_t hot_tier, hot_fold_prev, cold_mode;
};

__global__ __launch_bounds__(256) void k_sync_cut(const SyncCutArgs a) {
  const EpilogueCutArgs& e = a.e;
  if (blockIdx.y == gridDim.y - 1) {  // the statistics row
    if (blockIdx.x != 0) return;
    for (int k = threadIdx.x; k < 2 * e.d; k += 256) e.sig_acc[k] = 0.0;
    if (e.out == nullptr) return;
    __shared__ double red[256][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < e.n_blocks; b += 256)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += (double)e.partials[(int64_t)b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
      __syncthreads();
    }
    if (threadIdx.x < 4) e.out[threadIdx.x] += (float)red[0][threadIdx.x];
    return;
  }
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  const int f0 = blockIdx.y * 32;
  const int d = e.d;
  const int64_t hd = (int64_t)e.H * d;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int64_t i = i0 + ty + r;
    const int f = f0 + tx;
    float v = 0.f;
    if (i < e.I && f < d) {
      const int64_t q = i * d + f;
      v = e.Q[q];
      const int32_t s = e.hot_slot != nullptr ? e.hot_slot[i] : -1;
      if (s >= 0) {
        float dl = 0.f;
        for (int rep = 0; rep < e.R; ++rep) {
          float* p = e.delta + rep * hd + (int64_t)s * d + f;
          dl += *p;
          *p = 0.f;
        }
        if (a.hot_tier) {
          const int64_t c = (int64_t)a.canon[s] * d + f;
          float b = a.hb[c];
          if (a.hot_fold_prev) b += a.htot[c];
          a.hb[c] = b;
          a.htot[c] = dl;
          v = b + dl;
          if (a.cold_mode != 0) {
            a.base[q] = v;
            a.own[q] = 0.f;
            a.tot[q] = 0.f;
          }
        } else {
          v += dl;
        }
        e.Q[q] = v;
      }
      if (s < 0 || !a.hot_tier) {
        if (a.cold_mode == 2) {
          const float st = a.scale * a.tot[q];
          const float nb = a.base[q] + st;
          const float nq = v + (st - a.own[q]);
          const float dl = nq - nb;
          a.base[q] = nb;
          a.own[q] = dl;
          a.tot[q] = dl;
          v = nq;
          e.Q[q] = v;
        } else if (a.cold_mode == 1) {
          const float dl = v - a.base[q];
          a.own[q] = dl;
          a.tot[q] = dl;
        }
      }
    }
    tile[ty + r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int f = f0 + ty + r;
    const int64_t i = i0 + tx;
    if (f < d && i < e.I) e.T[(int64_t)f * e.I + i] = tile[tx][ty + r];
  }
}

// ---------------------------------------------------------------------------------------------
// STREAM: the throughput kernel.
//
// The chunk [0, n) is cut into runs of `run_len` consecutive triples; group g walks run g, g+NG, …
// While consecutive triples share the user, the user row lives in registers (read once, written
// once).  When the chunk is grouped by user (bpr_plan_epoch) a user whose triples all fall inside
// one run is owned exclusively by that group for the whole launch → plain store, no atomics and no
// cross-XCD coherence question; a user whose triples straddle a run boundary gets the group's
// accumulated delta added atomically instead.  Item rows are shared by everybody → one full-line
// fp32 atomic add per 128 B of row.
// (r3: runs whose boundaries bend to user boundaries — a user's short tail finished by the run that
// started it — were built, held to the oracle, and measured: 0.5 line-atomics per triple fewer,
// but the two groups of a wave then walk ranges of different length in lockstep, and the kernel
// got SLOWER, 0.224 -> 0.236 ms per chunk with the adaptive sampler, 0.2003 vs 0.2008 ms with
// given negatives: profiles/r03_sweep_late_atomics.txt, r03_sweep_kstream_v1.txt.  Not kept.)
// ---------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) uint32_t bpr_smem[];

// kernel arguments of k_stream only (kept small: every field costs SGPRs for the whole kernel)
// The item_bias k_stream works on is a table of our own with ONE ITEM PER 128-B LINE (element
// i * BIAS_LINE): in the caller's dense vector 32 items share a line, every line takes the adds of
// all of them, memory-side atomics drop the line from L2, and each bias LOAD — in the dependent
// chain of the logit — then queues behind those adds at the memory side (46 us of a 264-us launch,
// profiles/shapes_r04.txt).  launch_stream fills it from the caller's vector before the launch
// (k_bias_widen) and writes it back after (k_bias_narrow): same arithmetic, another address.
constexpr int BIAS_LINE = 32;  // (the epilogues above index the wide table with this stride)
struct StreamArgs {
  float* P;
  float* Q;
  float* bias;  // the WIDE table (BIAS_LINE floats per item), or NULL
  const int64_t* indptr;
  const int32_t* indices;
  const int32_t* order;
  const float* sigma;
  const int32_t* users;
  const int32_t* pos;
  int32_t* neg;
  float* partials;  // NULL = no statistics
  uint64_t seed, offset;
  int32_t n, I, d;
  int32_t pad_user, pad_item;
  int32_t run_len, grouped, bm_words;
  int32_t gpw_active;  // groups of a wave that work (G = 32: 2; 1 = one triple at a time, tests)
  float au, ai, an, lr, inv_log1mp;
  // hot item rows: updates of row i with hot_slot[i] = s >= 0 go to the replica delta row
  // hot_delta[wave & hot_rmask][s]; its value is Q[i] + the sum of its replicas (NULL = off)
  const int32_t* hot_slot;
  float* hot_delta;
  int32_t hot_H, hot_rmask;
  ItemWeights iw;  // uniform sampler with item weights (NULL: uniform)
  // heavy users' precomputed seen bitmaps (bpr_device.h): word offset per user, ~0u = light
  const uint32_t* heavy_off;
  const uint32_t* heavy_bits;
  int32_t heavy_T;
  // a PARTIAL adaptive snapshot (bpr_refresh.hip k_sort_partial): {kt, kb} per column and the key columns
  // the snapshot was sorted from (NULL: `order` is sorted whole)
  const int32_t* snap_meta;
  const float* snap_keys;
};

// A hot row's value is its base row plus its replica delta rows; returns where this wave adds its
// update: an encoded row offset — >= 0: element offset into Q, < 0: ~(element offset into the
// delta block) — one register instead of a 64-bit pointer.
template <int G, int E>
__device__ __forceinline__ int32_t hot_row(float (&q)[E], const float* __restrict__ delta,
                                           int32_t slot, int32_t H, int32_t rmask, int wave, int d,
                                           int gl) {
  for (int32_t r = 0; r <= rmask; ++r) {
    const float* __restrict__ row = delta + (uint32_t)(r * H + slot) * (uint32_t)d;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int f = e * G + gl;
      if (f < d) q[e] += row[f];
    }
  }
  return ~(int32_t)((uint32_t)((wave & rmask) * H + slot) * (uint32_t)d);
}
__device__ __forceinline__ float* row_at(float* Q, float* delta, int32_t enc) {
  return enc >= 0 ? Q + (uint32_t)enc : delta + (uint32_t)(~enc);
}

// the snapshot's sigma, kept in LDS (k_stream): entry e of lane gl is factor e*G + gl
template <int G>
struct SigmaLds {
  const float* s;
  int gl;
  __device__ __forceinline__ float operator[](int e) const { return s[e * G + gl]; }
};

// FULL: d == G*E (32, 64, 128, 256, 512, 1024) — every `f < d` predicate folds away.
// (occupancy: the adaptive sampler over the staged-list structure needs a few registers more than
// 5 waves per SIMD leave; measured on MI355X, 4 and 5 waves run the kernel equally fast — it is not
// bound by occupancy — so that variant asks for 4 instead of spilling)
// PART: the adaptive snapshot may be partial (bpr_refresh.hip k_sort_partial) — its own instantiations
// (its in-bin finish needs ~20 registers more than 5 waves per SIMD leave: it asks for 4 — the kernel is
// not bound by occupancy, see above — instead of spilling)
template <int G, int E, int SAMPLER, int SEEN, bool FULL, bool PART = false>
__global__ __launch_bounds__(256, (E <= 4 ? ((SAMPLER == NEG_ADAPTIVE && SEEN == SEEN_LIST) || PART
                                                 ? BPR_STREAM_WAVES_PER_EU - 1
                                                 : BPR_STREAM_WAVES_PER_EU)
                                          : (E <= 8 ? 3 : 2)))
void k_stream(const StreamArgs a) {
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int d = FULL ? G * E : a.d;
  const int L = a.run_len;
  const int n_runs = (a.n + L - 1) / L;
  const bool stats = a.partials != nullptr;
  float s_loss = 0.f, s_reg = 0.f, s_abs = 0.f, s_cnt = 0.f;
  // per-group LDS scratch of W words: the seen-items bitmap of the current user (I bits, one
  // ds_read answers "seen?") or, for large item tables, the user's sorted seen list
  constexpr bool BM = SEEN == SEEN_BITMAP;
  const int W = a.bm_words;
  uint32_t* bm = bpr_smem + (threadIdx.x / G) * W;
  if constexpr (BM) {
    uint4* bm4 = reinterpret_cast<uint4*>(bm);
    for (int k = gl; k < (W >> 2); k += G) bm4[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  int32_t list_n = -1;
  uint32_t hoff = NOT_HEAVY;       // current user's row in the heavy users' HBM bitmaps
  int64_t cur_lo = 0;              // CSR slice of the current user: indices[cur_lo .. cur_lo + cur_n)
  int32_t cur_n = 0;
  __shared__ float s_sigma[SAMPLER == NEG_ADAPTIVE ? G * E : 1];  // snapshot sigma (adaptive only)
  if constexpr (SAMPLER == NEG_ADAPTIVE) {
    for (int k = threadIdx.x; k < G * E; k += blockDim.x) s_sigma[k] = k < d ? a.sigma[k] : 0.f;
    __syncthreads();
  }
  const SigmaLds<G> sg{s_sigma, gl};

  const int gpw = GPW == 1 ? 1 : a.gpw_active;
  for (int rbase = wave * gpw; rbase < n_runs; rbase += n_waves * gpw) {
    const int run = rbase + gw;
    const bool run_act = gw < gpw && run < n_runs;
    const int t0 = run_act ? run * L : 0;
    const int t1 = run_act ? min(t0 + L, a.n) : 0;
    // ---- run prologue: ONE coalesced load brings the ids of the whole run (lane k holds triple
    // t0+k; lane L the successor, lane G-1 the predecessor) and one more the users' CSR bounds,
    // so the per-triple dependent chain starts at the row gathers instead of at the ids.
    int32_t my_u = 0, my_i = 0, my_si = -1;
    int64_t my_lo = 0;
    int32_t my_cnt = 0;  // the user's seen items: indices[my_lo .. my_lo + my_cnt)
    {
      const int tk = (gl == G - 1) ? t0 - 1 : t0 + gl;
      const bool in_run = run_act && gl < L && tk < t1;
      const bool neighbour = run_act && ((gl == L && tk < a.n) || (gl == G - 1 && tk >= 0));
      if (in_run || neighbour) my_u = a.users[tk];
      if (in_run) {
        my_i = a.pos[tk];
        if (a.hot_slot != nullptr) my_si = a.hot_slot[my_i];
        if constexpr (SAMPLER != NEG_GIVEN) {
          my_lo = a.indptr[my_u];
          my_cnt = (int32_t)(a.indptr[my_u + 1] - my_lo);
        }
      }
    }
    // the model-independent draws of the run's triples, all at once (lane k = triple t0+k)
    AdaptiveRandoms my_rnd = {0.f, 0};
    if constexpr (SAMPLER == NEG_ADAPTIVE) {
      my_rnd = adaptive_randoms(a.seed, a.offset + (uint64_t)(t0 + gl), a.inv_log1mp,
                                (int64_t)(a.I - 1) - (int64_t)my_cnt);
    }
    // uniform: the first Philox block (candidates 0..3) of every triple of the run, all at once
    u32x4 my_w = {0u, 0u, 0u, 0u};
    if constexpr (SAMPLER == NEG_UNIFORM) {
      const uint64_t tc = a.offset + (uint64_t)(t0 + gl);
      my_w = philox4x32_10((uint32_t)tc, (uint32_t)(tc >> 32), 0u, PURPOSE_UNIFORM,
                           (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    }
    const int32_t prev_u = (run_act && t0 > 0) ? group_bcast<G>(my_u, G - 1, lane) : -1;
    const int32_t next_u = (run_act && t1 < a.n) ? group_bcast<G>(my_u, t1 - t0, lane) : -1;
    int32_t cur_u = -1;
    bool cur_starts_inside = false;
    // pl = live user row (memory value + this group's pending updates dp)
    float pl[E], dp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) pl[e] = dp[e] = 0.f;

    float x_mine = 0.f;  // statistics: logit of step gl of this run
    bool x_have = false;
    for (int step = 0; step < L; ++step) {
      const int t = t0 + step;
      const bool act = run_act && t < t1;
      const int tt = act ? t : (a.n - 1);
      const int32_t u = group_bcast<G>(my_u, step, lane);
      const int32_t i = group_bcast<G>(my_i, step, lane);
      const int64_t u_lo = group_bcast<G>(my_lo, step, lane);
      const int32_t u_cnt = group_bcast<G>(my_cnt, step, lane);
      if (act && u != cur_u) {
        // ---- user change: write the previous user's row back, fetch the new one
        if (cur_u >= 0 && cur_u != a.pad_user) {
          float* row = a.P + (uint32_t)cur_u * (uint32_t)d;
          if (a.grouped && cur_starts_inside) {  // run of cur_u ended here, inside my range
            store_row<G, E>(row, pl, d, gl);
          } else {
            atomic_add_row<G, E>(row, dp, d, gl);
          }
        }
        load_row<G, E>(pl, a.P + (uint32_t)u * (uint32_t)d, d, gl);
#pragma unroll
        for (int e = 0; e < E; ++e) dp[e] = 0.f;
        if constexpr (SAMPLER != NEG_GIVEN) {
          cur_lo = u_lo;
          cur_n = u_cnt;
          hoff = NOT_HEAVY;
          if (a.heavy_off != nullptr && cur_n > a.heavy_T) hoff = a.heavy_off[u];
          if (hoff == NOT_HEAVY) {
            if constexpr (BM) {
              // wipe: the whole bitmap with 16-byte LDS stores (W is a multiple of 4: 5 stores per
              // lane for ML-20M) — cheaper than re-reading the previous user's indices from HBM
              uint4* bm4 = reinterpret_cast<uint4*>(bm);
              for (int k = gl; k < (W >> 2); k += G) bm4[k] = make_uint4(0u, 0u, 0u, 0u);
              // set: 4 index loads in flight per trip
              const int32_t* __restrict__ ids = a.indices + cur_lo;
              for (int32_t k = gl; k < cur_n; k += 4 * G) {
                int32_t it[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int32_t kk = k + q * G;
                  it[q] = kk < cur_n ? ids[kk] : 0;  // bit 0 of word 0 = the pad item: harmless
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicOr(&bm[it[q] >> 5], 1u << (it[q] & 31));
              }
            }
            if constexpr (SEEN == SEEN_LIST) {
              list_n = cur_n <= W ? cur_n : -1;
              for (int32_t k = gl; k < list_n; k += 4 * G) {
                uint32_t it[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  it[q] = k + q * G < list_n ? (uint32_t)a.indices[cur_lo + k + q * G] : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (k + q * G < list_n) bm[k + q * G] = it[q];
              }
            }
          }
        }
        cur_u = u;
        cur_starts_inside = (step > 0) || (t == 0) || (prev_u != u);
      }

      int32_t j;
      if constexpr (SAMPLER == NEG_GIVEN) {
        j = a.neg[tt];
      } else {
        using Seen = typename std::conditional<
            BM, SeenBitmap,
            typename std::conditional<SEEN == SEEN_LIST, SeenList, SeenCsr>::type>::type;
        Seen seen;
        if constexpr (BM) {
          seen = SeenBitmap{bm, a.heavy_bits, hoff};
        } else if constexpr (SEEN == SEEN_LIST) {
          seen = SeenList{reinterpret_cast<const int32_t*>(bm), list_n, a.indices, cur_lo,
                          cur_lo + cur_n, a.heavy_bits, hoff};
        } else {
          seen = SeenCsr{a.indices, cur_lo, cur_lo + cur_n};
        }
        if constexpr (SAMPLER == NEG_UNIFORM) {
          // candidates 0..3 were drawn in the run prologue; lanes 0..3 test them.  All four seen
          // (probability (n_seen / I)^4) -> the general generator, which starts over and rejects
          // the same four first.
          const uint32_t w0 = group_bcast<G>(my_w.x, step, lane), w1 = group_bcast<G>(my_w.y, step, lane);
          const uint32_t w2 = group_bcast<G>(my_w.z, step, lane), w3 = group_bcast<G>(my_w.w, step, lane);
          const uint32_t wsel = (gl & 2) ? ((gl & 1) ? w3 : w2) : ((gl & 1) ? w1 : w0);
          const int32_t c = uniform_candidate(wsel, a.I, a.iw);
          const bool is_seen = seen(c);
          const Ballot b = wave_ballot(gl < 4 && !is_seen);
          if (group_first<G>(b, lane) >= 0 && act) {
            j = group_pick<G>(b, c, lane);
          } else {
            j = sample_uniform<G>(seen, (int64_t)cur_n, a.indices + cur_lo, a.I, a.seed,
                                  a.offset + (uint64_t)tt, lane, a.iw);
          }
        } else {
          const AdaptiveRandoms rnd = {group_bcast<G>(my_rnd.uf, step, lane),
                                       group_bcast<G>(my_rnd.r, step, lane)};
          const AdaptiveDraw dr = sample_adaptive<G, E, Seen, SigmaLds<G>, PART>(
              pl, d, sg, a.order, a.I, seen, (int64_t)cur_n, rnd, lane, a.snap_meta, a.snap_keys);
          j = dr.item;
          if constexpr (PART) {
            if (__builtin_expect(__any(dr.mid), 0)) {
              // the walk ended in the bucketed middle of a partial snapshot: finish inside the bin
              const int32_t kt = a.snap_meta[2 * dr.factor], kb = a.snap_meta[2 * dr.factor + 1];
              const int32_t zlo = dr.from_top ? kt : kb, zhi = a.I - (dr.from_top ? kb : kt);
              j = adaptive_finish_in_bin<G>(a.order + (int64_t)dr.factor * a.I, a.snap_keys + (int64_t)dr.factor * a.I,
                                            a.I, seen, dr.from_top, dr.kres, dr.mid, zlo, zhi, j, lane);
            }
          }
        }
        if (a.neg != nullptr && act && gl == 0) a.neg[t] = j;
      }
      // both item rows are gathered together, after the negative is known: the positive row would
      // only sit in registers while the sampler runs (they, not issue slots, bound the occupancy)
      int32_t irow = (int32_t)((uint32_t)i * (uint32_t)d);  // encoded row offsets (hot_row)
      int32_t jrow = (int32_t)((uint32_t)j * (uint32_t)d);
      float qi[E], qj[E];
      load_row<G, E>(qi, a.Q + (uint32_t)irow, d, gl);
      load_row<G, E>(qj, a.Q + (uint32_t)jrow, d, gl);
      const int32_t si = group_bcast<G>(my_si, step, lane);
      if (si >= 0) irow = hot_row<G, E>(qi, a.hot_delta, si, a.hot_H, a.hot_rmask, wave, d, gl);
      if (a.hot_slot != nullptr) {
        const int32_t sj = a.hot_slot[j];
        if (sj >= 0) jrow = hot_row<G, E>(qj, a.hot_delta, sj, a.hot_H, a.hot_rmask, wave, d, gl);
      }
      float bi = 0.f, bj = 0.f;
      if (a.bias != nullptr) {
        bi = a.bias[(uint32_t)i * (uint32_t)BIAS_LINE];
        bj = a.bias[(uint32_t)j * (uint32_t)BIAS_LINE];
      }

      // x_uij = <p_u, q_i - q_j> (+ bias difference): one group sum
      float xl = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) xl = fmaf(pl[e], qi[e] - qj[e], xl);
      float x = group_sum<G>(xl, lane);
      x += bi - bj;
      if (stats && act) {
        // the L2 term is accumulated per lane (its slice of the rows): no group sums; lane k keeps
        // the logit of step k and the loss terms are evaluated once per run, lane-parallel
        s_reg += 0.5f * (a.ai * dot<E>(qi, qi) + a.an * dot<E>(qj, qj) + a.au * dot<E>(pl, pl));
        if (gl == step) {
          x_mine = x;
          x_have = true;
        }
      }
      // ---- SGD on the three rows (SURVEY §3.3 gradients), w = σ(−x); every gradient uses the
      // pre-update values of this triple's rows
      const float w = 1.0f / (1.0f + expf(x));
      const float lr = a.lr;
      if (act) {
        // pad rows stay exactly zero: their update value is masked to 0 instead of branching
        const float mi = (i != a.pad_item) ? -lr : 0.f;
        const float mj = (j != a.pad_item) ? -lr : 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int f = e * G + gl;
          const float pe = pl[e];
          const float du = -lr * (-w * (qi[e] - qj[e]) + a.au * pe);
          dp[e] += du;
          pl[e] = pe + du;
          if (f < d) {
            atomic_add_f32(row_at(a.Q, a.hot_delta, irow) + f, mi * (-w * pe + a.ai * qi[e]));
            atomic_add_f32(row_at(a.Q, a.hot_delta, jrow) + f, mj * (w * pe + a.an * qj[e]));
          }
        }
        if (a.bias != nullptr && gl == 0) {
          atomic_add_f32(a.bias + (uint32_t)i * (uint32_t)BIAS_LINE, lr * w);
          atomic_add_f32(a.bias + (uint32_t)j * (uint32_t)BIAS_LINE, -lr * w);
        }
      }
    }
    if (stats && x_have) {
      s_loss += neg_logsigmoid(x_mine);
      s_abs += fabsf(x_mine);
      s_cnt += 1.f;
    }
    // ---- end of run: flush the last user
    if (run_act && cur_u >= 0 && cur_u != a.pad_user) {
      float* row = a.P + (uint32_t)cur_u * (uint32_t)d;
      const bool ends_inside = (t1 == a.n) || (next_u != cur_u);
      if (a.grouped && cur_starts_inside && ends_inside) {
        store_row<G, E>(row, pl, d, gl);
      } else {
        atomic_add_row<G, E>(row, dp, d, gl);
      }
    }
  }
  if (stats) reduce_scalars(a.partials, s_loss, s_reg, s_abs, s_cnt, lane);
}

// dense item_bias -> one item per line, and back (launch_stream, around every k_stream launch of a
// model with an item_bias)
__global__ __launch_bounds__(256) void k_bias_widen(const float* __restrict__ b, float* __restrict__ w,
                                                    int32_t I) {
  const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < I) w[(size_t)i * BIAS_LINE] = b[i];
}
__global__ __launch_bounds__(256) void k_bias_narrow(const float* __restrict__ w, float* __restrict__ b,
                                                     int32_t I) {
  const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < I) b[i] = w[(size_t)i * BIAS_LINE];
}

// ---------------------------------------------------------------------------------------------
// STRICT phase A / forward-only: one triple per group, negatives given.
// ---------------------------------------------------------------------------------------------
template <int G, int E, int MODE>
__global__ __launch_bounds__(256) void k_triples(const TripleArgs a) {
  constexpr int GPW = 64 / G;  // groups (triples) per wave
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int d = a.d;
  const bool stats = a.scalars != nullptr || a.acc_partials != 0;
  float s_loss = 0.f, s_reg = 0.f, s_abs = 0.f, s_cnt = 0.f;

  for (int64_t base = wave * GPW; base < a.n; base += n_waves * GPW) {
    const int64_t t = base + gw;
    const bool act = t < a.n;
    const int64_t tt = act ? t : a.n - 1;
    const int32_t u = a.users[tt];
    const int32_t i = a.pos[tt];
    const int32_t j = a.neg[tt];
    float p[E], qi[E], qj[E];
    load_row<G, E>(p, a.P + (int64_t)u * d, d, gl);
    load_row<G, E>(qi, a.Q + (int64_t)i * d, d, gl);
    load_row<G, E>(qj, a.Q + (int64_t)j * d, d, gl);
    float bias_i = 0.f, bias_j = 0.f;
    if (a.bias != nullptr) {
      bias_i = a.bias[i];
      bias_j = a.bias[j];
    }
    if (a.o.kind != OPT_SGD && a.lastP != nullptr) {
      const int64_t su = a.lastP[u], si = a.lastQ[i], sj = a.lastQ[j];
      const int64_t now = a.o.t - 1;
      catch_up_row<G, E>(p, a.mP, a.vP, u, d, gl, su, now - su, a.o);
      catch_up_row<G, E>(qi, a.mQ, a.vQ, i, d, gl, si, now - si, a.o);
      catch_up_row<G, E>(qj, a.mQ, a.vQ, j, d, gl, sj, now - sj, a.o);
      if (a.bias != nullptr) {
        float m = a.mb ? a.mb[i] : 0.f, v = a.vb ? a.vb[i] : 0.f;
        opt_replay(bias_i, m, v, si, now - si, a.o);
        m = a.mb ? a.mb[j] : 0.f;
        v = a.vb ? a.vb[j] : 0.f;
        opt_replay(bias_j, m, v, sj, now - sj, a.o);
      }
    }
    // ---- MF.forward (model.py:131-145) and BPR logits (model.py:48-64)
    const float xp = group_sum<G>(dot<E>(p, qi), lane) + bias_i;
    const float xn = group_sum<G>(dot<E>(p, qj), lane) + bias_j;
    const float x = xp - xn;
    if (act && gl == 0) {
      if (a.lpos != nullptr) a.lpos[t] = xp;
      if (a.lneg != nullptr) a.lneg[t] = xn;
    }
    if (stats) {
      // Loss (loss.py:20) and Model.regularization (model.py:87-93)
      const float np2 = group_sum<G>(dot<E>(p, p), lane);
      const float ni2 = group_sum<G>(dot<E>(qi, qi), lane);
      const float nj2 = group_sum<G>(dot<E>(qj, qj), lane);
      if (act && gl == 0) {
        s_loss += neg_logsigmoid(x);
        s_reg += 0.5f * (a.ai * ni2 + a.an * nj2 + a.au * np2);
        s_abs += fabsf(x);
        s_cnt += 1.f;
      }
    }
    if constexpr (MODE == MODE_GRAD) {
      // ---- pairwise gradient, w = σ(−x), accumulated per row
      const float w = 1.0f / (1.0f + expf(x));
      float gp[E], gi[E], gj[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        gp[e] = -w * (qi[e] - qj[e]) + a.au * p[e];
        gi[e] = -w * p[e] + a.ai * qi[e];
        gj[e] = w * p[e] + a.an * qj[e];
      }
      if (act) {
        if (u != a.pad_user) atomic_add_row<G, E>(a.GP + (int64_t)u * d, gp, d, gl);
        if (i != a.pad_item) atomic_add_row<G, E>(a.GQ + (int64_t)i * d, gi, d, gl);
        if (j != a.pad_item) atomic_add_row<G, E>(a.GQ + (int64_t)j * d, gj, d, gl);
        if (gl == 0) {
          if (a.bias != nullptr) {
            atomic_add_f32(a.Gb + i, -w);
            atomic_add_f32(a.Gb + j, w);
          }
          if (u != a.pad_user) mark_touched(a.flagP, (uint32_t)u, 0u, a.touched, a.touched_cnt);
          mark_touched(a.flagQ, (uint32_t)i, 1u, a.touched, a.touched_cnt);
          mark_touched(a.flagQ, (uint32_t)j, 1u, a.touched, a.touched_cnt);
        }
      }
    }
  }
  if (stats) reduce_scalars(a.partials, s_loss, s_reg, s_abs, s_cnt, lane, a.acc_partials != 0);
}

// ---------------------------------------------------------------------------------------------
// sampler-only kernels (UniformSampler.sample / AdaptiveSampler.sample behind the Python API)
// ---------------------------------------------------------------------------------------------
struct SampleArgs {
  const float* P;
  int64_t I;
  int d;
  const int64_t* indptr;
  const int32_t* indices;
  const int32_t* order;
  const float* sigma;
  float inv_log1mp;
  uint64_t seed, offset;
  const int32_t* users;
  const int32_t* factor_in;
  const int32_t* rank_in;
  int64_t n;
  int32_t* neg;
  int32_t* factor_out;
  int32_t* rank_out;
  // stateful optimizers: read the user row "as of now" (lazy dense-optimizer replay)
  const float *mP, *vP;
  const int32_t* lastP;
  OptDev o;
  int32_t bm_words;  // words per group of the LDS seen-bitmap (k_sample<..., BM = true>)
  ItemWeights iw;
};

enum { SAMPLE_UNIFORM = 0, SAMPLE_ADAPTIVE = 1, SAMPLE_PICK = 2 };

// BM: the walk's "seen?" tests go to a per-group LDS bitmap of the user's seen items (dynamic LDS,
// a.bm_words words per group) built once per pick with 8 index loads in flight — a launch lasts
// as long as its slowest pick, and that is a user with thousands of seen items whose candidates
// would otherwise each cost a binary search of dependent HBM loads.
template <int G, int E, int WHAT, bool BM>
__global__ __launch_bounds__(256) void k_sample(const SampleArgs a) {
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * GPW; base < a.n; base += n_waves * GPW) {
    const int64_t t = base + gw;
    const bool act = t < a.n;
    const int64_t tt = act ? t : a.n - 1;
    const int32_t u = a.users[tt];
    const int64_t lo = a.indptr[u], hi = a.indptr[u + 1];
    using Seen = typename std::conditional<BM, SeenBitmap, SeenCsr>::type;
    Seen seen;
    if constexpr (BM) {
      uint32_t* bm = bpr_smem + (threadIdx.x / G) * a.bm_words;
      uint4* bm4 = reinterpret_cast<uint4*>(bm);
      for (int k = gl; k < (a.bm_words >> 2); k += G) bm4[k] = make_uint4(0u, 0u, 0u, 0u);
      for (int64_t k = lo + gl; k < hi; k += 8 * G) {
        int32_t it[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) it[q] = k + q * G < hi ? a.indices[k + q * G] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) atomicOr(&bm[it[q] >> 5], 1u << (it[q] & 31));
      }
      seen = SeenBitmap{bm, nullptr, NOT_HEAVY};
    } else {
      seen = SeenCsr{a.indices, lo, hi};
    }
    if constexpr (WHAT == SAMPLE_UNIFORM) {
      const int32_t j = sample_uniform<G>(seen, hi - lo, a.indices + lo, a.I, a.seed,
                                          a.offset + (uint64_t)tt, lane, a.iw);
      if (act && gl == 0) a.neg[t] = j;
    } else if constexpr (WHAT == SAMPLE_ADAPTIVE) {
      float p[E];
      load_row<G, E>(p, a.P + (int64_t)u * a.d, a.d, gl);
      if (a.o.kind != OPT_SGD && a.lastP != nullptr) {
        const int64_t su = a.lastP[u];
        catch_up_row<G, E>(p, a.mP, a.vP, u, a.d, gl, su, (a.o.t - 1) - su, a.o);
      }
      float sg[E];
      load_row<G, E>(sg, a.sigma, a.d, gl);
      const AdaptiveDraw r =
          sample_adaptive<G, E>(p, a.d, sg, a.order, a.I, seen, hi - lo,
                                adaptive_randoms(a.seed, a.offset + (uint64_t)tt, a.inv_log1mp,
                                                 (a.I - 1) - (hi - lo)),
                                lane);
      if (act && gl == 0) {
        a.neg[t] = r.item;
        if (a.factor_out != nullptr) a.factor_out[t] = r.factor;
        if (a.rank_out != nullptr) a.rank_out[t] = r.rank;
      }
    } else {
      const int32_t f = a.factor_in[tt];
      const int32_t rk = a.rank_in[tt];
      const int32_t j = adaptive_walk<G>(a.order + (int64_t)f * a.I, a.I, seen, true, rk, lane);
      if (act && gl == 0) a.neg[t] = j;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// STRICT phase B: torch.optim step on the touched rows, with lazy replay of the zero-gradient
// steps a dense torch optimizer would have applied to the row since it was last touched (H2).
// ---------------------------------------------------------------------------------------------
template <int G, int E>
__global__ __launch_bounds__(256) void k_apply(const ApplyArgs a) {
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t cnt = *a.touched_cnt;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.next_cnt = 0u;  // saves a memset launch per step
  if (grp >= (int64_t)cnt) return;
  const uint32_t en = a.touched[grp];
  const bool is_item = (en & 1u) != 0u;
  const int64_t row = (int64_t)(en >> 1);
  const int d = a.d;
  float* W = (is_item ? a.Q : a.P) + row * d;
  float* Gr = (is_item ? a.GQ : a.GP) + row * d;
  float* M = is_item ? a.mQ : a.mP;
  float* V = is_item ? a.vQ : a.vP;
  int32_t* last = is_item ? a.lastQ : a.lastP;
  int32_t* flag = is_item ? a.flagQ : a.flagP;
  const int pad = is_item ? a.pad_item : a.pad_user;
  const OptDev& o = a.o;
  const bool stateful = o.kind != OPT_SGD;
  const int64_t s0 = stateful ? (int64_t)last[row] : 0;
  const int64_t k = stateful ? (o.t - 1) - s0 : 0;
  const float adam_step = o.adam_step, adam_bc2_sqrt = o.adam_bc2;
  if (row != pad) {
    float w[E], g[E], m[E], v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int f = e * G + gl;
      const bool in = f < d;
      w[e] = in ? W[f] : 0.f;
      g[e] = in ? Gr[f] : 0.f;
      m[e] = (in && M != nullptr) ? M[row * d + f] : 0.f;
      v[e] = (in && V != nullptr) ? V[row * d + f] : 0.f;
    }
    opt_replay_row<E>(w, m, v, s0, k, o);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int f = e * G + gl;
      if (f >= d) continue;
      opt_update(w[e], g[e], m[e], v[e], o, adam_step, adam_bc2_sqrt);
      W[f] = w[e];
      Gr[f] = 0.f;
      if (M != nullptr) M[row * d + f] = m[e];
      if (V != nullptr) V[row * d + f] = v[e];
    }
  }
  if (gl == 0) {
    if (is_item && a.bias != nullptr) {
      float w = a.bias[row], m = a.mb ? a.mb[row] : 0.f, v = a.vb ? a.vb[row] : 0.f;
      const float g = a.Gb[row];
      opt_replay(w, m, v, s0, k, o);
      opt_update(w, g, m, v, o, adam_step, adam_bc2_sqrt);
      a.bias[row] = w;
      a.Gb[row] = 0.f;
      if (a.mb) a.mb[row] = m;
      if (a.vb) a.vb[row] = v;
    }
    flag[row] = 0;
    if (stateful) last[row] = (int32_t)o.t;
  }
}

// drop accumulated gradients of the touched rows
template <int G, int E>
__global__ __launch_bounds__(256) void k_discard(const ApplyArgs a) {
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.next_cnt = 0u;
  if (grp >= (int64_t)*a.touched_cnt) return;
  const uint32_t en = a.touched[grp];
  const bool is_item = (en & 1u) != 0u;
  const int64_t row = (int64_t)(en >> 1);
  float* Gr = (is_item ? a.GQ : a.GP) + row * a.d;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    if (f < a.d) Gr[f] = 0.f;
  }
  if (gl == 0) {
    if (is_item && a.Gb != nullptr) a.Gb[row] = 0.f;
    (is_item ? a.flagQ : a.flagP)[row] = 0;
  }
}

// bring every row of one table to step o.t (dense sweep; before eval / checkpoint / all-reduce)
template <int G, int E>
__global__ __launch_bounds__(256) void k_flush_lazy(const ApplyArgs a, const int is_item) {
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) / G;
  const int64_t rows = is_item ? a.I : a.U;
  const int d = a.d;
  float* Wt = is_item ? a.Q : a.P;
  float* M = is_item ? a.mQ : a.mP;
  float* V = is_item ? a.vQ : a.vP;
  int32_t* last = is_item ? a.lastQ : a.lastP;
  const OptDev& o = a.o;
  for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; row < rows;
       row += n_groups) {
    const int64_t s0 = last[row];
    const int64_t k = o.t - s0;
    if (k <= 0) continue;
    float w[E], m[E], v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int f = e * G + gl;
      const bool in = f < d;
      w[e] = in ? Wt[row * d + f] : 0.f;
      m[e] = (in && M != nullptr) ? M[row * d + f] : 0.f;
      v[e] = (in && V != nullptr) ? V[row * d + f] : 0.f;
    }
    opt_replay_row<E>(w, m, v, s0, k, o);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int f = e * G + gl;
      if (f >= d) continue;
      Wt[row * d + f] = w[e];
      if (M != nullptr) M[row * d + f] = m[e];
      if (V != nullptr) V[row * d + f] = v[e];
    }
    if (gl == 0) {
      if (is_item && a.bias != nullptr) {
        float w = a.bias[row], m = a.mb ? a.mb[row] : 0.f, v = a.vb ? a.vb[row] : 0.f;
        opt_replay(w, m, v, s0, k, o);
        a.bias[row] = w;
        if (a.mb) a.mb[row] = m;
        if (a.vb) a.vb[row] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) last[row] = (int32_t)o.t;
  }
}

// ---------------------------------------------------------------------------------------------
// item-table reconciliation (multi-GPU): the two elementwise passes around the RCCL all-reduce,
// each ONE pass over the table (HBM bound: 12 B / 24 B moved per element)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_item_delta(const float* __restrict__ q,
                                                    const float* __restrict__ base,
                                                    float* __restrict__ own,
                                                    float* __restrict__ tot, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    const float dlt = q[k] - base[k];
    own[k] = dlt;
    tot[k] = dlt;
  }
}

__global__ __launch_bounds__(256) void k_item_fold(float* __restrict__ q, float* __restrict__ base,
                                                   const float* __restrict__ own,
                                                   const float* __restrict__ tot, float scale,
                                                   int rebase, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    const float st = scale * tot[k];
    const float nb = base[k] + st;  // identical on every rank: the bases never drift apart
    base[k] = nb;
    q[k] = rebase ? nb : q[k] + (st - own[k]);
  }
}

// fold of the previous reconciliation and cut of the next delta in one pass (nothing trained in
// between): q += st - own; base += st; own = tot = q - base
__global__ __launch_bounds__(256) void k_item_fold_delta(float* __restrict__ q,
                                                         float* __restrict__ base,
                                                         float* __restrict__ own,
                                                         float* __restrict__ tot, float scale,
                                                         int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    const float st = scale * tot[k];
    const float nb = base[k] + st;
    const float nq = q[k] + (st - own[k]);
    const float dl = nq - nb;
    base[k] = nb;
    q[k] = nq;
    own[k] = dl;
    tot[k] = dl;
  }
}

// ---------------------------------------------------------------------------------------------
// two-tier reconciliation, HOT tier (multi-GPU): the hot block of a launch is exchanged on its own,
// every launch (or sub-launch), while the cold rows wait for the per-period all-reduce.
//   hb  [H, d]  the hot rows as of the last reconciled exchange — bit-identical on every rank
//               (only all-reduced sums are ever added to it), canonical row order
//   tot [H, d]  in: the all-reduced sum of every rank's deltas of the PREVIOUS exchange (fold_prev);
//               out: this rank's deltas of the launch just finished (cut), to be all-reduced next
// One pass: hb += tot_prev;  dl = delta[slot] (summed over replicas), delta = 0;  tot = dl;
//           Q[item] = hb + dl   — the reconciled value plus what only this rank knows so far.
// cold_base (optional): the cold tier's base of the same rows is set to the new Q, so that the cold
// tier's delta of a hot row is exactly zero (hot rows travel in the hot tier only).
// ---------------------------------------------------------------------------------------------
struct HotStepArgs {
  float* Q;
  float* delta;
  const int32_t* hot_items;
  const int32_t* canon;
  float* hb;
  float* tot;
  float* cold_base;
  int32_t H, R, d, fold_prev, cut;
};

__global__ __launch_bounds__(256) void k_hot_step(const HotStepArgs a) {
  const int64_t n = (int64_t)a.H * a.d;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(k / a.d), f = (int)(k % a.d);
    const int32_t it = a.hot_items[s];
    const int64_t c = (int64_t)a.canon[s] * a.d + f;
    float b = a.hb[c];
    if (a.fold_prev) b += a.tot[c];
    float dl = 0.f;
    if (a.cut) {
      for (int r = 0; r < a.R; ++r) {
        dl += a.delta[(int64_t)r * n + k];
        a.delta[(int64_t)r * n + k] = 0.f;
      }
      a.tot[c] = dl;
    }
    a.hb[c] = b;
    if (it >= 0) {
      const int64_t q = (int64_t)it * a.d + f;
      a.Q[q] = b + dl;
      if (a.cold_base != nullptr) a.cold_base[q] = b + dl;
    }
  }
}

// hb = the hot rows of Q in canonical order (start of the hot tier: replicas are identical)
__global__ __launch_bounds__(256) void k_hot_gather(const float* __restrict__ Q,
                                                    const int32_t* __restrict__ hot_items,
                                                    const int32_t* __restrict__ canon,
                                                    float* __restrict__ hb, int H, int d) {
  const int64_t n = (int64_t)H * d;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(k / d), f = (int)(k % d);
    hb[(int64_t)canon[s] * d + f] = Q[(int64_t)hot_items[s] * d + f];
  }
}

}  // namespace bpr
