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
#include "bpr_stream.h"

namespace bpr {

enum { MODE_FORWARD = 0, MODE_GRAD = 1 };

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
