// bpr_stream.h — k_stream, THE hot kernel of libbprcore (STREAM mode), and what it needs; a header of its own
// because two translation units instantiate it (bprcore.hip: the plain kernel; bpr_hotlds.hip: the LDS tier of
// the hot block).  See bpr_kernels.h for the kernels around it and DESIGN.md §4.1 for its roofline.
#pragma once
#include <type_traits>

#include "bpr_device.h"

namespace bpr {

// occupancy request for k_stream (waves per SIMD). Measured on MI355X, ML-20M shape, adaptive: 4 → 0.325 ms, 5 → 0.305 ms, 6 → 0.444 ms, 8 → 0.544 ms per chunk (above 5 the allocator spills to scratch)
#ifndef BPR_STREAM_WAVES_PER_EU
#define BPR_STREAM_WAVES_PER_EU 5
#endif
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
  // LDS tier of the hot block (k_stream<..., LDSHOT = true>, bpr_hotlds.hip): `hot_slot` then holds CODES —
  // -1 = cold, else (popularity rank << 16 | slot) — the lds_L most popular rows take this workgroup's
  // updates in a delta block in LDS, flushed into slot hot_by_rank[rank] of the global block at exit
  int32_t lds_L;
  const int32_t* hot_by_rank;
  int32_t lds_only;      // 1: hot rows past lds_L are updated in Q like cold rows (no hot tier that wants their deltas)
  int32_t tail1, tail2;  // LDSHOT: first triple of the zones of runs of run_len / 2 and run_len / 4 (n: no such zone)
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
// LDS tier (LDSHOT): a third kind of encoded row — ~(LDS_ROW | element offset into the workgroup's LDS
// delta block); offsets into the global block stay below 2^30 (H * R * d)
constexpr uint32_t LDS_ROW = 0x40000000u;
__device__ __forceinline__ bool is_lds_row(int32_t enc) { return enc < 0 && ((uint32_t)(~enc) & LDS_ROW) != 0u; }
__device__ __forceinline__ uint32_t lds_row_off(int32_t enc) { return (uint32_t)(~enc) & (LDS_ROW - 1u); }
template <int G, int E>
__device__ __forceinline__ int32_t lds_row(float (&q)[E], const float* hl, int32_t rank, int d, int gl) {
  const float* row = hl + (uint32_t)rank * (uint32_t)d;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    if (f < d) q[e] += row[f];
  }
  return ~(int32_t)(LDS_ROW | ((uint32_t)rank * (uint32_t)d));
}
__device__ __forceinline__ void lds_add_f32(float* addr, float v) {  // ds_add_f32, no return
  __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
//
// LDSHOT (r6, bpr_hotlds.hip): ONE workgroup of up to 1,024 threads per CU, persistent over its share of the
// runs, with a private fp32 delta block for the lds_L most popular item rows in LDS beside the seen bitmaps.
// On a trained model 40-60 % of the adaptive negatives and half of the positives fall on a few hundred rows
// (profiles/r05_trained_state.md): every such update was 4 memory-side line requests plus a read of a delta
// row the atomics keep dropping from the L2s.  Here it is a ds_add_f32, and the row's value is Q + this
// workgroup's own LDS delta: the other workgroups' updates of the launch arrive with the epilogue's fold,
// i.e. a hot row is at most one launch stale per CU — the staleness the multi-GPU budget rule prices
// (fast.lag_within_budget).  Dirty rows are flushed into the global delta block with line atomics at exit, so
// the epilogues, the hot tier and the exact-sum property are unchanged.  Runs are dealt to workgroups
// interleaved (run pair k of workgroup b = k * gridDim + b: a heavy user's consecutive runs spread over the
// CUs) and to the waves of a workgroup by an LDS ticket (no static tail inside the workgroup).
template <int G, int E, int SAMPLER, int SEEN, bool FULL, bool PART = false, bool LDSHOT = false>
__global__ __launch_bounds__(LDSHOT ? 1024 : 256,
                             (LDSHOT ? (E <= 4 ? 4 : (E <= 8 ? 3 : 2))
                                     : (E <= 4 ? ((SAMPLER == NEG_ADAPTIVE && SEEN == SEEN_LIST) || PART
                                                      ? BPR_STREAM_WAVES_PER_EU - 1
                                                      : BPR_STREAM_WAVES_PER_EU)
                                               : (E <= 8 ? 3 : 2))))
void k_stream(const StreamArgs a) {
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int d = FULL ? G * E : a.d;
  const int L = a.run_len;
  // The chunk as runs: [0, tail1) in runs of L triples, [tail1, tail2) in runs of L / 2, [tail2, n) in runs of
  // L / 4 (LDSHOT: a persistent workgroup is over when its last run is, so the last tickets are short ones —
  // guided self-scheduling; the plain kernel's launches have tail1 = tail2 = n).  The zones hold whole wave-loads
  // of runs, so the run length is wave-uniform.
  const int L2 = L >= 2 ? L >> 1 : 1, L3 = L >= 4 ? L >> 2 : 1;
  const int R1 = LDSHOT ? a.tail1 / L : (a.n + L - 1) / L;
  const int R2 = LDSHOT ? R1 + (a.tail2 - a.tail1) / L2 : R1;
  const int n_runs = LDSHOT ? R2 + (a.n - a.tail2 + L3 - 1) / L3 : R1;
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
  // LDSHOT: [lds_L, d] fp32 deltas + a dirty word per row, behind the groups' seen structures
  float* const hl = reinterpret_cast<float*>(bpr_smem + (SAMPLER == NEG_GIVEN ? 0 : (blockDim.x / G) * W));
  uint32_t* const hl_dirty = reinterpret_cast<uint32_t*>(hl + (LDSHOT ? a.lds_L * d : 0));
  __shared__ int s_ticket;
  if constexpr (LDSHOT) {
    for (int k = threadIdx.x; k < a.lds_L * d + a.lds_L; k += blockDim.x) hl[k] = 0.f;
    if (threadIdx.x == 0) s_ticket = 0;
    __syncthreads();
  }

  const int gpw = GPW == 1 ? 1 : a.gpw_active;
  for (int it = 0;; ++it) {
    int rbase;
    if constexpr (LDSHOT) {
      int tk = 0;
      if (lane == 0) tk = atomicAdd(&s_ticket, 1);
      tk = __builtin_amdgcn_readfirstlane(tk);
      rbase = (tk * (int)gridDim.x + (int)blockIdx.x) * gpw;
    } else {
      rbase = (wave + it * n_waves) * gpw;
    }
    if (rbase >= n_runs) break;
    const int run = rbase + gw;
    const bool run_act = gw < gpw && run < n_runs;
    int Lr = L, t0r = run * L;
    if constexpr (LDSHOT) {
      if (rbase >= R2) {
        Lr = L3;
        t0r = a.tail2 + (run - R2) * Lr;
      } else if (rbase >= R1) {
        Lr = L2;
        t0r = a.tail1 + (run - R1) * Lr;
      }
    }
    const int t0 = run_act ? t0r : 0;
    const int t1 = run_act ? min(t0 + Lr, a.n) : 0;
    // ---- run prologue: ONE coalesced load brings the ids of the whole run (lane k holds triple
    // t0+k; lane L the successor, lane G-1 the predecessor) and one more the users' CSR bounds,
    // so the per-triple dependent chain starts at the row gathers instead of at the ids.
    int32_t my_u = 0, my_i = 0, my_si = -1;
    int64_t my_lo = 0;
    int32_t my_cnt = 0;  // the user's seen items: indices[my_lo .. my_lo + my_cnt)
    {
      const int tk = (gl == G - 1) ? t0 - 1 : t0 + gl;
      const bool in_run = run_act && gl < Lr && tk < t1;
      const bool neighbour = run_act && ((gl == Lr && tk < a.n) || (gl == G - 1 && tk >= 0));
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
    for (int step = 0; step < Lr; ++step) {
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
      if constexpr (LDSHOT) {
        // codes (StreamArgs::lds_L): rank < lds_L -> this workgroup's LDS delta row, else the global block
        const int32_t sj = a.hot_slot[j];
        // (without a hot tier the rest of the block is left alone: beside the LDS rows a global delta row costs a
        // read under atomic fire and buys no balance — 733 against 708 M triples/s, profiles/r06_hotlds_sweep.txt)
        if (si >= 0) {
          if ((si >> 16) < a.lds_L) irow = lds_row<G, E>(qi, hl, si >> 16, d, gl);
          else if (!a.lds_only) irow = hot_row<G, E>(qi, a.hot_delta, si & 0xFFFF, a.hot_H, a.hot_rmask, wave, d, gl);
        }
        if (sj >= 0) {
          if ((sj >> 16) < a.lds_L) jrow = lds_row<G, E>(qj, hl, sj >> 16, d, gl);
          else if (!a.lds_only) jrow = hot_row<G, E>(qj, a.hot_delta, sj & 0xFFFF, a.hot_H, a.hot_rmask, wave, d, gl);
        }
      } else {
        if (si >= 0) irow = hot_row<G, E>(qi, a.hot_delta, si, a.hot_H, a.hot_rmask, wave, d, gl);
        if (a.hot_slot != nullptr) {
          const int32_t sj = a.hot_slot[j];
          if (sj >= 0) jrow = hot_row<G, E>(qj, a.hot_delta, sj, a.hot_H, a.hot_rmask, wave, d, gl);
        }
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
        const bool i_lds = LDSHOT && is_lds_row(irow), j_lds = LDSHOT && is_lds_row(jrow);
        if constexpr (LDSHOT) {
          if (gl == 0) {
            if (i_lds) hl_dirty[lds_row_off(irow) / d] = 1u;
            if (j_lds) hl_dirty[lds_row_off(jrow) / d] = 1u;
          }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int f = e * G + gl;
          const float pe = pl[e];
          const float du = -lr * (-w * (qi[e] - qj[e]) + a.au * pe);
          dp[e] += du;
          pl[e] = pe + du;
          if (f < d) {
            const float gi = mi * (-w * pe + a.ai * qi[e]), gj = mj * (w * pe + a.an * qj[e]);
            if (i_lds) lds_add_f32(hl + lds_row_off(irow) + f, gi);
            else atomic_add_f32(row_at(a.Q, a.hot_delta, irow) + f, gi);
            if (j_lds) lds_add_f32(hl + lds_row_off(jrow) + f, gj);
            else atomic_add_f32(row_at(a.Q, a.hot_delta, jrow) + f, gj);
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
  if constexpr (LDSHOT) {
    // flush: the rows this workgroup touched go into the global delta block (replica blockIdx & rmask),
    // a group per row, full-line atomics
    __syncthreads();
    float* const dst = a.hot_delta + (uint32_t)(((int)blockIdx.x & a.hot_rmask) * a.hot_H) * (uint32_t)d;
    const int n_groups = (int)(blockDim.x / G);
    for (int k = (int)(threadIdx.x / G); k < a.lds_L; k += n_groups) {
      // (workgroups finish together: each starts its flush at another row)
      const int l = (int)(((uint32_t)k + blockIdx.x * 7u) % (uint32_t)a.lds_L);
      if (hl_dirty[l] == 0u) continue;
      float* const row = dst + (uint32_t)a.hot_by_rank[l] * (uint32_t)d;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int f = e * G + gl;
        if (f < d) atomic_add_f32(row + f, hl[l * d + f]);
      }
    }
  }
  if (stats) reduce_scalars(a.partials, s_loss, s_reg, s_abs, s_cnt, lane);
}

}  // namespace bpr
