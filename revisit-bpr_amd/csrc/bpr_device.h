// bpr_device.h — device-side building blocks of libbprcore (gfx950 / CDNA4, wave64).
//
// Work decomposition: a *group* of G lanes (G = 32 for d <= 128, else 64) owns one BPR triple.
// Lane gl of the group holds elements {e*G + gl : e < E} of each embedding row, so every row access
// — load OR fp32 atomic — is a run of G consecutive dwords per instruction: 128 B (one cache line)
// per group at G = 32, 256 B at G = 64.  That layout is dictated by the atomics: the chip retires
// device-scope fp32 atomics at a fixed rate of ~9-10 G cache-line requests/s regardless of how many
// lanes hit the line (measured, tools/ubench/gs_bench.hip: float4-per-lane rows → 4 partial-line
// requests per row slice, 188 M triples/s; dword-per-lane rows → full lines, 740 M triples/s).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bpr {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al. SC'11 — the algorithm behind rocRAND's
// philox4x32_10).  Keyed by (seed, global triple index), so a triple's draws do not depend on
// launch geometry, group width or GPU count.
// ---------------------------------------------------------------------------------------------
struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

// word `word` of block `block` of stream `purpose` for triple t
__device__ __forceinline__ uint32_t draw(uint64_t seed, uint64_t t, uint32_t block,
                                         uint32_t purpose, int word) {
  const u32x4 o = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), block, purpose, (uint32_t)seed,
                                (uint32_t)(seed >> 32));
  return word == 0 ? o.x : word == 1 ? o.y : word == 2 ? o.z : o.w;
}

enum : uint32_t { PURPOSE_UNIFORM = 0u, PURPOSE_ADAPTIVE = 1u };
enum { NEG_GIVEN = 0, NEG_UNIFORM = 1, NEG_ADAPTIVE = 2 };  // == bpr_sampler_kind
constexpr int UNIFORM_MAX_CAND = 4096;  // rejection candidates tried before the exact rank pick

// ---------------------------------------------------------------------------------------------
// group (sub-wave) collectives.  Sums and scans run on the VALU with DPP row operations
// (v_add_f32 + row_shr / row_bcast modifiers: 5-6 full-rate instructions, no LDS round trips);
// the kernels that use them are instruction-issue bound, and a ds_bpermute butterfly costs five
// dependent LDS-pipe latencies per sum.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v) {  // source lane's v, 0 where there is none
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
// inclusive prefix sum over the lanes of a group (lane order)
template <int G>
__device__ __forceinline__ float group_scan_incl(float v) {
  static_assert(G == 32 || G == 64, "groups are half or whole waves");
  v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8   — scan inside each row of 16
  v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 → rows 1, 3 add the total of rows 0, 2
  if constexpr (G == 64) v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 → rows 2, 3 add row 1
  return v;
}
// the value lane G-1 of the caller's group holds
template <int G>
__device__ __forceinline__ float group_last(float v, int lane) {
  const int x = __builtin_bit_cast(int, v);
  if constexpr (G == 64) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 63));
  } else {
    const int a = __builtin_amdgcn_readlane(x, 31), b = __builtin_amdgcn_readlane(x, 63);
    return __builtin_bit_cast(float, (lane & 32) ? b : a);
  }
}
template <int G>
__device__ __forceinline__ float group_sum(float v, int lane) {
  return group_last<G>(group_scan_incl<G>(v), lane);
}
// value held by lane `src` of the caller's group
template <int G, typename T>
__device__ __forceinline__ T group_bcast(T v, int src, int lane) {
  return __shfl(v, (lane & ~(G - 1)) + src, 64);
}
// Wave ballot as two scalar halves: the comparison mask itself, no per-lane work.  For G = 32 the
// low half belongs to the first group of the wave, the high half to the second.
struct Ballot {
  uint32_t lo, hi;
};
__device__ __forceinline__ Ballot wave_ballot(bool pred) {
  const uint64_t m = __builtin_amdgcn_ballot_w64(pred);
  return {(uint32_t)m, (uint32_t)(m >> 32)};
}
// lowest lane (0..G-1) of the caller's group whose predicate is set, -1 if none
template <int G>
__device__ __forceinline__ int group_first(const Ballot& b, int lane) {
  if constexpr (G == 64) {
    return b.lo ? (int)__builtin_ctz(b.lo) : (b.hi ? 32 + (int)__builtin_ctz(b.hi) : -1);
  } else {
    const int a = b.lo ? (int)__builtin_ctz(b.lo) : -1, c = b.hi ? (int)__builtin_ctz(b.hi) : -1;
    return (lane & 32) ? c : a;
  }
}
// highest such lane, -1 if none
template <int G>
__device__ __forceinline__ int group_last_set(const Ballot& b, int lane) {
  if constexpr (G == 64) {
    return b.hi ? 63 - (int)__builtin_clz(b.hi) : (b.lo ? 31 - (int)__builtin_clz(b.lo) : -1);
  } else {
    const int a = b.lo ? 31 - (int)__builtin_clz(b.lo) : -1;
    const int c = b.hi ? 31 - (int)__builtin_clz(b.hi) : -1;
    return (lane & 32) ? c : a;
  }
}
// v of the lowest lane of the caller's group whose predicate is set (v of the group's lane 0 if
// none): two v_readlane with scalar lane numbers instead of a ds_bpermute
template <int G>
__device__ __forceinline__ int32_t group_pick(const Ballot& b, int32_t v, int lane) {
  if constexpr (G == 64) {
    const int src = b.lo ? (int)__builtin_ctz(b.lo) : (b.hi ? 32 + (int)__builtin_ctz(b.hi) : 0);
    return __builtin_amdgcn_readlane(v, src);
  } else {
    const int32_t a = __builtin_amdgcn_readlane(v, b.lo ? (int)__builtin_ctz(b.lo) : 0);
    const int32_t c = __builtin_amdgcn_readlane(v, b.hi ? 32 + (int)__builtin_ctz(b.hi) : 32);
    return (lane & 32) ? c : a;
  }
}

// ---------------------------------------------------------------------------------------------
// seen-items CSR
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool csr_contains(const int32_t* __restrict__ indices, int64_t lo,
                                             int64_t hi, int32_t item) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int32_t v = indices[mid];
    if (v == item) return true;
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return false;
}

// "has user u seen item c?" — two interchangeable answers
struct SeenCsr {  // binary search in the user's sorted CSR slice (≈ log2(n_u) dependent loads)
  const int32_t* __restrict__ indices;
  int64_t lo, hi;
  __device__ __forceinline__ bool operator()(int32_t c) const {
    return csr_contains(indices, lo, hi, c);
  }
};
// Heavy users (more seen items than `heavy_T`): their I-bit seen bitmaps are precomputed ONCE per
// seen CSR in HBM (288 GB make that cheap: a few thousand users x I/8 bytes) — building the LDS
// structure costs a dependent HBM round trip per 128 seen items at every user change, and a user
// with thousands of them is rebuilt by every group that walks one of its runs.  hoff = word offset
// of the user's row in gbits, ~0u = light user (LDS structure).
constexpr uint32_t NOT_HEAVY = 0xFFFFFFFFu;
struct SeenBitmap {  // one LDS read: the group's I-bit bitmap of the current user (k_stream)
  const uint32_t* bm;
  const uint32_t* __restrict__ gbits;
  uint32_t hoff;
  __device__ __forceinline__ bool operator()(int32_t c) const {
    const uint32_t w = hoff != NOT_HEAVY ? gbits[hoff + (uint32_t)(c >> 5)] : bm[c >> 5];
    return ((w >> (c & 31)) & 1u) != 0u;
  }
};

// the user's sorted seen list staged in LDS (k_stream, item tables too large for per-group
// bitmaps): ≈ log2(n_u) LDS reads; users with more than the staged capacity use their HBM bitmap
// (heavy users) or search the CSR in HBM
struct SeenList {
  const int32_t* lst;
  int32_t n;  // < 0: not staged, search the CSR slice
  const int32_t* __restrict__ indices;
  int64_t lo, hi;
  const uint32_t* __restrict__ gbits;
  uint32_t hoff;
  __device__ __forceinline__ bool operator()(int32_t c) const {
    if (hoff != NOT_HEAVY) return ((gbits[hoff + (uint32_t)(c >> 5)] >> (c & 31)) & 1u) != 0u;
    if (n < 0) return csr_contains(indices, lo, hi, c);
    int32_t b = 0, len = n;
    while (len > 0) {
      const int32_t half = len >> 1;
      const bool right = lst[b + half] < c;
      b = right ? b + half + 1 : b;
      len = right ? len - half - 1 : half;
    }
    return b < n && lst[b] == c;
  }
};
enum { SEEN_CSR = 0, SEEN_BITMAP = 1, SEEN_LIST = 2 };

// ---------------------------------------------------------------------------------------------
// Uniform negative: UniformSampler.sample (reference revisit_bpr/modules/neg_samplers.py:31-37),
// i.e. uniform over items ∉ seen(u) ∪ {0}.  Candidate k of triple t is
//   c_k = 1 + mulhi32(philox(seed, t, k>>2, UNIFORM)[k&3], I-1);
// the result is the first accepted candidate.  A group tests G candidates per round.
// All lanes of the wave must call this together (wave-uniform loop).
// ---------------------------------------------------------------------------------------------
// Candidate from one 32-bit draw r: column c = 1 + floor(r (I-1) / 2^32).  With item weights
// (BPRExperiment._static_sampling with count_i ** neg_sampling_alpha, experiments/bpr/exp.py:85-91,
// 282-293) a Walker alias table decides between the column and its alias on the FRACTIONAL part of
// r (I-1) / 2^32, so candidates are ~ w and one draw still makes one candidate.
struct ItemWeights {
  const float* __restrict__ accept;  // [I] (entry 0 unused); NULL = uniform
  const int32_t* __restrict__ alias;
};
__device__ __forceinline__ int32_t uniform_candidate(uint32_t r, int64_t I, const ItemWeights& w) {
  const uint32_t n = (uint32_t)(I - 1);
  const int32_t c = 1 + (int32_t)__umulhi(r, n);
  if (w.accept == nullptr) return c;
  const float frac = (float)(r * n) * (1.0f / 4294967296.0f);
  return frac < w.accept[c] ? c : w.alias[c];
}

// Item number r (0-based) among the items 1..I-1 that are NOT in the sorted list seen[0..n_seen):
// seen[k] - 1 - k items below seen[k] are unseen, so the answer is r + 1 + #{k : seen[k] - 1 - k <= r}
// (binary search: the predicate is monotone in k).
__device__ __forceinline__ int32_t nth_unseen(const int32_t* __restrict__ seen_ids, int64_t n_seen,
                                              int64_t r) {
  int64_t lo = 0, hi = n_seen;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)seen_ids[mid] - 1 - mid <= r) lo = mid + 1; else hi = mid;
  }
  return (int32_t)(r + 1 + lo);
}

// `seen_ids` / `n_seen`: the user's sorted CSR slice (ids >= 1).  When UNIFORM_MAX_CAND candidates
// in a row were all seen (a user who has seen nearly every item) the pick becomes exact: one more
// draw (the next Philox block) selects uniformly among the user's unseen items by rank, so the
// result is always a valid unseen item, as the reference's multinomial over the masked weights
// (neg_samplers.py:31-37); item 0 only when nothing is unseen (the reference would raise).
template <int G, typename Seen>
__device__ __forceinline__ int32_t sample_uniform(const Seen& seen, int64_t n_seen,
                                                  const int32_t* __restrict__ seen_ids, int64_t I,
                                                  uint64_t seed, uint64_t t, int lane,
                                                  const ItemWeights& iw = ItemWeights{nullptr, nullptr}) {
  const int gl = lane & (G - 1);
  int32_t result = 0;
  bool done = false;
  constexpr int ROUNDS = UNIFORM_MAX_CAND / G;
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t k = (uint32_t)(round * G + gl);
    const uint32_t r = draw(seed, t, k >> 2, PURPOSE_UNIFORM, (int)(k & 3u));
    const int32_t c = uniform_candidate(r, I, iw);
    const bool ok = !seen(c);
    const Ballot b = wave_ballot(ok);
    const int32_t cand = group_pick<G>(b, c, lane);
    if (!done && group_first<G>(b, lane) >= 0) {
      result = cand;
      done = true;
    }
    if (__all(done)) break;
  }
  if (!done) {
    const int64_t n_unseen = (I - 1) - n_seen;
    if (n_unseen > 0) {
      const uint32_t r = draw(seed, t, (uint32_t)(UNIFORM_MAX_CAND / 4), PURPOSE_UNIFORM, 0);
      result = nth_unseen(seen_ids, n_seen, (int64_t)__umulhi(r, (uint32_t)n_unseen));
    }
  }
  return result;
}

// ---------------------------------------------------------------------------------------------
// Adaptive negative: AdaptiveSampler.sample (neg_samplers.py:74-124).
//   order [d, I]  per-factor descending item order of the last snapshot (bpr_adaptive_refresh)
//   sigma [d]     per-factor unbiased std of the snapshot
// p[] are the caller's register copies of the LIVE user row, sigma[] of the snapshot's per-factor
// std (both in the element layout of this file: entry e is factor e*G + gl).
// ---------------------------------------------------------------------------------------------
struct AdaptiveDraw {
  int32_t factor;
  int32_t rank;  // 0-based from the top (neg_samplers.py:96-100)
  int32_t item;
  // partial snapshots (PART): the candidate sits in the bucketed middle of its column at walk entry `kres`
  // and the caller still has to finish inside its bin (adaptive_finish_in_bin) — the finish is the caller's
  // so that it can get its own live registers out of the way first
  int32_t kres;
  bool mid, from_top;
};

// item number `skip` (0-based) among the items of order_f that are not in seen(u) ∪ {0}, counting
// from the top (from_top) or from the bottom of the order.  WALK_UNROLL chunks of G order entries
// are fetched per trip so the (coalesced, independent) loads overlap.
#ifndef BPR_WALK_UNROLL
#define BPR_WALK_UNROLL 4
#endif
constexpr int WALK_UNROLL = BPR_WALK_UNROLL;

struct __attribute__((packed, aligned(4))) OrderVec {  // 4 consecutive order entries, any dword
  int32_t v[WALK_UNROLL];                               // alignment (columns start at f*I)
};
static_assert(WALK_UNROLL == 4, "the walk fetches one dwordx4 per lane and trip");

// One trip covers 4*G consecutive entries of the walk with ONE 16-byte load per lane (lane gl
// holds entries base + 4*gl + {0..3}; 512 contiguous bytes per group of 32).  `order` carries
// ORDER_PAD entries of slack on both ends, so the last vector of the first / last column may
// overhang; overhanging entries are masked to 0 (the pad item: never a candidate).
constexpr int ORDER_PAD = WALK_UNROLL;  // == BPR_ORDER_PAD (bpr_ctx.h), checked in bprcore.hip

// PARTIAL snapshots (bpr_refresh.hip k_sort_partial): only the two ends of a column are in exact order,
// [0, kt) and [I - kb, I); the middle is bucketed — bins in order, any order inside a bin, the first
// entry of a bin flagged in its top bit (item ids are < 2^30).  `keys_f` = the column of keys the
// snapshot was sorted from (NULL: an ordinary, fully sorted snapshot — kt, kb unused).
constexpr uint32_t ORDER_FLAG = 0x80000000u;
struct PartialColumn {
  const float* __restrict__ keys_f;  // NULL = the column is sorted whole
  int32_t kt, kb;
};

// The walk found its candidate at walk entry `kres` inside the bucketed middle: counting unseen entries
// is exact up to the candidate's BIN, the order inside the bin is not.  Fetch the 4*G entries around the
// candidate (a bin holds at most 64: it lies inside), find the bin's unseen entries, and take them out
// one by one in exact walk order — (key descending, id ascending) from the top, reversed from the bottom
// — until as many are out as the position order had put before the candidate: the next one is the answer.
// Kept lean on purpose (one bit mask per lane, one group-wide maximum per round): it is compiled into
// the hot kernel, whose register allocation it must not disturb.  Wave-uniform call; `mine` = this group
// needs it (the other group of the wave idles through).
template <int G, typename Seen>
__device__ __forceinline__ int32_t adaptive_finish_in_bin(const int32_t* __restrict__ order_f,
                                                          const float* __restrict__ keys_f, int32_t I,
                                                          const Seen& seen, bool from_top, int32_t kres,
                                                          bool mine, int32_t zlo, int32_t zhi,
                                                          int32_t fallback, int lane) {
  const int gl = lane & (G - 1);
  const bool second = G == 32 && (lane & 32) != 0;
  const int32_t w0 = mine ? max(kres - 2 * G, 0) : 0;
  const int32_t k0 = w0 + WALK_UNROLL * gl;
  const int32_t kc = min(k0, I - 1);
  const OrderVec vec = *reinterpret_cast<const OrderVec*>(order_f + (from_top ? kc : I - WALK_UNROLL - kc));
  uint32_t raw[WALK_UNROLL];
#pragma unroll
  for (int c = 0; c < WALK_UNROLL; ++c)
    raw[c] = (k0 + c < I) ? (uint32_t)(from_top ? vec.v[c] : vec.v[WALK_UNROLL - 1 - c]) : 0u;
  // bin number of my entries = flags counted up to them (walking from the top a flagged entry OPENS its
  // bin: inclusive count; from the bottom it CLOSES it: exclusive count)
  int32_t run;
  {
    uint32_t below = 0u, cnt_lo = 0u;
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      const Ballot b = wave_ballot((raw[c] & ORDER_FLAG) != 0u);
      below = __builtin_amdgcn_mbcnt_hi(b.hi, __builtin_amdgcn_mbcnt_lo(b.lo, below));
      cnt_lo += (uint32_t)__builtin_popcount(b.lo);
    }
    run = (int32_t)(below - (second ? cnt_lo : 0u));  // flags of the lower lanes of my group
  }
  const int32_t rel = kres - w0;  // the candidate's place in the window: lane rel / 4, slot rel % 4
  int32_t b_cand = 0;             // its bin number (valid in its lane)
  uint32_t in_bin = 0u;           // per lane: bit c = my entry c might be of the candidate's bin (by number)
  int32_t bnum[WALK_UNROLL];
#pragma unroll
  for (int c = 0; c < WALK_UNROLL; ++c) {
    const bool fl = (raw[c] & ORDER_FLAG) != 0u;
    if (from_top) run += fl ? 1 : 0;
    bnum[c] = run;
    if (!from_top) run += fl ? 1 : 0;
    b_cand = (rel & 3) == c ? bnum[c] : b_cand;
  }
  const int32_t bsel = group_bcast<G>(b_cand, mine ? (rel >> 2) : 0, lane);
  // alive = unseen entries of that bin (inside the bucketed zone); key / id of each for the order
  uint32_t alive = 0u;
  int32_t before = 0;  // of them, those the position order puts before the candidate
  float key[WALK_UNROLL];
#pragma unroll
  for (int c = 0; c < WALK_UNROLL; ++c) {
    const int32_t it = (int32_t)(raw[c] & ~ORDER_FLAG);
    const bool m = mine && bnum[c] == bsel && k0 + c >= zlo && k0 + c < zhi && it != 0;
    key[c] = m ? keys_f[it] + 0.f : 0.f;  // (+0: a -0 key orders as +0, as in the full sort's comparisons)
    const bool u = m && !seen(it);
    alive |= u ? (1u << c) : 0u;
    before += (u && k0 + c < kres) ? 1 : 0;
  }
  (void)in_bin;
  // s = unseen entries of the bin before the candidate (sum over the group's lanes)
  int32_t s = before;
#pragma unroll
  for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off, G);
  // take the bin's unseen entries out in walk order; the (s+1)-th is the answer
  int32_t result = fallback;
  const int32_t rounds = mine ? s : -1;
  int32_t rmax = rounds;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) rmax = max(rmax, __shfl_xor(rmax, off, 64));  // both groups of the wave
  for (int32_t round = 0; round <= rmax; ++round) {
    // my best alive entry: composite (orderable key, id) — larger = earlier in walk order
    uint32_t bh = 0u, bl = 0u;
    int bc = -1;
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      uint32_t kb = __float_as_uint(key[c]);
      kb = (kb & 0x80000000u) ? ~kb : (kb | 0x80000000u);  // ascending uint = ascending float
      const uint32_t id = raw[c] & ~ORDER_FLAG;
      const uint32_t h = from_top ? kb : ~kb, l = from_top ? ~id : id;
      const bool better = ((alive >> c) & 1u) && (bc < 0 || h > bh || (h == bh && l > bl));
      bh = better ? h : bh;
      bl = better ? l : bl;
      bc = better ? c : bc;
    }
    if (bc < 0) bh = bl = 0u;
    // the group's best (ids are unique: no ties between lanes)
    uint32_t gh = bh, gl_ = bl;
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
      const uint32_t oh = __shfl_xor(gh, off, G), ol = __shfl_xor(gl_, off, G);
      const bool take = oh > gh || (oh == gh && ol > gl_);
      gh = take ? oh : gh;
      gl_ = take ? ol : gl_;
    }
    const bool winner = bc >= 0 && bh == gh && bl == gl_ && round <= rounds;
    if (winner) alive &= ~(1u << bc);
    const uint32_t wid = from_top ? ~gl_ : gl_;  // the winner's item id, known to the whole group
    if (round == rounds) result = (int32_t)wid;
  }
  return result;
}

// PART: the snapshot may be partial (its own instantiations of the kernels that sample: an ordinary
// snapshot's walk carries none of this)
template <int G, typename Seen, bool PART = false>
__device__ __forceinline__ int32_t adaptive_walk(const int32_t* __restrict__ order_f, int64_t I64,
                                                 const Seen& seen, bool from_top, int32_t skip,
                                                 int lane, const PartialColumn pc = PartialColumn{nullptr, 0, 0},
                                                 int32_t* kres_out = nullptr, bool* mid_out = nullptr) {
  const int gl = lane & (G - 1);
  const int32_t I = (int32_t)I64;  // d*I < 2^31 (bpr_adaptive_refresh)
  int32_t result = 0, kres = 0;
  bool done = false;
  for (int32_t base = 0; base < I; base += G * WALK_UNROLL) {
    // entry k of the walk is order_f[from_top ? k : I-1-k]; k0 = my first entry
    const int32_t k0 = base + WALK_UNROLL * gl;
    const int32_t kc = min(k0, I - 1);  // lanes wholly past the end re-read the last vector
    const OrderVec vec = *reinterpret_cast<const OrderVec*>(
        order_f + (from_top ? kc : I - WALK_UNROLL - kc));
    int32_t items[WALK_UNROLL];
    bool is_seen[WALK_UNROLL], unseen[WALK_UNROLL];
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      const int32_t v = (int32_t)((uint32_t)(from_top ? vec.v[c] : vec.v[WALK_UNROLL - 1 - c]) & ~ORDER_FLAG);
      items[c] = (k0 + c < I) ? v : 0;
    }
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) is_seen[c] = seen(items[c]);  // 4 lookups in flight
    // unseen entries of the trip (run) and of lower lanes (lanes_below): scalar popcounts of the
    // ballots and v_mbcnt, corrected for the group's position in the wave
    uint32_t wave_below = 0u, cnt_lo = 0u, cnt_hi = 0u;
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      unseen[c] = (items[c] != 0) & !is_seen[c];
      const Ballot b = wave_ballot(unseen[c]);
      wave_below = __builtin_amdgcn_mbcnt_hi(b.hi, __builtin_amdgcn_mbcnt_lo(b.lo, wave_below));
      cnt_lo += (uint32_t)__builtin_popcount(b.lo);
      cnt_hi += (uint32_t)__builtin_popcount(b.hi);
    }
    const bool second = G == 32 && (lane & 32) != 0;
    const int32_t run = (int32_t)(G == 64 ? cnt_lo + cnt_hi : (second ? cnt_hi : cnt_lo));
    const int32_t lanes_below = (int32_t)(wave_below - (second ? cnt_lo : 0u));
    // unseen entries before mine in walk order = those of lower lanes + my own earlier ones
    const bool here = !done && skip < run;
    int32_t before = lanes_below, isel = 0, ksel = 0;
    bool mine = false;
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      const bool hit_c = unseen[c] && before == skip;
      isel = hit_c ? items[c] : isel;
      ksel = hit_c ? k0 + c : ksel;
      mine |= hit_c;
      before += unseen[c] ? 1 : 0;
    }
    const Ballot hb = wave_ballot(here && mine);
    const int32_t got = group_pick<G>(hb, isel, lane);
    result = here ? got : result;
    if constexpr (PART) {  // remember WHERE the candidate sits
      const int32_t gotk = group_pick<G>(hb, ksel, lane);
      kres = here ? gotk : kres;
    }
    skip = (done | here) ? skip : skip - run;
    done |= here;
    if (__all(done)) break;
  }
  if constexpr (PART) {
    // the bucketed middle in walk-entry coordinates
    const int32_t zlo = from_top ? pc.kt : pc.kb, zhi = I - (from_top ? pc.kb : pc.kt);
    *mid_out = done && kres >= zlo && kres < zhi;
    *kres_out = kres;
  }
  return result;
}

// The two draws of one adaptive negative that do not depend on the model: the uniform that picks
// the factor, and r ~ Geometric(p) on {1,2,…} clamped to #unseen (neg_samplers.py:90-94).  One
// Philox block per triple; k_stream evaluates it for all triples of a run at once (lane k = triple
// t0+k), k_sample per triple.
struct AdaptiveRandoms {
  float uf;   // in [0, 1)
  int32_t r;  // 1-based rank, <= n_unseen (0 when the user has nothing unseen)
};
__device__ __forceinline__ AdaptiveRandoms adaptive_randoms(uint64_t seed, uint64_t t,
                                                            float inv_log1mp, int64_t n_unseen) {
  const u32x4 rnd = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), 0u, PURPOSE_ADAPTIVE,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
  AdaptiveRandoms o;
  o.uf = (float)(rnd.x >> 8) * (1.0f / 16777216.0f);
  const float ug = (float)((rnd.y >> 8) + 1u) * (1.0f / 16777216.0f);
  const float rr = ceilf(logf(ug) * inv_log1mp);
  int64_t r = rr < 1.0f ? 1 : (rr > 2.0e9f ? 2000000000ll : (int64_t)rr);
  if (r > n_unseen) r = n_unseen;
  o.r = (int32_t)r;
  return o;
}

// `sigma` is any indexable holder of the snapshot's per-factor std in the element layout of this
// file (a register array, or an LDS view: k_stream keeps it in LDS to save registers).
// `meta` / `keysT`: a PARTIAL snapshot's per-column {kt, kb} and the key columns it was sorted from
// (k_stream only; NULL = fully sorted).
template <int G, int E, typename Seen, typename Sigma, bool PART = false>
__device__ __forceinline__ AdaptiveDraw sample_adaptive(
    const float (&p)[E], int d, const Sigma& sigma,
    const int32_t* __restrict__ order, int64_t I, const Seen& seen, int64_t n_seen,
    const AdaptiveRandoms& rnd, int lane, const int32_t* __restrict__ meta = nullptr,
    const float* __restrict__ keysT = nullptr) {
  const int gl = lane & (G - 1);
  // ---- factor ~ Categorical(|p_uf|·σ_f)  (neg_samplers.py:84-88) by inverse CDF, lane by lane:
  // the factors are enumerated in the order (lane, chunk) = (f mod G, f div G) — any enumeration
  // gives the same distribution, and this one needs ONE scan over the lanes' sums instead of a
  // scan per chunk (the per-chunk version was ~100 of the kernel's ~290 VALU instructions per
  // wave-iteration, and instruction issue is what the sampler costs on a chip that is otherwise
  // waiting for its atomics).  The oracle enumerates the same way.
  float w[E];
  float lane_sum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    w[e] = (f < d) ? fabsf(p[e]) * sigma[e] : 0.f;
    lane_sum += w[e];
  }
  const float incl = group_scan_incl<G>(lane_sum);
  const float total = group_last<G>(incl, lane);
  const float thr = rnd.uf * total;
  // the first lane whose inclusive sum exceeds thr holds the factor; inside it, the first chunk
  // whose running sum does.  thr rounded up to (or past) the total: the last factor with weight of
  // this enumeration.  (Static indices throughout: a dynamically indexed p[] would be moved to
  // scratch memory by the compiler.)
  const int lsel = group_first<G>(wave_ballot(lane_sum > 0.f && incl > thr), lane);
  const int llast = group_last_set<G>(wave_ballot(lane_sum > 0.f), lane);
  float cum = incl - lane_sum;
  int e_first = -1, e_last = 0;
  float p_first = p[0], p_last = p[0];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    cum += w[e];
    const bool pos = w[e] > 0.f;
    const bool take = e_first < 0 && pos && cum > thr;
    e_first = take ? e : e_first;
    p_first = take ? p[e] : p_first;
    e_last = pos ? e : e_last;
    p_last = pos ? p[e] : p_last;
  }
  const bool first_ok = lsel >= 0 && e_first >= 0;
  const int e_loc = first_ok ? e_first : e_last;
  const float p_loc = first_ok ? p_first : p_last;
  const int src = lsel >= 0 ? lsel : (llast >= 0 ? llast : 0);
  const int fsel = group_bcast<G>(e_loc, src, lane) * G + src;
  const float psel = group_bcast<G>(p_loc, src, lane);
  // ---- the user's factor value decides the orientation (:96-100)
  const float pv = psel;
  const int32_t n_unseen = (int32_t)((I - 1) - n_seen);
  const int32_t r = min(rnd.r, n_unseen);
  const bool from_top = pv > 0.f;
  AdaptiveDraw out;
  out.factor = fsel;
  out.rank = from_top ? r - 1 : n_unseen - r;
  PartialColumn pc{nullptr, 0, 0};
  if constexpr (PART) {
    pc.kt = meta[2 * fsel];
    pc.kb = meta[2 * fsel + 1];
    pc.keys_f = keysT + (int64_t)fsel * I;
  }
  out.kres = 0;
  out.mid = false;
  out.from_top = from_top;
  out.item = (n_unseen > 0) ? adaptive_walk<G, Seen, PART>(order + (int64_t)fsel * I, I, seen, from_top,
                                                           r - 1, lane, pc, &out.kres, &out.mid)
                            : 0;
  return out;
}

// ---------------------------------------------------------------------------------------------
// rows: lane gl holds elements e*G + gl
// ---------------------------------------------------------------------------------------------
template <int G, int E>
__device__ __forceinline__ void load_row(float (&r)[E], const float* __restrict__ row, int d,
                                         int gl) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    r[e] = (f < d) ? row[f] : 0.f;
  }
}

template <int G, int E>
__device__ __forceinline__ void store_row(float* __restrict__ row, const float (&r)[E], int d,
                                          int gl) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    if (f < d) row[f] = r[e];
  }
}

// fire-and-forget fp32 atomic add at device scope (global_atomic_add_f32; resolved below the
// per-XCD L2s, so updates from different XCDs to one row are never lost)
__device__ __forceinline__ void atomic_add_f32(float* addr, float v) {
  __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G, int E>
__device__ __forceinline__ void atomic_add_row(float* __restrict__ row, const float (&v)[E], int d,
                                               int gl) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    if (f < d) atomic_add_f32(row + f, v[e]);
  }
}

template <int E>
__device__ __forceinline__ float dot(const float (&a)[E], const float (&b)[E]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) s = fmaf(a[e], b[e], s);
  return s;
}

// −logσ(x) = softplus(−x)
__device__ __forceinline__ float neg_logsigmoid(float x) {
  return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
}

// Loss statistics: wave shuffle → LDS → ONE plain store of the block's partial sums.  (Adding them
// to the caller's 4 floats with atomics from every wave serialises ~10^4 same-line atomics per
// launch — measured 0.3 ms — so the final sum is a separate one-block kernel, k_sum_partials.)
__device__ __forceinline__ void reduce_scalars(float* partials, float s_loss, float s_reg,
                                               float s_abs, float s_cnt, int lane,
                                               bool accumulate = false) {
  __shared__ float red[16][4];  // up to 1,024 threads (k_stream LDSHOT)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_loss += __shfl_xor(s_loss, off, 64);
    s_reg += __shfl_xor(s_reg, off, 64);
    s_abs += __shfl_xor(s_abs, off, 64);
    s_cnt += __shfl_xor(s_cnt, off, 64);
  }
  const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) {
    red[wv][0] = s_loss;
    red[wv][1] = s_reg;
    red[wv][2] = s_abs;
    red[wv][3] = s_cnt;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int k = 0; k < nw; ++k) v += red[k][threadIdx.x];
    float* slot = partials + (int64_t)blockIdx.x * 4 + threadIdx.x;
    *slot = accumulate ? *slot + v : v;
  }
}

}  // namespace bpr
