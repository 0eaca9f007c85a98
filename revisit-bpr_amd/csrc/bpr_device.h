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
constexpr int UNIFORM_MAX_CAND = 4096;  // candidates tried before giving up (returns item 0)

// ---------------------------------------------------------------------------------------------
// group (sub-wave) collectives
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int G>
__device__ __forceinline__ int group_min(int v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
template <int G>
__device__ __forceinline__ int group_max(int v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}
// inclusive prefix sum over the lanes of a group
template <int G>
__device__ __forceinline__ float group_scan_incl(float v, int gl) {
#pragma unroll
  for (int off = 1; off < G; off <<= 1) {
    const float t = __shfl_up(v, off, G);
    if (gl >= off) v += t;
  }
  return v;
}
// ballot restricted to the caller's group, bit k = lane k of the group
template <int G>
__device__ __forceinline__ uint64_t group_ballot(bool pred, int lane) {
  const uint64_t full = __ballot(pred);
  if constexpr (G == 64) {
    return full;
  } else {
    return (full >> (lane & ~(G - 1))) & ((1ull << G) - 1ull);
  }
}
// value held by lane `src` of the caller's group
template <int G, typename T>
__device__ __forceinline__ T group_bcast(T v, int src, int lane) {
  return __shfl(v, (lane & ~(G - 1)) + src, 64);
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
struct SeenBitmap {  // one LDS read: the group's I-bit bitmap of the current user (k_stream)
  const uint32_t* bm;
  __device__ __forceinline__ bool operator()(int32_t c) const {
    return ((bm[c >> 5] >> (c & 31)) & 1u) != 0u;
  }
};

// the user's sorted seen list staged in LDS (k_stream, item tables too large for per-group
// bitmaps): ≈ log2(n_u) LDS reads; users with more than the staged capacity use the CSR in HBM
struct SeenList {
  const int32_t* lst;
  int32_t n;  // < 0: not staged, search the CSR slice
  const int32_t* __restrict__ indices;
  int64_t lo, hi;
  __device__ __forceinline__ bool operator()(int32_t c) const {
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
template <int G, typename Seen>
__device__ __forceinline__ int32_t sample_uniform(const Seen& seen, int64_t I, uint64_t seed,
                                                  uint64_t t, int lane) {
  const int gl = lane & (G - 1);
  int32_t result = 0;
  bool done = false;
  constexpr int ROUNDS = UNIFORM_MAX_CAND / G;
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t k = (uint32_t)(round * G + gl);
    const uint32_t r = draw(seed, t, k >> 2, PURPOSE_UNIFORM, (int)(k & 3u));
    const int32_t c = 1 + (int32_t)__umulhi(r, (uint32_t)(I - 1));
    const bool ok = !seen(c);
    const uint64_t m = group_ballot<G>(ok, lane);
    const int first = m ? (__ffsll((unsigned long long)m) - 1) : 0;
    const int32_t cand = group_bcast<G>(c, first, lane);
    if (!done && m) {
      result = cand;
      done = true;
    }
    if (__all(done)) break;
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
};

// item number `skip` (0-based) among the items of order_f that are not in seen(u) ∪ {0}, counting
// from the top (from_top) or from the bottom of the order.  WALK_UNROLL chunks of G order entries
// are fetched per trip so the (coalesced, independent) loads overlap.
#ifndef BPR_WALK_UNROLL
#define BPR_WALK_UNROLL 4
#endif
constexpr int WALK_UNROLL = BPR_WALK_UNROLL;

template <int G, typename Seen>
__device__ __forceinline__ int32_t adaptive_walk(const int32_t* __restrict__ order_f, int64_t I,
                                                 const Seen& seen, bool from_top, int32_t skip,
                                                 int lane) {
  const int gl = lane & (G - 1);
  int32_t result = 0;
  bool done = false;
  for (int64_t base = 0; base < I; base += (int64_t)G * WALK_UNROLL) {
    int32_t items[WALK_UNROLL];
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      const int64_t k = base + (int64_t)c * G + gl;
      const int64_t tpos = from_top ? k : (I - 1 - k);
      items[c] = (k < I) ? order_f[tpos] : 0;  // 0 = pad item = never a candidate
    }
#pragma unroll
    for (int c = 0; c < WALK_UNROLL; ++c) {
      const int32_t item = items[c];
      const bool unseen = item != 0 && !seen(item);
      const uint64_t m = group_ballot<G>(unseen, lane);
      const int cnt = __popcll((unsigned long long)m);
      const int below = __popcll((unsigned long long)(m & ((1ull << gl) - 1ull)));
      const bool mine = !done && unseen && (below == skip);
      const uint64_t hit = group_ballot<G>(mine, lane);
      const int src = hit ? (__ffsll((unsigned long long)hit) - 1) : 0;
      const int32_t got = group_bcast<G>(item, src, lane);
      if (!done) {
        if (hit) {
          result = got;
          done = true;
        } else {
          skip -= cnt;
        }
      }
    }
    if (__all(done)) break;
  }
  return result;
}

template <int G, int E, typename Seen>
__device__ __forceinline__ AdaptiveDraw sample_adaptive(
    const float (&p)[E], int d, const float (&sigma)[E],
    const int32_t* __restrict__ order, int64_t I, const Seen& seen, int64_t n_seen,
    float inv_log1mp, uint64_t seed, uint64_t t, int lane) {
  const int gl = lane & (G - 1);
  const u32x4 rnd = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), 0u, PURPOSE_ADAPTIVE,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
  // ---- factor ~ Categorical(|p_uf|·σ_f)  (neg_samplers.py:84-88): inverse CDF in factor order
  // f = e*G + gl, i.e. chunk e after chunk e-1, lanes in order inside a chunk
  float w[E], incl[E], carry[E];
  float total = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    w[e] = (f < d) ? fabsf(p[e]) * sigma[e] : 0.f;
    incl[e] = group_scan_incl<G>(w[e], gl);
    carry[e] = total;
    total += group_bcast<G>(incl[e], G - 1, lane);
  }
  const float uf = (float)(rnd.x >> 8) * (1.0f / 16777216.0f);
  const float thr = uf * total;
  int fsel = 0x7fffffff, flast = -1;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    if (w[e] > 0.f) {
      flast = max(flast, f);
      if (carry[e] + incl[e] > thr) fsel = min(fsel, f);
    }
  }
  fsel = group_min<G>(fsel);
  flast = group_max<G>(flast);
  if (fsel == 0x7fffffff) fsel = max(flast, 0);
  // ---- the user's factor value decides the orientation (:96-100)
  float pv = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (fsel == e * G + gl) pv = p[e];
  pv = group_sum<G>(pv);
  // ---- r ~ Geometric(p) on {1,2,…}, clamped to #unseen (:90-94)
  const int64_t n_unseen = (I - 1) - n_seen;
  const float ug = (float)((rnd.y >> 8) + 1u) * (1.0f / 16777216.0f);
  const float rr = ceilf(logf(ug) * inv_log1mp);
  int64_t r = rr < 1.0f ? 1 : (rr > 2.0e9f ? 2000000000ll : (int64_t)rr);
  if (r > n_unseen) r = n_unseen;
  const bool from_top = pv > 0.f;
  AdaptiveDraw out;
  out.factor = fsel;
  out.rank = (int32_t)(from_top ? r - 1 : n_unseen - r);
  out.item = (n_unseen > 0) ? adaptive_walk<G>(order + (int64_t)fsel * I, I, seen, from_top,
                                               (int32_t)(r - 1), lane)
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

}  // namespace bpr
