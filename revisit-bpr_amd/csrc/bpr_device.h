// bpr_device.h — device-side building blocks of libbprcore (gfx950 / CDNA4, wave64).
//
// Work decomposition: a *group* of G lanes (G = 2..64, a power of two dividing the 64-lane wave)
// owns one BPR triple.  Lane gl of the group holds 16-byte slices [c*4G + 4gl, +4) of each of the
// three embedding rows (NV slices per row), so every row access is a run of consecutive 16-B
// lanes: one fully coalesced 64*16 B = 1 KiB wave transaction covers 64/G rows' slices.
// d=128 → G=32, NV=1 (two triples per wave); d=256 → G=64; d=1024 → G=64, NV=4.
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
  if constexpr (G == 64) return full;
  return (full >> (lane & ~(G - 1))) & ((1ull << G) - 1ull);
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

// ---------------------------------------------------------------------------------------------
// Uniform negative: UniformSampler.sample (reference revisit_bpr/modules/neg_samplers.py:31-37),
// i.e. uniform over items ∉ seen(u) ∪ {0}.  Candidate k of triple t is
//   c_k = 1 + mulhi32(philox(seed, t, k>>2, UNIFORM)[k&3], I-1);
// the result is the first accepted candidate.  A group tests G candidates per round.
// All lanes of the wave must call this together (wave-uniform loop).
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ int32_t sample_uniform(const int64_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ indices, int64_t I,
                                                  int32_t user, uint64_t seed, uint64_t t,
                                                  int lane) {
  const int gl = lane & (G - 1);
  const int64_t lo = indptr[user], hi = indptr[user + 1];
  int32_t result = 0;
  bool done = false;
  constexpr int ROUNDS = UNIFORM_MAX_CAND / G;
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t k = (uint32_t)(round * G + gl);
    const uint32_t r = draw(seed, t, k >> 2, PURPOSE_UNIFORM, (int)(k & 3u));
    const int32_t c = 1 + (int32_t)__umulhi(r, (uint32_t)(I - 1));
    const bool ok = !csr_contains(indices, lo, hi, c);
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
// p[] are the caller's register copies of the LIVE user row (slice layout of this file).
// ---------------------------------------------------------------------------------------------
struct AdaptiveDraw {
  int32_t factor;
  int32_t rank;  // 0-based from the top (neg_samplers.py:96-100)
  int32_t item;
};

// rank-th (0-based, from the top) item of order[f] that is not in seen(u) ∪ {0}; walks from the
// nearer end: `from_top` selects the direction, `skip` = unseen items to pass first.
template <int G>
__device__ __forceinline__ int32_t adaptive_walk(const int32_t* __restrict__ order_f, int64_t I,
                                                 const int32_t* __restrict__ indices, int64_t lo,
                                                 int64_t hi, bool from_top, int32_t skip,
                                                 int lane) {
  const int gl = lane & (G - 1);
  int32_t result = 0;
  bool done = false;
  for (int64_t base = 0; base < I; base += G) {
    const int64_t k = base + gl;
    const bool valid = k < I;
    const int64_t tpos = from_top ? k : (I - 1 - k);
    const int32_t item = valid ? order_f[tpos] : 0;
    const bool unseen = valid && item != 0 && !csr_contains(indices, lo, hi, item);
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
    if (__all(done)) break;
  }
  return result;
}

template <int G, int NV>
__device__ __forceinline__ AdaptiveDraw sample_adaptive(
    const float4 (&p)[NV], int d, const float* __restrict__ sigma,
    const int32_t* __restrict__ order, int64_t I, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, int32_t user, float inv_log1mp, uint64_t seed, uint64_t t,
    int lane) {
  const int gl = lane & (G - 1);
  const u32x4 rnd = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), 0u, PURPOSE_ADAPTIVE,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
  // ---- factor ~ Categorical(|p_uf|·σ_f)  (neg_samplers.py:84-88), inverse CDF in element order
  float w[NV][4];
  float incl[NV];  // inclusive scan (over lanes) of this lane's slice sum, per chunk
  float carry[NV]; // total weight of all earlier chunks
  float total = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int f0 = c * 4 * G + 4 * gl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f0 < d) s = *reinterpret_cast<const float4*>(sigma + f0);
    w[c][0] = fabsf(p[c].x) * s.x;
    w[c][1] = fabsf(p[c].y) * s.y;
    w[c][2] = fabsf(p[c].z) * s.z;
    w[c][3] = fabsf(p[c].w) * s.w;
    const float local = (w[c][0] + w[c][1]) + (w[c][2] + w[c][3]);
    incl[c] = group_scan_incl<G>(local, gl);
    carry[c] = total;
    total += group_bcast<G>(incl[c], G - 1, lane);
  }
  const float uf = (float)(rnd.x >> 8) * (1.0f / 16777216.0f);
  const float thr = uf * total;
  int fsel = 0x7fffffff, flast = -1;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    float cum = carry[c] + (incl[c] - ((w[c][0] + w[c][1]) + (w[c][2] + w[c][3])));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = c * 4 * G + 4 * gl + e;
      cum += w[c][e];
      if (w[c][e] > 0.f) {
        flast = max(flast, f);
        if (cum > thr) fsel = min(fsel, f);
      }
    }
  }
  fsel = group_min<G>(fsel);
  flast = group_max<G>(flast);
  if (fsel == 0x7fffffff) fsel = max(flast, 0);
  // ---- the user's factor value decides the orientation (:96-100)
  float pv = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int f0 = c * 4 * G + 4 * gl;
    if (fsel == f0 + 0) pv = p[c].x;
    if (fsel == f0 + 1) pv = p[c].y;
    if (fsel == f0 + 2) pv = p[c].z;
    if (fsel == f0 + 3) pv = p[c].w;
  }
  pv = group_sum<G>(pv);
  // ---- r ~ Geometric(p) on {1,2,…}, clamped to #unseen (:90-94)
  const int64_t lo = indptr[user], hi = indptr[user + 1];
  const int64_t n_unseen = (I - 1) - (hi - lo);
  const float ug = (float)((rnd.y >> 8) + 1u) * (1.0f / 16777216.0f);
  const float rr = ceilf(logf(ug) * inv_log1mp);
  int64_t r = rr < 1.0f ? 1 : (rr > 2.0e9f ? 2000000000ll : (int64_t)rr);
  if (r > n_unseen) r = n_unseen;
  const bool from_top = pv > 0.f;
  AdaptiveDraw out;
  out.factor = fsel;
  out.rank = (int32_t)(from_top ? r - 1 : n_unseen - r);
  out.item = (n_unseen > 0)
                 ? adaptive_walk<G>(order + (int64_t)fsel * I, I, indices, lo, hi, from_top,
                                    (int32_t)(r - 1), lane)
                 : 0;
  return out;
}

// ---------------------------------------------------------------------------------------------
// row slices
// ---------------------------------------------------------------------------------------------
template <int G, int NV>
__device__ __forceinline__ void load_row(float4 (&r)[NV], const float* __restrict__ row, int d,
                                         int gl) {
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int f0 = c * 4 * G + 4 * gl;
    r[c] = (f0 < d) ? *reinterpret_cast<const float4*>(row + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// fire-and-forget fp32 atomic add at device scope (global_atomic_add_f32, executed at L2 /
// memory side; no lost updates across XCDs).
__device__ __forceinline__ void atomic_add_f32(float* addr, float v) {
  __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G, int NV>
__device__ __forceinline__ void atomic_add_row(float* __restrict__ row, const float4 (&v)[NV],
                                               int d, int gl) {
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int f0 = c * 4 * G + 4 * gl;
    if (f0 < d) {
      atomic_add_f32(row + f0 + 0, v[c].x);
      atomic_add_f32(row + f0 + 1, v[c].y);
      atomic_add_f32(row + f0 + 2, v[c].z);
      atomic_add_f32(row + f0 + 3, v[c].w);
    }
  }
}

// −logσ(x) = softplus(−x)
__device__ __forceinline__ float neg_logsigmoid(float x) {
  return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
}

}  // namespace bpr
