// bpr_vstream.h — BATCHED STREAM: the single-launch throughput path for EVERY torch.optim kind the
// reference configs use (SGD, SGD-momentum / Nesterov, Adam, RMSprop), gfx950.
//
// The reference steps once per mini-batch of B triples: all B gradients are evaluated at the
// pre-step parameters, duplicates are summed, and a DENSE optimizer then moves every row
// (experiments/trainer.py:64-83, example.py:172-180, torch.optim.*).  One launch here consumes a
// whole chunk of the shuffled triple stream; triple k belongs to the VIRTUAL step
// t = t_base + k / B.  Rows carry the bookkeeping that lets triples of many virtual steps be in
// flight at once without a grid barrier:
//
//   header H[row] (8 bytes, one atomic word):  last | gstep | slot | lock
//     (w, m, v)[row] are exact as of optimizer step `last`;
//     G[slot][row] accumulates the gradient of virtual step `gstep` (> last while one is pending);
//     it has not been applied yet.
//   view(row, t)   the row "as of step t-1", in registers: the pending step is applied on the fly
//                  when it is older than t, then the zero-gradient steps a dense optimizer would
//                  have taken since are replayed (bpr_opt.h) — nothing is written.
//   contribute(row, t, g)
//       t <= gstep : the row's pending step is this one (or a later one: a straggler joins it):
//                    atomic add into G[slot].
//       t >  gstep : this triple CLOSES the pending step: it takes the row's lock with one CAS
//                    (gstep <- t, slot flips, so the batch-mates that follow add into the other
//                    buffer right away), consumes G[old slot] with atomic exchanges, applies the
//                    optimizer step `gstep` to (w, m, v) — ONE step with the summed gradient, as
//                    torch does — publishes last = gstep and releases the lock.
//
// With ONE group walking the stream this is the reference's mini-batch algorithm exactly (every
// gradient of a batch sees the pre-step rows, one optimizer step per batch on the summed gradient,
// dense semantics through the replay); tests/test_gpu_vstream.py holds that limit to the oracle's
// dense torch.optim restatement for every optimizer.  With the chip full, triples of a window of
// consecutive virtual steps run concurrently: a gradient may be evaluated on rows that are a few
// steps stale and a straggler's gradient is counted one step late — bounded by the number of
// triples in flight (`max_inflight`), the same kind of asynchrony as the SGD STREAM kernel.
//
// Cross-CU visibility (MI355X_MICROARCH.md "inter-workgroup visibility"): per-CU L1s and per-XCD
// L2s are not coherent, so every access to shared row state is an agent-scope (sc1) load / store /
// atomic, and the closer drains its stores (s_waitcnt vmcnt(0)) before the sc1 header store that
// releases the lock.
#pragma once
#include <type_traits>

#include "bpr_device.h"
#include "bpr_opt.h"

namespace bpr {

// ---- row header -----------------------------------------------------------------------------
struct VHdr {
  uint64_t raw;
  __host__ __device__ int64_t last() const { return (int64_t)(raw >> 33); }
  __host__ __device__ int64_t gstep() const { return (int64_t)((raw >> 2) & 0x7fffffffull); }
  __host__ __device__ int slot() const { return (int)((raw >> 1) & 1ull); }
  __host__ __device__ bool locked() const { return (raw & 1ull) != 0ull; }
};
__host__ __device__ inline uint64_t vhdr_pack(int64_t last, int64_t gstep, int slot, int lock) {
  return ((uint64_t)last << 33) | ((uint64_t)gstep << 2) | ((uint64_t)slot << 1) | (uint64_t)lock;
}
constexpr int64_t VSTEP_MAX = 0x7fffffffll;  // 31-bit step fields

__device__ __forceinline__ uint64_t ld_hdr(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_hdr(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xchg_zero(float* p) {
  return __hip_atomic_exchange(p, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G, int E>
__device__ __forceinline__ void load_row_sc1(float (&r)[E], const float* __restrict__ row, int d,
                                             int gl) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    r[e] = (f < d) ? ld_sc1(row + f) : 0.f;
  }
}
template <int G, int E>
__device__ __forceinline__ void store_row_sc1(float* __restrict__ row, const float (&r)[E], int d,
                                              int gl) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    if (f < d) st_sc1(row + f, r[e]);
  }
}

// one embedding table with its optimizer state and batched-stream bookkeeping
struct VTable {
  float* W;
  float* M;     // momentum_buffer / exp_avg        (NULL when the optimizer has none)
  float* V;     // exp_avg_sq / square_avg          (NULL when the optimizer has none)
  float* Gacc;  // [2, rows, d] gradient accumulators
  uint64_t* H;  // [rows]
  int64_t rows;
  // per-row scalar riding on the row's header (item_bias; all NULL for the user table)
  float* b;
  float* mb;
  float* vb;
  float* Gb;  // [2, rows]
};

// one optimizer step `gs` with gradient g on a row held in registers, after the k = gs-1-a
// zero-gradient steps it missed since `a`
template <int E, bool STATEFUL>
__device__ __forceinline__ void vs_apply(float (&w)[E], float (&m)[E], float (&v)[E],
                                         const float (&g)[E], int64_t a, int64_t gs,
                                         const OptDev& o) {
  if constexpr (STATEFUL) opt_replay_row<E, true>(w, m, v, a, gs - 1 - a, o);
  float stp, bc2;
  adam_step_consts(o, gs, stp, bc2);
#pragma unroll
  for (int e = 0; e < E; ++e) opt_update_at(w[e], g[e], m[e], v[e], o, gs == 1, stp, bc2);
}
template <bool STATEFUL>
__device__ __forceinline__ void vs_apply1(float& w, float& m, float& v, float g, int64_t a,
                                          int64_t gs, const OptDev& o) {
  if constexpr (STATEFUL) opt_replay(w, m, v, a, gs - 1 - a, o);
  float stp, bc2;
  adam_step_consts(o, gs, stp, bc2);
  opt_update_at(w, g, m, v, o, gs == 1, stp, bc2);
}

// The row as of virtual step t-1 (nothing is written).  wb: the riding scalar's view (0 if none).
template <int G, int E, bool STATEFUL>
__device__ __forceinline__ void vs_view(float (&w)[E], float& wb, const VTable& T, uint32_t row,
                                        int d, int gl, int64_t t, const OptDev& o) {
  const VHdr h{ld_hdr(T.H + row)};
  const size_t off = (size_t)row * (size_t)d;
  load_row_sc1<G, E>(w, T.W + off, d, gl);
  float m[E], v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) m[e] = v[e] = 0.f;
  if constexpr (STATEFUL) {
    if (T.M != nullptr) load_row_sc1<G, E>(m, T.M + off, d, gl);
    if (T.V != nullptr) load_row_sc1<G, E>(v, T.V + off, d, gl);
  }
  float bm = 0.f, bv = 0.f;
  wb = 0.f;
  if (T.b != nullptr) {
    wb = ld_sc1(T.b + row);
    if constexpr (STATEFUL) {
      if (T.mb != nullptr) bm = ld_sc1(T.mb + row);
      if (T.vb != nullptr) bv = ld_sc1(T.vb + row);
    }
  }
  int64_t a = h.last();
  const int64_t gs = h.gstep();
  if (gs > a && gs < t) {  // a closed step nobody has applied yet: apply it in this view
    const size_t goff = ((size_t)h.slot() * (size_t)T.rows + row) * (size_t)d;
    float g[E];
    load_row_sc1<G, E>(g, T.Gacc + goff, d, gl);
    vs_apply<E, STATEFUL>(w, m, v, g, a, gs, o);
    if (T.b != nullptr) {
      const float gb = ld_sc1(T.Gb + (size_t)h.slot() * (size_t)T.rows + row);
      vs_apply1<STATEFUL>(wb, bm, bv, gb, a, gs, o);
    }
    a = gs;
  }
  if constexpr (STATEFUL) {
    const int64_t k = (t - 1) - a;
    opt_replay_row<E, false>(w, m, v, a, k, o);
    if (T.b != nullptr) opt_replay(wb, bm, bv, a, k, o);
  }
}

// Add this triple's gradient g (and gb for the riding scalar) to row's virtual step t.
template <int G, int E, bool STATEFUL>
__device__ __forceinline__ void vs_contribute(const VTable& T, uint32_t row, int d, int gl,
                                              int lane, int64_t t, const float (&g)[E], float gb,
                                              bool act, const OptDev& o) {
  const size_t off = (size_t)row * (size_t)d;
  bool done = !act;
  while (!__all(done)) {
    if (!done) {
      const VHdr h{ld_hdr(T.H + row)};
      const int64_t gs = h.gstep();
      if (t <= gs) {
        // the row's pending step is mine, or already a later one (I am a straggler: join it)
        float* acc = T.Gacc + ((size_t)h.slot() * (size_t)T.rows + row) * (size_t)d;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int f = e * G + gl;
          if (f < d) atomic_add_f32(acc + f, g[e]);
        }
        if (T.b != nullptr && gl == 0)
          atomic_add_f32(T.Gb + (size_t)h.slot() * (size_t)T.rows + row, gb);
        done = true;
      } else if (!h.locked()) {
        const int ns = h.slot() ^ 1;
        int won = 0;
        if (gl == 0) {
          uint64_t expect = h.raw;
          won = __hip_atomic_compare_exchange_strong(T.H + row, &expect,
                                                     vhdr_pack(h.last(), t, ns, 1),
                                                     __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)
                    ? 1 : 0;
        }
        won = group_bcast<G>(won, 0, lane);
        if (won) {
          const int64_t a = h.last();
          int64_t new_last = a;
          if (gs > a) {  // close step gs: ONE optimizer step with the summed gradient
            float* old = T.Gacc + ((size_t)h.slot() * (size_t)T.rows + row) * (size_t)d;
            float go[E], w[E], m[E], v[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const int f = e * G + gl;
              go[e] = (f < d) ? xchg_zero(old + f) : 0.f;
              m[e] = v[e] = 0.f;
            }
            load_row_sc1<G, E>(w, T.W + off, d, gl);
            if constexpr (STATEFUL) {
              if (T.M != nullptr) load_row_sc1<G, E>(m, T.M + off, d, gl);
              if (T.V != nullptr) load_row_sc1<G, E>(v, T.V + off, d, gl);
            }
            vs_apply<E, STATEFUL>(w, m, v, go, a, gs, o);
            store_row_sc1<G, E>(T.W + off, w, d, gl);
            if constexpr (STATEFUL) {
              if (T.M != nullptr) store_row_sc1<G, E>(T.M + off, m, d, gl);
              if (T.V != nullptr) store_row_sc1<G, E>(T.V + off, v, d, gl);
            }
            if (T.b != nullptr && gl == 0) {
              float wb = ld_sc1(T.b + row), bm = 0.f, bv = 0.f;
              if constexpr (STATEFUL) {
                if (T.mb != nullptr) bm = ld_sc1(T.mb + row);
                if (T.vb != nullptr) bv = ld_sc1(T.vb + row);
              }
              const float gbo = xchg_zero(T.Gb + (size_t)h.slot() * (size_t)T.rows + row);
              vs_apply1<STATEFUL>(wb, bm, bv, gbo, a, gs, o);
              st_sc1(T.b + row, wb);
              if constexpr (STATEFUL) {
                if (T.mb != nullptr) st_sc1(T.mb + row, bm);
                if (T.vb != nullptr) st_sc1(T.vb + row, bv);
              }
            }
            new_last = gs;
          }
          // open step t with my own gradient
          float* acc = T.Gacc + ((size_t)ns * (size_t)T.rows + row) * (size_t)d;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int f = e * G + gl;
            if (f < d) atomic_add_f32(acc + f, g[e]);
          }
          if (T.b != nullptr && gl == 0) atomic_add_f32(T.Gb + (size_t)ns * (size_t)T.rows + row, gb);
          // publish: every store above has left this CU before the header says so
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (gl == 0) st_hdr(T.H + row, vhdr_pack(new_last, t, ns, 0));
          done = true;
        }
      }
      // else: another group is closing this row — look again
    }
  }
}

// ---- LDS "seen" structures built per triple (the stream is NOT grouped by user here) -----------
template <int G>
__device__ __forceinline__ void seen_bitmap_build(uint32_t* bm, int W,
                                                  const int32_t* __restrict__ indices, int64_t lo,
                                                  int64_t hi, int gl) {
  uint4* bm4 = reinterpret_cast<uint4*>(bm);
  for (int k = gl; k < (W >> 2); k += G) bm4[k] = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t k = lo + gl; k < hi; k += 4 * G) {
    int32_t it[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t kk = k + q * G;
      it[q] = kk < hi ? indices[kk] : 0;  // bit 0 of word 0 = the pad item: harmless
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicOr(&bm[it[q] >> 5], 1u << (it[q] & 31));
  }
}
template <int G>
__device__ __forceinline__ int32_t seen_list_build(uint32_t* lst, int W,
                                                   const int32_t* __restrict__ indices, int64_t lo,
                                                   int64_t hi, int gl) {
  const int64_t cnt = hi - lo;
  const int32_t n = cnt <= (int64_t)W ? (int32_t)cnt : -1;  // -1: too long, search the CSR
  for (int32_t k = gl; k < n; k += G) lst[k] = (uint32_t)indices[lo + k];
  return n;
}

extern __shared__ __attribute__((aligned(16))) uint32_t bpr_vsmem[];

struct VStreamArgs {
  VTable P, Q;
  const int64_t* indptr;
  const int32_t* indices;
  const int32_t* order;
  const float* sigma;
  const int32_t* users;
  const int32_t* pos;
  int32_t* neg;
  float* partials;  // NULL = no statistics
  uint64_t seed, offset;
  int64_t t_base;   // virtual step of the chunk's first batch
  int32_t n, I, d, B;
  int32_t pad_user, pad_item;
  int32_t bm_words, gpw_active;
  float au, ai, an, inv_log1mp;
  OptDev o;
};

template <int G, int E, int SAMPLER, int SEEN, bool STATEFUL>
__global__ __launch_bounds__(256) void k_vstream(const VStreamArgs a) {
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int d = a.d;
  const int gpw = a.gpw_active;  // 1: only the first group of each wave works (sequential limit)
  const bool stats = a.partials != nullptr;
  float s_loss = 0.f, s_reg = 0.f, s_abs = 0.f, s_cnt = 0.f;
  uint32_t* lds = bpr_vsmem + (threadIdx.x / G) * a.bm_words;
  float sg[E];
  if constexpr (SAMPLER == NEG_ADAPTIVE) load_row<G, E>(sg, a.sigma, d, gl);

  for (int base = wave * gpw; base < a.n; base += n_waves * gpw) {
    const int k = base + gw;
    const bool act = gw < gpw && k < a.n;
    const int kk = act ? k : a.n - 1;
    const uint32_t u = (uint32_t)a.users[kk];
    const uint32_t i = (uint32_t)a.pos[kk];
    const int64_t t = a.t_base + (int64_t)(kk / a.B);
    float p[E], qi[E], qj[E];
    float bu_unused, bi, bj;
    vs_view<G, E, STATEFUL>(p, bu_unused, a.P, u, d, gl, t, a.o);
    vs_view<G, E, STATEFUL>(qi, bi, a.Q, i, d, gl, t, a.o);
    int32_t j;
    if constexpr (SAMPLER == NEG_GIVEN) {
      j = a.neg[kk];
    } else {
      const int64_t lo = a.indptr[u], hi = a.indptr[u + 1];
      using Seen = typename std::conditional<
          SEEN == SEEN_BITMAP, SeenBitmap,
          typename std::conditional<SEEN == SEEN_LIST, SeenList, SeenCsr>::type>::type;
      Seen seen;
      if constexpr (SEEN == SEEN_BITMAP) {
        seen_bitmap_build<G>(lds, a.bm_words, a.indices, lo, hi, gl);
        seen = SeenBitmap{lds};
      } else if constexpr (SEEN == SEEN_LIST) {
        const int32_t ln = seen_list_build<G>(lds, a.bm_words, a.indices, lo, hi, gl);
        seen = SeenList{reinterpret_cast<const int32_t*>(lds), ln, a.indices, lo, hi};
      } else {
        seen = SeenCsr{a.indices, lo, hi};
      }
      const uint64_t ctr = a.offset + (uint64_t)kk;
      if constexpr (SAMPLER == NEG_UNIFORM) {
        j = sample_uniform<G>(seen, a.I, a.seed, ctr, lane);
      } else {
        const AdaptiveRandoms rnd =
            adaptive_randoms(a.seed, ctr, a.inv_log1mp, (int64_t)(a.I - 1) - (hi - lo));
        j = sample_adaptive<G, E>(p, d, sg, a.order, a.I, seen, hi - lo, rnd, lane).item;
      }
      if (a.neg != nullptr && act && gl == 0) a.neg[k] = j;
    }
    vs_view<G, E, STATEFUL>(qj, bj, a.Q, (uint32_t)j, d, gl, t, a.o);

    // x_uij = <p_u, q_i - q_j> + b_i - b_j   (model.py:48-64, 131-145)
    float xl = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) xl = fmaf(p[e], qi[e] - qj[e], xl);
    const float x = group_sum<G>(xl, lane) + (bi - bj);
    if (stats && act) {
      s_reg += 0.5f * (a.ai * dot<E>(qi, qi) + a.an * dot<E>(qj, qj) + a.au * dot<E>(p, p));
      if (gl == 0) {
        s_loss += neg_logsigmoid(x);
        s_abs += fabsf(x);
        s_cnt += 1.f;
      }
    }
    // per-triple gradients, w = sigma(-x); summed per row and virtual step
    const float w = 1.0f / (1.0f + expf(x));
    float gu[E], gi[E], gj[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      gu[e] = -w * (qi[e] - qj[e]) + a.au * p[e];
      gi[e] = -w * p[e] + a.ai * qi[e];
      gj[e] = w * p[e] + a.an * qj[e];
    }
    // padding_idx drops the gradient of the pad EMBEDDING row (torch's embedding backward); the
    // pad item's bias is an ordinary parameter and keeps its gradient
    const bool pad_i = (int32_t)i == a.pad_item, pad_j = j == a.pad_item;
    if (pad_i | pad_j) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        gi[e] = pad_i ? 0.f : gi[e];
        gj[e] = pad_j ? 0.f : gj[e];
      }
    }
    const bool has_bias = a.Q.b != nullptr;
    vs_contribute<G, E, STATEFUL>(a.P, u, d, gl, lane, t, gu, 0.f,
                                  act && (int32_t)u != a.pad_user, a.o);
    vs_contribute<G, E, STATEFUL>(a.Q, i, d, gl, lane, t, gi, -w, act && (!pad_i || has_bias),
                                  a.o);
    vs_contribute<G, E, STATEFUL>(a.Q, (uint32_t)j, d, gl, lane, t, gj, w,
                                  act && (!pad_j || has_bias), a.o);
  }
  if (stats) reduce_scalars(a.partials, s_loss, s_reg, s_abs, s_cnt, lane);
}

// Bring every row of one table to step `now`: apply the pending step, replay the zero-gradient
// steps up to `now`, clear both accumulators.  Runs between launches (nothing else touches the
// table), dense sweep.  Rows with nothing pending that are already current are skipped; with a
// stateless optimizer (plain SGD) so are all rows with nothing pending.
struct VFlushArgs {
  VTable T;
  int32_t d, pad;
  int64_t now;
  OptDev o;
};
template <int G, int E, bool STATEFUL>
__global__ __launch_bounds__(256) void k_vflush(const VFlushArgs a) {
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) / G;
  const int d = a.d;
  const VTable& T = a.T;
  for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; row < T.rows;
       row += n_groups) {
    const VHdr h{T.H[row]};
    const int64_t a0 = h.last(), gs = h.gstep();
    const bool pending = gs > a0;
    if (!pending && (!STATEFUL || a0 >= a.now)) continue;
    const size_t off = (size_t)row * (size_t)d;
    float w[E], m[E], v[E];
    load_row<G, E>(w, T.W + off, d, gl);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = v[e] = 0.f;
    if constexpr (STATEFUL) {
      if (T.M != nullptr) load_row<G, E>(m, T.M + off, d, gl);
      if (T.V != nullptr) load_row<G, E>(v, T.V + off, d, gl);
    }
    float wb = 0.f, bm = 0.f, bv = 0.f;
    if (T.b != nullptr) {
      wb = T.b[row];
      if constexpr (STATEFUL) {
        if (T.mb != nullptr) bm = T.mb[row];
        if (T.vb != nullptr) bv = T.vb[row];
      }
    }
    int64_t cur = a0;
    if (pending) {
      // both buffers: a straggler's late add may sit in the other one
      float* g0 = T.Gacc + off;
      float* g1 = T.Gacc + (size_t)T.rows * (size_t)d + off;
      float g[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int f = e * G + gl;
        g[e] = (f < d) ? g0[f] + g1[f] : 0.f;
        if (f < d) {
          g0[f] = 0.f;
          g1[f] = 0.f;
        }
      }
      vs_apply<E, STATEFUL>(w, m, v, g, a0, gs, a.o);  // (the pad row's g is all zero)
      if (T.b != nullptr) {
        const float gb = T.Gb[row] + T.Gb[T.rows + row];
        if (gl == 0) {
          T.Gb[row] = 0.f;
          T.Gb[T.rows + row] = 0.f;
        }
        vs_apply1<STATEFUL>(wb, bm, bv, gb, a0, gs, a.o);
      }
      cur = gs;
    }
    if constexpr (STATEFUL) {
      opt_replay_row<E, true>(w, m, v, cur, a.now - cur, a.o);
      if (T.b != nullptr) opt_replay(wb, bm, bv, cur, a.now - cur, a.o);
    }
    store_row<G, E>(T.W + off, w, d, gl);
    if constexpr (STATEFUL) {
      if (T.M != nullptr) store_row<G, E>(T.M + off, m, d, gl);
      if (T.V != nullptr) store_row<G, E>(T.V + off, v, d, gl);
    }
    if (gl == 0) {
      if (T.b != nullptr) {
        T.b[row] = wb;
        if constexpr (STATEFUL) {
          if (T.mb != nullptr) T.mb[row] = bm;
          if (T.vb != nullptr) T.vb[row] = bv;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t nl = a.now > cur ? a.now : cur;
    if (gl == 0) T.H[row] = vhdr_pack(nl, nl, 0, 0);
  }
}

__global__ __launch_bounds__(256) void k_vfill_hdr(uint64_t* __restrict__ H, int64_t rows,
                                                   uint64_t value) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < rows;
       k += (int64_t)gridDim.x * blockDim.x)
    H[k] = value;
}

// final sum of the per-block loss statistics (added to the caller's four floats)
__global__ __launch_bounds__(256) void k_vsum_partials(const float* __restrict__ partials,
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

// seeded pseudo-random permutation of the triple list (DataLoader(shuffle=True) stand-in for the
// batched stream, whose virtual batches must be random subsets as the reference's are — NOT
// grouped by user: Adam normalises per step, so a user's triples must spread over the batches)
__device__ __forceinline__ uint32_t vmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__global__ void k_shuffle_scatter(const int32_t* __restrict__ users_in,
                                  const int32_t* __restrict__ pos_in, int64_t n, int half_bits,
                                  uint64_t seed, int32_t* __restrict__ users_out,
                                  int32_t* __restrict__ pos_out) {
  const uint32_t mask = (half_bits >= 32) ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)t;
    do {  // keyed 6-round Feistel network on [0, 4^half_bits), cycle-walked into [0, n)
      uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
      for (int round = 0; round < 6; ++round) {
        const uint32_t key = (uint32_t)(seed >> (16 * (round & 1))) +
                             0x9E3779B9u * (uint32_t)(round + 1) + (uint32_t)(seed >> 32);
        const uint32_t f = vmix32(r ^ key) & mask;
        const uint32_t nl = r;
        r = l ^ f;
        l = nl;
      }
      x = ((uint64_t)l << half_bits) | r;
    } while (x >= (uint64_t)n);
    users_out[x] = users_in[t];
    pos_out[x] = pos_in[t];
  }
}

}  // namespace bpr
