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
// r3: views are seqlock reads (vs_view), Adam's closed-form replay also serves short tails
// (closed_min).  Two restructurings were built, passed the sequential-limit tests and were dropped
// after measurement: closing AND opening at view time (fast, 277 M triples/s on
// configs[4], but it lets triples far ahead in the stream claim rows for their step and merges the
// batches in between into one optimizer step: -0.005 nDCG@100 with Adam), and closing at view time /
// opening when contributing (parity restored, but no faster than this flow for Adam and 15 % slower
// for the cheap optimizers).
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

// ---------------------------------------------------------------------------------------------
// Optimizer math of the batched stream: fp32 only, templated on the optimizer kind.
// (The STRICT kernels' replay — bpr_opt.h — carries its bias-correction powers in double and one
// code path for every kind; inlined into this kernel it took 200+ VGPRs, i.e. one or two waves
// per SIMD for a kernel whose throughput is the number of triples in flight.)
//
// Bias corrections y_t = 1 - beta^t: seeded with -expm1f(t ln beta) (accurate to ~1e-7 relative
// for every t, no cancellation) and advanced by y_{t+1} = y_t beta + (1 - beta), a contraction
// whose rounding errors do not accumulate.  Exactly 1 from step t_sat on.
// ---------------------------------------------------------------------------------------------
constexpr int VADAM_SERIES = 3;
struct VOpt {
  float lr, mu, damp;
  int32_t nesterov;
  float b1, b2, eps, alpha;
  float ln_b1, ln_b2;                    // natural logs (-1e30 for beta = 0)
  float log2_mu, log2_alpha, log2_b1, log2_b2;
  float sqrt_b2, mom_c;                  // mom_c = (nesterov ? mu : 1) mu / (1 - mu)
  int32_t kmax;                          // Adam: replayed terms beyond are < 1e-8 of the first
  int32_t t_sat;                         // Adam: 1 - beta^t == 1.0f for t >= t_sat (INT32_MAX: never)
  // Adam, gaps >= closed_min steps past t_sat, in closed form (~45 instructions per row slice; the
  // step loop costs ~6 per element and step, so the crossover is 16 steps for torch's default
  // beta1 = 0.9 — a 176-step tail — and 3 steps for the paper's beta1 = 0.1, whose whole tail is 8
  // steps: r2 ran the loop on EVERY row access of BASELINE configs[4]): the replayed movement is
  //   lr (m / sqrt v) sum_s q^s / (1 + e r^-s),  q = b1 / r, r = sqrt b2, e = eps / sqrt v
  //   = lr (m / sqrt v) sum_j (-e)^j G_j(k),     G_j(k) = z_j (1 - z_j^k) / (1 - z_j), z_j = b1 / r^(j+1)
  // three terms; needs e r^-k <= 0.02 (truncation < 1e-5 of the movement), else the step loop
  float zc[VADAM_SERIES], log2_z[VADAM_SERIES];  // z_j / (1 - z_j), log2 z_j   (zc[0] < 0: off)
  float sv_min;                                  // sqrt v >= eps r^-kmax / 0.02
  int32_t closed_min;                            // gaps of at least this many steps take the closed form
};

template <int KIND>
__device__ __forceinline__ void vo_step_consts(const VOpt& o, int64_t t, float& step, float& ibc2) {
  step = o.lr;
  ibc2 = 1.f;  // 1 / sqrt(1 - b2^t)
  if constexpr (KIND == OPT_ADAM) {
    if (t < (int64_t)o.t_sat) {
      const float tf = (float)t;
      step = o.lr / (-expm1f(tf * o.ln_b1));
      ibc2 = 1.0f / sqrtf(-expm1f(tf * o.ln_b2));
    }
  }
}

// the optimizer step proper (torch.optim single-tensor formulas, as bpr_opt.h: opt_update_at)
template <int KIND>
__device__ __forceinline__ void vo_update(float& w, float g, float& m, float& v, const VOpt& o,
                                          bool first, float adam_step, float adam_ibc2) {
  if constexpr (KIND == OPT_SGD) {
    w = w - o.lr * g;
  } else if constexpr (KIND == OPT_MOMENTUM) {
    const float buf = first ? g : o.mu * m + (1.0f - o.damp) * g;
    m = buf;
    const float eff = o.nesterov ? g + o.mu * buf : buf;
    w = w - o.lr * eff;
  } else if constexpr (KIND == OPT_ADAM) {
    const float wgt = 1.0f - o.b1;
    m = (wgt < 0.5f) ? m + wgt * (g - m) : g - (g - m) * (1.0f - wgt);
    v = o.b2 * v + (1.0f - o.b2) * g * g;
    // 1-ulp hardware sqrt / rcp (as the replay): the IEEE sequences cost ~20 instructions per
    // element and this kernel is bound by VALU issue (profiles/r02_sq_counters_vstream.txt)
    const float denom = __builtin_amdgcn_sqrtf(v) * adam_ibc2 + o.eps;
    w = w - adam_step * (m * __builtin_amdgcn_rcpf(denom));
  } else {
    v = o.alpha * v + (1.0f - o.alpha) * g * g;
    const float iavg = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + o.eps);
    if (o.mu > 0.f) {  // torch.optim.RMSprop(momentum > 0)
      m = o.mu * m + g * iavg;
      w = w - o.lr * m;
    } else {
      w = w - o.lr * (g * iavg);
    }
  }
}

// k zero-gradient steps s0+1 .. s0+k of a dense torch optimizer on the E elements a lane holds of
// one row (STATE = false: only w is wanted — a view)
template <int KIND, int E, bool STATE>
__device__ __forceinline__ void vo_replay(float (&w)[E], float (&m)[E], float (&v)[E], int64_t s0,
                                          int64_t k, const VOpt& o) {
  if (k <= 0) return;
  if constexpr (KIND == OPT_MOMENTUM) {
    const float muk = __builtin_amdgcn_exp2f(fmaxf((float)k * o.log2_mu, -126.f));
    const float c = o.lr * o.mom_c * (1.0f - muk);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      w[e] -= c * m[e];
      if constexpr (STATE) m[e] *= muk;
    }
  } else if constexpr (KIND == OPT_RMSPROP) {
    if (o.mu > 0.f) {  // the momentum buffer keeps moving the row
      const float muk = __builtin_amdgcn_exp2f(fmaxf((float)k * o.log2_mu, -126.f));
      const float c = o.lr * o.mom_c * (1.0f - muk);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        w[e] -= c * m[e];
        if constexpr (STATE) m[e] *= muk;
      }
    }
    if constexpr (STATE) {
      const float ak = __builtin_amdgcn_exp2f(fmaxf((float)k * o.log2_alpha, -126.f));
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] *= ak;
    }
  } else if constexpr (KIND == OPT_ADAM) {
    bool any = false;
#pragma unroll
    for (int e = 0; e < E; ++e) any |= m[e] != 0.f;
    const int kk = (int)(k < (int64_t)o.kmax ? k : (int64_t)o.kmax);
    bool closed = any && o.zc[0] >= 0.f && s0 >= (int64_t)o.t_sat && k >= (int64_t)o.closed_min;
    if (closed) {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (m[e] != 0.f) closed = closed && (v[e] >= o.sv_min * o.sv_min);
    }
    if (closed) {
      float Gs[VADAM_SERIES];
#pragma unroll
      for (int j = 0; j < VADAM_SERIES; ++j)
        Gs[j] = o.zc[j] * (1.0f - __builtin_amdgcn_exp2f(fmaxf((float)kk * o.log2_z[j], -126.f)));
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (m[e] != 0.f) {
          const float isv = __builtin_amdgcn_rsqf(v[e]);
          const float ne = -o.eps * isv;
          const float acc = fmaf(fmaf(Gs[2], ne, Gs[1]), ne, Gs[0]);
          w[e] -= o.lr * (m[e] * isv) * acc;
        }
      }
    } else if (any && kk > 0) {
      float y1 = 1.f, y2 = 1.f;  // 1 - b1^s, 1 - b2^s of the step being replayed
      const bool warm = s0 < (int64_t)o.t_sat;
      if (warm) {
        const float sf = (float)s0;
        y1 = -expm1f(sf * o.ln_b1);
        y2 = -expm1f(sf * o.ln_b2);
      }
      float ms[E], sv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        ms[e] = m[e];
        sv[e] = sqrtf(v[e]);
      }
      const float c1 = 1.0f - o.b1, c2 = 1.0f - o.b2;
      for (int s = 0; s < kk; ++s) {
        float step = o.lr, ibc2 = 1.f;
        if (warm) {
          y1 = fmaf(y1, o.b1, c1);
          y2 = fmaf(y2, o.b2, c2);
          step = o.lr * __builtin_amdgcn_rcpf(y1);
          ibc2 = __builtin_amdgcn_rsqf(y2);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          ms[e] *= o.b1;
          sv[e] *= o.sqrt_b2;
          const float denom = fmaf(sv[e], ibc2, o.eps);
          w[e] -= step * (ms[e] * __builtin_amdgcn_rcpf(denom));  // (m = 0: ms stays 0)
        }
      }
    }
    if constexpr (STATE) {
      const float mk = __builtin_amdgcn_exp2f(fmaxf((float)k * o.log2_b1, -126.f));
      const float vk = __builtin_amdgcn_exp2f(fmaxf((float)k * o.log2_b2, -126.f));
#pragma unroll
      for (int e = 0; e < E; ++e) {
        m[e] *= mk;
        v[e] *= vk;
      }
    }
  }
}

// one optimizer step `gs` with gradient g on a row held in registers, after the k = gs-1-a
// zero-gradient steps it missed since `a`
template <int KIND, int E>
__device__ __forceinline__ void vs_apply(float (&w)[E], float (&m)[E], float (&v)[E],
                                         const float (&g)[E], int64_t a, int64_t gs,
                                         const VOpt& o) {
  vo_replay<KIND, E, true>(w, m, v, a, gs - 1 - a, o);
  float stp, ibc2;
  vo_step_consts<KIND>(o, gs, stp, ibc2);
#pragma unroll
  for (int e = 0; e < E; ++e) vo_update<KIND>(w[e], g[e], m[e], v[e], o, gs == 1, stp, ibc2);
}
// the same for the scalar riding on an item row (item_bias), held by the group's lane 0
template <int KIND, bool STATE>
__device__ __forceinline__ void vo_replay1(float& w, float& m, float& v, int64_t s0, int64_t k,
                                           const VOpt& o) {
  float w1[1] = {w}, m1[1] = {m}, v1[1] = {v};
  vo_replay<KIND, 1, STATE>(w1, m1, v1, s0, k, o);
  w = w1[0];
  m = m1[0];
  v = v1[0];
}
template <int KIND>
__device__ __forceinline__ void vs_apply1(float& w, float& m, float& v, float g, int64_t a,
                                          int64_t gs, const VOpt& o) {
  vo_replay1<KIND, true>(w, m, v, a, gs - 1 - a, o);
  float stp, ibc2;
  vo_step_consts<KIND>(o, gs, stp, ibc2);
  vo_update<KIND>(w, g, m, v, o, gs == 1, stp, ibc2);
}

// The row as of virtual step t-1 (nothing is written).  wb: the riding scalar's view (0 if none).
// SEQLOCK read (r3): header, row words, header again.  A closer takes the row's lock (one
// CAS on the header), stores the row and publishes a new header; a view that read the header
// before that and the row words after it would apply the pending step a second time on top of a row
// that already has it (and may read a half-written row).  So the header is read again AFTER the
// row loads have completed; if it moved — or the row is locked — the view is taken again, at most
// VS_READ_TRIES times (then the possibly stale view is accepted: the asynchrony the full-chip path
// has anyway; the sequential limit never retries).
constexpr int VS_READ_TRIES = 6;
template <int G, int E, int KIND>
__device__ __forceinline__ uint64_t vs_view(float (&w)[E], float& wb, const VTable& T, uint32_t row,
                                            int d, int gl, int64_t t, const VOpt& o) {
  constexpr bool STATEFUL = KIND != OPT_SGD;
  const size_t off = (size_t)row * (size_t)d;
  float m[E], v[E], g[E];
  float bm = 0.f, bv = 0.f, gb = 0.f;
  VHdr h{0};
  bool pend = false;
  bool ok = false;
  int tries = 0;
  while (!__all(ok)) {
    if (!ok) {
      h.raw = ld_hdr(T.H + row);
      if (h.locked() && tries < VS_READ_TRIES) {
        ++tries;
        __builtin_amdgcn_s_sleep(2);  // a closer is storing this row
      } else {
        load_row_sc1<G, E>(w, T.W + off, d, gl);
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = v[e] = g[e] = 0.f;
        if constexpr (STATEFUL) {
          if (T.M != nullptr) load_row_sc1<G, E>(m, T.M + off, d, gl);
          if (T.V != nullptr) load_row_sc1<G, E>(v, T.V + off, d, gl);
        }
        bm = bv = gb = 0.f;
        wb = 0.f;
        if (T.b != nullptr) {
          wb = ld_sc1(T.b + row);
          if constexpr (STATEFUL) {
            if (T.mb != nullptr) bm = ld_sc1(T.mb + row);
            if (T.vb != nullptr) bv = ld_sc1(T.vb + row);
          }
        }
        pend = h.gstep() > h.last() && h.gstep() < t;  // a closed step nobody has applied yet
        if (pend) {
          const size_t goff = ((size_t)h.slot() * (size_t)T.rows + row) * (size_t)d;
          load_row_sc1<G, E>(g, T.Gacc + goff, d, gl);
          if (T.b != nullptr) gb = ld_sc1(T.Gb + (size_t)h.slot() * (size_t)T.rows + row);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-read must not overtake the row reads
        const uint64_t h2 = ld_hdr(T.H + row);
        if (h2 == h.raw || tries >= VS_READ_TRIES) ok = true;
        else ++tries;
      }
    }
  }
  int64_t a = h.last();
  const int64_t gs = h.gstep();
  // Two stretches of zero-gradient steps, the pending step between them: [a+1, gs-1], step gs with the
  // parked gradient, [gs+1, t-1].  ONE copy of the replay serves both (r5: the kernel waits for
  // instruction fetch — profiles/r04_vstream_direct.md — and the replay is its largest block; the
  // second stretch advances the view's private m, v as well, which nobody reads afterwards).
#pragma unroll 1
  for (int ph = pend ? 0 : 1; ph < 2; ++ph) {
    const int64_t k = ph == 0 ? gs - 1 - a : (t - 1) - a;
    if constexpr (STATEFUL) {
      vo_replay<KIND, E, true>(w, m, v, a, k, o);
      if (T.b != nullptr) vo_replay1<KIND, true>(wb, bm, bv, a, k, o);
    }
    if (ph == 0) {  // apply the pending step in this view
      float stp, ibc2;
      vo_step_consts<KIND>(o, gs, stp, ibc2);
#pragma unroll
      for (int e = 0; e < E; ++e) vo_update<KIND>(w[e], g[e], m[e], v[e], o, gs == 1, stp, ibc2);
      if (T.b != nullptr) vo_update<KIND>(wb, gb, bm, bv, o, gs == 1, stp, ibc2);
      a = gs;
    }
  }
  return h.raw;  // the header this view was taken under (vs_contribute's first guess)
}

// Add this triple's gradient g (and gb for the riding scalar) to row's virtual step t.
// `hint`: the header the row's view was taken under.  When it says "t closes the pending step" the
// CAS is tried on it directly (the CAS itself validates it: one round trip saved on the common
// path — a user row, a cold item row); a "join" is always decided on a fresh header, so that the
// add lands in the accumulator that is current NOW.
//
// r4 — OWNED rows (`owned`: the user table while the launch runs with k_valone's flags).  Every
// change to an owned row — joins too — happens under the row's lock, so no add can slip past the
// exchange that consumes an accumulator, and a row may be left with NOTHING pending:
//  * a triple whose row is ALONE in its virtual batch (`alone`: no other triple of batch t touches
//    it) takes its step at once — the pending step (if any), then step t, one load and one store
//    of (w, m, v) — instead of parking the gradient for the next visitor to exchange out and apply:
//    a third of the protocol's atomic bytes, and the next view finds the row current.  The
//    arithmetic is the deferred path's (the same vs_apply on the same operands): the sequential
//    limit is unchanged bit for bit;
//  * a straggler that arrives when the row's last step is already applied (t <= gstep == last)
//    applies its gradient there and then as one more optimizer step at index gstep (for SGD the
//    same as joining; with state, one extra decay — the deferred path counts a straggler one step
//    late instead);
//  * everything else as before (join a pending step, or close it and open step t), with the lock
//    taken by setting the lock bit alone.
// Item rows keep the lock-free join: they are never left applied inside a launch, and the flush
// sums both accumulators of a pending row, so a late add is counted, not lost.
template <int G, int E, int KIND>
__device__ __forceinline__ void vs_contribute(const VTable& T, uint32_t row, int d, int gl,
                                              int lane, int64_t t, const float (&g)[E], float gb,
                                              bool act, const VOpt& o, uint64_t hint, bool owned,
                                              bool alone) {
  constexpr bool STATEFUL = KIND != OPT_SGD;
  const size_t off = (size_t)row * (size_t)d;
  bool done = !act;
  bool guess = true;
  while (!__all(done)) {
    if (!done) {
      VHdr h{hint};
      if (!guess || t <= h.gstep() || h.locked()) h.raw = ld_hdr(T.H + row);
      guess = false;
      const int64_t gs = h.gstep();
      const bool late = t <= gs;
      if (late && !owned) {
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
        const bool applied = gs == h.last();  // nothing pending on this row
        const bool join = late && !applied;   // (owned) join the pending step, under the lock
        const bool direct = owned && (late ? applied : alone);
        const int64_t te = late ? gs : t;
        const int ns = (join || direct) ? h.slot() : (h.slot() ^ 1);
        int won = 0;
        if (gl == 0) {
          uint64_t expect = h.raw;
          won = __hip_atomic_compare_exchange_strong(
                    T.H + row, &expect, owned ? (h.raw | 1ull) : vhdr_pack(h.last(), te, ns, 1),
                    __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                    ? 1 : 0;
        }
        won = group_bcast<G>(won, 0, lane);
        if (won) {
          int64_t cur = h.last();
          const bool close = !late && gs > cur;  // a pending step older than mine
          if (close || direct) {
            // close step gs (ONE optimizer step with the summed gradient), then — direct — step te
            // with this triple's own gradient.  (A loop, not two inlined applies: the second copy
            // of the optimizer math cost 25 % of the launch — profiles/r04_vstream_direct.md.)
            float* old = T.Gacc + ((size_t)h.slot() * (size_t)T.rows + row) * (size_t)d;
            float go[E], w[E], m[E], v[E];
            float gbo = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const int f = e * G + gl;
              go[e] = (close && f < d) ? xchg_zero(old + f) : 0.f;
              m[e] = v[e] = 0.f;
            }
            load_row_sc1<G, E>(w, T.W + off, d, gl);
            if constexpr (STATEFUL) {
              if (T.M != nullptr) load_row_sc1<G, E>(m, T.M + off, d, gl);
              if (T.V != nullptr) load_row_sc1<G, E>(v, T.V + off, d, gl);
            }
            float wb = 0.f, bm = 0.f, bv = 0.f;
            const bool bias = T.b != nullptr && gl == 0;
            if (bias) {
              wb = ld_sc1(T.b + row);
              if constexpr (STATEFUL) {
                if (T.mb != nullptr) bm = ld_sc1(T.mb + row);
                if (T.vb != nullptr) bv = ld_sc1(T.vb + row);
              }
              if (close) gbo = xchg_zero(T.Gb + (size_t)h.slot() * (size_t)T.rows + row);
            }
#pragma unroll 1
            for (int ph = close ? 0 : 1; ph < (direct ? 2 : 1); ++ph) {
              const int64_t st = ph == 0 ? gs : te;
              if (ph == 1) {
#pragma unroll
                for (int e = 0; e < E; ++e) go[e] = g[e];
                gbo = gb;
              }
              vs_apply<KIND, E>(w, m, v, go, cur, st, o);
              if (bias) vs_apply1<KIND>(wb, bm, bv, gbo, cur, st, o);
              cur = st;
            }
            store_row_sc1<G, E>(T.W + off, w, d, gl);
            if constexpr (STATEFUL) {
              if (T.M != nullptr) store_row_sc1<G, E>(T.M + off, m, d, gl);
              if (T.V != nullptr) store_row_sc1<G, E>(T.V + off, v, d, gl);
            }
            if (bias) {
              st_sc1(T.b + row, wb);
              if constexpr (STATEFUL) {
                if (T.mb != nullptr) st_sc1(T.mb + row, bm);
                if (T.vb != nullptr) st_sc1(T.vb + row, bv);
              }
            }
          }
          if (!direct) {  // my gradient into step te's accumulator (joined, or opened just now)
            float* acc = T.Gacc + ((size_t)ns * (size_t)T.rows + row) * (size_t)d;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const int f = e * G + gl;
              if (f < d) atomic_add_f32(acc + f, g[e]);
            }
            if (T.b != nullptr && gl == 0) atomic_add_f32(T.Gb + (size_t)ns * (size_t)T.rows + row, gb);
          }
          // publish: every store above has left this CU before the header says so
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (gl == 0) st_hdr(T.H + row, vhdr_pack(cur, te, ns, 0));
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
  const uint8_t* alone;  // [n] 1 = the triple's user occurs once in its virtual batch (k_valone); DIRECT launches
  float* partials;  // NULL = no statistics
  uint64_t seed, offset;
  int64_t t_base;   // virtual step of the chunk's first batch
  int32_t n, I, d, B;
  int32_t pad_user, pad_item;
  int32_t bm_words, gpw_active;
  float au, ai, an, inv_log1mp;
  ItemWeights iw;
  VOpt o;
};

// rows of the triple as of step t-1 are staged in LDS ([3][G*E] floats per group: p_u, q_i, q_j),
// so that view() and contribute() each exist ONCE in the instruction stream (a loop over the
// three rows) instead of three times with all three rows live in registers around them: the
// optimizer replay inlined six times cost 256 VGPRs (one wave per SIMD); the kernel is bound by
// the latency of its dependent memory round trips, i.e. by how many triples are in flight.
#ifndef VS_BLOCKS_E4
#define VS_BLOCKS_E4 4  // blocks of 256 per CU at E <= 4 (= waves per SIMD)
#endif
// DIRECT: user rows are OWNED (vs_contribute) and take the steps of batches they are alone in at once.
// BIAS = false: a model without item_bias — the scalar copies of the optimizer math that serve the
// riding scalar are compiled out (a fifth of the Adam kernel's code, and code size is what this
// kernel is sensitive to: profiles/r04_vstream_direct.md).
template <int G, int E, int SAMPLER, int SEEN, int KIND, bool DIRECT, bool BIAS>
__global__ __launch_bounds__(256, (E <= 4 ? VS_BLOCKS_E4 : (E <= 8 ? 2 : 1))) void k_vstream(const VStreamArgs a) {
  constexpr int DP = G * E;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int d = a.d;
  const int gpw = a.gpw_active;  // 1: only the first group of each wave works (sequential limit)
  const bool stats = a.partials != nullptr;
  float s_loss = 0.f, s_reg = 0.f, s_abs = 0.f, s_cnt = 0.f;
  uint32_t* lds = bpr_vsmem + (threadIdx.x / G) * (a.bm_words + 3 * DP);
  float* rows = reinterpret_cast<float*>(lds + a.bm_words);  // [3][DP]

  auto table = [&](int r) {  // uniform selects: the user table for r == 0, else the item table
    VTable T;
    T.W = r == 0 ? a.P.W : a.Q.W;
    T.M = r == 0 ? a.P.M : a.Q.M;
    T.V = r == 0 ? a.P.V : a.Q.V;
    T.Gacc = r == 0 ? a.P.Gacc : a.Q.Gacc;
    T.H = r == 0 ? a.P.H : a.Q.H;
    T.rows = r == 0 ? a.P.rows : a.Q.rows;
    T.b = (!BIAS || r == 0) ? nullptr : a.Q.b;
    T.mb = (!BIAS || r == 0) ? nullptr : a.Q.mb;
    T.vb = (!BIAS || r == 0) ? nullptr : a.Q.vb;
    T.Gb = (!BIAS || r == 0) ? nullptr : a.Q.Gb;
    return T;
  };

  for (int base = wave * gpw; base < a.n; base += n_waves * gpw) {
    const int k = base + gw;
    const bool act = gw < gpw && k < a.n;
    const int kk = act ? k : a.n - 1;
    const uint32_t u = (uint32_t)a.users[kk];
    const uint32_t i = (uint32_t)a.pos[kk];
    const int64_t t = a.t_base + (int64_t)(kk / a.B);
    bool alone = false;
    if constexpr (DIRECT) alone = a.alone[kk] != 0;
    int32_t j = 0;
    float bi = 0.f, bj = 0.f;
    uint64_t hu = 0, hi_ = 0, hj = 0;  // headers the three views were taken under
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
      if (r == 2) {  // the negative: drawn from the user's row as of t-1, like the reference
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
            seen = SeenBitmap{lds, nullptr, NOT_HEAVY};
          } else if constexpr (SEEN == SEEN_LIST) {
            const int32_t ln = seen_list_build<G>(lds, a.bm_words, a.indices, lo, hi, gl);
            seen = SeenList{reinterpret_cast<const int32_t*>(lds), ln, a.indices, lo, hi, nullptr,
                            NOT_HEAVY};
          } else {
            seen = SeenCsr{a.indices, lo, hi};
          }
          const uint64_t ctr = a.offset + (uint64_t)kk;
          if constexpr (SAMPLER == NEG_UNIFORM) {
            j = sample_uniform<G>(seen, hi - lo, a.indices + lo, a.I, a.seed, ctr, lane, a.iw);
          } else {
            float p[E], sg[E];
#pragma unroll
            for (int e = 0; e < E; ++e) p[e] = rows[e * G + gl];
            load_row<G, E>(sg, a.sigma, d, gl);
            const AdaptiveRandoms rnd =
                adaptive_randoms(a.seed, ctr, a.inv_log1mp, (int64_t)(a.I - 1) - (hi - lo));
            j = sample_adaptive<G, E>(p, d, sg, a.order, a.I, seen, hi - lo, rnd, lane).item;
          }
          if (a.neg != nullptr && act && gl == 0) a.neg[k] = j;
        }
      }
      const uint32_t row = r == 0 ? u : (r == 1 ? i : (uint32_t)j);
      float w[E], wb;
      const uint64_t hr = vs_view<G, E, KIND>(w, wb, table(r), row, d, gl, t, a.o);
      hu = r == 0 ? hr : hu;
      hi_ = r == 1 ? hr : hi_;
      hj = r == 2 ? hr : hj;
#pragma unroll
      for (int e = 0; e < E; ++e) rows[r * DP + e * G + gl] = w[e];
      bi = r == 1 ? wb : bi;
      bj = r == 2 ? wb : bj;
    }
    // x_uij = <p_u, q_i - q_j> + b_i - b_j   (model.py:48-64, 131-145)
    float xl = 0.f, reg = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float pe = rows[e * G + gl], qie = rows[DP + e * G + gl], qje = rows[2 * DP + e * G + gl];
      xl = fmaf(pe, qie - qje, xl);
      reg += a.ai * qie * qie + a.an * qje * qje + a.au * pe * pe;
    }
    const float x = group_sum<G>(xl, lane) + (bi - bj);
    if (stats && act) {
      s_reg += 0.5f * reg;
      if (gl == 0) {
        s_loss += neg_logsigmoid(x);
        s_abs += fabsf(x);
        s_cnt += 1.f;
      }
    }
    // per-triple gradients, w = sigma(-x): row r gets c_p p + c_i q_i + c_j q_j.
    // padding_idx drops the gradient of the pad EMBEDDING row (torch's embedding backward); the
    // pad item's bias is an ordinary parameter and keeps its gradient.
    const float w = 1.0f / (1.0f + expf(x));
    const bool has_bias = BIAS && a.Q.b != nullptr;
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
      const uint32_t row = r == 0 ? u : (r == 1 ? i : (uint32_t)j);
      const bool pad = (int32_t)row == (r == 0 ? a.pad_user : a.pad_item);
      const float cp = pad ? 0.f : (r == 0 ? a.au : (r == 1 ? -w : w));
      const float ci = pad ? 0.f : (r == 0 ? -w : (r == 1 ? a.ai : 0.f));
      const float cj = pad ? 0.f : (r == 0 ? w : (r == 1 ? 0.f : a.an));
      float g[E];
#pragma unroll
      for (int e = 0; e < E; ++e)
        g[e] = cp * rows[e * G + gl] + ci * rows[DP + e * G + gl] + cj * rows[2 * DP + e * G + gl];
      const float gb = r == 0 ? 0.f : (r == 1 ? -w : w);
      vs_contribute<G, E, KIND>(table(r), row, d, gl, lane, t, g, gb,
                                act && (!pad || (r != 0 && has_bias)), a.o,
                                r == 0 ? hu : (r == 1 ? hi_ : hj), DIRECT && r == 0,
                                DIRECT && r == 0 && alone);
    }
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
  VOpt o;
};
template <int G, int E, int KIND>
__global__ __launch_bounds__(256) void k_vflush(const VFlushArgs a) {
  constexpr bool STATEFUL = KIND != OPT_SGD;
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
      vs_apply<KIND, E>(w, m, v, g, a0, gs, a.o);  // (the pad row's g is all zero)
      if (T.b != nullptr) {
        const float gb = T.Gb[row] + T.Gb[T.rows + row];
        if (gl == 0) {
          T.Gb[row] = 0.f;
          T.Gb[T.rows + row] = 0.f;
        }
        vs_apply1<KIND>(wb, bm, bv, gb, a0, gs, a.o);
      }
      cur = gs;
    }
    if constexpr (STATEFUL) {
      vo_replay<KIND, E, true>(w, m, v, cur, a.now - cur, a.o);
      if (T.b != nullptr) vo_replay1<KIND, true>(wb, bm, bv, cur, a.now - cur, a.o);
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
// alone[k] = 1 when users[k] occurs exactly once among the users of its virtual batch
// [b * B, (b + 1) * B) — one block per batch, an open-addressing set in LDS (S >= 2 B slots, a power
// of two: keys[S] then dup[S]).  B <= VALONE_MAX_B; the launch leaves the direct path off above that.
constexpr int VALONE_MAX_B = 2048;
__global__ __launch_bounds__(256) void k_valone(const int32_t* __restrict__ users, int32_t n, int32_t B,
                                                int32_t S, uint8_t* __restrict__ alone) {
  extern __shared__ uint32_t valone_lds[];
  uint32_t* keys = valone_lds;
  uint32_t* dup = valone_lds + S;
  const int64_t lo = (int64_t)blockIdx.x * B;
  const int32_t cnt = (int32_t)((lo + B <= n ? lo + B : (int64_t)n) - lo);
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    keys[s] = 0xFFFFFFFFu;
    dup[s] = 0u;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
    const uint32_t u = (uint32_t)users[lo + k];
    uint32_t s = ((u * 0x9E3779B1u) >> 7) & (uint32_t)(S - 1);
    while (true) {
      const uint32_t prev = atomicCAS(&keys[s], 0xFFFFFFFFu, u);
      if (prev == 0xFFFFFFFFu) break;
      if (prev == u) {
        dup[s] = 1u;
        break;
      }
      s = (s + 1) & (uint32_t)(S - 1);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
    const uint32_t u = (uint32_t)users[lo + k];
    uint32_t s = ((u * 0x9E3779B1u) >> 7) & (uint32_t)(S - 1);
    while (keys[s] != u) s = (s + 1) & (uint32_t)(S - 1);
    alone[lo + k] = dup[s] ? 0 : 1;
  }
}

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
