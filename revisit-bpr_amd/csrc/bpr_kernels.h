// bpr_kernels.h — the kernels of libbprcore (gfx950).  See DESIGN.md for the roofline of each.
//
//   k_triples<G,NV,MODE,SAMPLER>   the hot path: [sample j] → gather p_u,q_i,q_j → x, σ(−x) →
//                                  MODE_FORWARD: logits + loss scalars only
//                                  MODE_GRAD   : + accumulate per-row gradients (STRICT phase A)
//                                  MODE_STREAM : + apply SGD in place with fp32 atomics
//   k_apply<G,NV>                  STRICT phase B: one optimizer step per touched row
//   k_discard<G,NV>, k_flush_lazy<G,NV>, samplers-only kernels.
#pragma once
#include "bpr_device.h"

namespace bpr {

enum { MODE_FORWARD = 0, MODE_GRAD = 1, MODE_STREAM = 2 };
enum { NEG_GIVEN = 0, NEG_UNIFORM = 1, NEG_ADAPTIVE = 2 };
enum { OPT_SGD = 0, OPT_MOMENTUM = 1, OPT_ADAM = 2, OPT_RMSPROP = 3 };

struct TripleArgs {
  float* P;
  float* Q;
  float* bias;
  int64_t I;
  int d;
  int pad_user, pad_item;
  float au, ai, an, lr;
  // sampling
  const int64_t* indptr;
  const int32_t* indices;
  const int32_t* order;
  const float* sigma;
  float inv_log1mp;
  uint64_t seed, offset;
  // triple stream
  const int32_t* users;
  const int32_t* pos;
  int32_t* neg;
  int64_t n;
  // outputs
  float* lpos;
  float* lneg;
  float* scalars;
  // STRICT accumulators
  float* GP;
  float* GQ;
  float* Gb;
  int32_t* flagP;
  int32_t* flagQ;
  uint32_t* touched;
  uint32_t* touched_cnt;
};

__device__ __forceinline__ void mark_touched(int32_t* flag, uint32_t row, uint32_t table,
                                             uint32_t* touched, uint32_t* cnt) {
  if (atomicExch(&flag[row], 1) == 0) {
    const uint32_t slot = atomicAdd(cnt, 1u);
    touched[slot] = row * 2u + table;
  }
}

template <int G, int NV, int MODE, int SAMPLER>
__global__ __launch_bounds__(256) void k_triples(const TripleArgs a) {
  constexpr int GPW = 64 / G;  // groups (triples) per wave
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gw = lane / G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int d = a.d;
  const bool stats = a.scalars != nullptr;
  float s_loss = 0.f, s_reg = 0.f, s_abs = 0.f, s_cnt = 0.f;

  for (int64_t base = wave * GPW; base < a.n; base += n_waves * GPW) {
    const int64_t t = base + gw;
    const bool act = t < a.n;
    const int64_t tt = act ? t : a.n - 1;
    const int32_t u = a.users[tt];
    const int32_t i = a.pos[tt];
    float* __restrict__ prow = a.P + (int64_t)u * d;
    float* __restrict__ irow = a.Q + (int64_t)i * d;
    float4 p[NV], qi[NV], qj[NV];
    load_row<G, NV>(p, prow, d, gl);
    load_row<G, NV>(qi, irow, d, gl);

    int32_t j;
    if constexpr (SAMPLER == NEG_GIVEN) {
      j = a.neg[tt];
    } else if constexpr (SAMPLER == NEG_UNIFORM) {
      j = sample_uniform<G>(a.indptr, a.indices, a.I, u, a.seed, a.offset + (uint64_t)tt, lane);
    } else {
      j = sample_adaptive<G, NV>(p, d, a.sigma, a.order, a.I, a.indptr, a.indices, u,
                                 a.inv_log1mp, a.seed, a.offset + (uint64_t)tt, lane).item;
    }
    if constexpr (SAMPLER != NEG_GIVEN) {
      if (a.neg != nullptr && act && gl == 0) a.neg[t] = j;
    }
    float* __restrict__ jrow = a.Q + (int64_t)j * d;
    load_row<G, NV>(qj, jrow, d, gl);

    // ---- MF.forward (model.py:131-145) and BPR logits (model.py:48-64)
    float dpi = 0.f, dpj = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      dpi += dot4(p[c], qi[c]);
      dpj += dot4(p[c], qj[c]);
    }
    dpi = group_sum<G>(dpi);
    dpj = group_sum<G>(dpj);
    float xp = dpi, xn = dpj;
    if (a.bias != nullptr) {
      xp += a.bias[i];
      xn += a.bias[j];
    }
    const float x = xp - xn;
    if (act && gl == 0) {
      if (a.lpos != nullptr) a.lpos[t] = xp;
      if (a.lneg != nullptr) a.lneg[t] = xn;
    }
    if (stats) {
      // Loss (loss.py:20) and Model.regularization (model.py:87-93)
      float np2 = 0.f, ni2 = 0.f, nj2 = 0.f;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        np2 += dot4(p[c], p[c]);
        ni2 += dot4(qi[c], qi[c]);
        nj2 += dot4(qj[c], qj[c]);
      }
      np2 = group_sum<G>(np2);
      ni2 = group_sum<G>(ni2);
      nj2 = group_sum<G>(nj2);
      if (act && gl == 0) {
        s_loss += neg_logsigmoid(x);
        s_reg += 0.5f * (a.ai * ni2 + a.an * nj2 + a.au * np2);
        s_abs += fabsf(x);
        s_cnt += 1.f;
      }
    }
    if constexpr (MODE == MODE_FORWARD) continue;

    // ---- pairwise gradient, w = σ(−x)
    const float w = 1.0f / (1.0f + expf(x));
    // gradient sign convention: MODE_GRAD accumulates +g, MODE_STREAM adds −lr·g
    const float sc = (MODE == MODE_STREAM) ? -a.lr : 1.0f;
    float4 gp[NV], gi[NV], gj[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      gp[c].x = sc * (-w * (qi[c].x - qj[c].x) + a.au * p[c].x);
      gp[c].y = sc * (-w * (qi[c].y - qj[c].y) + a.au * p[c].y);
      gp[c].z = sc * (-w * (qi[c].z - qj[c].z) + a.au * p[c].z);
      gp[c].w = sc * (-w * (qi[c].w - qj[c].w) + a.au * p[c].w);
      gi[c].x = sc * (-w * p[c].x + a.ai * qi[c].x);
      gi[c].y = sc * (-w * p[c].y + a.ai * qi[c].y);
      gi[c].z = sc * (-w * p[c].z + a.ai * qi[c].z);
      gi[c].w = sc * (-w * p[c].w + a.ai * qi[c].w);
      gj[c].x = sc * (w * p[c].x + a.an * qj[c].x);
      gj[c].y = sc * (w * p[c].y + a.an * qj[c].y);
      gj[c].z = sc * (w * p[c].z + a.an * qj[c].z);
      gj[c].w = sc * (w * p[c].w + a.an * qj[c].w);
    }
    if (!act) continue;
    if constexpr (MODE == MODE_STREAM) {
      if (u != a.pad_user) atomic_add_row<G, NV>(prow, gp, d, gl);
      if (i != a.pad_item) atomic_add_row<G, NV>(irow, gi, d, gl);
      if (j != a.pad_item) atomic_add_row<G, NV>(jrow, gj, d, gl);
      if (a.bias != nullptr && gl == 0) {
        atomic_add_f32(a.bias + i, a.lr * w);
        atomic_add_f32(a.bias + j, -a.lr * w);
      }
    } else {
      if (u != a.pad_user) atomic_add_row<G, NV>(a.GP + (int64_t)u * d, gp, d, gl);
      if (i != a.pad_item) atomic_add_row<G, NV>(a.GQ + (int64_t)i * d, gi, d, gl);
      if (j != a.pad_item) atomic_add_row<G, NV>(a.GQ + (int64_t)j * d, gj, d, gl);
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

  if (stats) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s_loss += __shfl_xor(s_loss, off, 64);
      s_reg += __shfl_xor(s_reg, off, 64);
      s_abs += __shfl_xor(s_abs, off, 64);
      s_cnt += __shfl_xor(s_cnt, off, 64);
    }
    if (lane == 0 && s_cnt > 0.f) {
      atomic_add_f32(a.scalars + 0, s_loss);
      atomic_add_f32(a.scalars + 1, s_reg);
      atomic_add_f32(a.scalars + 2, s_abs);
      atomic_add_f32(a.scalars + 3, s_cnt);
    }
  }
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
};

enum { SAMPLE_UNIFORM = 0, SAMPLE_ADAPTIVE = 1, SAMPLE_PICK = 2 };

template <int G, int NV, int WHAT>
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
    if constexpr (WHAT == SAMPLE_UNIFORM) {
      const int32_t j =
          sample_uniform<G>(a.indptr, a.indices, a.I, u, a.seed, a.offset + (uint64_t)tt, lane);
      if (act && gl == 0) a.neg[t] = j;
    } else if constexpr (WHAT == SAMPLE_ADAPTIVE) {
      float4 p[NV];
      load_row<G, NV>(p, a.P + (int64_t)u * a.d, a.d, gl);
      const AdaptiveDraw r =
          sample_adaptive<G, NV>(p, a.d, a.sigma, a.order, a.I, a.indptr, a.indices, u,
                                 a.inv_log1mp, a.seed, a.offset + (uint64_t)tt, lane);
      if (act && gl == 0) {
        a.neg[t] = r.item;
        if (a.factor_out != nullptr) a.factor_out[t] = r.factor;
        if (a.rank_out != nullptr) a.rank_out[t] = r.rank;
      }
    } else {
      const int32_t f = a.factor_in[tt];
      const int32_t rk = a.rank_in[tt];
      const int64_t lo = a.indptr[u], hi = a.indptr[u + 1];
      const int32_t j =
          adaptive_walk<G>(a.order + (int64_t)f * a.I, a.I, a.indices, lo, hi, true, rk, lane);
      if (act && gl == 0) a.neg[t] = j;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// STRICT phase B: torch.optim step on the touched rows, with lazy replay of the zero-gradient
// steps a dense torch optimizer would have applied to the row since it was last touched (H2).
// ---------------------------------------------------------------------------------------------
struct OptDev {
  int kind;
  float lr, mu, damp;
  int nesterov;
  float b1, b2, eps, alpha;
  int64_t t;  // 1-based number of the step being applied (flush: steps applied so far)
  double log_b1, log_b2, log_mu, log_alpha;
  int kmax;   // Adam replay truncation (terms beyond are < 1e-8 of the first)
};

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
  int64_t U, I;
  int d;
  int pad_user, pad_item;
  OptDev o;
};

// zero-gradient steps s0+1 … s0+k applied to one element
__device__ __forceinline__ void opt_replay(float& w, float& m, float& v, int64_t s0, int64_t k,
                                           const OptDev& o) {
  if (k <= 0) return;
  if (o.kind == OPT_MOMENTUM) {
    const float muk = (float)exp((double)k * o.log_mu);
    const float c = o.nesterov ? o.mu : 1.0f;
    w -= o.lr * c * m * o.mu * (1.0f - muk) / (1.0f - o.mu);
    m *= muk;
  } else if (o.kind == OPT_ADAM) {
    if (m != 0.f) {
      double b1p = exp((double)s0 * o.log_b1), b2p = exp((double)s0 * o.log_b2);
      float ms = m, vs = v;
      const int64_t kk = k < (int64_t)o.kmax ? k : (int64_t)o.kmax;
      for (int64_t s = 0; s < kk; ++s) {
        b1p *= (double)o.b1;
        b2p *= (double)o.b2;
        ms *= o.b1;
        vs *= o.b2;
        const float step = (float)((double)o.lr / (1.0 - b1p));
        const float denom = sqrtf(vs) / (float)sqrt(1.0 - b2p) + o.eps;
        w -= step * (ms / denom);
      }
    }
    m *= (float)exp((double)k * o.log_b1);
    v *= (float)exp((double)k * o.log_b2);
  } else if (o.kind == OPT_RMSPROP) {
    v *= (float)exp((double)k * o.log_alpha);
  }
}

// the step with gradient g (torch.optim single-tensor formulas, pinned via the oracle)
__device__ __forceinline__ void opt_update(float& w, float g, float& m, float& v, const OptDev& o,
                                           float adam_step, float adam_bc2_sqrt) {
  if (o.kind == OPT_SGD) {
    w = w - o.lr * g;
  } else if (o.kind == OPT_MOMENTUM) {
    const float buf = (o.t == 1) ? g : o.mu * m + (1.0f - o.damp) * g;
    m = buf;
    const float eff = o.nesterov ? g + o.mu * buf : buf;
    w = w - o.lr * eff;
  } else if (o.kind == OPT_ADAM) {
    const float wgt = 1.0f - o.b1;
    m = (wgt < 0.5f) ? m + wgt * (g - m) : g - (g - m) * (1.0f - wgt);
    v = o.b2 * v + (1.0f - o.b2) * g * g;
    const float denom = sqrtf(v) / adam_bc2_sqrt + o.eps;
    w = w - adam_step * (m / denom);
  } else {
    v = o.alpha * v + (1.0f - o.alpha) * g * g;
    const float avg = sqrtf(v) + o.eps;
    w = w - o.lr * (g / avg);
  }
}

#define BPR_FOR4(EXPR_X, EXPR_Y, EXPR_Z, EXPR_W) \
  { EXPR_X; EXPR_Y; EXPR_Z; EXPR_W; }

template <int G, int NV>
__global__ __launch_bounds__(256) void k_apply(const ApplyArgs a) {
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t cnt = *a.touched_cnt;
  if (grp >= (int64_t)cnt) return;
  const uint32_t e = a.touched[grp];
  const bool is_item = (e & 1u) != 0u;
  const int64_t row = (int64_t)(e >> 1);
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
  float adam_step = 0.f, adam_bc2_sqrt = 1.f;
  if (o.kind == OPT_ADAM) {
    adam_step = (float)((double)o.lr / (1.0 - exp((double)o.t * o.log_b1)));
    adam_bc2_sqrt = (float)sqrt(1.0 - exp((double)o.t * o.log_b2));
  }
  if (row != pad) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int f0 = c * 4 * G + 4 * gl;
      if (f0 >= d) continue;
      float4 w4 = *reinterpret_cast<float4*>(W + f0);
      const float4 g4 = *reinterpret_cast<float4*>(Gr + f0);
      float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = m4;
      if (M != nullptr) m4 = *reinterpret_cast<float4*>(M + row * d + f0);
      if (V != nullptr) v4 = *reinterpret_cast<float4*>(V + row * d + f0);
      BPR_FOR4(opt_replay(w4.x, m4.x, v4.x, s0, k, o), opt_replay(w4.y, m4.y, v4.y, s0, k, o),
               opt_replay(w4.z, m4.z, v4.z, s0, k, o), opt_replay(w4.w, m4.w, v4.w, s0, k, o));
      BPR_FOR4(opt_update(w4.x, g4.x, m4.x, v4.x, o, adam_step, adam_bc2_sqrt),
               opt_update(w4.y, g4.y, m4.y, v4.y, o, adam_step, adam_bc2_sqrt),
               opt_update(w4.z, g4.z, m4.z, v4.z, o, adam_step, adam_bc2_sqrt),
               opt_update(w4.w, g4.w, m4.w, v4.w, o, adam_step, adam_bc2_sqrt));
      *reinterpret_cast<float4*>(W + f0) = w4;
      *reinterpret_cast<float4*>(Gr + f0) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (M != nullptr) *reinterpret_cast<float4*>(M + row * d + f0) = m4;
      if (V != nullptr) *reinterpret_cast<float4*>(V + row * d + f0) = v4;
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
template <int G, int NV>
__global__ __launch_bounds__(256) void k_discard(const ApplyArgs a) {
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (grp >= (int64_t)*a.touched_cnt) return;
  const uint32_t e = a.touched[grp];
  const bool is_item = (e & 1u) != 0u;
  const int64_t row = (int64_t)(e >> 1);
  float* Gr = (is_item ? a.GQ : a.GP) + row * a.d;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int f0 = c * 4 * G + 4 * gl;
    if (f0 < a.d) *reinterpret_cast<float4*>(Gr + f0) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (gl == 0) {
    if (is_item && a.Gb != nullptr) a.Gb[row] = 0.f;
    (is_item ? a.flagQ : a.flagP)[row] = 0;
  }
}

// bring every row of one table to step o.t (dense sweep; before eval / checkpoint / all-reduce)
template <int G, int NV>
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
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int f0 = c * 4 * G + 4 * gl;
      if (f0 >= d) continue;
      float4 w4 = *reinterpret_cast<float4*>(Wt + row * d + f0);
      float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = m4;
      if (M != nullptr) m4 = *reinterpret_cast<float4*>(M + row * d + f0);
      if (V != nullptr) v4 = *reinterpret_cast<float4*>(V + row * d + f0);
      BPR_FOR4(opt_replay(w4.x, m4.x, v4.x, s0, k, o), opt_replay(w4.y, m4.y, v4.y, s0, k, o),
               opt_replay(w4.z, m4.z, v4.z, s0, k, o), opt_replay(w4.w, m4.w, v4.w, s0, k, o));
      *reinterpret_cast<float4*>(Wt + row * d + f0) = w4;
      if (M != nullptr) *reinterpret_cast<float4*>(M + row * d + f0) = m4;
      if (V != nullptr) *reinterpret_cast<float4*>(V + row * d + f0) = v4;
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

}  // namespace bpr
