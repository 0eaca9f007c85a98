// bpr_opt.h — torch.optim update rules on registers (SGD / momentum / Adam / RMSprop) and the lazy
// replay of the zero-gradient steps a DENSE torch optimizer applies to rows nobody touched
// (SURVEY H2).  Shared by the STRICT kernels (bpr_kernels.h) and the batched STREAM kernels
// (bpr_vstream.h).  Pinned through the oracle to the reference's torch.optim trajectories.
#pragma once
#include "bpr_device.h"

namespace bpr {

enum { OPT_SGD = 0, OPT_MOMENTUM = 1, OPT_ADAM = 2, OPT_RMSPROP = 3 };

constexpr int ADAM_SERIES = 12;

struct OptDev {
  int kind;
  float lr, mu, damp;
  int nesterov;
  float b1, b2, eps, alpha;
  int64_t t;  // 1-based number of the step being applied (flush: steps applied so far)
  double log_b1, log_b2, log_mu, log_alpha;
  int kmax;   // Adam replay truncation (terms beyond are < 1e-8 of the first)
  // Adam replay in closed form: from step t_sat on both bias corrections are exactly 1 in fp32
  // (-1 = never / disabled); G = the geometric series of the eps expansion, for kk = kmax
  int64_t t_sat;
  float G[ADAM_SERIES];  // G(z_j) = z_j (1 - z_j^kmax) / (1 - z_j), z_j = b1 / b2^((j+1)/2)
  float adam_step, adam_bc2;  // lr / (1 - b1^t), sqrt(1 - b2^t) of the step being applied (host)
  float sv_min;          // closed form needs sqrt(v) >= eps * r^-kmax / 0.2 (series argument)
  float log2_z[ADAM_SERIES], zc[ADAM_SERIES];  // log2(z_j), z_j / (1 - z_j): G_j(k) for k < kmax
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
    if (o.mu > 0.f) {  // RMSprop(momentum): buf <- mu buf, w <- w - lr buf on every zero-gradient step
      const float muk = (float)exp((double)k * o.log_mu);
      w -= o.lr * m * o.mu * (1.0f - muk) / (1.0f - o.mu);
      m *= muk;
    }
    v *= (float)exp((double)k * o.log_alpha);
  }
}

// The same k zero-gradient steps for the E elements a lane holds of ONE row: the per-step scalars
// (bias corrections — double-precision exp / divide / sqrt) are computed once per step instead of
// once per element and step; element arithmetic is identical to opt_replay, bit for bit.
// STATE = false: only w is wanted (a forward pass reading the row "as of now"): the decayed
// moments — two double-precision exps per row — are not computed.
template <int E, bool STATE = true>
__device__ __forceinline__ void opt_replay_row(float (&w)[E], float (&m)[E], float (&v)[E],
                                               int64_t s0, int64_t k, const OptDev& o) {
  if (k <= 0) return;
  if (o.kind == OPT_MOMENTUM) {
    const float muk = (float)exp((double)k * o.log_mu);
    const float c = o.nesterov ? o.mu : 1.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      w[e] -= o.lr * c * m[e] * o.mu * (1.0f - muk) / (1.0f - o.mu);
      m[e] *= muk;
    }
  } else if (o.kind == OPT_ADAM) {
    bool any = false;
#pragma unroll
    for (int e = 0; e < E; ++e) any |= m[e] != 0.f;
    // Closed form for gaps of 16 steps and more once the bias corrections have saturated (1 - b^t == 1 in fp32 for every
    // replayed step).  The replayed movement is then
    //   lr * sum_{s=1..kmax} m b1^s / (sqrt(v) r^s + eps),  r = sqrt(b2)
    //   = lr (m / sqrt(v)) * sum q^s / (1 + e r^-s),         q = b1 / r,  e = eps / sqrt(v)
    //   = lr (m / sqrt(v)) * sum_j (-e)^j G(q / r^j),        G(z) = z (1 - z^kmax) / (1 - z)
    // with host-side constants G_j; twelve terms, e r^-kmax <= 0.2 required (truncation < 5e-9).
    // Lanes holding an element with a larger e, shorter gaps and the warm-up take the loop.
    bool closed = any && o.t_sat >= 0 && s0 >= o.t_sat && k >= 16;
    if (closed) {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (m[e] != 0.f) closed = closed && (sqrtf(v[e]) >= o.sv_min);
    }
    if (closed) {
      // G_j for this gap: the host's constants for k >= kmax, else z_j (1 - z_j^k) / (1 - z_j) with
      // z^k = 2^(k log2 z) in fp32 (|k log2 z| < 30: relative error ~1e-6, of the replayed movement)
      float G[ADAM_SERIES];
#pragma unroll
      for (int jj = 0; jj < ADAM_SERIES; ++jj)
        G[jj] = k >= (int64_t)o.kmax
                    ? o.G[jj]
                    : o.zc[jj] * (1.0f - __builtin_amdgcn_exp2f((float)k * o.log2_z[jj]));
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (m[e] != 0.f) {
          const float isv = 1.0f / sqrtf(v[e]);
          const float ne = -o.eps * isv;
          float acc = G[ADAM_SERIES - 1];
#pragma unroll
          for (int jj = ADAM_SERIES - 2; jj >= 0; --jj) acc = fmaf(acc, ne, G[jj]);
          w[e] -= o.lr * (m[e] * isv) * acc;
        }
      }
    } else if (any) {
      double b1p = exp((double)s0 * o.log_b1), b2p = exp((double)s0 * o.log_b2);
      // ms = m b1^s and sv = sqrt(v b2^s) = sqrt(v) sqrt(b2)^s are carried as float products
      float ms[E], sv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        ms[e] = m[e];
        sv[e] = sqrtf(v[e]);
      }
      const float sqrt_b2 = (float)sqrt((double)o.b2);
      const int kk = (int)(k < (int64_t)o.kmax ? k : (int64_t)o.kmax);
      // With the default beta1 = 0.9 a row untouched for a while replays 176 steps, and these
      // single-batch kernels run one wave per SIMD: the loop is bound by its instruction count and
      // by the quarter-rate transcendental unit.  The bias-correction powers are carried in double
      // (1 - beta^t cancels for small t); everything after that uses the hardware's 1-ulp v_rcp /
      // v_rsq instead of correctly rounded divide / sqrt sequences, and one multiply replaces the
      // square root per element and step (6 instead of ~45 instructions per element and step, one
      // transcendental instead of three).  The terms are independent given (m, v), so the error
      // does not compound beyond the float products' drift (~1e-6 of the replayed movement, itself
      // ~1e-2 of a weight: far inside the 2e-6 parity tolerance); the optimizer step itself
      // (opt_update) stays exactly rounded.
#pragma unroll 4
      for (int s = 0; s < kk; ++s) {
        b1p *= (double)o.b1;
        b2p *= (double)o.b2;
        const float step = o.lr * __builtin_amdgcn_rcpf((float)(1.0 - b1p));
        const float inv_bc2 = __builtin_amdgcn_rsqf((float)(1.0 - b2p));
#pragma unroll
        for (int e = 0; e < E; ++e) {
          ms[e] *= o.b1;
          sv[e] *= sqrt_b2;
          const float denom = fmaf(sv[e], inv_bc2, o.eps);
          if (m[e] != 0.f) w[e] -= step * (ms[e] * __builtin_amdgcn_rcpf(denom));
        }
      }
    }
    if constexpr (STATE) {
      const float mk = (float)exp((double)k * o.log_b1), vk = (float)exp((double)k * o.log_b2);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        m[e] *= mk;
        v[e] *= vk;
      }
    }
  } else if (o.kind == OPT_RMSPROP) {
    if (o.mu > 0.f) {  // momentum buffer keeps moving the row (torch.optim.RMSprop, momentum > 0)
      const float muk = (float)exp((double)k * o.log_mu);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        w[e] -= o.lr * m[e] * o.mu * (1.0f - muk) / (1.0f - o.mu);
        m[e] *= muk;
      }
    }
    if constexpr (STATE) {
      const float ak = (float)exp((double)k * o.log_alpha);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] *= ak;
    }
  }
}

// the step with gradient g (torch.optim single-tensor formulas, pinned via the oracle)
// `first`: this is optimizer step 1 (torch's SGD seeds the momentum buffer with the gradient)
__device__ __forceinline__ void opt_update_at(float& w, float g, float& m, float& v,
                                              const OptDev& o, bool first, float adam_step,
                                              float adam_bc2_sqrt) {
  if (o.kind == OPT_SGD) {
    w = w - o.lr * g;
  } else if (o.kind == OPT_MOMENTUM) {
    const float buf = first ? g : o.mu * m + (1.0f - o.damp) * g;
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
    if (o.mu > 0.f) {  // torch: buf.mul_(momentum).addcdiv_(grad, avg); param.add_(buf, alpha=-lr)
      m = o.mu * m + g / avg;
      w = w - o.lr * m;
    } else {
      w = w - o.lr * (g / avg);
    }
  }
}
__device__ __forceinline__ void opt_update(float& w, float g, float& m, float& v, const OptDev& o,
                                           float adam_step, float adam_bc2_sqrt) {
  opt_update_at(w, g, m, v, o, o.t == 1, adam_step, adam_bc2_sqrt);
}

// Adam's two per-step scalars for an arbitrary step t (the STRICT kernels get them from the host
// for the one step a launch applies): lr / (1 - b1^t) and sqrt(1 - b2^t), in double as torch does;
// both are exactly (lr, 1) in fp32 once t >= t_sat.
__device__ __forceinline__ void adam_step_consts(const OptDev& o, int64_t t, float& step,
                                                 float& bc2_sqrt) {
  if (o.kind != OPT_ADAM) {
    step = o.lr;
    bc2_sqrt = 1.f;
  } else if (o.t_sat >= 0 && t > o.t_sat) {
    step = o.lr;
    bc2_sqrt = 1.f;
  } else {
    step = (float)((double)o.lr / (1.0 - exp((double)t * o.log_b1)));
    bc2_sqrt = (float)sqrt(1.0 - exp((double)t * o.log_b2));
  }
}


// Bring the register copy of a row to "now" (all steps before o.t applied): a dense torch
// optimizer has been moving the row since it was last touched, and the forward pass must see
// those moves (the state tensors themselves are only rewritten by k_apply / k_flush_lazy).
template <int G, int E>
__device__ __forceinline__ void catch_up_row(float (&r)[E], const float* __restrict__ M,
                                             const float* __restrict__ V, int64_t row, int d,
                                             int gl, int64_t s0, int64_t k, const OptDev& o) {
  if (k <= 0) return;
  float m[E], v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int f = e * G + gl;
    m[e] = (M != nullptr && f < d) ? M[row * d + f] : 0.f;
    v[e] = (V != nullptr && f < d) ? V[row * d + f] : 0.f;
  }
  opt_replay_row<E, false>(r, m, v, s0, k, o);
}

}  // namespace bpr
