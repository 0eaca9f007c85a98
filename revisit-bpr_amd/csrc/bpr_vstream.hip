// bpr_vstream.hip — host side of the batched STREAM path (kernels: bpr_vstream.h).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <limits.h>

#include <algorithm>
#include <string>

#include "bpr_host.h"
#include "bpr_vstream.h"

namespace bpr {

static VTable table_P(const bpr_ctx* c) {
  VTable t;
  memset(&t, 0, sizeof(t));
  t.W = c->P; t.M = c->mP; t.V = c->vP; t.Gacc = c->vGP; t.H = c->vHP; t.rows = c->U;
  return t;
}
static VTable table_Q(const bpr_ctx* c) {
  VTable t;
  memset(&t, 0, sizeof(t));
  t.W = c->Q; t.M = c->mQ; t.V = c->vQ; t.Gacc = c->vGQ; t.H = c->vHQ; t.rows = c->I;
  if (c->bias != nullptr) {
    t.b = c->bias; t.mb = c->mb; t.vb = c->vb; t.Gb = c->vGb;
  }
  return t;
}

// fp32 view of the ctx's optimizer for the batched-stream kernels (bpr_vstream.h: VOpt)
static VOpt vopt(const bpr_ctx* c) {
  VOpt o;
  memset(&o, 0, sizeof(o));
  const bpr_opt_params& p = c->opt;
  o.lr = p.lr; o.mu = p.momentum; o.damp = p.dampening; o.nesterov = p.nesterov;
  o.b1 = p.beta1; o.b2 = p.beta2; o.eps = p.eps; o.alpha = p.alpha;
  auto ln = [](double x) { return x > 0.0 ? log(x) : -1.0e30; };
  auto l2 = [](double x) { return x > 0.0 ? log2(x) : -1.0e30; };
  o.ln_b1 = (float)ln(p.beta1); o.ln_b2 = (float)ln(p.beta2);
  o.log2_mu = (float)l2(p.momentum); o.log2_alpha = (float)l2(p.alpha);
  o.log2_b1 = (float)l2(p.beta1); o.log2_b2 = (float)l2(p.beta2);
  o.sqrt_b2 = (float)sqrt((double)p.beta2);
  o.mom_c = (float)((p.nesterov ? (double)p.momentum : 1.0) * (double)p.momentum /
                    (1.0 - (double)p.momentum));
  o.kmax = 0;
  o.t_sat = INT32_MAX;
  o.zc[0] = -1.f;
  if (c->opt_kind == BPR_OPT_ADAM) {
    const double b1 = p.beta1, b2 = p.beta2;
    // 1 - beta^t rounds to 1.0f once beta^t < 2^-25
    if (b2 > 0.0 && b2 < 1.0 && b1 < 1.0) {
      const double lim = log(ldexp(1.0, -25));
      const double t1 = b1 > 0.0 ? ceil(lim / log(b1)) : 1.0, t2 = ceil(lim / log(b2));
      const double ts = t1 > t2 ? t1 : t2;
      if (ts < 2.0e9) o.t_sat = (int32_t)ts;
    }
    if (b1 > 0.0) {
      const double ratio = b1 / sqrt(b2);
      o.kmax = ratio < 1.0 ? (int)ceil(log(1e-8) / log(ratio)) : 1 << 20;
      const bool no_closed = c->tune_adam_closed == 0;  // bpr_set_tuning("adam_closed", 0)
      const double zmax = b1 / pow(b2, 0.5 * VADAM_SERIES);
      o.closed_min = o.kmax >= 64 ? 16 : 3;
      if (!no_closed && zmax < 0.999 && o.kmax >= 4 && o.kmax < (1 << 20)) {
        for (int j = 0; j < VADAM_SERIES; ++j) {
          const double lz = log(b1) - 0.5 * (j + 1) * log(b2), z = exp(lz);
          o.zc[j] = (float)(z / (1.0 - z));
          o.log2_z[j] = (float)(lz / log(2.0));
        }
        o.sv_min = (float)((double)p.eps * pow(b2, -0.5 * o.kmax) / 0.02);
      }
    }
  }
  return o;
}

void vs_free(bpr_ctx* c) {
  hipFree(c->vGP); hipFree(c->vGQ); hipFree(c->vGb); hipFree(c->vHP); hipFree(c->vHQ);
  hipFree(c->v_alone);
  c->v_alone = nullptr;
  c->v_alone_cap = 0;
  c->vGP = c->vGQ = c->vGb = nullptr;
  c->vHP = c->vHQ = nullptr;
  c->vs_active = false;
}

static int vs_ensure(bpr_ctx* c) {
  if (c->vHP != nullptr) return BPR_OK;
  const size_t nP = (size_t)c->U * c->d, nQ = (size_t)c->I * c->d;
  BPR_HIP_CHECK(hipMalloc(&c->vGP, sizeof(float) * 2 * nP));
  BPR_HIP_CHECK(hipMalloc(&c->vGQ, sizeof(float) * 2 * nQ));
  BPR_HIP_CHECK(hipMalloc(&c->vGb, sizeof(float) * 2 * (size_t)c->I));
  BPR_HIP_CHECK(hipMalloc(&c->vHP, sizeof(uint64_t) * (size_t)c->U));
  BPR_HIP_CHECK(hipMalloc(&c->vHQ, sizeof(uint64_t) * (size_t)c->I));
  BPR_HIP_CHECK(hipMemsetAsync(c->vGP, 0, sizeof(float) * 2 * nP, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->vGQ, 0, sizeof(float) * 2 * nQ, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->vGb, 0, sizeof(float) * 2 * (size_t)c->I, c->stream));
  return BPR_OK;
}

// rows tracked by lastP / lastQ (STRICT) -> rows tracked by the headers
static int vs_enter(bpr_ctx* c) {
  if (c->vs_active) return BPR_OK;
  if (int rc = vs_ensure(c)) return rc;
  if (int rc = strict_flush_impl(c)) return rc;  // every row current as of c->step
  const uint64_t h = vhdr_pack(c->step, c->step, 0, 0);
  hipLaunchKernelGGL(k_vfill_hdr, dim3(256), dim3(256), 0, c->stream, c->vHP, c->U, h);
  hipLaunchKernelGGL(k_vfill_hdr, dim3(256), dim3(256), 0, c->stream, c->vHQ, c->I, h);
  BPR_HIP_CHECK(hipGetLastError());
  c->vs_active = true;
  return BPR_OK;
}

int vs_flush(bpr_ctx* c, bool users, bool items) {
  c->bias_w_valid = false;  // (the batched path writes the item_bias)
  if (!c->vs_active) return BPR_OK;
  const bool stateful = c->opt_kind != BPR_OPT_SGD;
  if (stateful)
    if (int rc = check_opt_state(c, "bpr_flush_lazy")) return rc;
  VFlushArgs a;
  memset(&a, 0, sizeof(a));
  a.d = c->d;
  a.now = c->step;
  a.o = vopt(c);
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int G = T::G, E = T::E;
    auto go = [&](const VTable& t, int pad) {
      a.T = t;
      a.pad = pad;
      const int64_t per_block = 256 / G;
      const unsigned grid = (unsigned)std::min<int64_t>((t.rows + per_block - 1) / per_block, 8192);
      switch (c->opt_kind) {
        case BPR_OPT_SGD:
          hipLaunchKernelGGL((k_vflush<G, E, OPT_SGD>), dim3(grid), dim3(256), 0, c->stream, a);
          break;
        case BPR_OPT_MOMENTUM:
          hipLaunchKernelGGL((k_vflush<G, E, OPT_MOMENTUM>), dim3(grid), dim3(256), 0, c->stream, a);
          break;
        case BPR_OPT_ADAM:
          hipLaunchKernelGGL((k_vflush<G, E, OPT_ADAM>), dim3(grid), dim3(256), 0, c->stream, a);
          break;
        default:
          hipLaunchKernelGGL((k_vflush<G, E, OPT_RMSPROP>), dim3(grid), dim3(256), 0, c->stream, a);
      }
    };
    if (users) go(table_P(c), c->pad_user);
    if (items) go(table_Q(c), c->pad_item);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

// headers -> lastP / lastQ: everything applied, every row current as of c->step
int vs_leave(bpr_ctx* c) {
  if (c->vs_active) c->bias_w_valid = false;
  if (!c->vs_active) return BPR_OK;
  if (int rc = vs_flush(c, true, true)) return rc;
  if (c->lastP != nullptr) {
    BPR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->lastP, (int)c->step, (size_t)c->U, c->stream));
    BPR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->lastQ, (int)c->step, (size_t)c->I, c->stream));
  }
  c->vs_active = false;
  return BPR_OK;
}

static int launch_vstream(bpr_ctx* c, VStreamArgs a, int sampler, int64_t cap_groups,
                          float* out_scalars) {
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int G = T::G, E = T::E;
    constexpr int GPW = 64 / G;
    unsigned block = 256;
    a.gpw_active = GPW;
    if (cap_groups > 0 && cap_groups * G < 256) {
      block = (unsigned)(((cap_groups * G + 63) / 64) * 64);
      if (cap_groups < GPW) a.gpw_active = (int)cap_groups;  // one wave, fewer groups at work
    }
    // "seen?" per triple (the stream is not grouped by user): the user's sorted seen list staged
    // in LDS (LIST_CAP entries, binary search in LDS; longer lists search the CSR in HBM).
    // bpr_set_tuning("seen", 1) stages nothing (every lookup searches the CSR: tests).
    const std::string force = c->tune_seen == 1 ? "csr" : "";  // bpr_set_tuning("seen", 1)
#ifndef VS_LIST_CAP
#define VS_LIST_CAP 512
#endif
    constexpr int LIST_CAP = VS_LIST_CAP;
    const int lds_words = (sampler != NEG_GIVEN && force != "csr") ? LIST_CAP : 0;
    // per group: the seen list + the triple's three rows as of t-1 (bpr_vstream.h)
    const size_t shmem = (size_t)(block / G) * (size_t)(lds_words + 3 * G * E) * sizeof(uint32_t);
    const int64_t per_block = (int64_t)(block / 64) * a.gpw_active;
    int64_t want = a.n;
    if (cap_groups > 0 && want > cap_groups) want = cap_groups;
    int64_t nblk = (want + per_block - 1) / per_block;
    const int64_t resident = 256 * 8;  // 256 CUs x up to 8 blocks: more only queues
    if (nblk > resident) nblk = resident;
    const unsigned grid = (unsigned)(nblk < 1 ? 1 : nblk);
    a.bm_words = lds_words;
    {
      Timer tm(c, true);
      (void)tm;
      auto go = [&](auto smp, auto knd) {
        constexpr int SMP = decltype(smp)::value, KND = decltype(knd)::value;
        constexpr int SN = SMP == NEG_GIVEN ? SEEN_CSR : SEEN_LIST;
        auto with_bias = [&](auto dir) {
          constexpr bool DIR = decltype(dir)::value;
          if (a.Q.b != nullptr)
            hipLaunchKernelGGL((k_vstream<G, E, SMP, SN, KND, DIR, true>), dim3(grid), dim3(block), shmem,
                               c->stream, a);
          else
            hipLaunchKernelGGL((k_vstream<G, E, SMP, SN, KND, DIR, false>), dim3(grid), dim3(block), shmem,
                               c->stream, a);
        };
        if (a.alone != nullptr) with_bias(std::integral_constant<bool, true>{});
        else with_bias(std::integral_constant<bool, false>{});
      };
      using std::integral_constant;
      auto with_kind = [&](auto smp) {
        switch (c->opt_kind) {
          case BPR_OPT_SGD: go(smp, integral_constant<int, OPT_SGD>{}); break;
          case BPR_OPT_MOMENTUM: go(smp, integral_constant<int, OPT_MOMENTUM>{}); break;
          case BPR_OPT_ADAM: go(smp, integral_constant<int, OPT_ADAM>{}); break;
          default: go(smp, integral_constant<int, OPT_RMSPROP>{});
        }
      };
      if (sampler == NEG_GIVEN) with_kind(integral_constant<int, NEG_GIVEN>{});
      else if (sampler == NEG_UNIFORM) with_kind(integral_constant<int, NEG_UNIFORM>{});
      else with_kind(integral_constant<int, NEG_ADAPTIVE>{});
    }
    if (out_scalars != nullptr)
      hipLaunchKernelGGL(k_vsum_partials, dim3(1), dim3(256), 0, c->stream, a.partials, (int)grid,
                         out_scalars);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

}  // namespace bpr

using namespace bpr;

extern "C" {

int bpr_train_stream_batched(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg,
                             int64_t n, int64_t B, int32_t sampler, float adaptive_p,
                             uint64_t seed, uint64_t offset, int64_t max_inflight,
                             float* out_scalars) {
  if (c != nullptr) c->keys_cut = false;
  if (int rc = check_triples(c, "bpr_train_stream_batched", users, pos, n)) return rc;
  if (int rc = check_sampler(c, "bpr_train_stream_batched", sampler, adaptive_p, neg, n)) return rc;
  if (B < 1) return fail(BPR_ERR_INVALID, "bpr_train_stream_batched: B must be >= 1");
  if (int rc = check_opt_state(c, "bpr_train_stream_batched")) return rc;
  if (c->pending != 0)
    return fail(BPR_ERR_INVALID, "bpr_train_stream_batched: unapplied STRICT gradients pending");
  if (n == 0) return BPR_OK;
  const int64_t steps = (n + B - 1) / B;
  if (n >= ((int64_t)1 << 31) || B >= ((int64_t)1 << 31) || c->step + steps >= VSTEP_MAX)
    return fail(BPR_ERR_UNSUPPORTED,
                "bpr_train_stream_batched: n, B and the optimizer step count must be < 2^31");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = vs_enter(c)) return rc;
  VStreamArgs a;
  memset(&a, 0, sizeof(a));
  a.P = table_P(c);
  a.Q = table_Q(c);
  a.indptr = c->indptr; a.indices = c->indices;
  if (int rc = snapshot_complete_impl(c)) return rc;  // (a partial snapshot left by the SGD path's split refresh)
  a.order = c->order; a.sigma = c->sigma;
  a.users = users; a.pos = pos; a.neg = neg;
  a.partials = out_scalars != nullptr ? c->dev_scalars : nullptr;
  a.seed = seed; a.offset = offset;
  a.t_base = c->step + 1;
  a.n = (int32_t)n; a.I = (int32_t)c->I; a.d = c->d; a.B = (int32_t)B;
  a.pad_user = c->pad_user; a.pad_item = c->pad_item;
  a.au = c->au; a.ai = c->ai; a.an = c->an;
  a.inv_log1mp = sampler == BPR_NEG_ADAPTIVE ? inv_log1mp(adaptive_p) : 0.f;
  a.iw = ItemWeights{c->w_accept, c->w_alias};
  a.o = vopt(c);
  // DIRECT launches: user rows are owned, and alone in their virtual batch they take their step at
  // once (bpr_vstream.h, vs_contribute): +20 / +5 / +6 % for SGD / momentum / RMSprop on the Yelp
  // shape, +2 % for Adam without item_bias.  Off for Adam WITH item_bias, whose kernel is at the
  // edge of its registers and loses 3 % to the extra branches (profiles/r04_vstream_direct.md).
  // bpr_set_tuning("vs_direct", 0 / 1) forces it (tests run both).
  const bool direct = c->tune_vs_direct >= 0 ? c->tune_vs_direct == 1
                                             : !(c->opt_kind == BPR_OPT_ADAM && a.Q.b != nullptr);
  if (B >= 16 && B <= VALONE_MAX_B && direct) {  // (tiny batches: one block per batch would not pay)
    if (c->v_alone_cap < n) {
      hipFree(c->v_alone);
      c->v_alone = nullptr;
      c->v_alone_cap = 0;
      BPR_HIP_CHECK(hipMalloc(&c->v_alone, (size_t)n));
      c->v_alone_cap = n;
    }
    int S = 64;
    while (S < 2 * B) S <<= 1;
    hipLaunchKernelGGL(k_valone, dim3((unsigned)steps), dim3(256), 2 * (size_t)S * sizeof(uint32_t), c->stream,
                       users, (int32_t)n, (int32_t)B, (int32_t)S, c->v_alone);
    a.alone = c->v_alone;
  }
  if (int rc = launch_vstream(c, a, sampler, max_inflight, out_scalars)) return rc;
  c->step += steps;
  return BPR_OK;
}

int bpr_shuffle_epoch(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n,
                      uint64_t seed, int32_t* users_out, int32_t* pos_out) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_shuffle_epoch: ctx is NULL");
  if (n < 0 || (n > 0 && (!users_in || !pos_in || !users_out || !pos_out)))
    return fail(BPR_ERR_INVALID, "bpr_shuffle_epoch: bad argument");
  if (users_in == users_out || pos_in == pos_out)
    return fail(BPR_ERR_INVALID, "bpr_shuffle_epoch: outputs must not alias the inputs");
  if (n == 0) return BPR_OK;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  int bits = 1;
  while ((((uint64_t)(n - 1)) >> bits) != 0) ++bits;
  const int half_bits = std::max(1, (bits + 1) / 2);
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_shuffle_scatter, dim3(grid), dim3(256), 0, c->stream, users_in, pos_in, n,
                     half_bits, seed, users_out, pos_out);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

}  // extern "C"
