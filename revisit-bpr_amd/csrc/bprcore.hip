// bprcore.hip — C ABI of libbprcore.so (declared in include/bprcore.h) and kernel dispatch.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <type_traits>

#include "bpr_ctx.h"
#include "bpr_kernels.h"
#include "bpr_host.h"

namespace bpr {
static_assert(ORDER_PAD == BPR_ORDER_PAD, "walk vector width vs snapshot padding");

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

static int max_blocks() {
  static int v = [] {
    const char* e = getenv("BPR_MAX_BLOCKS");
    int n = e ? atoi(e) : 0;
    return n > 0 ? n : 256 * 8;  // 256 CUs x 8 resident 256-thread blocks
  }();
  return v;
}
// STREAM grids: ~1.5 runs per group and the hardware dispatcher balances the rest (measured:
// ML-20M 2,075 blocks 0.233 ms vs 0.24-0.25 for 1 or >= 2 runs per group; Yelp 10,922 blocks
// 1.22 ms vs 1.31 ms with a persistent 2,048-block grid).  BPR_MAX_BLOCKS caps it (experiments).
constexpr int64_t STREAM_MAX_GRID = 65536;  // also the size of the per-block partials scratch
static int64_t stream_grid_cap() {
  static const bool forced = getenv("BPR_MAX_BLOCKS") != nullptr;
  return forced ? (int64_t)max_blocks() : STREAM_MAX_GRID;
}

// grid for n groups of G lanes, 256-thread blocks, capped (grid-stride inside the kernels)
static unsigned grid_for(int64_t n_groups, int G, int64_t cap_groups) {
  if (cap_groups > 0 && n_groups > cap_groups) n_groups = cap_groups;
  const int64_t per_block = 256 / G;
  int64_t blocks = (n_groups + per_block - 1) / per_block;
  if (blocks > max_blocks()) blocks = max_blocks();
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

int check_bound(const bpr_ctx* c, const char* who, bool whole_table) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, std::string(who) + ": ctx is NULL");
  if (c->P == nullptr || c->Q == nullptr)
    return fail(BPR_ERR_INVALID, std::string(who) + ": tables not bound (bpr_bind_tables)");
  // every entry point sees the item table whole — but bpr_train_stream_acut itself and the calls of
  // its pipeline that never look at the table (commit; begin when the keys are already cut)
  if (whole_table && c->hot_unfolded && !c->acut_call) return hot_fold_impl(const_cast<bpr_ctx*>(c));
  return BPR_OK;
}

// Fold the hot deltas an asynchronous cut left in the block (after that cut has read them).
int hot_fold_impl(bpr_ctx* c) {
  if (!c->hot_unfolded) return BPR_OK;
  c->hot_unfolded = false;
  if (c->hot_H <= 0) return BPR_OK;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->ev_keys != nullptr) BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_keys, 0));
  c->acut_pending = false;
  EpilogueArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.Q = c->Q; ea.delta = c->hot_delta; ea.hot_items = c->hot_items;
  ea.H = c->hot_H; ea.R = c->hot_R; ea.d = c->d;
  ea.fold_blocks = (int)std::min<int64_t>(((int64_t)c->hot_H * c->d + 255) / 256, 64);
  hipLaunchKernelGGL(k_stream_epilogue, dim3(1 + ea.fold_blocks), dim3(256), 0, c->stream, ea);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;  // (Q + delta is what it was: keys cut from it stay valid)
}

// bpr_train_stream_cut under the hot tier leaves its loss partials to the bpr_sync_cut that should
// follow.  Whoever comes first instead — the next launch (it overwrites the partials), a hot exchange
// of the three-call protocol (bpr_hot_exchange / bpr_hot_sync), the end of the tier — sums them now.
int flush_deferred_stats(bpr_ctx* c) {
  if (!c->defer_pending) return BPR_OK;
  c->defer_pending = false;
  float* out = c->defer_out;
  c->defer_out = nullptr;
  if (out == nullptr || c->defer_blocks <= 0) return BPR_OK;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->dev_scalars, c->defer_blocks, out);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

static int ensure_strict_scratch(bpr_ctx* c) {
  if (c->GP != nullptr) return BPR_OK;
  const size_t nP = (size_t)c->U * c->d, nQ = (size_t)c->I * c->d;
  BPR_HIP_CHECK(hipMalloc(&c->GP, sizeof(float) * nP));
  BPR_HIP_CHECK(hipMalloc(&c->GQ, sizeof(float) * nQ));
  BPR_HIP_CHECK(hipMalloc(&c->Gb, sizeof(float) * c->I));
  BPR_HIP_CHECK(hipMalloc(&c->flagP, sizeof(int32_t) * c->U));
  BPR_HIP_CHECK(hipMalloc(&c->flagQ, sizeof(int32_t) * c->I));
  BPR_HIP_CHECK(hipMalloc(&c->lastP, sizeof(int32_t) * c->U));
  BPR_HIP_CHECK(hipMalloc(&c->lastQ, sizeof(int32_t) * c->I));
  BPR_HIP_CHECK(hipMalloc(&c->touched, sizeof(uint32_t) * (size_t)(c->U + c->I)));
  BPR_HIP_CHECK(hipMalloc(&c->touched_cnt, 2 * sizeof(uint32_t)));
  BPR_HIP_CHECK(hipMemsetAsync(c->GP, 0, sizeof(float) * nP, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->GQ, 0, sizeof(float) * nQ, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->Gb, 0, sizeof(float) * c->I, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->flagP, 0, sizeof(int32_t) * c->U, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->flagQ, 0, sizeof(int32_t) * c->I, c->stream));
  // rows are current as of the step counter (0, or a resumed checkpoint's: bpr_set_step)
  BPR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->lastP, (int)c->step, (size_t)c->U, c->stream));
  BPR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->lastQ, (int)c->step, (size_t)c->I, c->stream));
  BPR_HIP_CHECK(hipMemsetAsync(c->touched_cnt, 0, 2 * sizeof(uint32_t), c->stream));
  c->cnt_sel = 0;
  c->pending = 0;
  return BPR_OK;
}

static void free_strict_scratch(bpr_ctx* c) {
  hipFree(c->GP); hipFree(c->GQ); hipFree(c->Gb);
  hipFree(c->flagP); hipFree(c->flagQ); hipFree(c->lastP); hipFree(c->lastQ);
  hipFree(c->touched); hipFree(c->touched_cnt);
  c->GP = c->GQ = c->Gb = nullptr;
  c->flagP = c->flagQ = c->lastP = c->lastQ = nullptr;
  c->touched = c->touched_cnt = nullptr;
  c->pending = 0;
}


static TripleArgs triple_args(const bpr_ctx* c) {
  TripleArgs a;
  memset(&a, 0, sizeof(a));
  a.P = c->P; a.Q = c->Q; a.bias = c->bias;
  a.I = c->I; a.d = c->d;
  a.pad_user = c->pad_user; a.pad_item = c->pad_item;
  a.au = c->au; a.ai = c->ai; a.an = c->an;
  a.GP = c->GP; a.GQ = c->GQ; a.Gb = c->Gb;
  a.flagP = c->flagP; a.flagQ = c->flagQ;
  a.touched = c->touched; a.touched_cnt = c->touched_cnt + c->cnt_sel;
  a.partials = c->dev_scalars;
  a.acc_partials = c->defer_stats ? 1 : 0;
  a.mP = c->mP; a.vP = c->vP; a.mQ = c->mQ; a.vQ = c->vQ; a.mb = c->mb; a.vb = c->vb;
  a.lastP = c->lastP; a.lastQ = c->lastQ;
  a.o = opt_dev(c, c->step + 1);
  return a;
}

OptDev opt_dev(const bpr_ctx* c, int64_t t) {
  OptDev o;
  memset(&o, 0, sizeof(o));
  o.kind = c->opt_kind;
  o.lr = c->opt.lr; o.mu = c->opt.momentum; o.damp = c->opt.dampening;
  o.nesterov = c->opt.nesterov;
  o.b1 = c->opt.beta1; o.b2 = c->opt.beta2; o.eps = c->opt.eps; o.alpha = c->opt.alpha;
  o.t = t;
  auto safe_log = [](double x) { return x > 0.0 ? log(x) : -1.0e30; };
  o.log_b1 = safe_log((double)o.b1);
  o.log_b2 = safe_log((double)o.b2);
  o.log_mu = safe_log((double)o.mu);
  o.log_alpha = safe_log((double)o.alpha);
  // Adam replay: terms decay like (b1/sqrt(b2))^s; stop once below 1e-8 of the first
  o.kmax = 0;
  o.t_sat = -1;
  o.adam_step = 0.f;
  o.adam_bc2 = 1.f;
  if (o.kind == OPT_ADAM) {
    o.adam_step = (float)((double)o.lr / (1.0 - exp((double)t * o.log_b1)));
    o.adam_bc2 = (float)sqrt(1.0 - exp((double)t * o.log_b2));
  }
  if (o.kind == OPT_ADAM && o.b1 > 0.f) {
    const double ratio = (double)o.b1 / sqrt((double)o.b2);
    o.kmax = ratio < 1.0 ? (int)ceil(log(1e-8) / log(ratio)) : 1 << 20;
    // closed-form replay (bpr_kernels.h: opt_replay_row): needs 1 - b^t to round to 1.0f for every
    // replayed t (b^t < 2^-25) and all three series ratios b1 / b2^((j+1)/2) below 1
    const bool no_closed = c->tune_adam_closed == 0;  // bpr_set_tuning("adam_closed", 0): tests compare both routes
    const double zmax = (double)o.b1 / pow((double)o.b2, 0.5 * ADAM_SERIES);
    if (!no_closed && zmax < 0.999 && o.b1 < 1.f && o.b2 > 0.f && o.b2 < 1.f) {
      const double lim = log(ldexp(1.0, -25));
      const double t1 = ceil(lim / log((double)o.b1)), t2 = ceil(lim / log((double)o.b2));
      o.t_sat = (int64_t)(t1 > t2 ? t1 : t2);  // replayed steps are s0+1 .. : s0 >= t_sat suffices
      o.sv_min = (float)((double)o.eps * pow((double)o.b2, -0.5 * o.kmax) / 0.2);
      for (int j = 0; j < ADAM_SERIES; ++j) {
        const double lz = log((double)o.b1) - 0.5 * (j + 1) * log((double)o.b2), z = exp(lz);
        o.G[j] = (float)(z * (1.0 - exp((double)o.kmax * lz)) / (1.0 - z));
        o.log2_z[j] = (float)(lz / log(2.0));
        o.zc[j] = (float)(z / (1.0 - z));
      }
    }
  }
  return o;
}

static ApplyArgs apply_args(const bpr_ctx* c, int64_t t) {
  ApplyArgs a;
  memset(&a, 0, sizeof(a));
  a.P = c->P; a.Q = c->Q; a.bias = c->bias;
  a.GP = c->GP; a.GQ = c->GQ; a.Gb = c->Gb;
  a.mP = c->mP; a.vP = c->vP; a.mQ = c->mQ; a.vQ = c->vQ; a.mb = c->mb; a.vb = c->vb;
  a.lastP = c->lastP; a.lastQ = c->lastQ;
  a.flagP = c->flagP; a.flagQ = c->flagQ;
  a.touched = c->touched; a.touched_cnt = c->touched_cnt + c->cnt_sel;
  a.next_cnt = c->touched_cnt + (c->cnt_sel ^ 1);
  a.U = c->U; a.I = c->I; a.d = c->d;
  a.pad_user = c->pad_user; a.pad_item = c->pad_item;
  a.o = opt_dev(c, t);
  return a;
}

int check_opt_state(const bpr_ctx* c, const char* who) {
  const bool need_m = c->opt_kind == BPR_OPT_MOMENTUM || c->opt_kind == BPR_OPT_ADAM ||
                      (c->opt_kind == BPR_OPT_RMSPROP && c->opt.momentum > 0.f);
  const bool need_v = c->opt_kind == BPR_OPT_ADAM || c->opt_kind == BPR_OPT_RMSPROP;
  if ((need_m && (!c->mP || !c->mQ || (c->bias && !c->mb))) ||
      (need_v && (!c->vP || !c->vQ || (c->bias && !c->vb))))
    return fail(BPR_ERR_INVALID,
                std::string(who) + ": optimizer state not bound (bpr_bind_opt_state)");
  return BPR_OK;
}

static int drain_timing(bpr_ctx* c) {
  if (c->ev_used == 0) return BPR_OK;
  BPR_HIP_CHECK(hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < c->ev_used; ++k) {
    float ms = 0.f;
    BPR_HIP_CHECK(hipEventElapsedTime(&ms, c->ev_start[k], c->ev_stop[k]));
    c->timed_ms += (double)ms;
    c->timed_launches += 1;
  }
  c->ev_used = 0;
  return BPR_OK;
}

// ---- launches ---------------------------------------------------------------------------------
template <int MODE>
static int launch_triples(bpr_ctx* c, TripleArgs a, bool timed) {
  if (a.n <= 0) return BPR_OK;
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int G = T::G, E = T::E;
    const unsigned grid = grid_for(a.n, G, 0);
    Timer tm(c, timed);
    (void)tm;
    hipLaunchKernelGGL((k_triples<G, E, MODE>), dim3(grid), dim3(256), 0, c->stream, a);
    if (a.scalars != nullptr)
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, a.partials, (int)grid,
                         a.scalars);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

// CUs the ctx's launch stream may use: the popcount of its CU mask (bpr_stream_create), the whole
// chip for an unmasked stream.  Queried once per stream handle.
static int stream_cus(bpr_ctx* c) {
  if (c->stream_cus_of == (void*)c->stream && c->stream_cus > 0) return c->stream_cus;
  int n_cu = 0;
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
  int cus = n_cu;
  if (c->stream != nullptr) {
    uint32_t mask[16] = {0};
    if (hipExtStreamGetCUMask(c->stream, 16, mask) == hipSuccess) {
      int bits = 0;
      for (int w = 0; w < 16; ++w) bits += __builtin_popcount(mask[w]);
      if (bits > 0 && bits < n_cu) cus = bits;
    } else {
      (void)hipGetLastError();
    }
  }
  c->stream_cus = cus > 0 ? cus : 256;
  c->stream_cus_of = (void*)c->stream;
  return c->stream_cus;
}

// STREAM: one group per run of a.run_len triples; max_inflight caps the number of groups (= triples
// in flight).  A cap below one 256-thread block shrinks the block (whole waves), so
// max_inflight = 1 at G = 64 really is ONE wave walking the stream sequentially.
static int launch_stream(bpr_ctx* c, StreamArgs a, int sampler, int64_t cap_groups,
                         float* out_scalars, bool cut, bool acut = false) {
  if (a.n <= 0) return BPR_OK;
  if (cut)
    if (int rc = refresh_alloc(c)) return rc;
  if (int rc = flush_deferred_stats(c)) return rc;  // (this launch overwrites the partials)
  if (c->acut_pending && !acut) {
    // the asynchronous cut of the previous launch still reads that launch's partials (and Q + the hot
    // deltas) on the side stream: a launch outside the acut pipeline reuses the first set of partials
    // and folds / moves the rows, so it waits for that cut
    BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_keys, 0));
    c->acut_pending = false;
  }
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int G = T::G, E = T::E;
    unsigned block = 256;
    a.gpw_active = 64 / G;
    if (cap_groups > 0 && cap_groups * G < 256) {
      block = (unsigned)(((cap_groups * G + 63) / 64) * 64);
      if (cap_groups < 64 / G) a.gpw_active = (int)cap_groups;  // one wave, one group at work
    }
    // "seen?" answers (bpr_device.h): the LDS bitmap (I bits per group) while a full 256-thread
    // block's bitmaps fit 64 KiB (>= 2 blocks per CU at full width: I <= 65,536 for d <= 128,
    // 131,072 above); larger item tables stage the user's sorted seen list in LDS instead
    // (LIST_CAP entries per group, heavier users search the CSR in HBM).  bpr_set_tuning("seen", ...)
    // forces a structure (tests, measurements); a forced bitmap shrinks the block to fit.
    const int words = (int)(((c->I + 31) / 32 + 3) / 4 * 4);  // multiple of 4: 16-byte LDS wipes
    static const char* const seen_names[] = {"", "csr", "bitmap", "list"};
    const std::string force = seen_names[c->tune_seen];  // bpr_set_tuning (tests, measurements)
    constexpr int LIST_CAP = 512;
    int seen = SEEN_CSR;
    int lds_words = 0;
    if (sampler != NEG_GIVEN && force != "csr") {
      const bool bm_fits = (size_t)(block / G) * words * sizeof(uint32_t) <= 64 * 1024;
      if (force == "list" || (force != "bitmap" && !bm_fits)) {
        seen = SEEN_LIST;
        lds_words = LIST_CAP;
      } else {
        while (block > 64 && (size_t)(block / G) * words * sizeof(uint32_t) > 64 * 1024) block /= 2;
        if ((size_t)(block / G) * words * sizeof(uint32_t) <= 64 * 1024) {
          seen = SEEN_BITMAP;
          lds_words = words;
        } else {
          block = 256;
          seen = SEEN_LIST;
          lds_words = LIST_CAP;
        }
      }
    }
    const size_t shmem = (size_t)(block / G) * (size_t)lds_words * sizeof(uint32_t);
    const int64_t per_block = (int64_t)(block / 64) * a.gpw_active;
    // groups the launch stream's CUs hold at once (occupancy of THIS instantiation x its CUs)
    auto pick = [&](auto fn) {
      using std::integral_constant;
      auto with_seen = [&](auto smp) {
        if (seen == SEEN_BITMAP) fn(smp, integral_constant<int, SEEN_BITMAP>{});
        else if (seen == SEEN_LIST) fn(smp, integral_constant<int, SEEN_LIST>{});
        else fn(smp, integral_constant<int, SEEN_CSR>{});
      };
      if (sampler == NEG_GIVEN)
        fn(integral_constant<int, NEG_GIVEN>{}, integral_constant<int, SEEN_CSR>{});
      else if (sampler == NEG_UNIFORM) with_seen(integral_constant<int, NEG_UNIFORM>{});
      else with_seen(integral_constant<int, NEG_ADAPTIVE>{});
    };
    const auto occ_key = std::make_tuple(c->d, sampler, seen, (int)block, (int64_t)shmem);
    auto occ_it = c->stream_occ.find(occ_key);
    int occ = occ_it != c->stream_occ.end() ? occ_it->second : 0;
    if (occ < 1) pick([&](auto smp, auto sn) {
      constexpr int SMP = decltype(smp)::value, SN = decltype(sn)::value;
      if (c->d == G * E)
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_stream<G, E, SMP, SN, true>, (int)block,
                                                     shmem);
      else
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_stream<G, E, SMP, SN, false>,
                                                     (int)block, shmem);
    });
    if (occ < 1) {
      (void)hipGetLastError();
      occ = 1;
    }
    c->stream_occ[occ_key] = occ;
    const int64_t resident = (int64_t)occ * stream_cus(c) * per_block;
    // run_len 0 = by launch size.  A launch whose runs of 8 overfill the chip takes runs of 8 and a
    // grid of ~1.5 runs per group (the dispatcher balances the rest: r1 sweep).  A smaller launch
    // (Netflix-sized periods, a rank's share of a period at 8 ranks) is over when its slowest group
    // is: the shortest runs of >= 4 triples that still fit the chip in ONE residency, one run per
    // group (profiles/r03_sweep_small.txt: 40,704 triples d=64 0.0433 -> 0.0393 ms, 24,896 triples
    // d=128 0.0550 -> 0.0413 ms; runs shorter than 4 re-load the user row too often).
    if (a.run_len <= 0) {
      a.run_len = 8;
      // (max_inflight > 0 — what StreamTrainer passes — only changes this when the cap binds: the
      // residency bound is min(chip, cap))
      const int64_t room = cap_groups > 0 ? std::min<int64_t>(resident, cap_groups) : resident;
      if ((a.n + 7) / 8 < room) {
        a.run_len = 4;
        while (a.run_len < 8 && (a.n + a.run_len - 1) / a.run_len > room) ++a.run_len;
      }
    }
    c->last_run_len = a.run_len;
    const int64_t n_runs = (a.n + a.run_len - 1) / a.run_len;
    int64_t want = n_runs;
    if (cap_groups > 0 && want > cap_groups) want = cap_groups;
    int64_t nblk = (want + per_block - 1) / per_block;
    if ((cap_groups <= 0 || n_runs <= cap_groups) && n_runs > resident)
      nblk = (2 * nblk + 2) / 3;  // 1.5 runs per group
    const int64_t max_blk = std::min<int64_t>(stream_grid_cap() * (256 / block), STREAM_MAX_GRID);
    if (nblk > max_blk) nblk = max_blk;
    unsigned grid = (unsigned)(nblk < 1 ? 1 : nblk);
    a.bm_words = lds_words;
    const bool hot = a.hot_slot != nullptr;
    // ---- the LDS tier of the hot block (k_stream LDSHOT, bpr_hotlds.hip; bpr_set_tuning "hot_lds" = rows asked
    // for): ONE workgroup per CU — up to 1,024 threads, its groups' seen bitmaps and an [L, d] fp32 delta block in
    // LDS — persistent over its share of the runs.  Taken when the launch fills the chip at least twice (a smaller
    // one is over when its slowest group is: the plain kernel's short runs win there), the shape has a FULL
    // instantiation, the snapshot is sorted whole and the per-group bitmaps leave room for >= 8 rows.
    bool use_lds = false;
    int seen_l = SEEN_BITMAP;
    size_t shmem_l = 0;
    unsigned block_l = E <= 4 ? 1024 : 512;  // (E >= 8: 64+ registers of rows per lane — two waves per SIMD)
    if (c->tune_lds_block > 0) block_l = std::min<unsigned>(block_l, (unsigned)c->tune_lds_block);
    c->last_lds_rows = 0;
    // (not in the asynchronous-cut pipeline: there the deltas stay in the global block from launch to launch and a
    // hot row's value is Q + that block — which the LDS-tier kernel, reading Q + its own LDS delta, leaves out)
    const bool acut_fold = acut && c->tune_acut_fold != 0;  // r6: the asynchronous cut with the fold kept on this stream
    if (c->tune_hot_lds > 0 && hot && c->hot_code != nullptr && c->d == G * E && a.snap_meta == nullptr &&
        (!acut || acut_fold) && !c->hot_unfolded && (sampler == NEG_GIVEN || force != "csr")) {
      if (cap_groups > 0 && cap_groups * G < block_l) block_l = (unsigned)(((cap_groups * G + 63) / 64) * 64);
      // the groups' seen structure beside the rows: the I-bit bitmaps while they leave 32 KB for rows, else (or
      // forced) the staged sorted lists (LIST_CAP entries per group: item tables past ~60 k items)
      size_t bm_bytes = sampler == NEG_GIVEN ? 0 : (size_t)(block_l / G) * words * sizeof(uint32_t);
      if (sampler != NEG_GIVEN && (force == "list" || (force != "bitmap" && bm_bytes + 32 * 1024 > lds_tier_room(c->d)))) {
        seen_l = SEEN_LIST;
        bm_bytes = (size_t)(block_l / G) * LIST_CAP * sizeof(uint32_t);
      }
      const size_t row_bytes = sizeof(float) * (size_t)c->d + sizeof(uint32_t);
      const size_t room = lds_tier_room(c->d);
      int64_t L = bm_bytes < room ? (int64_t)((room - bm_bytes) / row_bytes) : 0;
      L = std::min<int64_t>(L, std::min<int64_t>(c->tune_hot_lds, c->hot_H));
      const int64_t per_block_l = (int64_t)(block_l / 64) * a.gpw_active;
      const int64_t runs8 = (a.n + 7) / 8;
      const bool fills = runs8 >= 2 * (int64_t)stream_cus(c) * per_block_l;
      if (L >= 8 && (fills || c->tune_hot_lds_force)) {
        use_lds = true;
        if (c->run_len <= 0) a.run_len = 8;
        c->last_run_len = a.run_len;
        // the last tickets of a persistent workgroup are short runs (k_stream: tail1 / tail2), whole wave-loads each
        // (zones hold whole wave-loads of runs: what is left of the chunk past the last whole wave-load of full
        // runs always goes in the shortest runs)
        const int64_t len2 = std::max(1, a.run_len / 2), len3 = std::max(1, a.run_len / 4);
        const int64_t wl = (int64_t)a.run_len * a.gpw_active;  // triples of a wave-load of full runs
        int64_t t1 = (int64_t)((double)a.n * (1.0 - c->tune_lds_tail / 100.0)) / wl * wl;
        int64_t t2 = t1 + (int64_t)((double)(a.n - t1) * 0.6) / (len2 * a.gpw_active) * (len2 * a.gpw_active);
        if (c->tune_lds_tail <= 0) t1 = t2 = a.n / wl * wl;
        a.tail1 = (int32_t)t1;
        a.tail2 = (int32_t)t2;
        const int64_t n_runs_l = t1 / a.run_len + (t2 - t1) / len2 + (a.n - t2 + len3 - 1) / len3;
        int64_t want_l = n_runs_l;
        if (cap_groups > 0 && want_l > cap_groups) want_l = cap_groups;
        grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(stream_cus(c), (want_l + per_block_l - 1) / per_block_l));
        a.bm_words = sampler == NEG_GIVEN ? 0 : (seen_l == SEEN_LIST ? LIST_CAP : words);
        a.lds_L = (int32_t)L;
        a.hot_by_rank = c->hot_by_rank;
        a.lds_only = c->hot_tier ? 0 : 1;
        a.hot_slot = c->hot_code;
        shmem_l = bm_bytes + (size_t)L * row_bytes;
        c->last_lds_rows = (int)L;
      }
    }
    // hot tier + cut: fold, reconciliation passes and snapshot cut are ONE pass, bpr_sync_cut, which the
    // caller issues next; the launch leaves it its loss partials too
    const bool defer = cut && hot && c->hot_tier;
    if ((cut || acut) && c->ev_keys == nullptr) {
      BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_keys, hipEventDisableTiming));
      BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming));
    }
    if (acut) {
      if (c->ev_launch == nullptr) BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_launch, hipEventDisableTiming));
      if (c->side == nullptr) {
        BPR_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        c->side_owned = true;
      }
      if (!acut_fold) {
        // the launch after next reuses this launch's partials: two sets, used alternately
        if (a.partials != nullptr) a.partials = c->dev_scalars + (size_t)c->acut_parity * 4 * (STREAM_MAX_GRID + 1);
        c->acut_parity ^= 1;
      }
    }
    if (a.bias != nullptr) {  // the launch works on the bias with one item per line (bpr_kernels.h)
      if (c->bias_w_rows != c->I) {
        hipFree(c->bias_w);
        c->bias_w = nullptr;
        c->bias_w_rows = 0;
        BPR_HIP_CHECK(hipMalloc(&c->bias_w, sizeof(float) * (size_t)c->I * BIAS_LINE));
        c->bias_w_rows = c->I;
      }
      // (re)fill the wide table — unless it is known to equal the dense vector still: the epilogue of
      // the previous launch wrote it back, and nobody has touched the vector since (every entry point
      // of the library that writes it says so; writes of the caller's own: bpr_bias_written)
      if (!(c->bias_track && c->bias_w_valid && c->bias_w_of == c->bias))
        hipLaunchKernelGGL(k_bias_widen, dim3((unsigned)((c->I + 255) / 256)), dim3(256), 0, c->stream, c->bias,
                           c->bias_w, (int32_t)c->I);
      c->bias_w_valid = false;
      c->bias_w_of = c->bias;
      a.bias = c->bias_w;
    }
    // the write-back rides on the launch's own epilogue where there is one on this stream
    const bool bias_in_epilogue = a.bias != nullptr && (!acut || acut_fold) && !(cut && hot && c->hot_tier);
    {
      Timer tm(c, true);
      (void)tm;
      hipEvent_t stop = (acut && !acut_fold) ? c->ev_launch : nullptr;
      if (use_lds) {
        if (int rc = launch_stream_lds(c, a, sampler, seen_l, grid, block_l, shmem_l, stop)) return rc;
      }
      auto go = [&](auto smp, auto sn) {
        constexpr int SMP = decltype(smp)::value, SN = decltype(sn)::value;
        if constexpr (SMP == NEG_ADAPTIVE) {
          if (a.snap_meta != nullptr) {  // a partial snapshot: the instantiations whose walk can finish inside a bin
            if (c->d == G * E)
              hipExtLaunchKernelGGL((k_stream<G, E, SMP, SN, true, true>), dim3(grid), dim3(block), shmem,
                                    c->stream, nullptr, stop, 0, a);
            else
              hipExtLaunchKernelGGL((k_stream<G, E, SMP, SN, false, true>), dim3(grid), dim3(block), shmem,
                                    c->stream, nullptr, stop, 0, a);
            return;
          }
        }
        if (c->d == G * E)
          hipExtLaunchKernelGGL((k_stream<G, E, SMP, SN, true>), dim3(grid), dim3(block), shmem,
                                c->stream, nullptr, stop, 0, a);
        else
          hipExtLaunchKernelGGL((k_stream<G, E, SMP, SN, false>), dim3(grid), dim3(block), shmem,
                                c->stream, nullptr, stop, 0, a);
      };
      if (!use_lds) pick(go);
    }
    if (a.bias != nullptr && !bias_in_epilogue)
      hipLaunchKernelGGL(k_bias_narrow, dim3((unsigned)((c->I + 255) / 256)), dim3(256), 0, c->stream,
                         c->bias_w, c->bias, (int32_t)c->I);
    if (a.bias != nullptr) c->bias_w_valid = true;  // dense == wide again once the write-back has run
    if (acut_fold) {
      // r6: the launch stream keeps only what the NEXT launch needs — the fold of the hot block (+ loss statistics,
      // + the item_bias write-back): k_stream_epilogue, a few microseconds — and the 2 x I x d x 4-byte transpose
      // that only the sorter reads goes to the side stream, behind this epilogue and beside the next launch (it
      // reads Q while that launch updates it: a snapshot cut during the launch instead of before it — a little
      // FRESHER than lag 1, inside the same budget).  Unlike the r4 form below the block is folded after every
      // launch, so the LDS-tier kernel (which reads a hot row as Q + its own LDS delta) stays valid.
      const bool fold = hot && !c->hot_tier;
      EpilogueArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.partials = a.partials; ea.n_blocks = (int)grid; ea.out = out_scalars;
      ea.Q = c->Q; ea.delta = c->hot_delta; ea.hot_items = c->hot_items;
      ea.H = fold ? c->hot_H : 0; ea.R = c->hot_R; ea.d = c->d;
      ea.fold_blocks = fold ? (int)std::min<int64_t>(((int64_t)c->hot_H * c->d + 255) / 256, 64) : 0;
      if (bias_in_epilogue) {
        ea.bias_w = c->bias_w; ea.bias = c->bias; ea.I = (int32_t)c->I;
        ea.bias_blocks = (int)std::min<int64_t>((c->I + 255) / 256, 256);
      }
      hipExtLaunchKernelGGL(k_stream_epilogue, dim3(1 + ea.fold_blocks + ea.bias_blocks), dim3(256), 0, c->stream, nullptr,
                            c->ev_launch, 0, ea);
      BPR_HIP_CHECK(hipGetLastError());
      EpilogueCutArgs ec;
      memset(&ec, 0, sizeof(ec));
      ec.Q = c->Q; ec.T = c->keysT; ec.sig_acc = c->sig_acc; ec.d = c->d; ec.I = (int32_t)c->I;  // no hot block, no sums
      dim3 eg((unsigned)((c->I + 31) / 32), (unsigned)((c->d + 31) / 32) + 1u);
      BPR_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_launch, 0));
      hipExtLaunchKernelGGL(k_stream_epilogue_cut, eg, dim3(256), 0, c->side, nullptr, c->ev_keys, 0, ec);
      BPR_HIP_CHECK(hipGetLastError());
      c->keys_cut = true;
      if (c->meta_front != nullptr && c->keysT == c->keys_front) c->keys_front_stale = true;
      c->keys_event = true;
      c->keys_on_side = true;
      c->acut_pending = true;  // (a launch outside the pipeline still waits for this cut: it reads Q)
      return BPR_OK;
    }
    if (acut) {
      // the cut of the next snapshot on the SIDE stream, behind this launch and beside the next
      // one: read-only (keys = Q + hot deltas, nothing folded), it also sums the loss partials
      EpilogueCutArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.partials = a.partials; ea.n_blocks = (int)grid; ea.out = out_scalars;
      ea.Q = c->Q; ea.delta = c->hot_delta; ea.hot_slot = hot ? c->hot_slot : nullptr;
      ea.T = c->keysT; ea.sig_acc = c->sig_acc;
      ea.H = hot ? c->hot_H : 0; ea.R = c->hot_R; ea.d = c->d; ea.I = (int32_t)c->I;
      ea.fold = 0;
      dim3 eg((unsigned)((c->I + 31) / 32), (unsigned)((c->d + 31) / 32) + 1u);
      BPR_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_launch, 0));
      hipExtLaunchKernelGGL(k_stream_epilogue_cut, eg, dim3(256), 0, c->side, nullptr, c->ev_keys, 0, ea);
      BPR_HIP_CHECK(hipGetLastError());
      c->keys_cut = true;
      if (c->meta_front != nullptr && c->keysT == c->keys_front) c->keys_front_stale = true;
      c->keys_event = true;
      c->keys_on_side = true;  // whoever sorts these keys on the launch stream waits for ev_keys
      c->acut_pending = true;
      c->hot_unfolded = hot;
      return BPR_OK;
    }
    if (defer) {
      c->defer_blocks = (int)grid;
      c->defer_out = out_scalars;
      c->defer_pending = true;
    } else if (cut) {
      // the epilogue also cuts the next snapshot's keys (k_stream_epilogue_cut) — into the key
      // buffer the split refresh that may still be sorting does NOT read (bpr_ctx.h keysT_buf)
      EpilogueCutArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.partials = a.partials; ea.n_blocks = (int)grid; ea.out = out_scalars;
      ea.Q = c->Q; ea.delta = c->hot_delta; ea.hot_slot = hot ? c->hot_slot : nullptr;
      ea.T = c->keysT; ea.sig_acc = c->sig_acc;
      ea.H = hot ? c->hot_H : 0; ea.R = c->hot_R; ea.d = c->d; ea.I = (int32_t)c->I;
      ea.fold = 1;
      if (bias_in_epilogue) { ea.bias_w = c->bias_w; ea.bias = c->bias; }
      dim3 eg((unsigned)((c->I + 31) / 32), (unsigned)((c->d + 31) / 32) + 1u);
      // the split refresh's side stream waits for this cut: the event rides on the kernel's own
      // completion signal (hipExtLaunchKernelGGL stop event) instead of a marker packet behind it
      static const bool ride = getenv("BPR_CUT_EVENT") == nullptr || atoi(getenv("BPR_CUT_EVENT")) != 0;
      if (ride) {
        hipExtLaunchKernelGGL(k_stream_epilogue_cut, eg, dim3(256), 0, c->stream, nullptr, c->ev_keys,
                              0, ea);
      } else {
        hipLaunchKernelGGL(k_stream_epilogue_cut, eg, dim3(256), 0, c->stream, ea);
      }
      c->keys_cut = true;
      if (c->meta_front != nullptr && c->keysT == c->keys_front) c->keys_front_stale = true;
      c->keys_event = ride;
    } else if (out_scalars != nullptr || (hot && !c->hot_tier) || bias_in_epilogue) {
      const bool fold = hot && !c->hot_tier;  // hot tier: the deltas stay for bpr_hot_exchange
      EpilogueArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.partials = a.partials; ea.n_blocks = (int)grid; ea.out = out_scalars;
      ea.Q = c->Q; ea.delta = c->hot_delta; ea.hot_items = c->hot_items;
      ea.H = fold ? c->hot_H : 0; ea.R = c->hot_R; ea.d = c->d;
      ea.fold_blocks =
          fold ? (int)std::min<int64_t>(((int64_t)c->hot_H * c->d + 255) / 256, 64) : 0;
      if (bias_in_epilogue) {
        ea.bias_w = c->bias_w; ea.bias = c->bias; ea.I = (int32_t)c->I;
        ea.bias_blocks = (int)std::min<int64_t>((c->I + 255) / 256, 256);
      }
      hipLaunchKernelGGL(k_stream_epilogue, dim3(1 + ea.fold_blocks + ea.bias_blocks), dim3(256), 0, c->stream, ea);
    }
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

float inv_log1mp(float p) { return (float)(1.0 / log1p(-(double)p)); }


int check_triples(const bpr_ctx* c, const char* who, const int32_t* users,
                         const int32_t* pos, int64_t B) {
  if (int rc = check_bound(c, who)) return rc;
  if (B < 0 || (B > 0 && (users == nullptr || pos == nullptr)))
    return fail(BPR_ERR_INVALID, std::string(who) + ": bad argument");
  return BPR_OK;
}

int check_sampler(const bpr_ctx* c, const char* who, int32_t sampler, float adaptive_p,
                         const int32_t* neg, int64_t B) {
  if (sampler == BPR_NEG_GIVEN) {
    if (neg == nullptr && B > 0) return fail(BPR_ERR_INVALID, std::string(who) + ": neg is NULL");
    return BPR_OK;
  }
  if (sampler != BPR_NEG_UNIFORM && sampler != BPR_NEG_ADAPTIVE)
    return fail(BPR_ERR_INVALID, std::string(who) + ": unknown sampler");
  if (c->indptr == nullptr) return fail(BPR_ERR_INVALID, std::string(who) + ": seen CSR not bound");
  if (sampler == BPR_NEG_ADAPTIVE) {
    if (!c->have_snapshot)
      return fail(BPR_ERR_INVALID, std::string(who) + ": call bpr_adaptive_refresh first");
    if (!(adaptive_p > 0.f && adaptive_p < 1.f))
      return fail(BPR_ERR_INVALID, std::string(who) + ": adaptive_p not in (0,1)");
  }
  return BPR_OK;
}

// STRICT lazy replay: bring every row to step c->step (rows tracked by lastP / lastQ)
int strict_flush_impl(bpr_ctx* c) {
  if (c->opt_kind == BPR_OPT_SGD || c->GP == nullptr || c->step == 0) return BPR_OK;
  c->bias_w_valid = false;
  if (int rc = check_opt_state(c, "bpr_flush_lazy")) return rc;
  // (accumulated gradients of a batch in flight are left alone: the flush only advances w / m / v
  // and the rows' last-step marks, bpr_apply then finds nothing left to replay)
  ApplyArgs a = apply_args(c, c->step);
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_flush_lazy<T::G, T::E>), dim3(grid_for(c->U, T::G, 0)), dim3(256), 0,
                       c->stream, a, 0);
    hipLaunchKernelGGL((k_flush_lazy<T::G, T::E>), dim3(grid_for(c->I, T::G, 0)), dim3(256), 0,
                       c->stream, a, 1);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

}  // namespace bpr

using namespace bpr;

extern "C" {

int bpr_version(void) { return BPRCORE_VERSION; }
const char* bpr_last_error(void) { return g_last_error.c_str(); }

int bpr_ctx_create(bpr_ctx** out, int device_id, void* hip_stream) {
  if (out == nullptr) return fail(BPR_ERR_INVALID, "bpr_ctx_create: out is NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BPR_ERR_HIP, "bpr_ctx_create: no HIP device visible (libbprcore has no CPU path)");
  if (device_id < 0 || device_id >= ndev)
    return fail(BPR_ERR_INVALID, "bpr_ctx_create: bad device id");
  BPR_HIP_CHECK(hipSetDevice(device_id));
  bpr_ctx* c = new (std::nothrow) bpr_ctx();
  if (c == nullptr) return fail(BPR_ERR_NOMEM, "bpr_ctx_create: out of host memory");
  c->device = device_id;
  c->stream = (hipStream_t)hip_stream;
  if (hipMalloc(&c->dev_scalars, sizeof(float) * 2 * 4 * (size_t)(STREAM_MAX_GRID + 1)) != hipSuccess) {
    delete c;
    return fail(BPR_ERR_HIP, "bpr_ctx_create: hipMalloc failed");
  }
  *out = c;
  return BPR_OK;
}

int bpr_ctx_destroy(bpr_ctx* c) {
  if (c == nullptr) return BPR_OK;
  hipSetDevice(c->device);
  // The caller's streams may be GONE by now: a garbage collector destroys the objects of a finished
  // trainer in any order, and hipStreamSynchronize on a destroyed (CU-masked) stream aborts inside the
  // runtime ("std::system_error: Invalid argument") or hangs — the intermittent failure of the two-rank
  // parity run (r5).  So: wait for the DEVICE, and do whatever cleanup work is left on the default stream.
  hipDeviceSynchronize();
  c->stream = nullptr;
  if (!c->side_owned) c->side = nullptr;
  free_strict_scratch(c);
  comm_free(c, false);  // (no fold into tables that may be gone)
  refresh_free(c);
  side_free(c);
  vs_free(c);
  hipFree(c->bias_w);
  hipFree(c->dev_scalars);
  if (c->ev_launch) hipEventDestroy(c->ev_launch);
  for (auto e : c->ev_start) hipEventDestroy(e);
  for (auto e : c->ev_stop) hipEventDestroy(e);
  delete c;
  return BPR_OK;
}

int bpr_set_stream(bpr_ctx* c, void* hip_stream) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_stream: ctx is NULL");
  c->stream = (hipStream_t)hip_stream;
  return BPR_OK;
}

int bpr_bind_tables(bpr_ctx* c, float* P, int64_t U, float* Q, int64_t I, int32_t d,
                    float* item_bias, int32_t pad_user, int32_t pad_item) {
  if (c == nullptr || P == nullptr || Q == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_bind_tables: NULL argument");
  if (U < 1 || I < 2 || U >= ((int64_t)1 << 30) || I >= ((int64_t)1 << 30))
    return fail(BPR_ERR_INVALID, "bpr_bind_tables: table sizes out of range");
  if (d < 1 || d > 1024)
    return fail(BPR_ERR_UNSUPPORTED, "bpr_bind_tables: embedding dim must be in [1, 1024]");
  if (((uintptr_t)P | (uintptr_t)Q) & 3u)
    return fail(BPR_ERR_INVALID, "bpr_bind_tables: tables must be 4-byte aligned");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->U != U || c->I != I || c->d != d) {
    BPR_HIP_CHECK(hipStreamSynchronize(c->stream));
    free_strict_scratch(c);
    refresh_free(c);
    vs_free(c);
  } else if (c->vs_active && (c->P != P || c->Q != Q || c->bias != item_bias)) {
    if (int rc = vs_leave(c)) return rc;  // pending steps belong to the tables bound so far
  }
  c->P = P; c->Q = Q; c->bias = item_bias;
  c->bias_w_valid = false;
  c->keys_cut = false;
  c->U = U; c->I = I; c->d = d;
  c->pad_user = pad_user; c->pad_item = pad_item;
  c->G = d <= 128 ? 32 : 64;
  const int per_lane = (d + c->G - 1) / c->G;
  c->E = c->G == 32 ? (per_lane <= 1 ? 1 : per_lane <= 2 ? 2 : 4)
                    : (per_lane <= 4 ? 4 : per_lane <= 8 ? 8 : 16);
  return BPR_OK;
}

int bpr_bind_seen_csr(bpr_ctx* c, const int64_t* indptr, const int32_t* indices) {
  if (c == nullptr || indptr == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_bind_seen_csr: NULL argument");
  c->indptr = indptr;
  c->indices = indices;
  c->heavy_for = nullptr;  // the heavy users' bitmaps belong to the CSR bound before: rebuilt lazily
  return BPR_OK;
}

int bpr_bind_item_weights(bpr_ctx* c, const float* accept, const int32_t* alias) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_bind_item_weights: ctx is NULL");
  if ((accept == nullptr) != (alias == nullptr))
    return fail(BPR_ERR_INVALID, "bpr_bind_item_weights: accept and alias go together");
  c->w_accept = accept;
  c->w_alias = alias;
  return BPR_OK;
}

int bpr_set_reg(bpr_ctx* c, float alpha_user, float alpha_item, float alpha_neg) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_reg: ctx is NULL");
  c->au = alpha_user; c->ai = alpha_item; c->an = alpha_neg;
  return BPR_OK;
}

int bpr_set_optimizer(bpr_ctx* c, int32_t kind, const bpr_opt_params* params) {
  if (c == nullptr || params == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_set_optimizer: NULL argument");
  if (kind < BPR_OPT_SGD || kind > BPR_OPT_RMSPROP)
    return fail(BPR_ERR_INVALID, "bpr_set_optimizer: unknown optimizer kind");
  if ((kind == BPR_OPT_MOMENTUM || kind == BPR_OPT_RMSPROP) &&
      !(params->momentum >= 0.f && params->momentum < 1.f))
    return fail(BPR_ERR_INVALID, "bpr_set_optimizer: momentum must be in [0, 1)");
  if (c->vs_active && (kind != c->opt_kind || memcmp(&c->opt, params, sizeof(*params)) != 0)) {
    // batched STREAM: pending steps and the replay of missed ones use the hyper-parameters in
    // force when they were taken — bring every row to "now" before those change
    BPR_HIP_CHECK(hipSetDevice(c->device));
    if (int rc = vs_leave(c)) return rc;
  }
  c->opt_kind = kind;
  c->opt = *params;
  return BPR_OK;
}

int bpr_bind_opt_state(bpr_ctx* c, float* m_P, float* v_P, float* m_Q, float* v_Q, float* m_bias,
                       float* v_bias) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_bind_opt_state: ctx is NULL");
  if (c->vs_active && (c->mP != m_P || c->vP != v_P || c->mQ != m_Q || c->vQ != v_Q ||
                       c->mb != m_bias || c->vb != v_bias)) {
    BPR_HIP_CHECK(hipSetDevice(c->device));
    if (int rc = vs_leave(c)) return rc;  // finish with the state tensors bound so far
  }
  c->mP = m_P; c->vP = v_P; c->mQ = m_Q; c->vQ = v_Q; c->mb = m_bias; c->vb = v_bias;
  return BPR_OK;
}

// ---- sampling -----------------------------------------------------------------------------------
static int launch_sample(bpr_ctx* c, int what, SampleArgs a) {
  if (a.n <= 0) return BPR_OK;
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int G = T::G, E = T::E;
    const unsigned grid = grid_for(a.n, G, 0);
    // adaptive picks walk hundreds of candidates: an LDS seen-bitmap per group when the block's
    // bitmaps fit 64 KiB; the uniform sampler tests a handful and searches the CSR directly
    const int words = (int)(((c->I + 31) / 32 + 3) / 4 * 4);
    const size_t lds = (size_t)(256 / G) * words * sizeof(uint32_t);
    const bool no_bm = c->tune_seen == 1;
    // (one block per CU is plenty for a single batch, so the bitmaps may take most of the 160 KB)
    constexpr size_t SAMPLE_LDS_MAX = 144 * 1024;
    const bool bm = what != SAMPLE_UNIFORM && lds <= SAMPLE_LDS_MAX && !no_bm;
    a.bm_words = bm ? words : 0;
    if (bm && lds > 64 * 1024) {
      static bool raised = false;  // per (G, E) instantiation of this lambda
      if (!raised) {
        BPR_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&k_sample<G, E, SAMPLE_ADAPTIVE, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLE_LDS_MAX));
        BPR_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&k_sample<G, E, SAMPLE_PICK, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLE_LDS_MAX));
        raised = true;
      }
    }
    if (what == SAMPLE_UNIFORM)
      hipLaunchKernelGGL((k_sample<G, E, SAMPLE_UNIFORM, false>), dim3(grid), dim3(256), 0,
                         c->stream, a);
    else if (what == SAMPLE_ADAPTIVE && bm)
      hipLaunchKernelGGL((k_sample<G, E, SAMPLE_ADAPTIVE, true>), dim3(grid), dim3(256), lds,
                         c->stream, a);
    else if (what == SAMPLE_ADAPTIVE)
      hipLaunchKernelGGL((k_sample<G, E, SAMPLE_ADAPTIVE, false>), dim3(grid), dim3(256), 0,
                         c->stream, a);
    else if (bm)
      hipLaunchKernelGGL((k_sample<G, E, SAMPLE_PICK, true>), dim3(grid), dim3(256), lds, c->stream,
                         a);
    else
      hipLaunchKernelGGL((k_sample<G, E, SAMPLE_PICK, false>), dim3(grid), dim3(256), 0, c->stream,
                         a);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

static SampleArgs sample_args(const bpr_ctx* c) {
  SampleArgs a;
  memset(&a, 0, sizeof(a));
  (void)snapshot_complete_impl(const_cast<bpr_ctx*>(c));  // the sampler kernels read a snapshot sorted whole
  a.P = c->P; a.I = c->I; a.d = c->d;
  a.indptr = c->indptr; a.indices = c->indices;
  a.order = c->order; a.sigma = c->sigma;
  a.iw = ItemWeights{c->w_accept, c->w_alias};
  a.mP = c->mP; a.vP = c->vP; a.lastP = c->lastP;
  a.o = opt_dev(c, c->step + 1);
  return a;
}

int bpr_sample_uniform(bpr_ctx* c, const int32_t* users, int64_t B, uint64_t seed, uint64_t offset,
                       int32_t* neg_out) {
  if (int rc = check_bound(c, "bpr_sample_uniform")) return rc;
  if (c->indptr == nullptr) return fail(BPR_ERR_INVALID, "bpr_sample_uniform: seen CSR not bound");
  if (users == nullptr || neg_out == nullptr || B < 0)
    return fail(BPR_ERR_INVALID, "bpr_sample_uniform: bad argument");
  SampleArgs a = sample_args(c);
  a.users = users; a.n = B; a.seed = seed; a.offset = offset; a.neg = neg_out;
  return launch_sample(c, SAMPLE_UNIFORM, a);
}

int bpr_adaptive_refresh(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_adaptive_refresh")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->vs_active)  // batched STREAM: the snapshot must see the item rows as of "now"
    if (int rc = vs_flush(c, false, true)) return rc;
  const int world = comm_world(c);
  if (world > 1 && c->d % world == 0 && !c->refresh_pending) {
    // several ranks behind one communicator: every rank sorts its share of the factors of its own
    // replica, an all-gather hands everybody every factor's order, the same snapshot is published
    const int per = c->d / world, r = comm_rank(c);
    if (int rc = refresh_impl(c, false, r * per, (r + 1) * per)) return rc;
    const int back = c->have_snapshot ? (c->snap_front ^ 1) : c->snap_front;
    if (int rc = comm_gather_snapshot(c, c->order_alloc[back] + BPR_ORDER_PAD, c->sigma_buf[back], per))
      return rc;
    return refresh_publish_impl(c);
  }
  return refresh_impl(c, false, 0, c->d);
}

int bpr_adaptive_refresh_begin(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_adaptive_refresh_begin", c == nullptr || !c->keys_cut)) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->vs_active)
    if (int rc = vs_flush(c, false, true)) return rc;
  return refresh_impl(c, true, 0, c->d);
}

int bpr_adaptive_refresh_commit(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_adaptive_refresh_commit", false)) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  return refresh_commit_impl(c);
}

int bpr_adaptive_refresh_part(bpr_ctx* c, int32_t f_lo, int32_t f_hi) {
  if (int rc = check_bound(c, "bpr_adaptive_refresh_part")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->vs_active)
    if (int rc = vs_flush(c, false, true)) return rc;
  return refresh_impl(c, false, f_lo, f_hi);
}

int bpr_adaptive_refresh_publish(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_adaptive_refresh_publish")) return rc;
  return refresh_publish_impl(c);
}

int bpr_adaptive_snapshot_ptrs(bpr_ctx* c, int32_t back, void** order_host, void** sigma_host) {
  if (int rc = check_bound(c, "bpr_adaptive_snapshot_ptrs")) return rc;
  if (order_host == nullptr || sigma_host == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_adaptive_snapshot_ptrs: NULL argument");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = refresh_alloc(c)) return rc;
  if (!back)
    if (int rc = snapshot_complete_impl(c)) return rc;
  const int front = c->snap_front;
  const int which = back ? (c->have_snapshot ? front ^ 1 : front) : front;
  *order_host = (void*)(c->order_alloc[which] + BPR_ORDER_PAD);
  *sigma_host = (void*)c->sigma_buf[which];
  return BPR_OK;
}

int bpr_adaptive_snapshot_partial(bpr_ctx* c, int32_t* partial_host) {
  if (c == nullptr || partial_host == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_adaptive_snapshot_partial: NULL argument");
  *partial_host = (c->have_snapshot && c->meta_front != nullptr) ? 1 : 0;
  return BPR_OK;
}

int bpr_adaptive_refresh_pending(bpr_ctx* c, int32_t* pending_host) {
  if (c == nullptr || pending_host == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_adaptive_refresh_pending: NULL argument");
  *pending_host = c->refresh_pending ? 1 : 0;
  return BPR_OK;
}

int bpr_set_side_stream(bpr_ctx* c, void* hip_stream) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_side_stream: ctx is NULL");
  if (c->refresh_pending)
    return fail(BPR_ERR_INVALID, "bpr_set_side_stream: a split refresh is pending");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->side != nullptr) {
    BPR_HIP_CHECK(hipStreamSynchronize(c->side));
    if (c->side_owned) hipStreamDestroy(c->side);
  }
  c->side = (hipStream_t)hip_stream;
  c->side_owned = false;
  return BPR_OK;
}

int bpr_stream_create(int device_id, const uint32_t* cu_mask, int32_t mask_words, void** stream_out) {
  if (stream_out == nullptr || mask_words < 0 || (mask_words > 0 && cu_mask == nullptr))
    return fail(BPR_ERR_INVALID, "bpr_stream_create: bad argument");
  BPR_HIP_CHECK(hipSetDevice(device_id));
  hipStream_t st = nullptr;
  if (mask_words == 0)
    BPR_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  else
    BPR_HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask_words, cu_mask));
  *stream_out = (void*)st;
  return BPR_OK;
}

int bpr_stream_destroy(void* hip_stream) {
  if (hip_stream == nullptr) return BPR_OK;
  BPR_HIP_CHECK(hipStreamSynchronize((hipStream_t)hip_stream));
  BPR_HIP_CHECK(hipStreamDestroy((hipStream_t)hip_stream));
  return BPR_OK;
}

int bpr_sample_adaptive(bpr_ctx* c, const int32_t* users, int64_t B, float p, uint64_t seed,
                        uint64_t offset, int32_t* neg_out, int32_t* factor_out,
                        int32_t* rank_out) {
  if (int rc = check_bound(c, "bpr_sample_adaptive")) return rc;
  if (c->indptr == nullptr) return fail(BPR_ERR_INVALID, "bpr_sample_adaptive: seen CSR not bound");
  if (!c->have_snapshot)
    return fail(BPR_ERR_INVALID, "bpr_sample_adaptive: call bpr_adaptive_refresh first");
  if (!(p > 0.f && p < 1.f)) return fail(BPR_ERR_INVALID, "bpr_sample_adaptive: p not in (0,1)");
  if (users == nullptr || neg_out == nullptr || B < 0)
    return fail(BPR_ERR_INVALID, "bpr_sample_adaptive: bad argument");
  if (int rc = vs_leave(c)) return rc;  // reads the live user rows
  if (int rc = snapshot_complete_impl(c)) return rc;  // (sample_args cannot report it)
  SampleArgs a = sample_args(c);
  a.users = users; a.n = B; a.seed = seed; a.offset = offset;
  a.neg = neg_out; a.factor_out = factor_out; a.rank_out = rank_out;
  a.inv_log1mp = inv_log1mp(p);
  return launch_sample(c, SAMPLE_ADAPTIVE, a);
}

int bpr_adaptive_pick(bpr_ctx* c, const int32_t* users, const int32_t* factor, const int32_t* rank,
                      int64_t B, int32_t* neg_out) {
  if (int rc = check_bound(c, "bpr_adaptive_pick")) return rc;
  if (c->indptr == nullptr || !c->have_snapshot)
    return fail(BPR_ERR_INVALID, "bpr_adaptive_pick: seen CSR / snapshot missing");
  if (!users || !factor || !rank || !neg_out || B < 0)
    return fail(BPR_ERR_INVALID, "bpr_adaptive_pick: bad argument");
  if (int rc = snapshot_complete_impl(c)) return rc;  // (sample_args cannot report it)
  SampleArgs a = sample_args(c);
  a.users = users; a.factor_in = factor; a.rank_in = rank; a.n = B; a.neg = neg_out;
  return launch_sample(c, SAMPLE_PICK, a);
}

int bpr_adaptive_get_snapshot(bpr_ctx* c, int32_t* order_out, float* sigma_out) {
  if (int rc = check_bound(c, "bpr_adaptive_get_snapshot")) return rc;
  if (!c->have_snapshot) return fail(BPR_ERR_INVALID, "bpr_adaptive_get_snapshot: no snapshot");
  if (int rc = snapshot_complete_impl(c)) return rc;
  if (order_out)
    BPR_HIP_CHECK(hipMemcpyAsync(order_out, c->order, sizeof(int32_t) * (size_t)c->d * c->I,
                                 hipMemcpyDeviceToDevice, c->stream));
  if (sigma_out)
    BPR_HIP_CHECK(hipMemcpyAsync(sigma_out, c->sigma, sizeof(float) * c->d,
                                 hipMemcpyDeviceToDevice, c->stream));
  return BPR_OK;
}

// ---- hot path -----------------------------------------------------------------------------------
int bpr_forward(bpr_ctx* c, const int32_t* users, const int32_t* pos, const int32_t* neg,
                int64_t B, float* out_logits_pos, float* out_logits_neg, float* out_scalars) {
  if (int rc = check_triples(c, "bpr_forward", users, pos, B)) return rc;
  if (neg == nullptr && B > 0) return fail(BPR_ERR_INVALID, "bpr_forward: neg is NULL");
  if (int rc = vs_leave(c)) return rc;
  TripleArgs a = triple_args(c);
  a.users = users; a.pos = pos; a.neg = neg; a.n = B;
  a.lpos = out_logits_pos; a.lneg = out_logits_neg; a.scalars = out_scalars;
  return launch_triples<MODE_FORWARD>(c, a, false);
}

int bpr_forward_grad(bpr_ctx* c, const int32_t* users, const int32_t* pos, const int32_t* neg,
                     int64_t B, float* out_logits_pos, float* out_logits_neg, float* out_scalars) {
  if (int rc = check_triples(c, "bpr_forward_grad", users, pos, B)) return rc;
  if (neg == nullptr && B > 0) return fail(BPR_ERR_INVALID, "bpr_forward_grad: neg is NULL");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = vs_leave(c)) return rc;
  if (int rc = ensure_strict_scratch(c)) return rc;
  TripleArgs a = triple_args(c);
  a.users = users; a.pos = pos; a.neg = neg; a.n = B;
  a.lpos = out_logits_pos; a.lneg = out_logits_neg; a.scalars = out_scalars;
  c->pending += 3 * B;
  if (c->pending > c->U + c->I) c->pending = c->U + c->I;
  return launch_triples<MODE_GRAD>(c, a, true);
}

int bpr_apply(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_apply")) return rc;
  c->keys_cut = false;
  c->bias_w_valid = false;  // writes the item_bias
  if (int rc = check_opt_state(c, "bpr_apply")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = ensure_strict_scratch(c)) return rc;
  c->step += 1;
  if (c->pending > 0) {
    ApplyArgs a = apply_args(c, c->step);
    int rc = dispatch_ge(c->G, c->E, [&](auto tag) -> int {
      using T = decltype(tag);
      const unsigned grid = (unsigned)((c->pending * T::G + 255) / 256);
      hipLaunchKernelGGL((k_apply<T::G, T::E>), dim3(grid), dim3(256), 0, c->stream, a);
      BPR_HIP_CHECK(hipGetLastError());
      return BPR_OK;
    });
    if (rc) return rc;
    c->cnt_sel ^= 1;  // the kernel cleared the other counter
    c->pending = 0;
  }
  return BPR_OK;
}

int bpr_discard_grad(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_discard_grad")) return rc;
  if (c->GP == nullptr || c->pending == 0) return BPR_OK;
  ApplyArgs a = apply_args(c, c->step);
  int rc = dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    const unsigned grid = (unsigned)((c->pending * T::G + 255) / 256);
    hipLaunchKernelGGL((k_discard<T::G, T::E>), dim3(grid), dim3(256), 0, c->stream, a);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
  if (rc) return rc;
  c->cnt_sel ^= 1;  // the kernel cleared the other counter
  c->pending = 0;
  return BPR_OK;
}

int bpr_get_grad(bpr_ctx* c, float* gP, float* gQ, float* gbias) {
  if (int rc = check_bound(c, "bpr_get_grad")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = ensure_strict_scratch(c)) return rc;
  if (gP)
    BPR_HIP_CHECK(hipMemcpyAsync(gP, c->GP, sizeof(float) * (size_t)c->U * c->d,
                                 hipMemcpyDeviceToDevice, c->stream));
  if (gQ)
    BPR_HIP_CHECK(hipMemcpyAsync(gQ, c->GQ, sizeof(float) * (size_t)c->I * c->d,
                                 hipMemcpyDeviceToDevice, c->stream));
  if (gbias)
    BPR_HIP_CHECK(hipMemcpyAsync(gbias, c->Gb, sizeof(float) * (size_t)c->I,
                                 hipMemcpyDeviceToDevice, c->stream));
  return BPR_OK;
}

static int train_stream_impl(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg,
                             int64_t n, int32_t sampler, float adaptive_p, uint64_t seed,
                             uint64_t offset, int64_t max_inflight, float* out_scalars, bool cut,
                             bool acut = false) {
  if (c != nullptr) c->acut_call = acut;  // (check_bound folds pending hot deltas for everybody else)
  const int rc0 = check_triples(c, "bpr_train_stream", users, pos, n);
  if (c != nullptr) c->acut_call = false;
  if (rc0) return rc0;
  c->keys_cut = false;  // the item table moves
  if (int rc = check_sampler(c, "bpr_train_stream", sampler, adaptive_p, neg, n)) return rc;
  if (c->opt_kind != BPR_OPT_SGD)
    return fail(BPR_ERR_UNSUPPORTED,
                "bpr_train_stream: this launch implements plain SGD only; momentum / Adam / RMSprop "
                "run through bpr_train_stream_batched (one launch per refresh period) or STRICT "
                "(bpr_forward_grad + bpr_apply)");
  if (c->pending != 0)
    return fail(BPR_ERR_INVALID, "bpr_train_stream: unapplied STRICT gradients pending");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = vs_leave(c)) return rc;
  if (n >= ((int64_t)1 << 31) || c->U * c->d >= ((int64_t)1 << 31) ||
      c->I * c->d >= ((int64_t)1 << 31))
    return fail(BPR_ERR_UNSUPPORTED,
                "bpr_train_stream: n and rows*d must be < 2^31 (32-bit offsets in the kernel)");
  StreamArgs a;
  memset(&a, 0, sizeof(a));
  a.P = c->P; a.Q = c->Q; a.bias = c->bias;
  a.indptr = c->indptr; a.indices = c->indices;
  a.order = c->order; a.sigma = c->sigma;
  if (c->meta_front != nullptr && c->keys_front_stale)
    return fail(BPR_ERR_INVALID, "bpr_train_stream: the partial snapshot in front was sorted from keys a later cut has "
                                 "overwritten; commit the pending refresh first");
  a.snap_meta = c->meta_front;
  a.snap_keys = c->meta_front != nullptr ? c->keys_front : nullptr;
  a.users = users; a.pos = pos; a.neg = neg;
  a.partials = out_scalars != nullptr ? c->dev_scalars : nullptr;
  a.seed = seed; a.offset = offset;
  a.n = (int32_t)n; a.I = (int32_t)c->I; a.d = c->d;
  a.pad_user = c->pad_user; a.pad_item = c->pad_item;
  a.run_len = c->run_len;  // 0: launch_stream picks it from the launch size and the chip's occupancy
  a.grouped = c->grouped;
  a.au = c->au; a.ai = c->ai; a.an = c->an; a.lr = c->opt.lr;
  a.iw = ItemWeights{c->w_accept, c->w_alias};
  a.inv_log1mp = sampler == BPR_NEG_ADAPTIVE ? inv_log1mp(adaptive_p) : 0.f;
  if (sampler != BPR_NEG_GIVEN) {
    if (int rc = heavy_build_impl(c)) return rc;
    a.heavy_off = c->heavy_off;
    a.heavy_bits = c->heavy_bits;
    a.heavy_T = c->heavy_T;
  }
  if (c->hot_H > 0) {
    a.hot_slot = c->hot_slot;
    a.hot_delta = c->hot_delta;
    a.hot_H = c->hot_H;
    a.hot_rmask = c->hot_R - 1;
  }
  if (acut && (c->hot_tier || c->vs_active))
    return fail(BPR_ERR_INVALID, "bpr_train_stream_acut: not under the hot tier / batched STREAM bookkeeping");
  if (acut)
    if (int rc = refresh_alloc(c)) return rc;
  return launch_stream(c, a, sampler, max_inflight, out_scalars, cut && !acut, acut);
}

int bpr_train_stream(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg, int64_t n,
                     int32_t sampler, float adaptive_p, uint64_t seed, uint64_t offset,
                     int64_t max_inflight, float* out_scalars) {
  return train_stream_impl(c, users, pos, neg, n, sampler, adaptive_p, seed, offset, max_inflight,
                           out_scalars, false);
}

int bpr_train_stream_cut(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg,
                         int64_t n, int32_t sampler, float adaptive_p, uint64_t seed, uint64_t offset,
                         int64_t max_inflight, float* out_scalars) {
  if (n <= 0)
    return fail(BPR_ERR_INVALID, "bpr_train_stream_cut: needs a non-empty launch (the cut rides "
                                 "on its epilogue)");
  return train_stream_impl(c, users, pos, neg, n, sampler, adaptive_p, seed, offset, max_inflight,
                           out_scalars, true);
}

int bpr_train_stream_acut(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg,
                          int64_t n, int32_t sampler, float adaptive_p, uint64_t seed, uint64_t offset,
                          int64_t max_inflight, float* out_scalars) {
  if (n <= 0)
    return fail(BPR_ERR_INVALID, "bpr_train_stream_acut: needs a non-empty launch");
  return train_stream_impl(c, users, pos, neg, n, sampler, adaptive_p, seed, offset, max_inflight,
                           out_scalars, true, true);
}

int bpr_hot_fold(bpr_ctx* c) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_hot_fold: ctx is NULL");
  return hot_fold_impl(c);
}

int bpr_step(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg, int64_t B,
             int32_t mode, int32_t sampler, float adaptive_p, uint64_t seed, uint64_t offset,
             float* out_logits_pos, float* out_logits_neg, float* out_scalars) {
  if (int rc = check_triples(c, "bpr_step", users, pos, B)) return rc;
  if (int rc = check_sampler(c, "bpr_step", sampler, adaptive_p, neg, B)) return rc;
  if (mode == BPR_MODE_STREAM) {
    if (out_logits_pos || out_logits_neg)
      return fail(BPR_ERR_UNSUPPORTED, "bpr_step: STREAM mode does not return logits");
    return bpr_train_stream(c, users, pos, neg, B, sampler, adaptive_p, seed, offset, 0,
                            out_scalars);
  }
  if (mode != BPR_MODE_STRICT) return fail(BPR_ERR_INVALID, "bpr_step: unknown mode");
  if (sampler != BPR_NEG_GIVEN) {
    if (neg == nullptr && B > 0)
      return fail(BPR_ERR_INVALID, "bpr_step: STRICT mode needs a neg buffer to sample into");
    int rc = sampler == BPR_NEG_UNIFORM
                 ? bpr_sample_uniform(c, users, B, seed, offset, neg)
                 : bpr_sample_adaptive(c, users, B, adaptive_p, seed, offset, neg, nullptr,
                                       nullptr);
    if (rc) return rc;
  }
  if (int rc = bpr_forward_grad(c, users, pos, neg, B, out_logits_pos, out_logits_neg,
                                out_scalars))
    return rc;
  return bpr_apply(c);
}

int bpr_train_strict(bpr_ctx* c, const int32_t* users, const int32_t* pos, int32_t* neg_scratch,
                     int64_t n, int64_t B, int32_t sampler, float adaptive_p, uint64_t seed,
                     uint64_t offset, int64_t refresh_every, float* out_scalars) {
  if (int rc = check_triples(c, "bpr_train_strict", users, pos, n)) return rc;
  if (B < 1) return fail(BPR_ERR_INVALID, "bpr_train_strict: B must be >= 1");
  if (int rc = check_sampler(c, "bpr_train_strict", sampler, adaptive_p, neg_scratch, n)) return rc;
  if (sampler != BPR_NEG_GIVEN && neg_scratch == nullptr && n > 0)
    return fail(BPR_ERR_INVALID, "bpr_train_strict: neg_scratch is NULL");
  // loss statistics: k_triples adds each batch's partial sums to the per-block scratch and ONE
  // k_sum_partials at the end adds them to out_scalars (instead of one extra launch per batch)
  const bool defer = out_scalars != nullptr && n > 0;
  const unsigned stat_blocks = grid_for(B < n ? B : n, c->G, 0);
  if (defer) {
    BPR_HIP_CHECK(hipSetDevice(c->device));
    BPR_HIP_CHECK(hipMemsetAsync(c->dev_scalars, 0, sizeof(float) * 4 * stat_blocks, c->stream));
    c->defer_stats = true;
  }
  // AdaptiveSampler.sample (neg_samplers.py:74-124): _iteration_cnt += 1; draw with the CURRENT
  // snapshot; if _iteration_cnt % every == 0: update_stats() — i.e. the snapshot is retaken after
  // the draws of that batch and BEFORE its optimizer step, and the counter lives as long as the
  // sampler (it is not reset at epoch boundaries): c->strict_iter carries it across calls.
  int rc = BPR_OK;
  for (int64_t lo = 0; lo < n && rc == BPR_OK; lo += B) {
    const int64_t b = n - lo < B ? n - lo : B;
    int32_t* neg = sampler == BPR_NEG_GIVEN ? neg_scratch + lo : neg_scratch;
    if (sampler == BPR_NEG_UNIFORM)
      rc = bpr_sample_uniform(c, users + lo, b, seed, offset + (uint64_t)lo, neg);
    else if (sampler == BPR_NEG_ADAPTIVE)
      rc = bpr_sample_adaptive(c, users + lo, b, adaptive_p, seed, offset + (uint64_t)lo, neg,
                               nullptr, nullptr);
    c->strict_iter += 1;
    if (rc == BPR_OK && refresh_every > 0 && c->strict_iter % refresh_every == 0) {
      rc = bpr_flush_lazy(c);
      if (rc == BPR_OK) rc = bpr_adaptive_refresh(c);
    }
    if (rc == BPR_OK)
      rc = bpr_forward_grad(c, users + lo, pos + lo, neg, b, nullptr, nullptr, nullptr);
    if (rc == BPR_OK) rc = bpr_apply(c);
  }
  c->defer_stats = false;
  if (rc != BPR_OK) return rc;
  if (defer) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->dev_scalars,
                       (int)stat_blocks, out_scalars);
    BPR_HIP_CHECK(hipGetLastError());
  }
  return BPR_OK;
}

int bpr_item_delta(const float* q, const float* base, float* own, float* tot, int64_t n,
                   void* hip_stream) {
  if (n < 0 || (n > 0 && (!q || !base || !own || !tot)))
    return fail(BPR_ERR_INVALID, "bpr_item_delta: bad argument");
  if (n == 0) return BPR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 1023) / 1024, 4096);
  hipLaunchKernelGGL(k_item_delta, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, q, base, own,
                     tot, n);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

int bpr_item_fold(float* q, float* base, const float* own, const float* tot, float scale,
                  int32_t rebase, int64_t n, void* hip_stream) {
  if (n < 0 || (n > 0 && (!q || !base || !own || !tot)))
    return fail(BPR_ERR_INVALID, "bpr_item_fold: bad argument");
  if (n == 0) return BPR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 1023) / 1024, 4096);
  hipLaunchKernelGGL(k_item_fold, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, q, base, own,
                     tot, scale, rebase, n);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

int bpr_item_fold_delta(float* q, float* base, float* own, float* tot, float scale, int64_t n,
                        void* hip_stream) {
  if (n < 0 || (n > 0 && (!q || !base || !own || !tot)))
    return fail(BPR_ERR_INVALID, "bpr_item_fold_delta: bad argument");
  if (n == 0) return BPR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 1023) / 1024, 4096);
  hipLaunchKernelGGL(k_item_fold_delta, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, q, base,
                     own, tot, scale, n);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

int bpr_set_hot_rows(bpr_ctx* c, int32_t hot_rows, int32_t replicas) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_hot_rows: ctx is NULL");
  if (hot_rows < 0 || hot_rows > 32768)
    return fail(BPR_ERR_INVALID, "bpr_set_hot_rows: hot_rows must be in [0, 32768]");
  if (hot_rows > 0 && replicas != 1 && replicas != 2 && replicas != 4 && replicas != 8)
    return fail(BPR_ERR_INVALID, "bpr_set_hot_rows: replicas must be 1, 2, 4 or 8");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  BPR_HIP_CHECK(hipStreamSynchronize(c->stream));
  hot_free(c);  // rebuilt by the next bpr_plan_epoch
  c->hot_rows_opt = hot_rows;
  c->hot_reps_opt = hot_rows > 0 ? replicas : 0;
  return BPR_OK;
}

int bpr_set_hot_lds(bpr_ctx* c, int32_t rows, int32_t always) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_hot_lds: ctx is NULL");
  if (rows < 0 || rows > 32767) return fail(BPR_ERR_INVALID, "bpr_set_hot_lds: rows must be in [0, 32767]");
  c->tune_hot_lds = rows;
  c->tune_hot_lds_force = always != 0;
  return BPR_OK;
}

int bpr_stream_lds_rows(bpr_ctx* c) {
  return c == nullptr ? 0 : c->last_lds_rows;
}

int bpr_set_hot_items(bpr_ctx* c, const int32_t* items_host, int32_t H, const uint32_t* counts_host) {
  if (int rc = check_bound(c, "bpr_set_hot_items")) return rc;
  if (H < 0 || H > 32768 || H >= c->I || (H > 0 && items_host == nullptr))
    return fail(BPR_ERR_INVALID, "bpr_set_hot_items: need 0 <= H <= min(32768, I - 1) item ids");
  if (c->hot_tier) return fail(BPR_ERR_INVALID, "bpr_set_hot_items: the hot tier is on (bpr_hot_tier_end first)");
  std::vector<char> seen((size_t)c->I, 0);
  for (int k = 0; k < H; ++k) {
    const int32_t it = items_host[k];
    if (it < 0 || it >= c->I || it == c->pad_item || seen[it])
      return fail(BPR_ERR_INVALID, "bpr_set_hot_items: ids must be distinct item rows, not the pad row");
    seen[it] = 1;
  }
  BPR_HIP_CHECK(hipSetDevice(c->device));
  BPR_HIP_CHECK(hipStreamSynchronize(c->stream));
  return hot_set_items_impl(c, items_host, H, counts_host);
}

int bpr_hot_rows(bpr_ctx* c, int32_t* rows_host) {
  if (c == nullptr || rows_host == nullptr) return fail(BPR_ERR_INVALID, "bpr_hot_rows: NULL argument");
  *rows_host = c->hot_H;
  return BPR_OK;
}

int bpr_hot_tier_begin(bpr_ctx* c, float* hot_base) {
  if (int rc = check_bound(c, "bpr_hot_tier_begin")) return rc;
  if (c->hot_H <= 0 || !c->hot_explicit)
    return fail(BPR_ERR_INVALID, "bpr_hot_tier_begin: no hot set (bpr_set_hot_items first: the ranks "
                                 "must agree on it)");
  if (hot_base == nullptr) return fail(BPR_ERR_INVALID, "bpr_hot_tier_begin: hot_base is NULL");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  const int64_t n = (int64_t)c->hot_H * c->d;
  hipLaunchKernelGGL(k_hot_gather, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256),
                     0, c->stream, c->Q, c->hot_items, c->hot_canon, hot_base, c->hot_H, c->d);
  BPR_HIP_CHECK(hipGetLastError());
  c->hot_tier = true;
  return BPR_OK;
}

int bpr_hot_tier_end(bpr_ctx* c) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_hot_tier_end: ctx is NULL");
  if (int rc = flush_deferred_stats(c)) return rc;
  c->hot_tier = false;  // (the caller has folded everything: bpr_hot_exchange with cut = 1 last)
  return BPR_OK;
}

int bpr_hot_exchange(bpr_ctx* c, float* hot_base, float* tot, int32_t fold_prev, int32_t cut,
                     float* cold_base) {
  if (int rc = check_bound(c, "bpr_hot_exchange")) return rc;
  if (!c->hot_tier) return fail(BPR_ERR_INVALID, "bpr_hot_exchange: the hot tier is off (bpr_hot_tier_begin)");
  if (hot_base == nullptr || tot == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_hot_exchange: NULL buffer");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = flush_deferred_stats(c)) return rc;  // a cut launch that was not followed by bpr_sync_cut
  c->keys_cut = false;  // the item table moves
  HotStepArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = c->Q; a.delta = c->hot_delta; a.hot_items = c->hot_items; a.canon = c->hot_canon;
  a.hb = hot_base; a.tot = tot; a.cold_base = cold_base;
  a.H = c->hot_H; a.R = c->hot_R; a.d = c->d; a.fold_prev = fold_prev != 0; a.cut = cut != 0;
  const int64_t n = (int64_t)c->hot_H * c->d;
  hipLaunchKernelGGL(k_hot_step, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0,
                     c->stream, a);
  BPR_HIP_CHECK(hipGetLastError());
  return BPR_OK;
}

int bpr_set_heavy_users(bpr_ctx* c, int32_t threshold, int64_t max_bytes) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_heavy_users: ctx is NULL");
  if (threshold < -1 || max_bytes < 0)
    return fail(BPR_ERR_INVALID, "bpr_set_heavy_users: threshold >= -1, max_bytes >= 0");
  c->heavy_T_opt = threshold;
  if (max_bytes > 0) c->heavy_max_bytes = max_bytes;
  c->heavy_for = nullptr;  // rebuilt by the next sampling STREAM launch
  return BPR_OK;
}

int bpr_sync_cut(bpr_ctx* c, float* hot_base, float* hot_tot, int32_t hot_fold_prev, float* cold_base,
                 float* cold_own, float* cold_tot, float scale, int32_t cold_mode) {
  if (int rc = check_bound(c, "bpr_sync_cut")) return rc;
  if (cold_mode < 0 || cold_mode > 2 || (cold_mode != 0 && (!cold_base || !cold_own || !cold_tot)))
    return fail(BPR_ERR_INVALID, "bpr_sync_cut: cold_mode 0 | 1 | 2 with base / own / tot");
  if (c->hot_tier && (hot_base == nullptr || hot_tot == nullptr))
    return fail(BPR_ERR_INVALID, "bpr_sync_cut: the hot tier is on: hot_base / hot_tot needed");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (int rc = refresh_alloc(c)) return rc;
  if (c->ev_keys == nullptr) {
    BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_keys, hipEventDisableTiming));
    BPR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming));
  }
  SyncCutArgs a;
  memset(&a, 0, sizeof(a));
  const bool hot = c->hot_H > 0;
  a.e.partials = c->dev_scalars;
  a.e.n_blocks = c->defer_pending ? c->defer_blocks : 0;
  a.e.out = c->defer_pending ? c->defer_out : nullptr;
  a.e.Q = c->Q; a.e.delta = c->hot_delta; a.e.hot_slot = hot ? c->hot_slot : nullptr;
  a.e.T = c->keysT; a.e.sig_acc = c->sig_acc;
  a.e.H = hot ? c->hot_H : 0; a.e.R = c->hot_R; a.e.d = c->d; a.e.I = (int32_t)c->I;
  a.canon = c->hot_canon; a.hb = hot_base; a.htot = hot_tot;
  a.base = cold_base; a.own = cold_own; a.tot = cold_tot; a.scale = scale;
  a.hot_tier = c->hot_tier ? 1 : 0; a.hot_fold_prev = hot_fold_prev != 0; a.cold_mode = cold_mode;
  dim3 eg((unsigned)((c->I + 31) / 32), (unsigned)((c->d + 31) / 32) + 1u);
  hipExtLaunchKernelGGL(k_sync_cut, eg, dim3(256), 0, c->stream, nullptr, c->ev_keys, 0, a);
  BPR_HIP_CHECK(hipGetLastError());
  c->defer_pending = false;
  c->keys_cut = true;   // the next bpr_adaptive_refresh_begin only queues the sort
  c->keys_event = true;
  return BPR_OK;
}

int bpr_set_tuning(bpr_ctx* c, const char* key, int32_t value) {
  if (c == nullptr || key == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_tuning: NULL argument");
  const std::string k = key;
  if (k == "seen" && value >= 0 && value <= 3) c->tune_seen = value;
  else if (k == "vs_direct" && value >= -1 && value <= 1) c->tune_vs_direct = value;
  else if (k == "adam_closed" && (value == 0 || value == 1)) c->tune_adam_closed = value;
  else if (k == "partial_snapshot" && (value == 0 || value == 1)) c->tune_partial = value;
  else if (k == "binned_sort" && (value == 0 || value == 1)) c->tune_binned = value;
  else if (k == "binned_split" && value >= 0 && value <= 4) c->tune_binned_split = value;
  else if (k == "partial_target" && value >= 1 && value <= 1024) c->partial_target = value;
  else if (k == "refresh_sub" && (value == 0 || value == 1 || value == 2 || value == 4)) c->tune_refresh_sub = value;
  else if (k == "lds_block" && value >= 0 && value <= 1024 && value % 64 == 0) c->tune_lds_block = value;
  else if (k == "lds_tail" && value >= 0 && value <= 50) c->tune_lds_tail = value;
  else if (k == "plan_input_sorted" && (value == 0 || value == 1)) c->tune_plan_sorted = value;
  else if (k == "acut_fold" && (value == 0 || value == 1)) c->tune_acut_fold = value;
  else return fail(BPR_ERR_INVALID, "bpr_set_tuning: unknown key or value out of range");
  c->stream_occ.clear();
  return BPR_OK;
}

int bpr_set_bias_tracking(bpr_ctx* c, int32_t on) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_bias_tracking: ctx is NULL");
  c->bias_track = on != 0;
  c->bias_w_valid = false;
  return BPR_OK;
}

int bpr_bias_written(bpr_ctx* c) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_bias_written: ctx is NULL");
  c->bias_w_valid = false;
  return BPR_OK;
}

int bpr_stream_run_len(bpr_ctx* c) {
  return c == nullptr ? 0 : c->last_run_len;
}

int bpr_set_stream_opts(bpr_ctx* c, int32_t grouped_by_user, int32_t run_len) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_set_stream_opts: ctx is NULL");
  if (run_len < 0 || run_len > 30)
    return fail(BPR_ERR_INVALID, "bpr_set_stream_opts: run_len must be 0 (by launch size) or in [1, 30] (one lane of a "
                                 "32-lane group per triple of the run, plus two neighbours)");
  c->grouped = grouped_by_user != 0;
  c->run_len = run_len;
  return BPR_OK;
}

int bpr_plan_epoch(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n,
                   int64_t chunk, uint64_t seed, int32_t* users_out, int32_t* pos_out) {
  if (int rc = check_bound(c, "bpr_plan_epoch")) return rc;
  if (n < 0 || chunk < 1 || (n > 0 && (!users_in || !pos_in || !users_out || !pos_out)))
    return fail(BPR_ERR_INVALID, "bpr_plan_epoch: bad argument");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  return plan_epoch_impl(c, users_in, pos_in, n, chunk, seed, users_out, pos_out);
}

int bpr_plan_chunk(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n, int64_t chunk,
                   uint64_t seed, int64_t index, int32_t* users_out, int32_t* pos_out, int32_t on_side) {
  if (int rc = check_bound(c, "bpr_plan_chunk")) return rc;
  if (n < 1 || n >= ((int64_t)1 << 31) || chunk < 1 || index < 0 || !users_in || !pos_in || !users_out ||
      !pos_out)
    return fail(BPR_ERR_INVALID, "bpr_plan_chunk: bad argument");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->hot_key_ptr != pos_in || c->hot_key_n != n) {  // new training set: measure popularity (as
    if (int rc = hot_build_impl(c, pos_in, n)) return rc;  // bpr_plan_epoch does)
  }
  if (!on_side) return plan_chunk_impl(c, users_in, pos_in, n, chunk, seed, index, users_out, pos_out, c->stream);
  // behind the split refresh's sort on the side stream; what bpr_adaptive_refresh_commit waits for
  // then covers this chunk too (one event, no extra packet on the launch stream)
  if (c->side == nullptr || !c->refresh_pending)
    return fail(BPR_ERR_INVALID, "bpr_plan_chunk: on_side needs a split refresh in flight "
                                 "(bpr_adaptive_refresh_begin first)");
  if (int rc = plan_chunk_impl(c, users_in, pos_in, n, chunk, seed, index, users_out, pos_out, c->side))
    return rc;
  BPR_HIP_CHECK(hipEventRecord(c->ev_sorted, c->side));
  return BPR_OK;
}

int bpr_flush_lazy(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_flush_lazy")) return rc;
  c->keys_cut = false;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->vs_active) return vs_flush(c, true, true);  // batched STREAM bookkeeping is live
  return strict_flush_impl(c);
}

int bpr_flush_items(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_flush_items")) return rc;
  c->keys_cut = false;
  c->bias_w_valid = false;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->vs_active) return vs_flush(c, false, true);
  if (c->opt_kind == BPR_OPT_SGD || c->GP == nullptr || c->step == 0) return BPR_OK;
  if (int rc = check_opt_state(c, "bpr_flush_items")) return rc;
  ApplyArgs a = apply_args(c, c->step);
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_flush_lazy<T::G, T::E>), dim3(grid_for(c->I, T::G, 0)), dim3(256), 0,
                       c->stream, a, 1);
    BPR_HIP_CHECK(hipGetLastError());
    return BPR_OK;
  });
}

int bpr_get_step_host(bpr_ctx* c, int64_t* step_host) {
  if (c == nullptr || step_host == nullptr)
    return fail(BPR_ERR_INVALID, "bpr_get_step_host: NULL argument");
  *step_host = c->step;
  return BPR_OK;
}

int bpr_set_step(bpr_ctx* c, int64_t step) {
  if (c == nullptr || step < 0) return fail(BPR_ERR_INVALID, "bpr_set_step: bad argument");
  if (c->vs_active) {
    BPR_HIP_CHECK(hipSetDevice(c->device));
    if (int rc = vs_leave(c)) return rc;
  }
  // rows are assumed flushed at `step` (a checkpoint is written after bpr_flush_lazy)
  c->step = step;
  if (c->lastP != nullptr) {
    BPR_HIP_CHECK(hipSetDevice(c->device));
    BPR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->lastP, (int)step, (size_t)c->U, c->stream));
    BPR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->lastQ, (int)step, (size_t)c->I, c->stream));
  }
  return BPR_OK;
}

int bpr_set_sampler_iter(bpr_ctx* c, int64_t iteration) {
  if (c == nullptr || iteration < 0)
    return fail(BPR_ERR_INVALID, "bpr_set_sampler_iter: bad argument");
  c->strict_iter = iteration;
  return BPR_OK;
}

int bpr_timing_enable(bpr_ctx* c, int32_t on) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_timing_enable: ctx is NULL");
  if (int rc = drain_timing(c)) return rc;
  c->timing = on != 0;
  c->timing_stride = on > 1 ? on : 1;
  c->timing_seen = 0;
  c->timed_ms = 0.0;
  c->timed_launches = 0;
  return BPR_OK;
}

int bpr_timing_read_host(bpr_ctx* c, double* avg_ms_host, int64_t* launches_host) {
  if (c == nullptr) return fail(BPR_ERR_INVALID, "bpr_timing_read_host: ctx is NULL");
  if (int rc = drain_timing(c)) return rc;
  if (avg_ms_host) *avg_ms_host = c->timed_launches ? c->timed_ms / (double)c->timed_launches : 0.0;
  if (launches_host) *launches_host = c->timed_launches;
  return BPR_OK;
}

}  // extern "C"
