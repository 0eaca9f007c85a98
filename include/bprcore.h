/*
 * bprcore.h — C ABI of libbprcore.so, the MI355X (gfx950) BPR-MF training engine.
 *
 * The reference (Nemexur/revisit-bpr) has no FFI: its hot path is a chain of stock torch ops
 * called from Python.  Each entry point below therefore cites the reference *Python call site*
 * it replaces (paths relative to the reference tree).  The binding a maintainer adds on the
 * reference side is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions across the boundary.
 *   - Every pointer argument is a DEVICE pointer (HBM) unless the name ends in `_host`.
 *   - The caller owns every buffer it passes in.  The library allocates only private scratch
 *     inside `bpr_ctx` (grad accumulators of the strict path, adaptive-sampler order arrays,
 *     scalar reduction slots) and frees it in bpr_ctx_destroy.
 *   - Return value: 0 (BPR_OK) on success, negative bpr_status on failure; the message for the
 *     calling thread is available from bpr_last_error().  The library never aborts.
 *   - All calls are asynchronous with respect to the host and are ordered on the HIP stream the
 *     ctx was created with (pass torch's current stream) — except bpr_ctx_create/destroy and the
 *     `*_host` getters, which synchronise that stream.
 *   - A ctx is single-threaded and tied to one HIP device + stream.
 *   - Embedding dim d in [1, 1024]; tables row-major fp32, 4-byte aligned.
 *   - Row 0 of both tables is the pad row (reference: nn.Embedding(padding_idx=0)); item ids
 *     handed to the library are in [0, I), user ids in [0, U).  Ids are int32.
 */
#ifndef BPRCORE_H
#define BPRCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPRCORE_VERSION 100 /* 0.1.0 */

typedef struct bpr_ctx bpr_ctx;

typedef enum bpr_status {
  BPR_OK = 0,
  BPR_ERR_INVALID = -1,     /* bad argument / state (message says which) */
  BPR_ERR_HIP = -2,         /* a HIP runtime call failed */
  BPR_ERR_UNSUPPORTED = -3, /* valid request the engine does not implement (e.g. d > 1024) */
  BPR_ERR_NOMEM = -4
} bpr_status;

/* torch.optim.* kinds used by the reference configs
 * (configs/RQ1/ours.yaml.j2:116-119, configs/RQ2/optimizers/, configs/RQ3/time-split/ada-sampling-adam.yaml.j2:169-175) */
typedef enum bpr_opt_kind {
  BPR_OPT_SGD = 0,      /* torch.optim.SGD(lr)                               */
  BPR_OPT_MOMENTUM = 1, /* torch.optim.SGD(lr, momentum, dampening, nesterov) */
  BPR_OPT_ADAM = 2,     /* torch.optim.Adam(lr, betas, eps)                   */
  BPR_OPT_RMSPROP = 3   /* torch.optim.RMSprop(lr, alpha, eps, momentum)      */
} bpr_opt_kind;

typedef struct bpr_opt_params {
  float lr;
  float momentum;  /* MOMENTUM; RMSPROP (0 = none) */
  float dampening; /* MOMENTUM */
  int32_t nesterov;
  float beta1, beta2; /* ADAM */
  float eps;          /* ADAM (1e-8), RMSPROP (1e-8) */
  float alpha;        /* RMSPROP smoothing constant (0.99) */
} bpr_opt_params;

/* How a step treats rows that occur several times in the same batch / chunk. */
typedef enum bpr_mode {
  /* Exact reference mini-batch semantics (revisit_bpr/models/bpr/model.py:40-68 + trainer.py:76-81):
   * all B gradients evaluated at the pre-step parameters, duplicates accumulated, ONE optimizer
   * step.  Two kernels (grad-accumulate, apply).  Launch-bound at B=256; used for parity. */
  BPR_MODE_STRICT = 0,
  /* Throughput path: one launch consumes a whole chunk of triples; every triple reads the latest
   * rows it can see and applies its update immediately with per-element fp32 atomics
   * (asynchronous SGD, bounded staleness).  SGD only; the other optimizers have their own
   * single-launch path, bpr_train_stream_batched. */
  BPR_MODE_STREAM = 1
} bpr_mode;

typedef enum bpr_sampler_kind {
  BPR_NEG_GIVEN = 0,   /* caller passes `neg` */
  BPR_NEG_UNIFORM = 1, /* revisit_bpr/modules/neg_samplers.py:31-37 (UniformSampler.sample) */
  BPR_NEG_ADAPTIVE = 2 /* revisit_bpr/modules/neg_samplers.py:74-124 (AdaptiveSampler.sample) */
} bpr_sampler_kind;

/* out_scalars layout written by bpr_forward_grad / bpr_step / bpr_train_stream (fp32, ADDED to —
 * zero it first):  [0] bpr_loss = Σ −logσ(x)   [1] l2_reg   [2] Σ|x|   [3] number of triples */
#define BPR_SCALARS 4

/* ---- lifecycle ------------------------------------------------------------------------- */
int bpr_version(void);
const char* bpr_last_error(void);
/* hip_stream: a hipStream_t (may be NULL = default stream). */
int bpr_ctx_create(bpr_ctx** out, int device_id, void* hip_stream);
int bpr_ctx_destroy(bpr_ctx* ctx);
int bpr_set_stream(bpr_ctx* ctx, void* hip_stream);

/* ---- model state (reference: MF parameters, revisit_bpr/models/bpr/model.py:97-116) ------ */
/* P [U,d], Q [I,d] row-major fp32, updated IN PLACE; item_bias [I] or NULL (MF(item_bias=...)).
 * pad_user / pad_item: the nn.Embedding padding_idx of each table (−1 = none): gradient of that
 * embedding row is dropped, as torch's embedding backward does. */
int bpr_bind_tables(bpr_ctx* ctx, float* P, int64_t U, float* Q, int64_t I, int32_t d,
                    float* item_bias, int32_t pad_user, int32_t pad_item);

/* Seen-items CSR over users: indptr [U+1] int64, indices [nnz] int32 sorted ascending inside each
 * row, no duplicates, no item 0.  Replaces the padded `seen_items` [B,S] batch tensor
 * (experiments/bpr/dataset.py:142-190, example.py:33-68) for on-device sampling.  The first
 * sampling STREAM launch after a bind derives private scratch from it (I-bit seen bitmaps in HBM
 * for users with more than 256 seen items: one-time, synchronous): bind again after changing the
 * arrays' contents. */
int bpr_bind_seen_csr(bpr_ctx* ctx, const int64_t* indptr, const int32_t* indices);

/* Model.regularization alphas after the resolution rules of model.py:74-86 were applied by the host. */
int bpr_set_reg(bpr_ctx* ctx, float alpha_user, float alpha_item, float alpha_neg);

/* torch.optim hyper-parameters (read from optimizer.param_groups by the host shim each step). */
int bpr_set_optimizer(bpr_ctx* ctx, int32_t kind, const bpr_opt_params* params);
/* Optimizer state, same shapes as the tables (caller-owned so checkpoints keep working):
 *   MOMENTUM: m_* = momentum_buffer;  ADAM: m_* = exp_avg, v_* = exp_avg_sq;  RMSPROP: v_* = square_avg
 *   (+ m_* = momentum_buffer when momentum > 0).
 * *_bias may be NULL when there is no item_bias.  All zero-initialised by the caller. */
int bpr_bind_opt_state(bpr_ctx* ctx, float* m_P, float* v_P, float* m_Q, float* v_Q,
                       float* m_bias, float* v_bias);

/* ---- negative sampling -------------------------------------------------------------------- */
/* Draw one negative per (user) uniformly over items the user has not seen, never item 0.
 * Randomness: Philox4x32-10 keyed by `seed`, counter = offset + position in the batch, so a triple's
 * negative does not depend on launch geometry or GPU count.  neg_out [B] int32. */
int bpr_sample_uniform(bpr_ctx* ctx, const int32_t* users, int64_t B, uint64_t seed,
                       uint64_t offset, int32_t* neg_out);

/* Item weights of the uniform sampler — BPRExperiment._static_sampling with
 * item_counts ** neg_sampling_alpha (experiments/bpr/exp.py:85-91, 282-293): P(negative = i) =
 * w_i / (sum of w over the user's unseen items), w_0 = 0.  The caller passes a Walker alias table
 * over items 1..I-1 (accept [I] fp32, alias [I] int32, entry 0 unused; the host shim builds it):
 * candidate = column c if frac(r (I-1) / 2^32) < accept[c] else alias[c]; seen candidates are
 * rejected as before.  NULL, NULL = uniform (the default).  Affects every uniform draw of the ctx
 * (bpr_sample_uniform, bpr_step, bpr_train_*).
 * Deviation: when 4,096 candidates in a row were all seen (a user who has seen nearly every item)
 * the exact fallback picks by rank UNIFORMLY among the user's unseen items — it does not carry the
 * weights, the reference's multinomial over the masked weights would (neg_samplers.py:31-37).  The
 * oracle restates the same rule; the chance of reaching the fallback is (seen weight share)^4096. */
int bpr_bind_item_weights(bpr_ctx* ctx, const float* accept, const int32_t* alias);

/* AdaptiveSampler.update_stats (neg_samplers.py:126-132): snapshot the item table as per-factor
 * descending item orders (private scratch, d*I int32) and sigma_f = unbiased std over rows 1.. */
int bpr_adaptive_refresh(bpr_ctx* ctx);
/* The same refresh in two halves, so that the sort does not stand between two STREAM launches
 * (the reference sorts inline: update_stats is called from sample(), neg_samplers.py:122-123).
 * _begin: the snapshot's keys are cut from the item table NOW, in the ctx stream's order — that
 * instant is the snapshot's point in time — and their per-factor sort is queued on the ctx's SIDE
 * stream; the ctx stream does not wait, and every sampler keeps reading the previous snapshot.
 * _commit: the ctx stream waits for that sort; launches after it read the new snapshot.
 * bpr_adaptive_refresh == _begin + _commit with nothing in between.  One split refresh may be
 * pending at a time (_begin while pending: BPR_ERR_INVALID; bpr_adaptive_refresh while pending
 * is refused as well — commit first).  What the caller puts between the two calls decides the
 * snapshot's age: `commit; begin; bpr_train_stream(chunk)` per chunk makes every chunk sample from
 * the item table as it was one chunk earlier (DESIGN.md section 4.3 holds the parity evidence). */
int bpr_adaptive_refresh_begin(bpr_ctx* ctx);
int bpr_adaptive_refresh_commit(bpr_ctx* ctx);
/* *pending_host (HOST pointer) = 1 while a split refresh awaits its commit. */
int bpr_adaptive_refresh_pending(bpr_ctx* ctx, int32_t* pending_host);
/* PARTIAL snapshots (r5, opt-in: bpr_set_tuning(ctx, "partial_snapshot", 1); "partial_target" = keys aimed at
 * per exact end, 1..1024, default 640).  All the sampler ever reads of a column is rank Geometric(p) +
 * seen-skips from either end (revisit_bpr/modules/neg_samplers.py:90-121), so the split refresh
 * (bpr_adaptive_refresh_begin) sorts only the two ends of every column exactly and BUCKETS the middle
 * (equi-depth bins cut along a coarse histogram of the column, bins in order): 39 us per 20 k-key column
 * instead of 98.  STREAM launches (bpr_train_stream*) read such a snapshot directly — a walk that leaves an
 * exact end finishes inside one bin by taking its <= 64 keys out in order, the same negative as from the
 * fully sorted column, triple for triple — every other reader (bpr_sample_adaptive, bpr_adaptive_pick, the
 * batched STREAM kernel, bpr_adaptive_get_snapshot / _snapshot_ptrs) first has it sorted whole, in place.
 * Columns of 2,048 .. 24,576 items; others — and a column with more than 64 equal-ranked keys in a bin, or an
 * end over 1,024 keys — are sorted whole as before.  Worth +1.6 % on the metric's configuration with the
 * sorter on 32 CUs (the finish costs the launch a wave of occupancy); off by default.
 * *partial_host = 1 while the snapshot the samplers read is a partial one. */
int bpr_adaptive_snapshot_partial(bpr_ctx* ctx, int32_t* partial_host);
/* The refresh SHARDED over the ranks of a multi-GPU job: every rank holds a replica of the item
 * table, so instead of every rank sorting all d columns (an Amdahl term: the sort does not shrink
 * with the number of ranks) rank r sorts columns [f_lo, f_hi) only — _part cuts the keys and
 * sorts those columns into the BACK snapshot —, the caller gathers every rank's columns into the
 * back snapshot of every rank (order [d, I] int32 and sigma [d] fp32: _snapshot_ptrs hands out the
 * device pointers, back != 0 for the back snapshot; an all-gather over RCCL in
 * revisit_bpr/distributed.py) and _publish makes it the snapshot the samplers read.  All in the ctx
 * stream's order.  The reference has no multi-device sampler to mirror. */
int bpr_adaptive_refresh_part(bpr_ctx* ctx, int32_t f_lo, int32_t f_hi);
int bpr_adaptive_refresh_publish(bpr_ctx* ctx);
int bpr_adaptive_snapshot_ptrs(bpr_ctx* ctx, int32_t back, void** order_host, void** sigma_host);
/* ---- multi-GPU inside the library (the reference's DDP launcher, experiments/launcher.py:
 * 35-73, is never enabled by a config, so there is no reference behaviour to mirror) ------------------
 * One RCCL communicator per ctx (librccl.so is opened at run time: no link-time dependency).
 * bpr_comm_unique_id: rank 0 fills BPR_COMM_ID_BYTES bytes of HOST memory (ncclGetUniqueId) and ships
 * them to the other ranks out of band; bpr_comm_init: every rank, collectively (ncclCommInitRank);
 * it also cuts the reconciliation BASE from the item table as it is now (identical on every rank).
 * bpr_item_sync: the steady-state step of the item-table reconciliation — fold the all-reduced
 * deltas of the reconciliation in flight into the live replica (one period late), cut this rank's
 * new delta, all-reduce it on the communicator's own stream under whatever the ctx stream does next
 * (Q <- Q_base + sum_r (Q_r - Q_base); the base is updated from the all-reduced sum only, so it
 * stays bit-identical on every rank).  bpr_item_sync_finish folds the last one in.  With a
 * communicator of world > 1, bpr_adaptive_refresh sorts d / world factors per rank and all-gathers
 * the orders (bpr_adaptive_refresh_part / _publish) when world divides d.
 * revisit_bpr/distributed.py runs the same protocol through torch.distributed (RCCL or gloo). */
#define BPR_COMM_ID_BYTES 128
int bpr_comm_unique_id(void* id_host);
int bpr_comm_init(bpr_ctx* ctx, const void* id_host, int32_t rank, int32_t world);
/* bpr_comm_destroy closes a hot tier still open (folds the exchange in flight and this rank's uncut deltas into the
 * item table) — call it while the tables are alive; bpr_ctx_destroy frees the communicator too but writes NOTHING
 * into the caller's tables (they may be gone by then). */
int bpr_comm_destroy(bpr_ctx* ctx);
int bpr_item_sync(bpr_ctx* ctx);
int bpr_item_sync_finish(bpr_ctx* ctx);
/* The bases follow the tables when they are re-bound (another shape or buffer).  A table overwritten
 * IN PLACE (checkpoint restore, restore-best) is invisible to the library: call this afterwards,
 * on every rank, with identical tables — nothing may be in flight that should still be applied. */
int bpr_item_sync_rebase(bpr_ctx* ctx);
/* Two tiers inside the library (see bpr_hot_exchange below): bpr_comm_hot_tier sets the hot set
 * (the same list on every rank), allocates the exchange buffers and turns the tier on;
 * bpr_hot_sync after every STREAM launch (or sub-launch) folds the hot exchange in flight, cuts the
 * launch's hot deltas and all-reduces them on the communicator's stream; bpr_item_sync stays the
 * per-period step of the cold rows (call it right after bpr_hot_sync); bpr_item_sync_finish folds
 * both.  Every collective of the ctx's communicator runs on the communicator's stream in program
 * order (bpr_adaptive_refresh's all-gather included: with a communicator it IS collective — every
 * rank must call it the same number of times, in the same place of the sequence). */
int bpr_comm_hot_tier(bpr_ctx* ctx, const int32_t* items_host, int32_t H, const uint32_t* counts_host);
int bpr_hot_sync(bpr_ctx* ctx);
/* The side stream of the split refresh (a hipStream_t of the ctx's device; NULL = a plain
 * non-blocking stream created by the library on first use).  The caller keeps ownership. */
int bpr_set_side_stream(bpr_ctx* ctx, void* hip_stream);
/* Streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask): cu_mask holds
 * mask_words 32-bit words, bit i of the mask = CU i in the driver's numbering (on MI300-class
 * parts bits are dealt round-robin over the 8 XCDs, so a contiguous bit range takes equally from
 * each); mask_words = 0 creates a plain non-blocking stream.  The STREAM kernel is bound by the L2
 * atomic units and leaves most CU cycles idle, so the split refresh's sort runs beside it on a
 * disjoint CU set: create two complementary streams, hand one to bpr_ctx_create / bpr_set_stream
 * and the other to bpr_set_side_stream.  No reference counterpart (the reference has no streams). */
int bpr_stream_create(int device_id, const uint32_t* cu_mask, int32_t mask_words, void** stream_out);
int bpr_stream_destroy(void* hip_stream);
/* AdaptiveSampler.sample: factor ~ |p_uf|·sigma_f, rank ~ Geometric(p) clamped to #unseen,
 * orientation by sign(p_uf), pick the rank-th unseen item of the snapshot order.
 * factor_out / rank_out (int32 [B], nullable) expose the intermediate draws for parity tests;
 * rank_out is the 0-based rank counted from the top, as in neg_samplers.py:96-100. */
int bpr_sample_adaptive(bpr_ctx* ctx, const int32_t* users, int64_t B, float p, uint64_t seed,
                        uint64_t offset, int32_t* neg_out, int32_t* factor_out, int32_t* rank_out);
/* Deterministic half of AdaptiveSampler.sample (neg_samplers.py:109-121): given factor and rank
 * (0-based from the top) return the rank-th unseen item in the snapshot order of that factor. */
int bpr_adaptive_pick(bpr_ctx* ctx, const int32_t* users, const int32_t* factor,
                      const int32_t* rank, int64_t B, int32_t* neg_out);
/* Copy the snapshot to caller buffers (either may be NULL): order [d*I] int32, sigma [d] fp32. */
int bpr_adaptive_get_snapshot(bpr_ctx* ctx, int32_t* order_out, float* sigma_out);

/* ---- the hot path ------------------------------------------------------------------------- */
/* BPR.forward train branch (model.py:48-68): logits_pos/neg [B] (nullable), scalars as above. No
 * parameter is modified.  Used by the eval-free "forward only" callers and by parity tests. */
int bpr_forward(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, const int32_t* neg,
                int64_t B, float* out_logits_pos, float* out_logits_neg, float* out_scalars);

/* STRICT phase A — BPR.forward + loss.backward() (model.py:48-68, trainer.py:76): as bpr_forward,
 * and accumulates the per-row gradients of the batch into the ctx's private accumulators. */
int bpr_forward_grad(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, const int32_t* neg,
                     int64_t B, float* out_logits_pos, float* out_logits_neg, float* out_scalars);
/* STRICT phase B — optimizer.step(); optimizer.zero_grad() (trainer.py:79-81): one optimizer step
 * on every row touched since the last apply, then clears the accumulators. */
int bpr_apply(bpr_ctx* ctx);
/* Drop accumulated gradients without applying them (a forward that was never stepped). */
int bpr_discard_grad(bpr_ctx* ctx);
/* Copy accumulated gradients to dense caller buffers for tests (any may be NULL):
 * gP [U,d], gQ [I,d], gbias [I].  Untouched rows read as zero. */
int bpr_get_grad(bpr_ctx* ctx, float* gP, float* gQ, float* gbias);

/* One reference training iteration (example.py:172-180): sample (if sampler != GIVEN) →
 * forward → backward → optimizer step.  mode STRICT = phases A+B; mode STREAM = one fused launch
 * (SGD only).  `neg` is read when sampler == BPR_NEG_GIVEN, otherwise (if non-NULL) it receives
 * the sampled negatives.  seed/offset as in bpr_sample_uniform; adaptive_p = Geometric p. */
int bpr_step(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, int32_t* neg, int64_t B,
             int32_t mode, int32_t sampler, float adaptive_p, uint64_t seed, uint64_t offset,
             float* out_logits_pos, float* out_logits_neg, float* out_scalars);

/* STRICT epoch driver — `for batch in loader: sample; forward; backward; step` (example.py:172-180,
 * trainer.py:64-83) with the loop on the host side of the library, so that the reference's exact
 * mini-batch semantics (any optimizer) cost three kernel launches per batch and no interpreter
 * time.  users/pos [n] are the epoch's triple stream (already shuffled); batches of B consecutive
 * triples; neg_scratch [>= B] int32 receives each batch's negatives (sampler != GIVEN) or, for
 * BPR_NEG_GIVEN, is the full [n] negative stream.  refresh_every > 0 calls bpr_adaptive_refresh
 * at every refresh_every-th batch exactly where AdaptiveSampler.sample does
 * (neg_samplers.py:75,122-123): after that batch's negatives are drawn and before its optimizer
 * step; the batch counter belongs to the ctx and keeps counting across calls (epochs), as the
 * sampler's _iteration_cnt does (bpr_set_sampler_iter rewinds it).  With lazy optimizers the
 * refresh flushes first.  out_scalars accumulates over the epoch. */
int bpr_train_strict(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, int32_t* neg_scratch,
                     int64_t n, int64_t B, int32_t sampler, float adaptive_p, uint64_t seed,
                     uint64_t offset, int64_t refresh_every, float* out_scalars);

/* STREAM throughput entry (train_one_epoch's inner loop, example.py:172-180, over n triples in ONE
 * launch): users/pos are the (already shuffled) triple stream resident in HBM.  `max_inflight`
 * bounds how many triples are processed concurrently (0 = fill the chip); see DESIGN.md §staleness.
 * A model's item_bias is read and updated, for the duration of the launch, in a table of the
 * library's own (one item per 128-byte line: DESIGN.md §9.5), filled from the bound vector in stream
 * order before the launch and written back after it: the bound vector is current once the launch
 * has completed on the ctx stream, not while it runs. */
int bpr_train_stream(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, int32_t* neg,
                     int64_t n, int32_t sampler, float adaptive_p, uint64_t seed, uint64_t offset,
                     int64_t max_inflight, float* out_scalars);
/* The same launch, whose epilogue ALSO cuts the keys of the next adaptive snapshot from the item
 * table as the launch leaves it — the first half of bpr_adaptive_refresh_begin, done in the pass
 * that folds the hot rows (one kernel and one kernel boundary less between two launches).  The
 * next bpr_adaptive_refresh_begin then only queues the sort, provided the item table was not
 * touched in between: every ctx call that moves it drops the cut, but writes the library cannot
 * see (bpr_item_fold*, the caller's own kernels) must not happen there.  n must be > 0. */
int bpr_train_stream_cut(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, int32_t* neg,
                         int64_t n, int32_t sampler, float adaptive_p, uint64_t seed,
                         uint64_t offset, int64_t max_inflight, float* out_scalars);
/* The same launch with the cut taken OFF the launch stream (r4): the keys of the next snapshot are cut
 * by a read-only pass on the split refresh's side stream, behind this launch and BESIDE the next one —
 * the launch stream runs launch after launch with nothing in between.  Two consequences, both inside
 * the asynchrony the STREAM mode already has: the snapshot's point in time is "after
 * this launch, plus whatever the next launch did in the ~20 us the cut takes"; and the hot rows'
 * deltas are NOT folded into the item table by the launch — every other entry point of the ctx folds
 * them first (the table is whole for whoever looks through the library), and a caller who reads the
 * item table's storage directly (eval, checkpoint) calls bpr_hot_fold before.  out_scalars is added
 * to by the side-stream pass: read it after bpr_hot_fold or a device synchronisation.
 * r6 (default; bpr_set_tuning "acut_fold" 0 restores the form above): only the TRANSPOSE of the keys leaves the launch
 * stream — the fold of the hot block, the loss sums and the item_bias write-back stay on it (k_stream_epilogue, a few
 * microseconds) — so the item table is whole after every launch (bpr_hot_fold is a no-op), out_scalars arrives on the
 * launch stream, and the LDS tier (bpr_set_hot_lds) stays in use. */
int bpr_train_stream_acut(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, int32_t* neg,
                          int64_t n, int32_t sampler, float adaptive_p, uint64_t seed,
                          uint64_t offset, int64_t max_inflight, float* out_scalars);
int bpr_hot_fold(bpr_ctx* ctx);

/* BATCHED STREAM — the single-launch throughput path for every optimizer kind (SGD, momentum /
 * Nesterov, Adam, RMSprop; configs/RQ3/time-split/ada-sampling-adam.yaml.j2:169-175 and 14 of the
 * 22 BPR configs step torch.optim.Adam at experiments/trainer.py:79-81).  users/pos [n] are the
 * shuffled triple stream (bpr_shuffle_epoch; NOT grouped by user); triple k belongs to virtual
 * mini-batch k / B, i.e. optimizer step (steps so far) + 1 + k / B.  Per row the gradients of one
 * virtual batch are summed and applied as ONE torch.optim step (dense semantics: untouched rows
 * catch up lazily), every gradient being evaluated on the rows as of the previous step.  Executed
 * by one group (max_inflight = 1) this IS the reference's mini-batch loop; with the chip full,
 * triples of up to max_inflight (0 = fill the chip) consecutive stream positions run concurrently
 * and see rows that may be a few steps stale.  Advances the step counter by
 * ceil(n / B); bpr_flush_lazy / bpr_adaptive_refresh bring rows to "now" (call bpr_flush_lazy
 * before reading the tables: eval, checkpoint, item all-reduce).  sampler / seed / offset / neg /
 * out_scalars as bpr_train_stream.
 * r4: for 16 <= B <= 2048 the launch first marks the triples whose user occurs once in its virtual
 * batch; such a user row takes its optimizer step at once, under the row's lock, instead of parking
 * the gradient for the row's next visitor (same arithmetic, bit-identical in the max_inflight = 1
 * limit).  On by default except for Adam on a model with item_bias;
 * bpr_set_tuning(ctx, "vs_direct", 0 / 1) forces it off / on (a measurement and test aid). */
int bpr_train_stream_batched(bpr_ctx* ctx, const int32_t* users, const int32_t* pos, int32_t* neg,
                             int64_t n, int64_t B, int32_t sampler, float adaptive_p,
                             uint64_t seed, uint64_t offset, int64_t max_inflight,
                             float* out_scalars);
/* DataLoader(shuffle=True, generator=manual_seed(seed)) stand-in (example.py:307-321,
 * experiments/bpr/exp.py:109-118) without the by-user grouping of bpr_plan_epoch: a seeded
 * pseudo-random permutation of the n training triples, computed on device.  Outputs must not alias
 * the inputs. */
int bpr_shuffle_epoch(bpr_ctx* ctx, const int32_t* users_in, const int32_t* pos_in, int64_t n,
                      uint64_t seed, int32_t* users_out, int32_t* pos_out);

/* ONE chunk of that plan without planning the epoch: the triples of chunk `index` (the same member
 * set bpr_plan_epoch(seed) puts there, grouped by user; the order of a user's triples may differ),
 * found through the INVERSE of the permutation — pi^-1 of [index * chunk, (index + 1) * chunk) — and
 * sorted by user: ~200 k keys instead of the whole epoch.  users_out / pos_out [min(chunk, n - index *
 * chunk)].  The plan does not depend on the model: with on_side != 0 it is queued on the split
 * refresh's side stream behind the sort in flight (bpr_adaptive_refresh_begin must have been
 * called), and bpr_adaptive_refresh_commit then waits for both — the chunk after next is planned in
 * the time the sorter idles, nothing is planned on the launch stream (fast.StreamTrainer(jit_plan)). */
int bpr_plan_chunk(bpr_ctx* ctx, const int32_t* users_in, const int32_t* pos_in, int64_t n, int64_t chunk,
                   uint64_t seed, int64_t index, int32_t* users_out, int32_t* pos_out, int32_t on_side);

/* STREAM options.  grouped_by_user = 1 promises that inside every chunk handed to
 * bpr_train_stream the triples of a user are contiguous (the output of bpr_plan_epoch): a user
 * whose triples all fall in one run of `run_len` consecutive triples is then owned by one
 * wavefront group for the whole launch and its row is written back with a plain store; with 0
 * (default) every user-row update is an atomic add.  run_len = consecutive triples one
 * group walks with the user row held in registers: 1..30, or 0 (default) = chosen per launch —
 * 8 when runs of 8 fill the launch stream's CUs more than once; for smaller launches the shortest
 * runs of 4..8 triples that fit those CUs in one residency, one run per group (a small launch
 * ends when its slowest group does; with max_inflight > 0 the bound is min(those CUs' capacity,
 * max_inflight) groups).  bpr_stream_run_len: what the last STREAM launch used. */
int bpr_set_stream_opts(bpr_ctx* ctx, int32_t grouped_by_user, int32_t run_len);
int bpr_stream_run_len(bpr_ctx* ctx);

/* Hot item rows.  On popularity-skewed data the STREAM kernel is limited by fp32 atomics queueing
 * on the memory channels that happen to hold the most popular item rows (rows are scattered over
 * the table, the load per channel is uneven).  bpr_plan_epoch therefore counts the training
 * positives per item (once per training set) and the `hot_rows` most popular rows (default 256)
 * take their STREAM updates in a compact block of delta rows — each placed in the slot whose
 * channels are least loaded, counting what the rows left in Q put on every channel, so that the
 * whole launch is spread evenly; optionally `replicas` (1, 2, 4 or 8; default 1) of the block, a
 * wavefront adds to one, every reader adds them all to the base row — folded into Q right after
 * every STREAM launch.  Same algebra
 * as updating Q directly, up to the association of fp32 sums.  hot_rows = 0 turns it off.  Takes
 * effect at the next bpr_plan_epoch. */
int bpr_set_hot_rows(bpr_ctx* ctx, int32_t hot_rows, int32_t replicas);
/* The LDS tier of the hot block (r6).  Once the adaptive sampler has a trained model to adapt to, its
 * negatives concentrate on the head of the dominant factors' orders (neg_samplers.py:84-121) — 40-60 % of
 * them, and half of the positives, fall on a few hundred item rows, ~800 updates per row and launch, each one
 * four memory-side line requests plus the read of a delta row those requests keep dropping from the L2s.
 * With rows > 0, STREAM launches that fill the chip at least twice run ONE workgroup per CU (up to 1,024
 * threads, persistent over its share of the runs) that keeps a private fp32 delta block for the `rows` most
 * popular hot rows in LDS beside its groups' seen bitmaps (as many as fit the CU's 160 KiB; d = 128, ML-20M:
 * 128 rows): an update of such a row is an LDS add, its value is Q + the workgroup's own LDS delta, and the
 * rows a workgroup touched are added to the global delta block at its exit (so the fold, the cut, the hot tier
 * of several GPUs and the exact-sum property are what they were).  SEMANTICS: a workgroup sees the other
 * workgroups' updates of these rows one launch late — the staleness `learning rate x triples per launch`
 * that the multi-GPU budget prices (revisit_bpr/fast.py lag_within_budget: inside at the
 * reference's lr 0.001 and 0.01, outside at 0.05), which is why it is off by default and the trainers switch
 * it on by that rule.  With max_inflight = 1 (one group) it is exactly sequential SGD.  always = 1: also for
 * launches that do not fill the chip (tests).  Needs a hot block (bpr_plan_epoch / bpr_set_hot_items), d in
 * {32, 64, 128, 256, 512, 1024}, a snapshot sorted whole (the groups' seen structure is the LDS bitmap, or the staged
 * sorted list when the bitmaps would not leave 32 KB for rows: item tables past ~60 k items); otherwise the plain
 * kernel runs.  bpr_stream_lds_rows: LDS rows of the last STREAM launch (0 = the plain kernel ran). */
int bpr_set_hot_lds(bpr_ctx* ctx, int32_t rows, int32_t always);
int bpr_stream_lds_rows(bpr_ctx* ctx);
/* Test and measurement aids, per ctx (nothing in the library reads the environment per launch):
 *   "seen"      0 = by shape (default), 1 = binary search in the CSR, 2 = LDS bitmap, 3 = staged list —
 *               the structure the sampling kernels answer "has u seen c?" from;
 *   "vs_direct" -1 = by optimizer (default), 0 / 1 = lonely user rows of the batched STREAM kernel step
 *               through the gradient buffer / directly under the row's lock;
 *   "adam_closed" 1 (default) / 0 = Adam's missed zero-gradient steps replayed in closed form / by the loop;
 *   "refresh_sub" 0 = by shape (default), 1 | 2 | 4 = workgroups per column of the in-LDS snapshot sort;
 *   "partial_snapshot" 0 / 1, "partial_target" 1..1024: bpr_adaptive_snapshot_partial above;
 *   "binned_sort" 1 (default) / 0: tables of 2,048 .. 20,480 items have their snapshot columns ordered by
 *               k_sort_binned (equi-depth bins + ranking inside the bin: the same stable descending order as
 *               the radix sort, bit for bit, in ~0.4 of its time; up to 65,535 items with several workgroups per
 *               column, k_sort_binned_split) / by the radix sort (a test and measurement aid);
 *   "binned_split" 0 (default: workgroups per column by table size) / 1..4 (tests force the split kernel on small tables);
 *   "lds_block" 0 (default: 1,024 threads for d <= 256 at 32-lane groups, 512 above) / a multiple of 64: threads per
 *               workgroup of the LDS-tier kernel (bpr_set_hot_lds) — fewer groups leave more LDS for hot rows;
 *   "plan_input_sorted" 0 (default) / 1: a PROMISE that the users_in handed to bpr_plan_epoch are sorted by user id (the
 *               triple list in CSR order, as every loader of this repository makes it): the plan — same output — then
 *               takes one radix pass over the chunk bits instead of three over (chunk, user);
 *   "acut_fold" 1 (default, r6) / 0: bpr_train_stream_acut keeps the fold of the hot block (+ loss sums, bias write-back)
 *               on the launch stream and sends only the transpose of the snapshot's keys to the side stream / r4's form
 *               (nothing folded until bpr_hot_fold: the LDS tier is then not used);
 *   "lds_tail"  0..50 (default 12): percent of a launch's triples the LDS-tier kernel deals in runs of run_len / 2 and
 *               run_len / 4 at the end of every persistent workgroup's share (a workgroup is over when its last run is). */
int bpr_set_tuning(bpr_ctx* ctx, const char* key, int32_t value);
/* The item_bias during STREAM launches (models/bpr/model.py:101-110; the RQ configs switch it on).
 * k_stream works on a table of its own with ONE item per 128-B line (in the dense vector 32 items share a
 * line and every bias load queues behind their adds at the memory side); each launch fills it from the
 * caller's vector and its epilogue writes it back.  bpr_set_bias_tracking(ctx, 1): the fill is skipped
 * while the table is known to equal the vector — written back by the previous launch, no entry point of
 * the library has written the vector since — which leaves writes the library cannot see: after any
 * write of the caller's own to item_bias (optimizer step on the tensor, checkpoint load, zero_()) call
 * bpr_bias_written before the next launch.  Off by default (every launch refills). */
int bpr_set_bias_tracking(bpr_ctx* ctx, int32_t on);
int bpr_bias_written(bpr_ctx* ctx);
/* Heavy users (more seen items than `threshold`, default 256; -1 = none) get an I-bit seen bitmap in
 * HBM, built synchronously by the first sampling STREAM launch after the seen CSR was bound; the
 * bitmaps take at most `max_bytes` (default 1 GiB; 0 keeps the current cap) — the threshold is
 * doubled until they fit. */
int bpr_set_heavy_users(bpr_ctx* ctx, int32_t threshold, int64_t max_bytes);

/* ---- two-tier item reconciliation, HOT tier (several GPUs; no reference counterpart — the
 * reference's latent DDP, experiments/launcher.py:35-73, would all-reduce every gradient of every
 * step).  At full-size launches per rank, N replicas each pour a whole launch into the same few
 * popular rows before anybody sees the others' updates; those rows — and only those — are therefore
 * exchanged after EVERY launch (or sub-launch) as a block of H x d floats (128 KB at H = 256,
 * d = 128), while the cold rows keep the per-period all-reduce of bpr_item_fold_delta.
 *
 * bpr_set_hot_items  the hot set as the caller gives it — the ranks must agree on it (the H most
 *                    popular items of the WHOLE training set); items_host [H] distinct item rows in
 *                    the canonical order of the exchange buffers, counts_host [I] (or NULL) only
 *                    steers the slot placement.  H = 0 returns to bpr_set_hot_rows' own choice.
 * bpr_hot_tier_begin hot_base [H, d] <- the hot rows of the item table (replicas identical at this
 *                    point); from here on STREAM launches leave their hot-row deltas in the block.
 * bpr_hot_exchange   one pass over the block on the ctx stream:
 *                      fold_prev: hot_base += tot   (tot = all-reduced sum of the previous exchange)
 *                      cut:       tot <- this rank's deltas of the launches since, deltas <- 0
 *                      Q[hot rows] <- hot_base + tot;  cold_base (optional, [I, d]): same rows <- same
 *                    The caller all-reduces `tot` (SUM) between two calls.  hot_base only ever takes
 *                    all-reduced sums: it stays bit-identical on every rank.  cold_base keeps the
 *                    cold tier's delta of a hot row at exactly zero.
 * bpr_hot_tier_end   launches fold their hot block themselves again (call bpr_hot_exchange with
 *                    fold_prev as due and cut = 1 first, all-reduce, then once more with cut = 0). */
int bpr_set_hot_items(bpr_ctx* ctx, const int32_t* items_host, int32_t H, const uint32_t* counts_host);
int bpr_hot_rows(bpr_ctx* ctx, int32_t* rows_host);
int bpr_hot_tier_begin(bpr_ctx* ctx, float* hot_base);
int bpr_hot_exchange(bpr_ctx* ctx, float* hot_base, float* tot, int32_t fold_prev, int32_t cut,
                     float* cold_base);
int bpr_hot_tier_end(bpr_ctx* ctx);
/* Everything a rank does to the item table between two launches, in ONE pass (r4): the hot-tier
 * exchange step (as bpr_hot_exchange with cut = 1), the cold tier's step (cold_mode 2: as
 * bpr_item_fold_delta; 1: as bpr_item_delta; 0: none) and the cut of the next adaptive snapshot's
 * keys (as bpr_train_stream_cut's epilogue: the next bpr_adaptive_refresh_begin only queues the
 * sort).  Call it right after a bpr_train_stream_cut issued under the hot tier — that launch leaves
 * its epilogue (the loss partials) to this pass — and all-reduce hot_tot and cold_tot afterwards.
 * Four kernels and their boundaries (53 us measured) become one. */
int bpr_sync_cut(bpr_ctx* ctx, float* hot_base, float* hot_tot, int32_t hot_fold_prev, float* cold_base,
                 float* cold_own, float* cold_tot, float scale, int32_t cold_mode);

/* Epoch order for STREAM mode — replaces DataLoader(shuffle=True, generator=manual_seed(seed))
 * (example.py:307-321; experiments/bpr/exp.py:109-118): a seeded pseudo-random partition of the n
 * training triples into ceil(n/chunk) chunks of `chunk` triples (the last may be shorter); inside a
 * chunk triples are grouped by user.  users_out/pos_out [n] must not alias the inputs. */
int bpr_plan_epoch(bpr_ctx* ctx, const int32_t* users_in, const int32_t* pos_in, int64_t n,
                   int64_t chunk, uint64_t seed, int32_t* users_out, int32_t* pos_out);

/* Dense-optimizer equivalence: torch's dense Adam / momentum move EVERY row on every
 * step.  The strict path replays the missed zero-gradient steps lazily when a row is next
 * touched; this brings all rows to the current step (call before eval / checkpoint / all-reduce). */
int bpr_flush_lazy(bpr_ctx* ctx);
/* The same for the item table (and item_bias) only: what the item reconciliation across GPUs and
 * the sampler snapshot need — the user shard of a rank is read by nobody else and stays lazy. */
int bpr_flush_items(bpr_ctx* ctx);
/* Global optimizer step counter t (number of bpr_apply calls); settable for checkpoint resume. */
int bpr_get_step_host(bpr_ctx* ctx, int64_t* step_host);
int bpr_set_step(bpr_ctx* ctx, int64_t step);
/* AdaptiveSampler._iteration_cnt of the bpr_train_strict loop (neg_samplers.py:75): batches drawn so
 * far; 0 at ctx creation.  Settable so a resumed run refreshes at the same iterations. */
int bpr_set_sampler_iter(bpr_ctx* ctx, int64_t iteration);

/* ---- evaluation (E3): ROC-AUC of a block of score rows — the quantity of the reference's RocAucMany /
 * RocAucManySlow (revisit_bpr/metrics/auc.py:70-130: every (positive, negative) pair of a row; 1 of the 14 metrics of
 * its ML-20M / MSD configs) without the [B, I, I] comparison or a sort per row: scores [n, I] fp32 row-major (masked
 * entries carry their mask value and count as negatives, as in the reference's eval loop), positives of row r =
 * pos_items[pos_indptr[r] .. pos_indptr[r + 1]); auc_out[r] = sum over the positives of #{negatives scored strictly
 * below} / (T (I - T)); NaN for a row without positives (0 / 0, as the metric classes) or with more than 4,096.
 * Context-free: runs on `hip_stream`. */
int bpr_auc_rows(const float* scores, int64_t n, int64_t I, const int64_t* pos_indptr, const int32_t* pos_items,
                 float* auc_out, void* hip_stream);

/* ---- multi-GPU item-table reconciliation (no reference counterpart: the reference's DDP path is
 * never enabled by a config, experiments/launcher.py:35-73).  The all-reduce itself is RCCL via
 * torch.distributed; these two fused elementwise kernels bracket it (revisit_bpr/distributed.py).
 * Context-free: arrays of n floats and the HIP stream to run on. */
/* own[k] = tot[k] = q[k] - base[k]   (this rank's delta since the last reconciliation) */
int bpr_item_delta(const float* q, const float* base, float* own, float* tot, int64_t n,
                   void* hip_stream);
/* tot = all-reduced sum of every rank's delta.  base += scale*tot (bit-identical on every rank);
 * rebase == 0: q += scale*tot - own   (asynchronous: q keeps what it learned since bpr_item_delta)
 * rebase == 1: q  = base              (blocking reconcile: every replica becomes the same cut) */
int bpr_item_fold(float* q, float* base, const float* own, const float* tot, float scale,
                  int32_t rebase, int64_t n, void* hip_stream);
/* bpr_item_fold (rebase = 0) immediately followed by bpr_item_delta, as one pass over the table:
 * the per-period reconciliation step when nothing trains between folding the previous all-reduce
 * and cutting the next delta.  tot holds the all-reduced sum on entry and the new delta on exit. */
int bpr_item_fold_delta(float* q, float* base, float* own, float* tot, float scale, int64_t n,
                        void* hip_stream);

/* ---- measurement --------------------------------------------------------------------------- */
/* Average duration (ms) of the dominant kernel over the launches recorded since the last reset,
 * measured with hipEvents on the ctx stream (bench.py's roofline.achieved uses this).
 * bpr_timing_enable(ctx, N) turns recording on for every N-th launch (N >= 1; 0 = off).  The two
 * event records around a timed launch cost ~6 us of stream idle time each on MI355X — ~4 % of a
 * 0.3 ms step when every launch is timed — so bench.py samples every 8th launch of the timed region. */
int bpr_timing_enable(bpr_ctx* ctx, int32_t on);
int bpr_timing_read_host(bpr_ctx* ctx, double* avg_ms_host, int64_t* launches_host);

#ifdef __cplusplus
}
#endif
#endif /* BPRCORE_H */
