/*
 * bpr_oracle.h — CPU restatement of the reference's BPR-MF hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker (or, in bench.py, as the timed CPU baseline) — never from the product path.
 *
 * Parity pin: the reference (Nemexur/revisit-bpr) ships no tests / golden vectors for this path
 * (SURVEY.md §4), so the oracle is pinned against outputs of the reference itself, generated in the
 * build container by tests/golden/make_golden.py (which imports /root/reference/revisit_bpr and
 * torch.optim) and committed as tests/golden/ (.npz files).  tests/test_oracle_golden.py checks every
 * function below against those vectors.  The Philox generator is pinned by the Random123
 * known-answer vectors.
 *
 * All pointers are host pointers.  Tables are row-major fp32.  Paths cited are relative to the
 * reference tree.
 */
#ifndef BPR_ORACLE_H
#define BPR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SGD = 0, ORC_MOMENTUM = 1, ORC_ADAM = 2, ORC_RMSPROP = 3 };
enum { ORC_NEG_GIVEN = 0, ORC_NEG_UNIFORM = 1, ORC_NEG_ADAPTIVE = 2 };

typedef struct orc_opt {
  int32_t kind;
  float lr, momentum, dampening;
  int32_t nesterov;
  float beta1, beta2, eps, alpha;
} orc_opt;

/* Philox4x32-10 (Salmon et al., SC'11; the generator behind rocRAND's philox4x32_10). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* MF.forward + Model.forward train branch + Loss + regularization
 * (revisit_bpr/models/bpr/model.py:48-68,70-93,131-145; loss.py:19-21).
 * scalars[0]=bpr_loss, [1]=l2_reg, [2]=sum|x|, [3]=B (double, overwritten). */
void orc_forward(const float* P, const float* Q, const float* item_bias, int32_t d,
                 const int32_t* users, const int32_t* pos, const int32_t* neg, int64_t B,
                 float a_user, float a_item, float a_neg, float* logits_pos, float* logits_neg,
                 double* scalars);

/* loss.backward() (trainer.py:76 / example.py:178): DENSE grads gP [U,d], gQ [I,d], gbias [I]
 * (overwritten; gbias may be NULL).  pad_user/pad_item: nn.Embedding padding_idx (−1 none). */
void orc_dense_grad(const float* P, const float* Q, const float* item_bias, int64_t U, int64_t I,
                    int32_t d, const int32_t* users, const int32_t* pos, const int32_t* neg,
                    int64_t B, float a_user, float a_item, float a_neg, int32_t pad_user,
                    int32_t pad_item, float* gP, float* gQ, float* gbias);

/* torch.optim.{SGD,Adam,RMSprop}.step on one dense tensor of n elements (A7-A9); `t` is the 1-based
 * step number of this call.  m / v are the state tensors (may be NULL when unused). */
void orc_opt_dense(const orc_opt* opt, int64_t t, float* w, const float* g, float* m, float* v,
                   int64_t n);

/* One reference iteration with given negatives: forward → dense backward → dense optimizer step on
 * P, Q (and bias).  Scratch for dense grads is allocated internally.  Returns 0 / −1 (alloc). */
int orc_step(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
             const int32_t* users, const int32_t* pos, const int32_t* neg, int64_t B, float a_user,
             float a_item, float a_neg, int32_t pad_user, int32_t pad_item, const orc_opt* opt,
             int64_t t, float* mP, float* vP, float* mQ, float* vQ, float* mb, float* vb,
             float* logits_pos, float* logits_neg, double* scalars);

/* Same iteration, but applying the optimizer only to rows that occur in the batch.  Identical to
 * orc_step for plain SGD (untouched rows have zero gradient and no state); used as the CPU
 * baseline so that the CPU side is not charged for the reference's dense sweep. */
int orc_step_sgd_sparse(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
                        const int32_t* users, const int32_t* pos, const int32_t* neg, int64_t B,
                        float a_user, float a_item, float a_neg, int32_t pad_user,
                        int32_t pad_item, float lr, float* logits_pos, float* logits_neg,
                        double* scalars);

/* _sampling_weights (revisit_bpr/modules/neg_samplers.py:135-141), literal: base [I], padded seen
 * [B,S] (0 = pad) → out [B,I] row-normalised. */
void orc_sampling_weights(const float* base, int64_t I, const int64_t* seen_padded, int64_t B,
                          int64_t S, float* out);

/* Uniform negative over unseen items via Philox rejection sampling (distribution of
 * UniformSampler.sample, neg_samplers.py:31-37; draw-for-draw identical to the HIP kernel).
 * CSR: indptr [U+1], indices sorted per row. */
void orc_sample_uniform(const int64_t* indptr, const int32_t* indices, int64_t I,
                        const int32_t* users, int64_t B, uint64_t seed, uint64_t offset,
                        int32_t* neg_out);

/* The same with item weights (BPRExperiment._static_sampling with count_i ** neg_sampling_alpha,
 * experiments/bpr/exp.py:85-91, 282-293): accept [I] / alias [I] are a Walker alias table over the
 * weights of items 1..I-1 (entry 0 unused); P(negative = i) = w_i / sum of w over the unseen. */
void orc_sample_weighted(const int64_t* indptr, const int32_t* indices, int64_t I,
                         const int32_t* users, int64_t B, uint64_t seed, uint64_t offset,
                         const float* accept, const int32_t* alias, int32_t* neg_out);

/* AdaptiveSampler.update_stats (neg_samplers.py:126-132), literal: QT [d,I] = transpose copy,
 * sigma [d] = unbiased std over rows 1..I-1. */
void orc_adaptive_stats(const float* Q, int64_t I, int32_t d, float* QT, float* sigma);
/* Per-factor descending item order of a snapshot, ties by ascending item id: order [d,I]. */
void orc_adaptive_order(const float* QT, int64_t I, int32_t d, int32_t* order);
/* neg_samplers.py:109-121, literal: copy row `factor` of QT, set seen∪{0} to −1e13, argsort
 * descending, take element `rank`.  seen given as CSR row of `user`. */
int32_t orc_adaptive_pick_literal(const float* QT, int64_t I, const int64_t* indptr,
                                  const int32_t* indices, int32_t user, int32_t factor,
                                  int32_t rank);
/* Same result through the precomputed order (walk, skipping seen∪{0}). */
int32_t orc_adaptive_pick(const int32_t* order, int64_t I, const int64_t* indptr,
                          const int32_t* indices, int32_t user, int32_t factor, int32_t rank);
/* Full AdaptiveSampler.sample (neg_samplers.py:74-124) with Philox draws, mirrored by the HIP
 * kernel: factor ~ |p_uf| sigma_f, r ~ Geometric(p) clamped to #unseen, orientation by sign.
 * factor_out / rank_out nullable. */
void orc_sample_adaptive(const float* P, int32_t d, const float* sigma, const int32_t* order,
                         int64_t I, const int64_t* indptr, const int32_t* indices,
                         const int32_t* users, int64_t B, float p, uint64_t seed, uint64_t offset,
                         int32_t* neg_out, int32_t* factor_out, int32_t* rank_out);

/* Sequential (B=1) SGD over a triple stream with on-the-fly sampling: the limit the STREAM path
 * approaches when max_inflight = 1 triple.  neg_io: read when sampler == GIVEN, else written
 * (nullable).  scalars as orc_forward but accumulated over the stream. */
void orc_train_stream_seq(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
                          const int32_t* users, const int32_t* pos, int32_t* neg_io, int64_t n,
                          int32_t sampler, float adaptive_p, const float* sigma,
                          const int32_t* order, const int64_t* indptr, const int32_t* indices,
                          uint64_t seed, uint64_t offset, float a_user, float a_item, float a_neg,
                          int32_t pad_user, int32_t pad_item, float lr, double* scalars);

/* CPU baseline (a) of SURVEY §8d: mini-batches of the same path with OpenMP over the triples of a
 * batch (sampling, gradients at the pre-step parameters, one sparse SGD step per batch) and over the
 * columns of the adaptive refresh (every `refresh_every` batches; QT [d,I] / sigma [d] / order [d,I]
 * are the caller's snapshot buffers).  seconds > 0: wall-clock budget; threads <= 0: OpenMP's default.
 * Returns the triples trained (-1: out of memory).  Timed by bench.py; equal to orc_step_sgd_sparse
 * batch by batch up to fp32 association (tests/test_oracle_golden.py). */
int orc_max_threads(void);
int64_t orc_train_batches_omp(float* P, float* Q, int64_t U, int64_t I, int32_t d, const int32_t* users,
                              const int32_t* pos, int64_t n, int64_t B, int32_t sampler, float adaptive_p,
                              float* QT, float* sigma, int32_t* order, int64_t refresh_every,
                              const int64_t* indptr, const int32_t* indices, uint64_t seed, uint64_t offset,
                              float a_user, float a_item, float a_neg, int32_t pad_user, int32_t pad_item,
                              float lr, double seconds, int32_t threads, double* scalars);

#ifdef __cplusplus
}
#endif
#endif
