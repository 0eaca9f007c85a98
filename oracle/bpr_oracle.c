/*
 * bpr_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see bpr_oracle.h for who may load it and how it is pinned).
 *
 * Arithmetic policy: inner products and duplicate accumulation are carried in double and rounded
 * to fp32 once, so the oracle sits between torch's fp32 CPU kernels and the HIP kernels; the
 * parity tolerance (1e-6 relative after one step) is stated in the tests.
 */
#include "bpr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10
 * ---------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* draw word `word` of block `block` of stream `purpose` for triple t */
static uint32_t draw(uint64_t seed, uint64_t t, uint32_t block, uint32_t purpose, int word) {
  uint32_t ctr[4] = {(uint32_t)t, (uint32_t)(t >> 32), block, purpose};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  orc_philox4x32_10(ctr, key, o);
  return o[word];
}

/* ------------------------------------------------------------------------------------------
 * forward / loss / regularisation   (model.py:48-68, 70-93, 131-145; loss.py:19-21)
 * ---------------------------------------------------------------------------------------- */
static double ddot(const float* a, const float* b, int32_t d) {
  double s = 0.0;
  for (int32_t k = 0; k < d; ++k) s += (double)a[k] * (double)b[k];
  return s;
}

/* −logσ(x) = softplus(−x), stable form */
static double neg_logsigmoid(double x) {
  return (x < 0 ? -x : 0.0) + log1p(exp(-fabs(x)));
}

void orc_forward(const float* P, const float* Q, const float* item_bias, int32_t d,
                 const int32_t* users, const int32_t* pos, const int32_t* neg, int64_t B,
                 float a_user, float a_item, float a_neg, float* logits_pos, float* logits_neg,
                 double* scalars) {
  double loss = 0.0, reg = 0.0, sabs = 0.0;
  for (int64_t b = 0; b < B; ++b) {
    const float* p = P + (int64_t)users[b] * d;
    const float* qi = Q + (int64_t)pos[b] * d;
    const float* qj = Q + (int64_t)neg[b] * d;
    /* MF.forward: einsum("bh,b...h->b...") (+ item_bias[item]) — model.py:137-140 */
    float xp = (float)(ddot(p, qi, d) + (item_bias ? (double)item_bias[pos[b]] : 0.0));
    float xn = (float)(ddot(p, qj, d) + (item_bias ? (double)item_bias[neg[b]] : 0.0));
    float x = xp - xn; /* model.py:63 */
    if (logits_pos) logits_pos[b] = xp;
    if (logits_neg) logits_neg[b] = xn;
    loss += neg_logsigmoid((double)x); /* loss.py:20, summed model.py:65 */
    sabs += fabs((double)x);
    /* model.py:87-93: ½(α_i‖q_i‖² + α_neg‖q_j‖² + α_u‖p_u‖²) */
    reg += 0.5 * ((double)a_item * ddot(qi, qi, d) + (double)a_neg * ddot(qj, qj, d) +
                  (double)a_user * ddot(p, p, d));
  }
  if (scalars) {
    scalars[0] = loss;
    scalars[1] = reg;
    scalars[2] = sabs;
    scalars[3] = (double)B;
  }
}

/* ------------------------------------------------------------------------------------------
 * dense backward (SURVEY §3.3 gradients; autograd of model.py:48-68)
 * ---------------------------------------------------------------------------------------- */
void orc_dense_grad(const float* P, const float* Q, const float* item_bias, int64_t U, int64_t I,
                    int32_t d, const int32_t* users, const int32_t* pos, const int32_t* neg,
                    int64_t B, float a_user, float a_item, float a_neg, int32_t pad_user,
                    int32_t pad_item, float* gP, float* gQ, float* gbias) {
  double* aP = (double*)calloc((size_t)U * d, sizeof(double));
  double* aQ = (double*)calloc((size_t)I * d, sizeof(double));
  double* ab = (double*)calloc((size_t)I, sizeof(double));
  for (int64_t b = 0; b < B; ++b) {
    int32_t u = users[b], i = pos[b], j = neg[b];
    const float* p = P + (int64_t)u * d;
    const float* qi = Q + (int64_t)i * d;
    const float* qj = Q + (int64_t)j * d;
    float xp = (float)(ddot(p, qi, d) + (item_bias ? (double)item_bias[i] : 0.0));
    float xn = (float)(ddot(p, qj, d) + (item_bias ? (double)item_bias[j] : 0.0));
    double x = (double)(xp - xn);
    double w = 1.0 / (1.0 + exp(x)); /* σ(−x) */
    for (int32_t k = 0; k < d; ++k) {
      double pk = p[k], qik = qi[k], qjk = qj[k];
      if (u != pad_user) aP[(int64_t)u * d + k] += -w * (qik - qjk) + (double)a_user * pk;
      if (i != pad_item) aQ[(int64_t)i * d + k] += -w * pk + (double)a_item * qik;
      if (j != pad_item) aQ[(int64_t)j * d + k] += w * pk + (double)a_neg * qjk;
    }
    ab[i] += -w; /* biases: plain Parameter indexing, no padding_idx, never regularised */
    ab[j] += w;
  }
  for (int64_t k = 0; k < U * d; ++k) gP[k] = (float)aP[k];
  for (int64_t k = 0; k < I * d; ++k) gQ[k] = (float)aQ[k];
  if (gbias)
    for (int64_t k = 0; k < I; ++k) gbias[k] = (float)ab[k];
  free(aP);
  free(aQ);
  free(ab);
}

/* ------------------------------------------------------------------------------------------
 * torch.optim single-tensor steps (torch/optim/{sgd,adam,rmsprop}.py, pinned by golden fixtures)
 * ---------------------------------------------------------------------------------------- */
void orc_opt_dense(const orc_opt* o, int64_t t, float* w, const float* g, float* m, float* v,
                   int64_t n) {
  switch (o->kind) {
    case ORC_SGD:
      for (int64_t k = 0; k < n; ++k) w[k] = w[k] - o->lr * g[k];
      break;
    case ORC_MOMENTUM:
      for (int64_t k = 0; k < n; ++k) {
        float gk = g[k];
        /* first step: buf = clone(grad); later: buf = μ·buf + (1−dampening)·grad */
        float buf = (t == 1) ? gk : o->momentum * m[k] + (1.0f - o->dampening) * gk;
        m[k] = buf;
        float eff = o->nesterov ? gk + o->momentum * buf : buf;
        w[k] = w[k] - o->lr * eff;
      }
      break;
    case ORC_ADAM: {
      double bc1 = 1.0 - pow((double)o->beta1, (double)t);
      double bc2 = 1.0 - pow((double)o->beta2, (double)t);
      float step_size = (float)((double)o->lr / bc1);
      float bc2_sqrt = (float)sqrt(bc2);
      for (int64_t k = 0; k < n; ++k) {
        float gk = g[k];
        /* exp_avg.lerp_(grad, 1−β1): ATen lerp uses the end-anchored form for weight >= 0.5 */
        float wgt = 1.0f - o->beta1;
        m[k] = (wgt < 0.5f) ? m[k] + wgt * (gk - m[k]) : gk - (gk - m[k]) * (1.0f - wgt);
        v[k] = o->beta2 * v[k] + (1.0f - o->beta2) * gk * gk;
        float denom = sqrtf(v[k]) / bc2_sqrt + o->eps;
        w[k] = w[k] - step_size * (m[k] / denom);
      }
    } break;
    case ORC_RMSPROP:
      for (int64_t k = 0; k < n; ++k) {
        float gk = g[k];
        v[k] = o->alpha * v[k] + (1.0f - o->alpha) * gk * gk;
        float avg = sqrtf(v[k]) + o->eps;
        if (o->momentum > 0.0f) {
          m[k] = o->momentum * m[k] + gk / avg;
          w[k] = w[k] - o->lr * m[k];
        } else {
          w[k] = w[k] - o->lr * (gk / avg);
        }
      }
      break;
    default:
      break;
  }
}

int orc_step(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
             const int32_t* users, const int32_t* pos, const int32_t* neg, int64_t B, float a_user,
             float a_item, float a_neg, int32_t pad_user, int32_t pad_item, const orc_opt* opt,
             int64_t t, float* mP, float* vP, float* mQ, float* vQ, float* mb, float* vb,
             float* logits_pos, float* logits_neg, double* scalars) {
  float* gP = (float*)malloc(sizeof(float) * (size_t)U * d);
  float* gQ = (float*)malloc(sizeof(float) * (size_t)I * d);
  float* gb = (float*)malloc(sizeof(float) * (size_t)I);
  if (!gP || !gQ || !gb) {
    free(gP); free(gQ); free(gb);
    return -1;
  }
  orc_forward(P, Q, item_bias, d, users, pos, neg, B, a_user, a_item, a_neg, logits_pos,
              logits_neg, scalars);
  orc_dense_grad(P, Q, item_bias, U, I, d, users, pos, neg, B, a_user, a_item, a_neg, pad_user,
                 pad_item, gP, gQ, gb);
  orc_opt_dense(opt, t, P, gP, mP, vP, U * d);
  orc_opt_dense(opt, t, Q, gQ, mQ, vQ, I * d);
  if (item_bias) orc_opt_dense(opt, t, item_bias, gb, mb, vb, I);
  free(gP); free(gQ); free(gb);
  return 0;
}

/* sparse-apply SGD: per-row accumulators keyed by slot, duplicates merged through a small
 * open-addressing map so cost is O(B·d), not O((U+I)·d). */
typedef struct { int64_t key; int32_t slot; } hent;

static int32_t hfind(hent* tab, int64_t cap, int64_t key, int32_t* nslots) {
  uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
  int64_t pos = (int64_t)(h & (uint64_t)(cap - 1));
  for (;;) {
    if (tab[pos].key == key) return tab[pos].slot;
    if (tab[pos].key < 0) {
      tab[pos].key = key;
      tab[pos].slot = (*nslots)++;
      return tab[pos].slot;
    }
    pos = (pos + 1) & (cap - 1);
  }
}

int orc_step_sgd_sparse(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
                        const int32_t* users, const int32_t* pos, const int32_t* neg, int64_t B,
                        float a_user, float a_item, float a_neg, int32_t pad_user,
                        int32_t pad_item, float lr, float* logits_pos, float* logits_neg,
                        double* scalars) {
  (void)U; (void)I;
  int64_t cap = 16;
  while (cap < 8 * B) cap <<= 1;
  hent* tab = (hent*)malloc(sizeof(hent) * (size_t)cap);
  double* acc = (double*)calloc((size_t)(3 * B) * (size_t)(d + 1), sizeof(double));
  int64_t* rowkey = (int64_t*)malloc(sizeof(int64_t) * (size_t)(3 * B));
  if (!tab || !acc || !rowkey) { free(tab); free(acc); free(rowkey); return -1; }
  for (int64_t k = 0; k < cap; ++k) tab[k].key = -1;
  int32_t nslots = 0;
  double loss = 0, reg = 0, sabs = 0;
  for (int64_t b = 0; b < B; ++b) {
    int32_t u = users[b], i = pos[b], j = neg[b];
    const float* p = P + (int64_t)u * d;
    const float* qi = Q + (int64_t)i * d;
    const float* qj = Q + (int64_t)j * d;
    float xp = (float)(ddot(p, qi, d) + (item_bias ? (double)item_bias[i] : 0.0));
    float xn = (float)(ddot(p, qj, d) + (item_bias ? (double)item_bias[j] : 0.0));
    float xf = xp - xn;
    double x = xf, w = 1.0 / (1.0 + exp(x));
    if (logits_pos) logits_pos[b] = xp;
    if (logits_neg) logits_neg[b] = xn;
    loss += neg_logsigmoid(x);
    sabs += fabs(x);
    reg += 0.5 * ((double)a_item * ddot(qi, qi, d) + (double)a_neg * ddot(qj, qj, d) +
                  (double)a_user * ddot(p, p, d));
    /* keys: users 2k, items 2k+1 */
    int32_t su = hfind(tab, cap, 2 * (int64_t)u, &nslots); rowkey[su] = 2 * (int64_t)u;
    int32_t si = hfind(tab, cap, 2 * (int64_t)i + 1, &nslots); rowkey[si] = 2 * (int64_t)i + 1;
    int32_t sj = hfind(tab, cap, 2 * (int64_t)j + 1, &nslots); rowkey[sj] = 2 * (int64_t)j + 1;
    double* gu = acc + (size_t)su * (d + 1);
    double* gi = acc + (size_t)si * (d + 1);
    double* gj = acc + (size_t)sj * (d + 1);
    for (int32_t k = 0; k < d; ++k) {
      double pk = p[k], qik = qi[k], qjk = qj[k];
      gu[k] += -w * (qik - qjk) + (double)a_user * pk;
      gi[k] += -w * pk + (double)a_item * qik;
      gj[k] += w * pk + (double)a_neg * qjk;
    }
    gi[d] += -w;
    gj[d] += w;
  }
  for (int32_t s = 0; s < nslots; ++s) {
    int64_t key = rowkey[s];
    int64_t row = key >> 1;
    const double* g = acc + (size_t)s * (d + 1);
    if (key & 1) {
      if (row != pad_item) {
        float* q = Q + row * d;
        for (int32_t k = 0; k < d; ++k) q[k] = q[k] - lr * (float)g[k];
      }
      if (item_bias) item_bias[row] = item_bias[row] - lr * (float)g[d];
    } else if (row != pad_user) {
      float* p = P + row * d;
      for (int32_t k = 0; k < d; ++k) p[k] = p[k] - lr * (float)g[k];
    }
  }
  if (scalars) { scalars[0] = loss; scalars[1] = reg; scalars[2] = sabs; scalars[3] = (double)B; }
  free(tab); free(acc); free(rowkey);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * negative sampling
 * ---------------------------------------------------------------------------------------- */
void orc_sampling_weights(const float* base, int64_t I, const int64_t* seen_padded, int64_t B,
                          int64_t S, float* out) {
  for (int64_t b = 0; b < B; ++b) {
    float* w = out + b * I;
    memcpy(w, base, sizeof(float) * (size_t)I);                    /* repeat(...)        :136 */
    for (int64_t s = 0; s < S; ++s) w[seen_padded[b * S + s]] = 0; /* scatter(value=0.0) :137 */
    w[0] = 0.0f;                                                   /* discard padding    :139 */
    float tot = 0.0f;
    for (int64_t k = 0; k < I; ++k) tot += w[k];
    float inv = 1.0f / tot; /* weights.sum().reciprocal() :140 */
    for (int64_t k = 0; k < I; ++k) w[k] *= inv;
  }
}

static int csr_contains(const int64_t* indptr, const int32_t* indices, int32_t user, int32_t item) {
  int64_t lo = indptr[user], hi = indptr[user + 1];
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    int32_t v = indices[mid];
    if (v == item) return 1;
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return 0;
}

#define ORC_UNIFORM_MAX_CAND 4096

/* candidate from one 32-bit draw r: column c = 1 + floor(r (I-1) / 2^32); with an alias table
 * (Walker / Vose) over item weights the fractional part of r (I-1) / 2^32 decides between the
 * column and its alias, so that P(candidate = i) = w_i / sum(w). */
static int32_t uniform_candidate(uint32_t r, int64_t I, const float* accept, const int32_t* alias) {
  uint32_t n = (uint32_t)(I - 1);
  int32_t c = 1 + (int32_t)(((uint64_t)r * (uint64_t)n) >> 32);
  if (accept == NULL) return c;
  float frac = (float)(uint32_t)(r * n) * (1.0f / 4294967296.0f);
  return frac < accept[c] ? c : alias[c];
}

static int32_t sample_weighted_one(const int64_t* indptr, const int32_t* indices, int64_t I,
                                   int32_t user, uint64_t seed, uint64_t t, const float* accept,
                                   const int32_t* alias) {
  for (uint32_t k = 0; k < ORC_UNIFORM_MAX_CAND; ++k) {
    uint32_t r = draw(seed, t, k >> 2, 0u, (int)(k & 3));
    int32_t c = uniform_candidate(r, I, accept, alias);
    if (!csr_contains(indptr, indices, user, c)) return c;
  }
  /* every candidate was seen (a user who has seen nearly all items): the reference's multinomial
   * over the masked weights (neg_samplers.py:31-37) still returns an unseen item — one more draw
   * picks the r-th unseen item by rank (literal scan over the ids) */
  {
    int64_t lo = indptr[user], hi = indptr[user + 1];
    int64_t n_unseen = (I - 1) - (hi - lo);
    if (n_unseen <= 0) return 0;
    uint32_t r = draw(seed, t, ORC_UNIFORM_MAX_CAND / 4, 0u, 0);
    int64_t want = (int64_t)(((uint64_t)r * (uint64_t)(uint32_t)n_unseen) >> 32);
    for (int32_t c = 1; c < I; ++c) {
      if (csr_contains(indptr, indices, user, c)) continue;
      if (want-- == 0) return c;
    }
  }
  return 0;
}

static int32_t sample_uniform_one(const int64_t* indptr, const int32_t* indices, int64_t I,
                                  int32_t user, uint64_t seed, uint64_t t) {
  return sample_weighted_one(indptr, indices, I, user, seed, t, NULL, NULL);
}

/* BPRExperiment._static_sampling with item weights count_i ** neg_sampling_alpha
 * (experiments/bpr/exp.py:85-91, 282-293): multinomial over w_i for unseen i, 0 for seen and item 0 —
 * here by rejection: candidates ~ w (alias table), the first unseen one wins. */
void orc_sample_weighted(const int64_t* indptr, const int32_t* indices, int64_t I,
                         const int32_t* users, int64_t B, uint64_t seed, uint64_t offset,
                         const float* accept, const int32_t* alias, int32_t* neg_out) {
  for (int64_t b = 0; b < B; ++b)
    neg_out[b] = sample_weighted_one(indptr, indices, I, users[b], seed, offset + (uint64_t)b,
                                     accept, alias);
}

void orc_sample_uniform(const int64_t* indptr, const int32_t* indices, int64_t I,
                        const int32_t* users, int64_t B, uint64_t seed, uint64_t offset,
                        int32_t* neg_out) {
  for (int64_t b = 0; b < B; ++b)
    neg_out[b] = sample_uniform_one(indptr, indices, I, users[b], seed, offset + (uint64_t)b);
}

void orc_adaptive_stats(const float* Q, int64_t I, int32_t d, float* QT, float* sigma) {
  for (int32_t f = 0; f < d; ++f) {
    for (int64_t i = 0; i < I; ++i) QT[(int64_t)f * I + i] = Q[i * d + f]; /* einsum("if->fi") :130 */
    double mean = 0.0;
    for (int64_t i = 1; i < I; ++i) mean += Q[i * d + f];
    mean /= (double)(I - 1);
    double ss = 0.0;
    for (int64_t i = 1; i < I; ++i) {
      double c = (double)Q[i * d + f] - mean;
      ss += c * c;
    }
    sigma[f] = (float)sqrt(ss / (double)(I - 2)); /* features["item"][1:].std(dim=0) unbiased :132 */
  }
}

typedef struct { float v; int32_t id; } vid;
static int cmp_desc(const void* a, const void* b) {
  const vid* x = (const vid*)a; const vid* y = (const vid*)b;
  if (x->v > y->v) return -1;
  if (x->v < y->v) return 1;
  return (x->id > y->id) - (x->id < y->id);
}

void orc_adaptive_order(const float* QT, int64_t I, int32_t d, int32_t* order) {
  vid* tmp = (vid*)malloc(sizeof(vid) * (size_t)I);
  for (int32_t f = 0; f < d; ++f) {
    for (int64_t i = 0; i < I; ++i) { tmp[i].v = QT[(int64_t)f * I + i]; tmp[i].id = (int32_t)i; }
    qsort(tmp, (size_t)I, sizeof(vid), cmp_desc);
    for (int64_t i = 0; i < I; ++i) order[(int64_t)f * I + i] = tmp[i].id;
  }
  free(tmp);
}

int32_t orc_adaptive_pick_literal(const float* QT, int64_t I, const int64_t* indptr,
                                  const int32_t* indices, int32_t user, int32_t factor,
                                  int32_t rank) {
  vid* tmp = (vid*)malloc(sizeof(vid) * (size_t)I);
  for (int64_t i = 0; i < I; ++i) { tmp[i].v = QT[(int64_t)factor * I + i]; tmp[i].id = (int32_t)i; }
  for (int64_t k = indptr[user]; k < indptr[user + 1]; ++k) tmp[indices[k]].v = -1e13f; /* :115 */
  tmp[0].v = -1e13f; /* hstack((seen, zeros)) :109-113 */
  qsort(tmp, (size_t)I, sizeof(vid), cmp_desc); /* argsort(−x) :114-117 */
  int32_t r = tmp[rank].id;                     /* gather(rank) :119 */
  free(tmp);
  return r;
}

int32_t orc_adaptive_pick(const int32_t* order, int64_t I, const int64_t* indptr,
                          const int32_t* indices, int32_t user, int32_t factor, int32_t rank) {
  const int32_t* o = order + (int64_t)factor * I;
  int32_t left = rank;
  for (int64_t t = 0; t < I; ++t) {
    int32_t item = o[t];
    if (item == 0 || csr_contains(indptr, indices, user, item)) continue;
    if (left == 0) return item;
    --left;
  }
  return 0;
}

static void adaptive_draw(const float* p, int32_t d, const float* sigma, int64_t I, int64_t n_seen,
                          float geo_p, uint64_t seed, uint64_t t, int32_t* factor, int32_t* rank) {
  uint32_t rf = draw(seed, t, 0u, 1u, 0);
  uint32_t rg = draw(seed, t, 0u, 1u, 1);
  /* factor ~ Categorical(|p_uf|·σ_f)  (neg_samplers.py:84-88) by inverse CDF.  The reference hands
   * the weights to torch.multinomial; any enumeration of the factors gives the same distribution,
   * and a draw only has to be reproducible between this oracle and the product.  The product's
   * wavefront group holds factor f in lane f mod G (G = 32 lanes for d <= 128, else 64), so its
   * CDF runs lane by lane: factors are enumerated in the order (f mod G, f div G). */
  const int32_t G = d <= 128 ? 32 : 64;
  double total = 0.0;
  for (int32_t l = 0; l < G; ++l)
    for (int32_t f = l; f < d; f += G) total += (double)(fabsf(p[f]) * sigma[f]);
  float uf = (float)(rf >> 8) * (1.0f / 16777216.0f);
  double thr = (double)uf * (double)(float)total;
  double cum = 0.0;
  int32_t fsel = -1, last_pos = 0;
  for (int32_t l = 0; l < G; ++l)
    for (int32_t f = l; f < d; f += G) {
      float wf = fabsf(p[f]) * sigma[f];
      cum += (double)wf;
      if (wf > 0.0f) last_pos = f;
      if (fsel < 0 && cum > thr && wf > 0.0f) fsel = f;
    }
  if (fsel < 0) fsel = last_pos;
  /* r ~ Geometric(p) on {1,2,…} (:90-93), clamped to the number of unseen items (:94) */
  float ug = (float)((rg >> 8) + 1u) * (1.0f / 16777216.0f);
  float inv = (float)(1.0 / log1p(-(double)geo_p));
  float rr = ceilf(logf(ug) * inv);
  int64_t n_unseen = (I - 1) - n_seen;
  int64_t r = rr < 1.0f ? 1 : (rr > 2.0e9f ? 2000000000 : (int64_t)rr);
  if (r > n_unseen) r = n_unseen;
  /* orientation by the sign of p_uf (:96-100) */
  *factor = fsel;
  *rank = (int32_t)(p[fsel] > 0.0f ? r - 1 : n_unseen - r);
}

void orc_sample_adaptive(const float* P, int32_t d, const float* sigma, const int32_t* order,
                         int64_t I, const int64_t* indptr, const int32_t* indices,
                         const int32_t* users, int64_t B, float p, uint64_t seed, uint64_t offset,
                         int32_t* neg_out, int32_t* factor_out, int32_t* rank_out) {
  for (int64_t b = 0; b < B; ++b) {
    int32_t u = users[b], f, rk;
    adaptive_draw(P + (int64_t)u * d, d, sigma, I, indptr[u + 1] - indptr[u], p, seed,
                  offset + (uint64_t)b, &f, &rk);
    if (factor_out) factor_out[b] = f;
    if (rank_out) rank_out[b] = rk;
    neg_out[b] = orc_adaptive_pick(order, I, indptr, indices, u, f, rk);
  }
}

/* ------------------------------------------------------------------------------------------
 * sequential stream (B = 1 SGD)
 * ---------------------------------------------------------------------------------------- */
static void stream_seq_impl(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
                          const int32_t* users, const int32_t* pos, int32_t* neg_io, int64_t n,
                          int32_t sampler, float adaptive_p, const float* sigma,
                          const int32_t* order, const int64_t* indptr, const int32_t* indices,
                          uint64_t seed, uint64_t offset, float a_user, float a_item, float a_neg,
                          int32_t pad_user, int32_t pad_item, float lr, double* scalars) {
  (void)U;
  double loss = 0, reg = 0, sabs = 0;
  for (int64_t b = 0; b < n; ++b) {
    int32_t u = users[b], i = pos[b], j;
    float* p = P + (int64_t)u * d;
    if (sampler == ORC_NEG_UNIFORM) {
      j = sample_uniform_one(indptr, indices, I, u, seed, offset + (uint64_t)b);
    } else if (sampler == ORC_NEG_ADAPTIVE) {
      int32_t f, rk;
      adaptive_draw(p, d, sigma, I, indptr[u + 1] - indptr[u], adaptive_p, seed,
                    offset + (uint64_t)b, &f, &rk);
      j = orc_adaptive_pick(order, I, indptr, indices, u, f, rk);
    } else {
      j = neg_io[b];
    }
    if (neg_io && sampler != ORC_NEG_GIVEN) neg_io[b] = j;
    float* qi = Q + (int64_t)i * d;
    float* qj = Q + (int64_t)j * d;
    float xp = (float)(ddot(p, qi, d) + (item_bias ? (double)item_bias[i] : 0.0));
    float xn = (float)(ddot(p, qj, d) + (item_bias ? (double)item_bias[j] : 0.0));
    double x = (double)(xp - xn), w = 1.0 / (1.0 + exp(x));
    loss += neg_logsigmoid(x);
    sabs += fabs(x);
    reg += 0.5 * ((double)a_item * ddot(qi, qi, d) + (double)a_neg * ddot(qj, qj, d) +
                  (double)a_user * ddot(p, p, d));
    int same = (i == j);
    for (int32_t k = 0; k < d; ++k) {
      double pk = p[k], qik = qi[k], qjk = qj[k];
      double gp = -w * (qik - qjk) + (double)a_user * pk;
      double gi = -w * pk + (double)a_item * qik;
      double gj = w * pk + (double)a_neg * qjk;
      if (u != pad_user) p[k] = (float)(pk - (double)lr * gp);
      if (same) {
        if (i != pad_item) qi[k] = (float)(qik - (double)lr * (gi + gj));
      } else {
        if (i != pad_item) qi[k] = (float)(qik - (double)lr * gi);
        if (j != pad_item) qj[k] = (float)(qjk - (double)lr * gj);
      }
    }
    if (item_bias && !same) {
      item_bias[i] = (float)((double)item_bias[i] + (double)lr * w);
      item_bias[j] = (float)((double)item_bias[j] - (double)lr * w);
    }
  }
  if (scalars) { scalars[0] = loss; scalars[1] = reg; scalars[2] = sabs; scalars[3] = (double)n; }
}

void orc_train_stream_seq(float* P, float* Q, float* item_bias, int64_t U, int64_t I, int32_t d,
                          const int32_t* users, const int32_t* pos, int32_t* neg_io, int64_t n,
                          int32_t sampler, float adaptive_p, const float* sigma,
                          const int32_t* order, const int64_t* indptr, const int32_t* indices,
                          uint64_t seed, uint64_t offset, float a_user, float a_item, float a_neg,
                          int32_t pad_user, int32_t pad_item, float lr, double* scalars) {
  stream_seq_impl(P, Q, item_bias, U, I, d, users, pos, neg_io, n, sampler, adaptive_p, sigma, order,
                  indptr, indices, seed, offset, a_user, a_item, a_neg, pad_user, pad_item, lr,
                  scalars);
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline (a) of SURVEY §8d: the same path — sample a mini-batch's negatives, evaluate its B
 * gradients at the parameters before the step, one sparse SGD step, the adaptive snapshot retaken
 * every `refresh_every` batches — with OpenMP over the triples of a batch (and over the columns
 * of a refresh).  bench.py times it; no test relies on it beyond equality with the serial route.
 * ---------------------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void order_parallel(const float* QT, int64_t I, int32_t d, int32_t* order) {
#pragma omp parallel
  {
    vid* tmp = (vid*)malloc(sizeof(vid) * (size_t)I);
#pragma omp for schedule(dynamic, 1)
    for (int32_t f = 0; f < d; ++f) {
      for (int64_t i = 0; i < I; ++i) { tmp[i].v = QT[(int64_t)f * I + i]; tmp[i].id = (int32_t)i; }
      qsort(tmp, (size_t)I, sizeof(vid), cmp_desc);
      for (int64_t i = 0; i < I; ++i) order[(int64_t)f * I + i] = tmp[i].id;
    }
    free(tmp);
  }
}

/* Trains mini-batches [0, n / B) of the stream; returns the number of triples trained, or -1.
 * sampler: ORC_NEG_UNIFORM | ORC_NEG_ADAPTIVE.  QT / sigma / order: snapshot buffers of the caller
 * ([d, I], [d], [d, I]), valid on entry for the adaptive sampler.  seconds > 0: stop after the batch
 * that crosses that wall-clock budget.  threads <= 0: the OpenMP default. */
int64_t orc_train_batches_omp(float* P, float* Q, int64_t U, int64_t I, int32_t d, const int32_t* users,
                              const int32_t* pos, int64_t n, int64_t B, int32_t sampler, float adaptive_p,
                              float* QT, float* sigma, int32_t* order, int64_t refresh_every,
                              const int64_t* indptr, const int32_t* indices, uint64_t seed, uint64_t offset,
                              float a_user, float a_item, float a_neg, int32_t pad_user, int32_t pad_item,
                              float lr, double seconds, int32_t threads, double* scalars) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  const double t_begin = omp_get_wtime();
#else
  (void)threads; (void)seconds;
#endif
  float* GP = (float*)calloc((size_t)U * (size_t)d, sizeof(float));
  float* GQ = (float*)calloc((size_t)I * (size_t)d, sizeof(float));
  int32_t* flagP = (int32_t*)calloc((size_t)U, sizeof(int32_t));
  int32_t* flagQ = (int32_t*)calloc((size_t)I, sizeof(int32_t));
  int64_t* touched = (int64_t*)malloc(sizeof(int64_t) * (size_t)(3 * B));
  int32_t* neg = (int32_t*)malloc(sizeof(int32_t) * (size_t)B);
  if (!GP || !GQ || !flagP || !flagQ || !touched || !neg) {
    free(GP); free(GQ); free(flagP); free(flagQ); free(touched); free(neg);
    return -1;
  }
  double loss = 0, reg = 0, sabs = 0;
  int64_t done = 0, iter = 0;
  for (int64_t lo = 0; lo + B <= n; lo += B) {
    const int32_t* us = users + lo;
    const int32_t* ps = pos + lo;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t b = 0; b < B; ++b) {
      const int32_t u = us[b];
      if (sampler == ORC_NEG_ADAPTIVE) {
        int32_t f, rk;
        adaptive_draw(P + (int64_t)u * d, d, sigma, I, indptr[u + 1] - indptr[u], adaptive_p, seed,
                      offset + (uint64_t)(lo + b), &f, &rk);
        neg[b] = orc_adaptive_pick(order, I, indptr, indices, u, f, rk);
      } else {
        neg[b] = sample_uniform_one(indptr, indices, I, u, seed, offset + (uint64_t)(lo + b));
      }
    }
    ++iter;
    if (sampler == ORC_NEG_ADAPTIVE && refresh_every > 0 && iter % refresh_every == 0) {
      orc_adaptive_stats(Q, I, d, QT, sigma);  /* AdaptiveSampler.update_stats (neg_samplers.py:122-132) */
      order_parallel(QT, I, d, order);
    }
    int64_t n_touched = 0;
#pragma omp parallel for schedule(static) reduction(+ : loss, reg, sabs)
    for (int64_t b = 0; b < B; ++b) {
      const int32_t u = us[b], i = ps[b], j = neg[b];
      const float* p = P + (int64_t)u * d;
      const float* qi = Q + (int64_t)i * d;
      const float* qj = Q + (int64_t)j * d;
      const double x = (double)((float)ddot(p, qi, d) - (float)ddot(p, qj, d));
      const float w = (float)(1.0 / (1.0 + exp(x)));
      loss += neg_logsigmoid(x);
      sabs += fabs(x);
      reg += 0.5 * ((double)a_item * ddot(qi, qi, d) + (double)a_neg * ddot(qj, qj, d) +
                    (double)a_user * ddot(p, p, d));
      float* gu = GP + (int64_t)u * d;
      float* gi = GQ + (int64_t)i * d;
      float* gj = GQ + (int64_t)j * d;
      for (int32_t k = 0; k < d; ++k) {
        const float pk = p[k], qik = qi[k], qjk = qj[k];
        const float du = -w * (qik - qjk) + a_user * pk, di = -w * pk + a_item * qik,
                    dj = w * pk + a_neg * qjk;
#pragma omp atomic
        gu[k] += du;
#pragma omp atomic
        gi[k] += di;
#pragma omp atomic
        gj[k] += dj;
      }
      int64_t slot;
      if (__atomic_exchange_n(&flagP[u], 1, __ATOMIC_RELAXED) == 0) {
        slot = __atomic_fetch_add(&n_touched, 1, __ATOMIC_RELAXED);
        touched[slot] = 2 * (int64_t)u;
      }
      if (__atomic_exchange_n(&flagQ[i], 1, __ATOMIC_RELAXED) == 0) {
        slot = __atomic_fetch_add(&n_touched, 1, __ATOMIC_RELAXED);
        touched[slot] = 2 * (int64_t)i + 1;
      }
      if (__atomic_exchange_n(&flagQ[j], 1, __ATOMIC_RELAXED) == 0) {
        slot = __atomic_fetch_add(&n_touched, 1, __ATOMIC_RELAXED);
        touched[slot] = 2 * (int64_t)j + 1;
      }
    }
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < n_touched; ++s) {
      const int64_t key = touched[s], row = key >> 1;
      float* w = (key & 1) ? Q + row * d : P + row * d;
      float* g = (key & 1) ? GQ + row * d : GP + row * d;
      const int pad = (key & 1) ? row == pad_item : row == pad_user;
      for (int32_t k = 0; k < d; ++k) {
        if (!pad) w[k] = w[k] - lr * g[k];
        g[k] = 0.0f;
      }
      if (key & 1) flagQ[row] = 0; else flagP[row] = 0;
    }
    done += B;
#ifdef _OPENMP
    if (seconds > 0.0 && omp_get_wtime() - t_begin >= seconds) break;
#endif
  }
  if (scalars) { scalars[0] = loss; scalars[1] = reg; scalars[2] = sabs; scalars[3] = (double)done; }
  free(GP); free(GQ); free(flagP); free(flagQ); free(touched); free(neg);
  return done;
}
