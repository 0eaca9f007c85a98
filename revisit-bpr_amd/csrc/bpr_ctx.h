// bpr_ctx.h — private state of a bpr_ctx (shared by bprcore.hip and bpr_refresh.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/bprcore.h"

constexpr int BPR_ORDER_PAD = 4;

struct bpr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // bound model state (caller-owned)
  float* P = nullptr;
  float* Q = nullptr;
  float* bias = nullptr;
  int64_t U = 0, I = 0;
  int d = 0, G = 0, E = 0;
  bool grouped = false;  // STREAM: chunks are grouped by user (bpr_plan_epoch output)
  int run_len = 0;       // STREAM: consecutive triples walked by one group (0: by launch size)
  int pad_user = -1, pad_item = -1;
  const int64_t* indptr = nullptr;
  const int32_t* indices = nullptr;
  float au = 0.f, ai = 0.f, an = 0.f;
  const float* w_accept = nullptr;  // item weights of the uniform sampler (alias table, caller-owned)
  const int32_t* w_alias = nullptr;
  int opt_kind = BPR_OPT_SGD;
  bpr_opt_params opt = {0.f, 0.f, 0.f, 0, 0.9f, 0.999f, 1e-8f, 0.99f};
  float *mP = nullptr, *vP = nullptr, *mQ = nullptr, *vQ = nullptr, *mb = nullptr, *vb = nullptr;
  // private scratch — STRICT path
  float *GP = nullptr, *GQ = nullptr, *Gb = nullptr;
  int32_t *flagP = nullptr, *flagQ = nullptr;
  int32_t *lastP = nullptr, *lastQ = nullptr;
  uint32_t* touched = nullptr;
  uint32_t* touched_cnt = nullptr;  // two counters: [cnt_sel] is live, k_apply clears the other
  int cnt_sel = 0;
  bool defer_stats = false;  // bpr_train_strict: k_triples accumulates into the partials scratch
  int64_t pending = 0;  // upper bound of entries in `touched`
  int64_t step = 0;     // optimizer steps applied so far
  int64_t flushed_at = 0;
  int64_t strict_iter = 0;  // batches drawn by bpr_train_strict so far (AdaptiveSampler._iteration_cnt)
  // private scratch — batched STREAM (bpr_vstream.hip): per-row headers (last | gstep | slot | lock)
  // and double-buffered gradient accumulators [2, rows, d]; while vs_active the headers — not
  // lastP / lastQ — say how far each row has been advanced
  float *vGP = nullptr, *vGQ = nullptr, *vGb = nullptr;
  uint64_t *vHP = nullptr, *vHQ = nullptr;
  bool vs_active = false;
  // private scratch — adaptive sampler snapshot
  int32_t* order = nullptr;  // [d, I], inside order_alloc with BPR_ORDER_PAD entries of slack on
  int32_t* order_alloc = nullptr;  // both ends (the sampler's walk reads 16-byte vectors)
  float* sigma = nullptr;    // [d]
  float* keysT = nullptr;    // [d, I] transposed item table
  float* keys_sorted = nullptr;  // 2 x [d, I] uint64 composite sort keys (in | out)
  int32_t* ids_in = nullptr;
  int32_t* seg_offsets = nullptr;  // [d+1]
  double* sig_acc = nullptr;       // [d, 2] shifted sum / sum of squares per factor
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  bool have_snapshot = false;
  // private scratch — epoch planner
  uint64_t* plan_keys = nullptr;
  uint64_t* plan_keys_sorted = nullptr;
  void* plan_tmp = nullptr;
  size_t plan_tmp_bytes = 0;
  int64_t plan_cap = 0;
  // STREAM with deferred positives (bpr_set_defer_positives): the plan also holds, per chunk, the
  // chunk's triples ordered by positive item; the hot kernel parks sigma(-x) per triple in wbuf and
  // k_pos_pass applies each positive row's summed update once
  int defer_pos = 0;                   // 0 off | 1 rows outside the hot block | 2 every positive row
  int32_t* plan_perm = nullptr;        // [plan_n] triple indices, every chunk sorted by positive
  int32_t* plan_iota = nullptr;        // [plan_perm_cap] 0, 1, 2, ...
  int32_t* plan_pos_sorted = nullptr;  // [plan_n] pos[plan_perm[k]]  (coalesced for k_pos_pass)
  int32_t* plan_users_bypos = nullptr; // [plan_n] users[plan_perm[k]]
  int64_t plan_perm_cap = 0;
  int32_t* plan_cnt = nullptr;         // [n_chunks, I] positives of an item inside a chunk
  int64_t plan_cnt_cap = 0;
  const int32_t* plan_users = nullptr;  // outputs of the last bpr_plan_epoch (caller-owned)
  const int32_t* plan_pos = nullptr;
  int64_t plan_n = 0, plan_chunk = 0;
  bool plan_perm_valid = false;
  float* wbuf = nullptr;               // [wbuf_cap] sigma(-x) of the triples of one launch
  int64_t wbuf_cap = 0;
  // hot item rows (bpr_set_hot_rows): the most popular rows take their STREAM updates in replica
  // delta rows, folded into Q right after every STREAM launch (all zero in between)
  int hot_rows_opt = 256, hot_reps_opt = 1;
  int hot_H = 0, hot_R = 0;
  int32_t* hot_slot = nullptr;   // [I] slot of an item row, -1 = not hot
  int32_t* hot_items = nullptr;  // [hot_H]
  float* hot_delta = nullptr;    // [hot_R, hot_H, d]
  const int32_t* hot_key_ptr = nullptr;  // training positives the popularity was measured on
  int64_t hot_key_n = 0;
  // scalar slots
  float* dev_scalars = nullptr;
  // timing of the dominant kernel
  bool timing = false;
  int timing_stride = 1;      // bpr_timing_enable(ctx, N): events around every N-th launch
  int64_t timing_seen = 0;
  std::vector<hipEvent_t> ev_start, ev_stop;
  size_t ev_used = 0;
  double timed_ms = 0.0;
  int64_t timed_launches = 0;
};

namespace bpr {
void set_error(const std::string& msg);
int refresh_impl(bpr_ctx* c);       // bpr_refresh.hip
void refresh_free(bpr_ctx* c);      // bpr_refresh.hip
int hot_build_impl(bpr_ctx* c, const int32_t* pos, int64_t n);  // bpr_refresh.hip
void hot_free(bpr_ctx* c);                                       // bpr_refresh.hip
int plan_epoch_impl(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n,
                    int64_t chunk, uint64_t seed, int32_t* users_out, int32_t* pos_out);
}  // namespace bpr

#define BPR_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      bpr::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return BPR_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
