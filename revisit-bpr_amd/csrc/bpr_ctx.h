// bpr_ctx.h — private state of a bpr_ctx (shared by bprcore.hip and bpr_refresh.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <map>
#include <string>
#include <tuple>
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
  int last_run_len = 0;  // what the last STREAM launch used
  int stream_cus = 0;    // CUs of `stream` (its CU mask's popcount), cached per stream handle
  void* stream_cus_of = nullptr;
  std::map<std::tuple<int, int, int, int, int64_t>, int> stream_occ;  // k_stream blocks per CU, per instantiation
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
  uint8_t* v_alone = nullptr;  // [v_alone_cap] per-triple "user alone in its virtual batch" (k_valone)
  int64_t v_alone_cap = 0;
  // private scratch — adaptive sampler snapshot.  Two snapshots: `order` / `sigma` point at the
  // FRONT one (what the samplers read); a refresh sorts into the back one and swaps.
  int32_t* order = nullptr;  // [d, I], inside order_alloc with BPR_ORDER_PAD entries of slack on
  float* sigma = nullptr;    // [d]       both ends (the sampler's walk reads 16-byte vectors)
  int32_t* order_alloc[2] = {nullptr, nullptr};
  float* sigma_buf[2] = {nullptr, nullptr};
  int snap_front = 0;
  float* keysT = nullptr;    // [d, I] transposed item table: the keys the NEXT cut writes / the
  float* keysT_buf[2] = {nullptr, nullptr};  // sort queued last reads (two buffers: a cut never
  int keys_w = 0;                            // waits for the sort still reading the other one)
  float* keys_sorted = nullptr;  // 2 x [d, I] uint64 composite sort keys (in | out)
  int32_t* ids_in = nullptr;
  int32_t* seg_offsets = nullptr;  // [d+1]
  double* sig_acc = nullptr;       // [d, 2] shifted sum / sum of squares per factor (of keysT)
  double* sig_acc_buf[2] = {nullptr, nullptr};
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  bool have_snapshot = false;
  // PARTIAL snapshots (bpr_set_tuning "partial_snapshot"; k_sort_partial): per snapshot buffer {kt, kb} of
  // every column, whether that buffer holds a partial order, and the key buffer it was sorted from (the
  // walk's in-bin finish reads it: it outlives the snapshot, see refresh_impl); *_front = the pair the
  // samplers read (meta_front NULL = sorted whole)
  int tune_binned = 1;        // 1: columns of 2,048 .. 20,480 keys are ordered by k_sort_binned (0: the radix sort)
  int tune_binned_split = 0;  // 0: workgroups per column of the binned sort by shape; 1..4: forced (tests)
  int tune_partial = 0;       // 1: the split refresh sorts partially when the shape allows
  int partial_target = 640;   // keys aimed at per exact end (at most 1,024 fit: k_sort_partial's PART_CAP)
  int32_t* snap_meta[2] = {nullptr, nullptr};
  bool snap_partial[2] = {false, false};
  const float* snap_keys[2] = {nullptr, nullptr};
  const int32_t* meta_front = nullptr;
  const float* keys_front = nullptr;
  bool keys_front_stale = false;  // a later cut has overwritten the key buffer the partial FRONT snapshot was sorted from
  // split refresh (bpr_adaptive_refresh_begin / _commit): the keys are cut on `stream`, the sort
  // runs on `side` while the caller keeps launching on `stream`, commit orders the swap
  hipStream_t side = nullptr;
  bool side_owned = false;
  hipEvent_t ev_keys = nullptr, ev_sorted = nullptr;
  bool refresh_pending = false;
  bool part_pending = false;  // bpr_adaptive_refresh_part: this rank's columns are sorted, publish pending
  // bpr_train_stream_cut: the launch's epilogue already cut the next snapshot's keys (keysT); the
  // next bpr_adaptive_refresh_begin only queues the sort.  Any call that moves the item table
  // afterwards clears it.
  bool keys_cut = false;
  bool keys_event = false;  // ev_keys was recorded by the cut kernel itself (hipExtLaunchKernelGGL)
  bool keys_on_side = false;  // ... and that cut ran on `side` (bpr_train_stream_acut): a sort on `stream` waits for it
  bool acut_pending = false;  // an asynchronous cut may still be reading Q, the hot deltas and a launch's partials
  // private scratch — epoch planner
  uint64_t* plan_keys = nullptr;
  uint64_t* plan_keys_sorted = nullptr;
  void* plan_tmp = nullptr;
  size_t plan_tmp_bytes = 0;
  int64_t plan_cap = 0;
  const int32_t* plan_users = nullptr;  // outputs of the last bpr_plan_epoch (caller-owned)
  const int32_t* plan_pos = nullptr;
  int64_t plan_n = 0, plan_chunk = 0;
  // bpr_plan_chunk scratch: keys / values of one chunk before the by-user sort
  uint32_t *pc_keys = nullptr, *pc_keys2 = nullptr, *pc_cnt = nullptr;
  int32_t *pc_vals = nullptr, *pc_vals2 = nullptr;
  void* pc_tmp = nullptr;
  size_t pc_tmp_bytes = 0;
  int64_t pc_cap = 0;
  // hot item rows (bpr_set_hot_rows): the most popular rows take their STREAM updates in replica
  // delta rows, folded into Q right after every STREAM launch (all zero in between)
  int hot_rows_opt = 256, hot_reps_opt = 1;
  int hot_H = 0, hot_R = 0;
  int32_t* hot_slot = nullptr;   // [I] slot of an item row, -1 = not hot
  int32_t* hot_items = nullptr;  // [hot_H]
  float* hot_delta = nullptr;    // [hot_R, hot_H, d], aligned to a channel round inside hot_delta_alloc
  void* hot_delta_alloc = nullptr;
  double hot_balance = 1.0;      // modelled max / mean channel load of a launch (hot_build_impl)
  const int32_t* hot_key_ptr = nullptr;  // training positives the popularity was measured on
  // two-tier item reconciliation (several GPUs): the hot set is given by the caller (the same on
  // every rank, bpr_set_hot_items), hot_canon[slot] = the row's position in that list, and while
  // hot_tier is on the launches leave their deltas in the block for bpr_hot_exchange
  bool hot_explicit = false;
  bool hot_tier = false;
  // bpr_train_stream_cut under the hot tier: the launch's epilogue (loss partials) is left to the
  // bpr_sync_cut that must follow
  // bpr_train_stream_acut: the cut of the next snapshot runs on the side stream beside the NEXT
  // launch and folds nothing — the hot deltas stay in the block until somebody needs the table
  // whole (hot_fold_impl: every other entry point, bpr_hot_fold)
  bool hot_unfolded = false;
  bool acut_call = false;
  int acut_parity = 0;
  hipEvent_t ev_launch = nullptr;
  int defer_blocks = 0;
  float* defer_out = nullptr;
  bool defer_pending = false;
  int32_t* hot_canon = nullptr;  // [hot_H]
  // LDS tier of the hot block (k_stream LDSHOT, bpr_hotlds.hip): hot_code[i] = -1 | (popularity rank << 16 | slot),
  // hot_by_rank[rank] = slot; tune_hot_lds = rows asked for per workgroup (0: off), tune_hot_lds_force: also for
  // launches that do not fill the chip (tests)
  int32_t* hot_code = nullptr;     // [I]
  int32_t* hot_by_rank = nullptr;  // [hot_H]
  int tune_hot_lds = 0, tune_hot_lds_force = 0;
  int tune_acut_fold = 1;          // bpr_train_stream_acut: 1 = the fold stays on the launch stream (r6), 0 = r4's read-only form
  int tune_plan_sorted = 0;        // 1: the caller promises bpr_plan_epoch's users_in sorted by user (one radix pass instead of three)
  int tune_lds_tail = 12;          // percent of a launch's triples dealt in short runs at the end (LDS-tier kernel)
  int tune_lds_block = 0;          // measurement aid: threads per workgroup of the LDS-tier kernel (0: 1,024 / 512 by shape)
  int last_lds_rows = 0;           // LDS rows of the last STREAM launch (0: the plain kernel ran)
  int64_t hot_key_n = 0;
  // heavy users' seen bitmaps (built once per seen CSR, by the first sampling STREAM launch)
  uint32_t* heavy_off = nullptr;   // [U] word offset of the user's row in heavy_bits, ~0u = light
  uint32_t* heavy_bits = nullptr;  // [n_heavy, words]
  int heavy_T = 0;                 // users with more seen items than this are heavy
  int heavy_T_opt = 256;           // bpr_set_heavy_users: requested threshold (-1: no table)
  int64_t heavy_max_bytes = (int64_t)1 << 30;  // ... and the most HBM the bitmaps may take
  int64_t heavy_n = 0;
  const int64_t* heavy_for = nullptr;  // the indptr the table was built from (NULL = not built)
  // test / measurement aids (bpr_set_tuning): which "seen?" structure the sampling kernels use
  // (0 = by shape, 1 = CSR search, 2 = LDS bitmap, 3 = staged list) and whether the batched STREAM
  // kernel steps lonely user rows directly (-1 = by optimizer, 0 / 1 = forced)
  int tune_seen = 0, tune_vs_direct = -1;
  int tune_adam_closed = 1;  // 0: Adam's missed steps are replayed by the step loop, never in closed form (tests compare both routes)
  int tune_refresh_sub = 0;  // 1 | 2 | 4: workgroups per column of the in-LDS snapshot sort (0 = by shape; tests force the merge paths)
  void* comm = nullptr;  // bpr_comm.hip: the RCCL communicator and the reconciliation buffers (NULL: one GPU)
  // scalar slots
  float* bias_w = nullptr;  // [I * BIAS_LINE] the item_bias k_stream works on, one item per 128-B line (bpr_kernels.h)
  int64_t bias_w_rows = 0;
  bool bias_track = false;    // bpr_set_bias_tracking: skip the refill while the wide table is known to be current
  bool bias_w_valid = false;  // wide table == dense vector (written back by the last launch, untouched since)
  const float* bias_w_of = nullptr;
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
int refresh_impl(bpr_ctx* c, bool split, int f_lo, int f_hi);  // bpr_refresh.hip: split = sort on c->side, no swap
int refresh_publish_impl(bpr_ctx* c);       // bpr_refresh.hip
int refresh_alloc(bpr_ctx* c);             // bpr_refresh.hip: the snapshot buffers (idempotent)
int refresh_commit_impl(bpr_ctx* c);        // bpr_refresh.hip
int snapshot_complete_impl(bpr_ctx* c);     // bpr_refresh.hip: a partial front snapshot is sorted whole, in place
void refresh_free(bpr_ctx* c);      // bpr_refresh.hip
void side_free(bpr_ctx* c);         // bpr_refresh.hip
int heavy_build_impl(bpr_ctx* c);   // bpr_refresh.hip
void heavy_free(bpr_ctx* c);        // bpr_refresh.hip
int hot_build_impl(bpr_ctx* c, const int32_t* pos, int64_t n);  // bpr_refresh.hip
int hot_set_items_impl(bpr_ctx* c, const int32_t* items, int H, const uint32_t* counts);  // bpr_refresh.hip
int flush_deferred_stats(bpr_ctx* c);  // bprcore.hip: loss partials a hot-tier cut launch left behind
int hot_fold_impl(bpr_ctx* c);  // bprcore.hip: fold deltas left by bpr_train_stream_acut (no-op otherwise)
void hot_free(bpr_ctx* c);                                       // bpr_refresh.hip
int plan_chunk_impl(bpr_ctx* c, const int32_t* users_in, const int32_t* pos_in, int64_t n, int64_t chunk,
                    uint64_t seed, int64_t index, int32_t* users_out, int32_t* pos_out, hipStream_t st);
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
