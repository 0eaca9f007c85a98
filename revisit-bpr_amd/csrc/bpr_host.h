// bpr_host.h — host-side helpers shared by the translation units of libbprcore (bprcore.hip,
// bpr_vstream.hip).
#pragma once
#include <string>

#include "bpr_ctx.h"
#include "bpr_opt.h"

namespace bpr {

int fail(int code, const std::string& msg);                 // sets bpr_last_error, returns code
OptDev opt_dev(const bpr_ctx* c, int64_t t);                // device view of the ctx's optimizer
int check_bound(const bpr_ctx* c, const char* who, bool whole_table = true);  // tables bound? (+ folds hot deltas an async cut left)
int check_opt_state(const bpr_ctx* c, const char* who);     // state tensors the kind needs bound?
int check_triples(const bpr_ctx* c, const char* who, const int32_t* users, const int32_t* pos,
                  int64_t B);
int check_sampler(const bpr_ctx* c, const char* who, int32_t sampler, float adaptive_p,
                  const int32_t* neg, int64_t B);
float inv_log1mp(float p);
int strict_flush_impl(bpr_ctx* c);                          // STRICT lazy-replay flush (bprcore.hip)
int vs_flush(bpr_ctx* c, bool users, bool items);           // bpr_vstream.hip
int vs_leave(bpr_ctx* c);                                   // bpr_vstream.hip
void vs_free(bpr_ctx* c);                                   // bpr_vstream.hip
void comm_free(bpr_ctx* c, bool tables_alive);              // bpr_comm.hip
int comm_world(const bpr_ctx* c);                           // bpr_comm.hip (1 without a communicator)
int comm_rank(const bpr_ctx* c);
int comm_gather_snapshot(bpr_ctx* c, int32_t* order_back, float* sigma_back, int per);  // bpr_comm.hip

// ---- (G, E) dispatch: G lanes per triple, E elements per lane (bpr_device.h) --------------------
template <int A, int B>
struct GE {
  static constexpr int G = A;
  static constexpr int E = B;
};

template <typename F>
inline int dispatch_ge(int G, int E, F&& f) {
  switch (G * 32 + E) {
    case 32 * 32 + 1: return f(GE<32, 1>{});
    case 32 * 32 + 2: return f(GE<32, 2>{});
    case 32 * 32 + 4: return f(GE<32, 4>{});
    case 64 * 32 + 4: return f(GE<64, 4>{});
    case 64 * 32 + 8: return f(GE<64, 8>{});
    case 64 * 32 + 16: return f(GE<64, 16>{});
    default: return fail(BPR_ERR_UNSUPPORTED, "unsupported embedding dim");
  }
}

// bpr_hotlds.hip: k_stream with the LDS tier of the hot block.  lds_tier_room(d) = what one workgroup may
// declare dynamically beside the kernels' static LDS (sigma: 4 d bytes, reduction scratch) of a CU's 160 KiB
// dynamic LDS a launch of the LDS-tier kernel may use at width d: the CU's 160 KiB less the kernel's static LDS
// (sigma: 4 d bytes rounded to the group layout, the loss reduction's 256 B, the ticket) and 256 B of slack
inline size_t lds_tier_room(int d) { return 160 * 1024 - ((size_t)4 * (size_t)((d + 63) / 64 * 64) + 256 + 16 + 256); }
struct StreamArgs;
int launch_stream_lds(bpr_ctx* c, const StreamArgs& a, int sampler, int seen, unsigned grid, unsigned block,
                      size_t shmem, hipEvent_t stop);

struct Timer {
  bpr_ctx* c;
  size_t slot = 0;
  bool on = false;
  Timer(bpr_ctx* ctx, bool enabled) : c(ctx) {
    if (!enabled || !c->timing) return;
    if ((c->timing_seen++ % c->timing_stride) != 0) return;  // every timing_stride-th launch
    if (c->ev_used == c->ev_start.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->ev_start.push_back(a);
      c->ev_stop.push_back(b);
    }
    slot = c->ev_used++;
    on = true;
    hipEventRecord(c->ev_start[slot], c->stream);
  }
  ~Timer() {
    if (on) hipEventRecord(c->ev_stop[slot], c->stream);
  }
};

}  // namespace bpr
