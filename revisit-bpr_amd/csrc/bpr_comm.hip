// bpr_comm.hip — multi-GPU behind the C ABI (bpr_comm_init / bpr_item_sync).
//
// The item table is replicated on every rank and reconciled by an all-reduce(SUM) of each rank's
// DELTA since the last reconciliation:  Q <- Q_base + sum_r (Q_r - Q_base)  (the
// reference has no working multi-device path to mirror: its DDP launcher,
// experiments/launcher.py:35-73, is never enabled by a config).  revisit_bpr/distributed.py runs
// this protocol through torch.distributed; here the same protocol runs inside the library over RCCL
// — one communicator per ctx, the collective on a side stream under the next STREAM launch — for
// hosts that are not torch.  RCCL is reached through dlopen (librccl.so.1: the copy the process
// already holds, e.g. torch's, or /opt/rocm/lib's), so libbprcore.so has no link-time dependency on
// it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <string>

#include <algorithm>
#include <new>

#include "bpr_host.h"

namespace bpr {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl* rccl() {
  // resolved once (thread-safe: a function-local static's initialiser)
  static const Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.handle != nullptr) break;
    }
    if (x.handle != nullptr) {
      x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
      x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
      x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
      x.AllReduce = (decltype(x.AllReduce))dlsym(x.handle, "ncclAllReduce");
      x.AllGather = (decltype(x.AllGather))dlsym(x.handle, "ncclAllGather");
      x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
      if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllReduce || !x.AllGather)
        x.handle = nullptr;
    }
    return x;
  }();
  return r.handle != nullptr ? const_cast<Rccl*>(&r) : nullptr;
}

#define BPR_NCCL_CHECK(expr)                                                              \
  do {                                                                                    \
    ncclResult_t _r = (expr);                                                             \
    if (_r != ncclSuccess) {                                                              \
      Rccl* _l = rccl();                                                                  \
      set_error(std::string(#expr) + ": " +                                               \
                (_l && _l->GetErrorString ? _l->GetErrorString(_r) : "RCCL error"));      \
      return BPR_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;  // the collectives' stream
  hipEvent_t ev_cut = nullptr, ev_done = nullptr;
  float *base = nullptr, *own = nullptr, *tot = nullptr;     // [I*d] each
  float *bbase = nullptr, *bown = nullptr, *btot = nullptr;  // [I] each (item bias), or NULL
  int64_t n = 0, nb = 0;
  bool pending = false;
  const float *q_at = nullptr, *bias_at = nullptr;  // the tables the bases were cut from
  // hot tier (bpr_comm_hot_tier / bpr_hot_sync): the hot block's own exchange, every launch
  float *hb = nullptr, *htot = nullptr;  // [H*d] each
  int64_t nh = 0;
  bool hot_pending = false;
  hipEvent_t ev_hot_cut = nullptr, ev_hot_done = nullptr, ev_gather = nullptr;
};

// Leave the hot tier with the item table whole: fold the exchange in flight, fold what the launches
// since left in the block (this rank's own deltas: nobody is asked for theirs), switch the tier off.
static int comm_hot_close(bpr_ctx* c, Comm* m) {
  if (m->hb == nullptr || !c->hot_tier) return BPR_OK;
  if (m->hot_pending) BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, m->ev_hot_done, 0));
  if (int rc = bpr_hot_exchange(c, m->hb, m->htot, m->hot_pending ? 1 : 0, 1, m->base)) return rc;
  m->hot_pending = false;
  return bpr_hot_tier_end(c);
}

// tables_alive: the caller's tables are known to exist (bpr_comm_destroy / a re-init: explicit calls) — only then is
// the hot tier closed by folding the exchange in flight and this rank's uncut deltas into Q.  From bpr_ctx_destroy
// (a garbage collector may have freed the caller's tensors already) nothing is written: a hot tier still
// open at that point is the caller's to close first (bpr_comm_destroy, or ItemSync.close in the Python layer).
void comm_free(bpr_ctx* c, bool tables_alive) {
  Comm* m = static_cast<Comm*>(c->comm);
  if (m == nullptr) return;
  if (m->stream) hipStreamSynchronize(m->stream);  // library-owned: always safe
  if (tables_alive) {
    // the launches must not go on leaving hot-row deltas in a block nobody folds
    if (c->P != nullptr && c->Q != nullptr) (void)comm_hot_close(c, m);
    hipStreamSynchronize(c->stream);
  }
  Rccl* l = rccl();
  if (m->comm && l) l->CommDestroy(m->comm);
  hipFree(m->base); hipFree(m->own); hipFree(m->tot);
  hipFree(m->bbase); hipFree(m->bown); hipFree(m->btot);
  hipFree(m->hb); hipFree(m->htot);
  if (m->ev_cut) hipEventDestroy(m->ev_cut);
  if (m->ev_done) hipEventDestroy(m->ev_done);
  if (m->ev_hot_cut) hipEventDestroy(m->ev_hot_cut);
  if (m->ev_hot_done) hipEventDestroy(m->ev_hot_done);
  if (m->ev_gather) hipEventDestroy(m->ev_gather);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
  c->comm = nullptr;
}

// (re)cut the bases from the tables as they are now
static int comm_rebase(bpr_ctx* c, Comm* m) {
  const int64_t n = c->I * c->d, nb = c->bias != nullptr ? c->I : 0;
  if (m->n != n || m->nb != nb) {
    hipFree(m->base); hipFree(m->own); hipFree(m->tot);
    hipFree(m->bbase); hipFree(m->bown); hipFree(m->btot);
    m->base = m->own = m->tot = m->bbase = m->bown = m->btot = nullptr;
    BPR_HIP_CHECK(hipMalloc(&m->base, sizeof(float) * n));
    BPR_HIP_CHECK(hipMalloc(&m->own, sizeof(float) * n));
    BPR_HIP_CHECK(hipMalloc(&m->tot, sizeof(float) * n));
    if (nb > 0) {
      BPR_HIP_CHECK(hipMalloc(&m->bbase, sizeof(float) * nb));
      BPR_HIP_CHECK(hipMalloc(&m->bown, sizeof(float) * nb));
      BPR_HIP_CHECK(hipMalloc(&m->btot, sizeof(float) * nb));
    }
    m->n = n;
    m->nb = nb;
  }
  BPR_HIP_CHECK(hipMemcpyAsync(m->base, c->Q, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  if (nb > 0)
    BPR_HIP_CHECK(hipMemcpyAsync(m->bbase, c->bias, sizeof(float) * nb, hipMemcpyDeviceToDevice,
                                 c->stream));
  m->pending = false;
  m->q_at = c->Q;
  m->bias_at = c->bias;
  if (m->hb != nullptr && c->hot_tier) {  // the hot base follows the table too
    if (m->hot_pending) BPR_HIP_CHECK(hipStreamSynchronize(m->stream));
    m->hot_pending = false;
    if (int rc = bpr_hot_tier_begin(c, m->hb)) return rc;
  }
  return BPR_OK;
}

// the all-reduce of the deltas cut on the ctx stream, on the comm stream
static int comm_launch_allreduce(bpr_ctx* c, Comm* m) {
  Rccl* l = rccl();
  BPR_HIP_CHECK(hipEventRecord(m->ev_cut, c->stream));
  BPR_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_cut, 0));
  BPR_NCCL_CHECK(l->AllReduce(m->tot, m->tot, (size_t)m->n, ncclFloat32, ncclSum, m->comm, m->stream));
  if (m->nb > 0)
    BPR_NCCL_CHECK(l->AllReduce(m->btot, m->btot, (size_t)m->nb, ncclFloat32, ncclSum, m->comm,
                                m->stream));
  BPR_HIP_CHECK(hipEventRecord(m->ev_done, m->stream));
  m->pending = true;
  return BPR_OK;
}

int comm_item_sync(bpr_ctx* c, bool finish_only) {
  Comm* m = static_cast<Comm*>(c->comm);
  if (m == nullptr) {
    set_error("bpr_item_sync: no communicator (bpr_comm_init first)");
    return BPR_ERR_INVALID;
  }
  if (m->n != c->I * c->d || m->nb != (c->bias != nullptr ? c->I : 0) || m->q_at != c->Q ||
      m->bias_at != c->bias) {
    // tables (re)bound since — another shape or another buffer: start from them.  (A table
    // overwritten IN PLACE, e.g. a checkpoint restored into the same storage, is invisible here:
    // bpr_item_sync_rebase.)
    if (m->pending) BPR_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (int rc = comm_rebase(c, m)) return rc;
  }
  c->keys_cut = false;  // the item table moves
  c->bias_w_valid = false;  // ... and the item_bias with it
  const float scale = 1.0f;
  if (m->pending) {
    // fold the other ranks' contribution of the reconciliation in flight ...
    BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, m->ev_done, 0));
    if (!finish_only) {  // ... and cut the next delta in the same pass (nothing trained in between)
      if (int rc = bpr_item_fold_delta(c->Q, m->base, m->own, m->tot, scale, m->n, c->stream)) return rc;
      if (m->nb > 0)
        if (int rc = bpr_item_fold_delta(c->bias, m->bbase, m->bown, m->btot, scale, m->nb, c->stream))
          return rc;
    } else {
      if (int rc = bpr_item_fold(c->Q, m->base, m->own, m->tot, scale, 0, m->n, c->stream)) return rc;
      if (m->nb > 0)
        if (int rc = bpr_item_fold(c->bias, m->bbase, m->bown, m->btot, scale, 0, m->nb, c->stream))
          return rc;
      m->pending = false;
      return BPR_OK;
    }
  } else {
    if (finish_only) return BPR_OK;
    if (int rc = bpr_item_delta(c->Q, m->base, m->own, m->tot, m->n, c->stream)) return rc;
    if (m->nb > 0)
      if (int rc = bpr_item_delta(c->bias, m->bbase, m->bown, m->btot, m->nb, c->stream)) return rc;
  }
  BPR_HIP_CHECK(hipGetLastError());
  return comm_launch_allreduce(c, m);
}

// hot tier: fold the exchange in flight, cut the launch's hot deltas (cut) and all-reduce them
static int comm_hot_sync(bpr_ctx* c, bool cut) {
  Comm* m = static_cast<Comm*>(c->comm);
  if (m == nullptr || m->hb == nullptr || !c->hot_tier) {
    set_error("bpr_hot_sync: no hot tier (bpr_comm_init, then bpr_comm_hot_tier)");
    return BPR_ERR_INVALID;
  }
  if (!cut && !m->hot_pending) return BPR_OK;
  if (m->hot_pending) BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, m->ev_hot_done, 0));
  if (int rc = bpr_hot_exchange(c, m->hb, m->htot, m->hot_pending ? 1 : 0, cut ? 1 : 0, m->base)) return rc;
  m->hot_pending = false;
  if (!cut) return BPR_OK;
  Rccl* l = rccl();
  BPR_HIP_CHECK(hipEventRecord(m->ev_hot_cut, c->stream));
  BPR_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_hot_cut, 0));
  BPR_NCCL_CHECK(l->AllReduce(m->htot, m->htot, (size_t)m->nh, ncclFloat32, ncclSum, m->comm, m->stream));
  BPR_HIP_CHECK(hipEventRecord(m->ev_hot_done, m->stream));
  m->hot_pending = true;
  return BPR_OK;
}

// all-gather of this rank's slice of the back snapshot (bpr_adaptive_refresh with a communicator).
// Every collective of a communicator goes to ITS stream, in program order — an all-gather on the
// ctx stream beside an all-reduce still in flight on the communicator's would be two concurrent
// operations on one communicator — with events ordering it against the ctx stream on both sides.
int comm_gather_snapshot(bpr_ctx* c, int32_t* order_back, float* sigma_back, int per) {
  Comm* m = static_cast<Comm*>(c->comm);
  Rccl* l = rccl();
  const size_t cnt = (size_t)per * (size_t)c->I;
  BPR_HIP_CHECK(hipEventRecord(m->ev_gather, c->stream));
  BPR_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_gather, 0));
  BPR_NCCL_CHECK(l->AllGather(order_back + (size_t)m->rank * cnt, order_back, cnt, ncclInt32, m->comm,
                              m->stream));
  BPR_NCCL_CHECK(l->AllGather(sigma_back + (size_t)m->rank * per, sigma_back, (size_t)per, ncclFloat32,
                              m->comm, m->stream));
  BPR_HIP_CHECK(hipEventRecord(m->ev_gather, m->stream));
  BPR_HIP_CHECK(hipStreamWaitEvent(c->stream, m->ev_gather, 0));
  return BPR_OK;
}

int comm_world(const bpr_ctx* c) {
  const Comm* m = static_cast<const Comm*>(c->comm);
  return m != nullptr ? m->world : 1;
}
int comm_rank(const bpr_ctx* c) {
  const Comm* m = static_cast<const Comm*>(c->comm);
  return m != nullptr ? m->rank : 0;
}

}  // namespace bpr

using namespace bpr;

extern "C" {

int bpr_comm_unique_id(void* id_host) {
  if (id_host == nullptr) return fail(BPR_ERR_INVALID, "bpr_comm_unique_id: NULL argument");
  Rccl* l = rccl();
  if (l == nullptr) return fail(BPR_ERR_UNSUPPORTED, "bpr_comm_unique_id: librccl.so not found");
  ncclUniqueId id;
  BPR_NCCL_CHECK(l->GetUniqueId(&id));
  static_assert(sizeof(ncclUniqueId) == BPR_COMM_ID_BYTES, "include/bprcore.h: BPR_COMM_ID_BYTES");
  memcpy(id_host, &id, sizeof(id));
  return BPR_OK;
}

int bpr_comm_init(bpr_ctx* c, const void* id_host, int32_t rank, int32_t world) {
  if (int rc = check_bound(c, "bpr_comm_init")) return rc;
  if (id_host == nullptr || world < 1 || rank < 0 || rank >= world)
    return fail(BPR_ERR_INVALID, "bpr_comm_init: bad argument");
  Rccl* l = rccl();
  if (l == nullptr) return fail(BPR_ERR_UNSUPPORTED, "bpr_comm_init: librccl.so not found");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  comm_free(c, true);
  Comm* m = new (std::nothrow) Comm();
  if (m == nullptr) return fail(BPR_ERR_NOMEM, "bpr_comm_init: out of host memory");
  c->comm = m;
  m->rank = rank;
  m->world = world;
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  BPR_NCCL_CHECK(l->CommInitRank(&m->comm, world, id, rank));
  BPR_HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  BPR_HIP_CHECK(hipEventCreateWithFlags(&m->ev_cut, hipEventDisableTiming));
  BPR_HIP_CHECK(hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming));
  BPR_HIP_CHECK(hipEventCreateWithFlags(&m->ev_hot_cut, hipEventDisableTiming));
  BPR_HIP_CHECK(hipEventCreateWithFlags(&m->ev_hot_done, hipEventDisableTiming));
  BPR_HIP_CHECK(hipEventCreateWithFlags(&m->ev_gather, hipEventDisableTiming));
  return comm_rebase(c, m);
}

int bpr_item_sync_rebase(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_item_sync_rebase")) return rc;
  Comm* m = static_cast<Comm*>(c->comm);
  if (m == nullptr) return fail(BPR_ERR_INVALID, "bpr_item_sync_rebase: no communicator");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (m->pending || m->hot_pending) BPR_HIP_CHECK(hipStreamSynchronize(m->stream));
  return comm_rebase(c, m);
}

int bpr_comm_hot_tier(bpr_ctx* c, const int32_t* items_host, int32_t H, const uint32_t* counts_host) {
  if (int rc = check_bound(c, "bpr_comm_hot_tier")) return rc;
  Comm* m = static_cast<Comm*>(c->comm);
  if (m == nullptr) return fail(BPR_ERR_INVALID, "bpr_comm_hot_tier: no communicator (bpr_comm_init first)");
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->hot_tier && m->hb != nullptr) {
    // re-tiering mid-run (every rank calls this): one last exchange of what the launches since left in
    // the block, folded at once, so that the replicas leave the old tier agreeing on its rows
    if (int rc = comm_hot_sync(c, true)) return rc;
    if (int rc = comm_hot_sync(c, false)) return rc;
    if (int rc = bpr_hot_tier_end(c)) return rc;
    BPR_HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  m->hot_pending = false;
  if (c->hot_tier) bpr_hot_tier_end(c);
  hipFree(m->hb); hipFree(m->htot);
  m->hb = m->htot = nullptr;
  m->nh = 0;
  if (int rc = bpr_set_hot_items(c, items_host, H, counts_host)) return rc;
  if (H <= 0) return BPR_OK;
  m->nh = (int64_t)c->hot_H * c->d;
  BPR_HIP_CHECK(hipMalloc(&m->hb, sizeof(float) * m->nh));
  BPR_HIP_CHECK(hipMalloc(&m->htot, sizeof(float) * m->nh));
  BPR_HIP_CHECK(hipMemsetAsync(m->htot, 0, sizeof(float) * m->nh, c->stream));
  return bpr_hot_tier_begin(c, m->hb);
}

int bpr_hot_sync(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_hot_sync")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  return comm_hot_sync(c, true);
}

int bpr_comm_destroy(bpr_ctx* c) {
  if (c == nullptr) return BPR_OK;
  hipSetDevice(c->device);
  comm_free(c, true);
  return BPR_OK;
}

int bpr_item_sync(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_item_sync")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  if (c->vs_active)  // batched STREAM: the reconciliation must see the item rows as of "now"
    if (int rc = vs_flush(c, false, true)) return rc;
  return comm_item_sync(c, false);
}

int bpr_item_sync_finish(bpr_ctx* c) {
  if (int rc = check_bound(c, "bpr_item_sync_finish")) return rc;
  BPR_HIP_CHECK(hipSetDevice(c->device));
  Comm* m = static_cast<Comm*>(c->comm);
  if (m != nullptr && m->hb != nullptr && c->hot_tier)
    if (int rc = comm_hot_sync(c, false)) return rc;
  return comm_item_sync(c, true);
}

}  // extern "C"
