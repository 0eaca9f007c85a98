// bpr_hotlds.hip — the LDS tier of k_stream's hot block (r6): the instantiations k_stream<..., LDSHOT = true>
// (bpr_kernels.h: one workgroup of up to 1,024 threads per CU, a private fp32 delta block for the most popular
// item rows in LDS, flushed into the global delta block at exit) and their launcher, in a translation unit of
// their own so that they compile beside bprcore.hip.  Reference: the rows concerned are the ones the adaptive
// sampler of revisit_bpr/modules/neg_samplers.py:84-121 concentrates on once the model has moved.
#include <hip/hip_runtime.h>

#include <set>

#include "bpr_host.h"
#include "bpr_stream.h"

namespace bpr {

// one workgroup may declare all of a CU's LDS; anything past 64 KiB of dynamic LDS must be asked for per function
template <typename K>
static int allow_lds(K kernel, size_t shmem, size_t room) {
  static std::set<const void*> done;  // (launches are issued from one host thread per ctx; the set is per instantiation)
  if (shmem <= 64 * 1024) return BPR_OK;
  const void* key = reinterpret_cast<const void*>(kernel);
  if (done.count(key)) return BPR_OK;
  BPR_HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room));
  done.insert(key);
  return BPR_OK;
}

// sampler: NEG_GIVEN (no seen structure) | NEG_UNIFORM | NEG_ADAPTIVE (per-group LDS bitmaps: a.bm_words);
// d == G * E (the FULL instantiations: every BASELINE shape)
// (r6 also measured every user's seen bitmap in HBM instead of the per-group LDS bitmaps — twice the LDS rows, and
// 612 against 832 M triples/s: the walk's lookups are LDS reads for a reason.  profiles/r06_hotlds.md.  Removed.)
int launch_stream_lds(bpr_ctx* c, const StreamArgs& a, int sampler, int seen, unsigned grid, unsigned block, size_t shmem,
                      hipEvent_t stop) {
  return dispatch_ge(c->G, c->E, [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int G = T::G, E = T::E;
    auto go = [&](auto kernel) -> int {
      if (int rc = allow_lds(kernel, shmem, lds_tier_room(G * E))) return rc;
      hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, c->stream, nullptr, stop, 0, a);
      BPR_HIP_CHECK(hipGetLastError());
      return BPR_OK;
    };
    if (sampler == NEG_GIVEN) return go(k_stream<G, E, NEG_GIVEN, SEEN_CSR, true, false, true>);
    if (seen == SEEN_LIST) {
      if (sampler == NEG_UNIFORM) return go(k_stream<G, E, NEG_UNIFORM, SEEN_LIST, true, false, true>);
      return go(k_stream<G, E, NEG_ADAPTIVE, SEEN_LIST, true, false, true>);
    }
    if (sampler == NEG_UNIFORM) return go(k_stream<G, E, NEG_UNIFORM, SEEN_BITMAP, true, false, true>);
    return go(k_stream<G, E, NEG_ADAPTIVE, SEEN_BITMAP, true, false, true>);
  });
}

}  // namespace bpr
