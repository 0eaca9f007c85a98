/* bprio.h — native loader of the reference's JSON-lines interaction files (host side, no GPU).
 *
 * On-disk formats (reference: bin/datasets/format-repro.sh:56-81, bin/datasets/jsonl.sh:77-83,
 * bin/datasets/split.py:110-115), one JSON object per line:
 *     {"user": u, "item": i}                  full-train-with-fold-in.jsonl       (one per pair)
 *     {"user": u, "seen_items": [i, ...]}     ...-user-seen-items.jsonl           (one per user)
 *     {"user": u, "item": [i, ...]}           test-grouped.jsonl                  (one per user)
 * The reference parses them with json.loads per line into a scipy dok matrix
 * (experiments/bpr/dataset.py:183-190; datasets/jsonl.py) — minutes on MSD.  Here the file is mapped,
 * cut at line boundaries into one piece per thread, scanned once, and the pairs go straight into
 * the seen-items CSR that bpr_bind_seen_csr (include/bprcore.h) consumes.
 *
 * Plain C ABI; every output buffer is malloc'ed by the library and released with bprio_free.
 * Functions return 0 on success, a negative code otherwise (message: bprio_last_error()).
 * Keys may come in any order, unknown keys with scalar / string / flat-array values are skipped,
 * ids must be non-negative integers < 2^31. */
#ifndef BPRIO_H
#define BPRIO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { BPRIO_OK = 0, BPRIO_ERR_IO = -1, BPRIO_ERR_PARSE = -2, BPRIO_ERR_INVALID = -3 };

int bprio_version(void);
const char* bprio_last_error(void);
void bprio_free(void* p);

/* {"user": u, <key>: i} per line -> users[n], values[n] in file order.  threads <= 0: all cores. */
int bprio_read_pairs(const char* path, const char* key, int threads, int32_t** users_out,
                     int32_t** values_out, int64_t* n_out);

/* {"user": u, <key>: [v, ...]} per line -> users[rows], offsets[rows + 1], values[offsets[rows]]
 * in file order (a scalar value counts as a one-element list). */
int bprio_read_ragged(const char* path, const char* key, int threads, int32_t** users_out,
                      int64_t** offsets_out, int32_t** values_out, int64_t* rows_out,
                      int64_t* n_values_out);

/* (user, item) pairs -> CSR over users [0, num_users): indptr[num_users + 1] (caller's buffer),
 * indices (malloc'ed) sorted ascending per row, duplicates removed (the reference's dok matrix
 * keeps one entry per pair), item 0 — the padding row — dropped when drop_item0 != 0.  Expanding
 * the CSR of the training file row by row gives the de-duplicated training pairs in (user, item)
 * order.  Fails if an id is out of range. */
int bprio_build_csr(const int32_t* users, const int32_t* items, int64_t n, int64_t num_users,
                    int64_t num_items, int drop_item0, int threads, int64_t* indptr_out,
                    int32_t** indices_out, int64_t* nnz_out);

#ifdef __cplusplus
}
#endif
#endif
