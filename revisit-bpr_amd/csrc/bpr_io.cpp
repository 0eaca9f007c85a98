// bpr_io.cpp — native loader of the reference's JSON-lines interaction files (include/bprio.h).
// Host-side IO only: g++ -O2 -pthread, no HIP.  Not a port of anything in the reference, whose
// loader is json.loads per line into a scipy dok matrix (experiments/bpr/dataset.py:183-190).
#include "../../include/bprio.h"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

struct Mapped {
  const char* p = nullptr;
  size_t n = 0;
  int fd = -1;
  ~Mapped() {
    if (p != nullptr && n > 0) munmap(const_cast<char*>(p), n);
    if (fd >= 0) close(fd);
  }
  int open_file(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(BPRIO_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) return fail(BPRIO_ERR_IO, std::string("cannot stat ") + path);
    n = (size_t)st.st_size;
    if (n == 0) return BPRIO_OK;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      n = 0;
      return fail(BPRIO_ERR_IO, std::string("cannot map ") + path);
    }
    p = static_cast<const char*>(m);
    madvise(m, n, MADV_SEQUENTIAL);
    return BPRIO_OK;
  }
};

int n_threads(int want, size_t bytes) {
  int t = want > 0 ? want : (int)std::thread::hardware_concurrency();
  if (t < 1) t = 1;
  const size_t by_size = bytes / (1u << 20) + 1;  // at least 1 MiB per thread
  if ((size_t)t > by_size) t = (int)by_size;
  return t;
}

// piece k of the file: [cut[k], cut[k+1]) with every cut just after a '\n'
std::vector<size_t> line_cuts(const char* p, size_t n, int parts) {
  std::vector<size_t> cut(parts + 1, n);
  cut[0] = 0;
  for (int k = 1; k < parts; ++k) {
    size_t at = n / parts * k;
    if (at < cut[k - 1]) at = cut[k - 1];
    const void* nl = at < n ? memchr(p + at, '\n', n - at) : nullptr;
    cut[k] = nl ? (size_t)(static_cast<const char*>(nl) - p) + 1 : n;
  }
  return cut;
}

struct Piece {
  std::vector<int32_t> users, values;
  std::vector<int64_t> lens;  // ragged: values per line
  std::string err;
};

inline const char* skip_ws(const char* c, const char* e) {
  while (c < e && (*c == ' ' || *c == '\t' || *c == '\r')) ++c;
  return c;
}
inline const char* parse_int(const char* c, const char* e, int64_t* out) {
  c = skip_ws(c, e);
  if (c >= e || *c < '0' || *c > '9') return nullptr;
  int64_t v = 0;
  while (c < e && *c >= '0' && *c <= '9') {
    v = v * 10 + (*c - '0');
    if (v > 0x7fffffffLL) return nullptr;
    ++c;
  }
  *out = v;
  return c;
}
// skip one JSON value that is not an object: number / string / literal / flat array
inline const char* skip_value(const char* c, const char* e) {
  c = skip_ws(c, e);
  if (c >= e) return nullptr;
  if (*c == '"') {
    for (++c; c < e && *c != '"'; ++c)
      if (*c == '\\') ++c;
    return c < e ? c + 1 : nullptr;
  }
  if (*c == '[') {
    for (++c; c < e && *c != ']'; ++c)
      if (*c == '"') {
        for (++c; c < e && *c != '"'; ++c)
          if (*c == '\\') ++c;
        if (c >= e) return nullptr;
      }
    return c < e ? c + 1 : nullptr;
  }
  while (c < e && *c != ',' && *c != '}') ++c;
  return c;
}

// one line: {"user": u, "<key>": v | [v, ...], ...}
bool parse_line(const char* c, const char* e, const char* key, size_t klen, bool ragged, Piece* out) {
  c = skip_ws(c, e);
  if (c >= e) return true;  // blank line
  if (*c != '{') return false;
  ++c;
  int64_t user = -1;
  bool have_val = false;
  const size_t v0 = out->values.size();
  while (true) {
    c = skip_ws(c, e);
    if (c < e && *c == '}') break;
    if (c >= e || *c != '"') return false;
    const char* k0 = ++c;
    while (c < e && *c != '"') ++c;
    if (c >= e) return false;
    const size_t kl = (size_t)(c - k0);
    ++c;
    c = skip_ws(c, e);
    if (c >= e || *c != ':') return false;
    ++c;
    if (kl == 4 && memcmp(k0, "user", 4) == 0) {
      c = parse_int(c, e, &user);
      if (c == nullptr) return false;
    } else if (kl == klen && memcmp(k0, key, klen) == 0) {
      c = skip_ws(c, e);
      if (c < e && *c == '[') {
        if (!ragged) return false;
        ++c;
        c = skip_ws(c, e);
        if (c < e && *c == ']') {
          ++c;
        } else {
          while (true) {
            int64_t v;
            c = parse_int(c, e, &v);
            if (c == nullptr) return false;
            out->values.push_back((int32_t)v);
            c = skip_ws(c, e);
            if (c < e && *c == ',') { ++c; continue; }
            if (c < e && *c == ']') { ++c; break; }
            return false;
          }
        }
      } else {
        int64_t v;
        c = parse_int(c, e, &v);
        if (c == nullptr) return false;
        out->values.push_back((int32_t)v);
      }
      have_val = true;
    } else {
      c = skip_value(c, e);
      if (c == nullptr) return false;
    }
    c = skip_ws(c, e);
    if (c < e && *c == ',') { ++c; continue; }
    if (c < e && *c == '}') break;
    return false;
  }
  if (user < 0 || !have_val) return false;
  out->users.push_back((int32_t)user);
  if (ragged) out->lens.push_back((int64_t)(out->values.size() - v0));
  else if (out->values.size() - v0 != 1) return false;
  return true;
}

int parse_file(const char* path, const char* key, int threads, bool ragged, std::vector<Piece>* pieces) {
  if (path == nullptr || key == nullptr) return fail(BPRIO_ERR_INVALID, "path / key is NULL");
  Mapped f;
  if (int rc = f.open_file(path)) return rc;
  const int T = n_threads(threads, f.n);
  const std::vector<size_t> cut = line_cuts(f.p, f.n, T);
  pieces->assign(T, Piece());
  const size_t klen = strlen(key);
  auto work = [&](int k) {
    Piece& pc = (*pieces)[k];
    const char* c = f.p + cut[k];
    const char* end = f.p + cut[k + 1];
    // a line is at least ~20 bytes: one allocation instead of repeated growth (page faults of
    // several threads serialise on the process's address-space lock)
    pc.users.reserve((size_t)(end - c) / 20 + 16);
    if (!ragged) pc.values.reserve((size_t)(end - c) / 20 + 16);
    else pc.lens.reserve((size_t)(end - c) / 20 + 16);
    while (c < end) {
      const char* nl = static_cast<const char*>(memchr(c, '\n', (size_t)(end - c)));
      const char* le = nl ? nl : end;
      if (!parse_line(c, le, key, klen, ragged, &pc)) {
        pc.err = std::string(path) + ": cannot parse the line at byte " + std::to_string(c - f.p) +
                 ": " + std::string(c, std::min<size_t>((size_t)(le - c), 80));
        return;
      }
      c = nl ? nl + 1 : end;
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < T; ++k) th.emplace_back(work, k);
  work(0);
  for (auto& t : th) t.join();
  for (const Piece& pc : *pieces)
    if (!pc.err.empty()) return fail(BPRIO_ERR_PARSE, pc.err);
  return BPRIO_OK;
}

template <typename T>
T* alloc(int64_t n) {
  return static_cast<T*>(malloc(sizeof(T) * (size_t)(n > 0 ? n : 1)));
}

}  // namespace

extern "C" {

int bprio_version(void) { return 100; }
const char* bprio_last_error(void) { return g_err.c_str(); }
void bprio_free(void* p) { free(p); }

int bprio_read_pairs(const char* path, const char* key, int threads, int32_t** users_out,
                     int32_t** values_out, int64_t* n_out) {
  if (!users_out || !values_out || !n_out) return fail(BPRIO_ERR_INVALID, "output pointer is NULL");
  std::vector<Piece> pieces;
  if (int rc = parse_file(path, key, threads, false, &pieces)) return rc;
  int64_t n = 0;
  for (const Piece& p : pieces) n += (int64_t)p.users.size();
  int32_t* u = alloc<int32_t>(n);
  int32_t* v = alloc<int32_t>(n);
  if (!u || !v) { free(u); free(v); return fail(BPRIO_ERR_IO, "out of memory"); }
  int64_t at = 0;
  for (const Piece& p : pieces) {
    if (!p.users.empty()) {
      memcpy(u + at, p.users.data(), sizeof(int32_t) * p.users.size());
      memcpy(v + at, p.values.data(), sizeof(int32_t) * p.values.size());
    }
    at += (int64_t)p.users.size();
  }
  *users_out = u; *values_out = v; *n_out = n;
  return BPRIO_OK;
}

int bprio_read_ragged(const char* path, const char* key, int threads, int32_t** users_out,
                      int64_t** offsets_out, int32_t** values_out, int64_t* rows_out,
                      int64_t* n_values_out) {
  if (!users_out || !offsets_out || !values_out || !rows_out || !n_values_out)
    return fail(BPRIO_ERR_INVALID, "output pointer is NULL");
  std::vector<Piece> pieces;
  if (int rc = parse_file(path, key, threads, true, &pieces)) return rc;
  int64_t rows = 0, nv = 0;
  for (const Piece& p : pieces) { rows += (int64_t)p.users.size(); nv += (int64_t)p.values.size(); }
  int32_t* u = alloc<int32_t>(rows);
  int64_t* off = alloc<int64_t>(rows + 1);
  int32_t* v = alloc<int32_t>(nv);
  if (!u || !off || !v) { free(u); free(off); free(v); return fail(BPRIO_ERR_IO, "out of memory"); }
  int64_t r = 0, at = 0;
  off[0] = 0;
  for (const Piece& p : pieces) {
    for (size_t k = 0; k < p.users.size(); ++k) {
      u[r] = p.users[k];
      off[r + 1] = off[r] + p.lens[k];
      ++r;
    }
    if (!p.values.empty()) memcpy(v + at, p.values.data(), sizeof(int32_t) * p.values.size());
    at += (int64_t)p.values.size();
  }
  *users_out = u; *offsets_out = off; *values_out = v; *rows_out = rows; *n_values_out = nv;
  return BPRIO_OK;
}

int bprio_build_csr(const int32_t* users, const int32_t* items, int64_t n, int64_t num_users,
                    int64_t num_items, int drop_item0, int threads, int64_t* indptr_out,
                    int32_t** indices_out, int64_t* nnz_out) {
  if (n < 0 || num_users < 1 || num_items < 1 || !indptr_out || !indices_out || !nnz_out ||
      (n > 0 && (!users || !items)))
    return fail(BPRIO_ERR_INVALID, "bprio_build_csr: bad argument");
  std::vector<int64_t> start((size_t)num_users + 1, 0);
  for (int64_t k = 0; k < n; ++k) {
    const int32_t u = users[k], i = items[k];
    if (u < 0 || u >= num_users || i < 0 || i >= num_items)
      return fail(BPRIO_ERR_INVALID, "ids out of range for the given num_users / num_items");
    if (drop_item0 && i == 0) continue;
    start[(size_t)u + 1] += 1;
  }
  for (int64_t u = 0; u < num_users; ++u) start[(size_t)u + 1] += start[(size_t)u];
  const int64_t total = start[(size_t)num_users];
  std::vector<int32_t> tmp((size_t)(total > 0 ? total : 1));
  {
    std::vector<int64_t> at(start.begin(), start.end() - 1);
    for (int64_t k = 0; k < n; ++k) {
      if (drop_item0 && items[k] == 0) continue;
      tmp[(size_t)at[(size_t)users[k]]++] = items[k];
    }
  }
  // sort + unique every row (rows are independent: a slice of users per thread)
  std::vector<int64_t> cnt((size_t)num_users, 0);
  int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (T < 1) T = 1;
  if ((int64_t)T > num_users) T = (int)num_users;
  auto work = [&](int k) {
    const int64_t u0 = num_users * k / T, u1 = num_users * (k + 1) / T;
    for (int64_t u = u0; u < u1; ++u) {
      int32_t* b = tmp.data() + start[(size_t)u];
      int32_t* e = tmp.data() + start[(size_t)u + 1];
      std::sort(b, e);
      cnt[(size_t)u] = (int64_t)(std::unique(b, e) - b);
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < T; ++k) th.emplace_back(work, k);
  work(0);
  for (auto& t : th) t.join();
  indptr_out[0] = 0;
  for (int64_t u = 0; u < num_users; ++u) indptr_out[u + 1] = indptr_out[u] + cnt[(size_t)u];
  const int64_t nnz = indptr_out[num_users];
  int32_t* idx = alloc<int32_t>(nnz);
  if (!idx) return fail(BPRIO_ERR_IO, "out of memory");
  for (int64_t u = 0; u < num_users; ++u)
    if (cnt[(size_t)u] > 0)
      memcpy(idx + indptr_out[u], tmp.data() + start[(size_t)u], sizeof(int32_t) * (size_t)cnt[(size_t)u]);
  *indices_out = idx;
  *nnz_out = nnz;
  return BPRIO_OK;
}

}  // extern "C"
