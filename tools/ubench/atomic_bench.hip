// atomic_bench.hip — what bounds the item-row updates of the STREAM kernel (gfx950)?
// Not part of the product.  One group of 32 lanes per triple, dword-per-lane rows (d = 128), the
// user row stored plainly, the two item rows updated by the variant under test:
//
//   agent / workgroup / wavefront / system   fp32 atomic add at that scope on the item table
//   xcd-replica-wg      workgroup-scope adds into a per-XCD replica of the item table (the block's
//                       XCC id picks the replica): do L2-local atomics retire faster?
//   neg-only            the positive row is NOT updated (what an item-major second pass would
//                       leave in the hot kernel), negative row agent-scope
//   store               both item rows read-modify-written with plain stores (no atomics: lost
//                       updates, the bandwidth floor)
//   none                no item update at all
//   u64-packed          two fixed-point int32 fields per 64-bit integer atomic (global_atomic_add_x2):
//                       half as many atomic operations for the same 128 values per row
//   pos-aggregated      what aggregating the positives of a 16 k-triple window per item would leave:
//                       24 % of the triples update their positive row with atomics, the others
//                       park a 512-B contribution vector (16-B stores) that is read back once
//   u32 / f64           32-bit integer adds (one per value), double adds (one per PAIR of values'
//                       worth of bytes) — is the unit's rate per operation or per byte?
//
//   hipcc -O3 --offload-arch=gfx950 atomic_bench.hip -o atomic_bench && ./atomic_bench [n] [U] [I] [hot]
//   llvm-objdump -d --offloading atomic_bench | grep global_atomic      (scope bits of each variant)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int D = 128, G = 32;

__device__ __forceinline__ float gsum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

enum { S_AGENT = 0, S_WG = 1, S_WAVE = 2, S_SYS = 3, S_XCD_WG = 4, S_NEG_ONLY = 5, S_STORE = 6, S_NONE = 7,
       S_AGENT_RET = 8, S_U64 = 9, S_U32 = 10, S_F64 = 11, S_AGG = 12 };

template <int V>
__device__ __forceinline__ void upd(float* p, float v) {
  if constexpr (V == S_AGENT || V == S_NEG_ONLY)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (V == S_WG || V == S_XCD_WG)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if constexpr (V == S_WAVE)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  else if constexpr (V == S_SYS)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int V>
__global__ __launch_bounds__(256) void k(float* P, float* Q, float* Qrep, int64_t rep_stride, const int* us,
                                         const int* is, const int* js, int64_t n, float lr, float* out) {
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), gw = lane / G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  float* Qw = Q;
  if constexpr (V == S_XCD_WG) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Qw = Qrep + (int64_t)(xcc & 7u) * rep_stride;
  }
  for (int64_t base = wave * 2; base < n; base += nw * 2) {
    const int64_t t = base + gw;
    const bool act = t < n;
    const int64_t tt = act ? t : n - 1;
    const int u = us[tt], i = is[tt], j = js[tt];
    float* pr = P + (int64_t)u * D;
    const float* ir = Q + (int64_t)i * D;
    const float* jr = Q + (int64_t)j * D;
    float p[4], qi[4], qj[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[e] = pr[e * G + gl]; qi[e] = ir[e * G + gl]; qj[e] = jr[e * G + gl]; }
    float x = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) x += p[e] * (qi[e] - qj[e]);
    x = gsum(x);
    const float w = 1.f / (1.f + __expf(x));
    acc += x;
    if (act) {
      float* iw = Qw + (int64_t)i * D;
      float* jw = Qw + (int64_t)j * D;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pr[e * G + gl] = p[e] + lr * (w * (qi[e] - qj[e]) - 0.01f * p[e]);
        const float di = lr * (w * p[e] - 0.01f * qi[e]), dj = lr * (-w * p[e] - 0.01f * qj[e]);
        if constexpr (V == S_STORE) {
          iw[e * G + gl] = qi[e] + di;
          jw[e * G + gl] = qj[e] + dj;
        } else if constexpr (V == S_AGENT_RET) {
          acc += __hip_atomic_fetch_add(iw + e * G + gl, di, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc += __hip_atomic_fetch_add(jw + e * G + gl, dj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if constexpr (V == S_U64 || V == S_F64) {
          // 64 qwords per row: two instructions of 32 lanes x 8 B (256 contiguous bytes) per row
          if (e < 2) {
            if constexpr (V == S_U64) {
              unsigned long long* i64 = reinterpret_cast<unsigned long long*>(iw) + e * G + gl;
              unsigned long long* j64 = reinterpret_cast<unsigned long long*>(jw) + e * G + gl;
              const long long vi = ((long long)(int)(di * 1e9f) << 32) + (long long)(int)(dj * 1e9f);
              __hip_atomic_fetch_add(i64, (unsigned long long)vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_fetch_add(j64, (unsigned long long)(vi + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              double* i64 = reinterpret_cast<double*>(iw) + e * G + gl;
              double* j64 = reinterpret_cast<double*>(jw) + e * G + gl;
              __hip_atomic_fetch_add(i64, (double)di, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_fetch_add(j64, (double)dj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        } else if constexpr (V == S_U32) {
          __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(iw) + e * G + gl, (unsigned)(int)(di * 1e9f),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(jw) + e * G + gl, (unsigned)(int)(dj * 1e9f),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if constexpr (V == S_AGG) {
          __hip_atomic_fetch_add(jw + e * G + gl, dj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (((uint32_t)tt * 2654435761u >> 8) % 100u < 24u)
            __hip_atomic_fetch_add(iw + e * G + gl, di, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else if (e == 3) {
            float4* cb = reinterpret_cast<float4*>(Qrep);
            cb[tt * G + gl] = make_float4(di, dj, di + dj, di - dj);
            const float4 o = cb[((tt * 7919) % n) * G + gl];
            acc += o.x + o.y + o.z + o.w;
          }
        } else if constexpr (V == S_NEG_ONLY) {
          acc += di;
          upd<V>(jw + e * G + gl, dj);
        } else if constexpr (V != S_NONE) {
          upd<V>(iw + e * G + gl, di);
          upd<V>(jw + e * G + gl, dj);
        } else {
          acc += di + dj;
        }
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int V>
void run(const char* name, int blocks, float* P, float* Q, float* Qrep, int64_t stride, int* us, int* is, int* js,
         int64_t n, float* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, P, Q, Qrep, stride, us, is, js, n, 1e-6f, out);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int it = 0; it < 20; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, P, Q, Qrep, stride, us, is, js, n, 1e-6f, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double ms = ts[ts.size() / 2];
  const double rows = (V == S_NEG_ONLY) ? 1.0 : (V == S_AGG ? 1.24 : ((V == S_NONE || V == S_STORE) ? 0.0 : 2.0));
  printf("%-16s %8.3f ms  %8.1f Mtriples/s   %6.2f G line-atomics/s  %6.1f G dword-atomics/s\n", name, ms,
         n / ms * 1e-3, rows * 4 * n / ms * 1e-6, rows * 128 * n / ms * 1e-6);
}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 199168;
  int64_t U = argc > 2 ? atoll(argv[2]) : 136678, I = argc > 3 ? atoll(argv[3]) : 20109;
  int hot = argc > 4 ? atoi(argv[4]) : 2;  // 0 uniform items, 1 Zipf on neighbouring rows, 2 Zipf scattered
  float *P, *Q, *Qrep, *out; int *us, *is, *js;
  CK(hipMalloc(&P, U * D * 4)); CK(hipMalloc(&Q, I * D * 4)); const size_t rep_bytes = (size_t)std::max<int64_t>(8 * I, n) * D * 4; CK(hipMalloc(&Qrep, rep_bytes)); CK(hipMalloc(&out, 16));
  CK(hipMemset(Qrep, 0, rep_bytes));
  CK(hipMalloc(&us, n * 4)); CK(hipMalloc(&is, n * 4)); CK(hipMalloc(&js, n * 4));
  std::mt19937_64 rng(13);
  std::vector<float> h(U * D); for (auto& x : h) x = ((rng() >> 40) * (1.0f / 16777216.0f) - 0.5f) / D;
  CK(hipMemcpy(P, h.data(), U * D * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Q, h.data(), I * D * 4, hipMemcpyHostToDevice));
  std::vector<int> hu(n), hi(n), hj(n);
  std::vector<double> cdf(I - 1); double tot = 0;
  for (int64_t r = 0; r < I - 1; ++r) { tot += pow(r + 1 + 60.0, -1.5); cdf[r] = tot; }
  std::uniform_real_distribution<double> ud(0, 1);
  std::vector<int> perm(I - 1);
  for (int64_t r = 0; r < I - 1; ++r) perm[r] = (int)r;
  if (hot == 2) std::shuffle(perm.begin(), perm.end(), rng);
  for (int64_t t = 0; t < n; ++t) {
    hu[t] = 1 + rng() % (U - 1);
    hi[t] = hot ? 1 + perm[std::min<int64_t>(I - 2, std::lower_bound(cdf.begin(), cdf.end(), ud(rng) * tot) - cdf.begin())]
                : 1 + rng() % (I - 1);
    hj[t] = 1 + rng() % (I - 1);
  }
  std::sort(hu.begin(), hu.end());  // the product's chunks are grouped by user
  CK(hipMemcpy(us, hu.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(is, hi.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(js, hj.data(), n * 4, hipMemcpyHostToDevice));
  printf("n=%lld U=%lld I=%lld d=%d item popularity=%s\n", (long long)n, (long long)U, (long long)I, D,
         hot == 0 ? "uniform" : hot == 1 ? "zipf, neighbouring rows" : "zipf, scattered rows");
  const int blocks = 2048;
  const int64_t stride = I * D;
  run<S_NONE>("none", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_STORE>("store", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_NEG_ONLY>("neg-only", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_AGENT>("agent", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_AGENT_RET>("agent-returning", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_WG>("workgroup", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_WAVE>("wavefront", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_SYS>("system", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_XCD_WG>("xcd-replica-wg", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_AGG>("pos-aggregated", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  // (integer / double variants scribble over the fp32 table: last)
  run<S_U32>("u32", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_U64>("u64-packed", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  run<S_F64>("f64", blocks, P, Q, Qrep, stride, us, is, js, n, out);
  return 0;
}
