// gs_bench.hip — micro-benchmark of gather/scatter structures for the BPR stream kernel (gfx950).
// Not part of the product; used to choose the update strategy (see DESIGN.md §kernel notes).
//   hipcc -O3 --offload-arch=gfx950 gs_bench.hip -o gs_bench && ./gs_bench [n_triples] [U] [I]
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
__device__ __forceinline__ void aadd(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void aadd_wg(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

enum { V_ATOMS_ONLYQ_256 = 10, V_ATOMS_SYS = 8, V_ATOMS_ONLYQ = 9, V_READ = 0, V_RMW = 1, V_ATOM4 = 2, V_ATOMS = 3, V_ATOM4_WG = 4, V_ATOM_LDS = 5, V_ATOMS_Q_RMW_P = 6, V_NT = 7 };

template <int V>
__global__ __launch_bounds__(256) void k(float* P, float* Q, const int* us, const int* is, const int* js,
                                         int64_t n, float lr, float* out) {
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), gw = lane / G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  __shared__ float4 lds[256];
  float acc = 0.f;
  for (int64_t base = wave * 2; base < n; base += nw * 2) {
    const int64_t t = base + gw;
    const bool act = t < n;
    const int64_t tt = act ? t : n - 1;
    const int u = us[tt], i = is[tt], j = js[tt];
    float* pr = P + (int64_t)u * D;
    float* ir = Q + (int64_t)i * D;
    float* jr = Q + (int64_t)j * D;
    if constexpr (V == V_ATOMS_ONLYQ_256) {
      // same work as V_ATOMS_ONLYQ, but every atomic instruction covers 256 contiguous bytes of ONE
      // row (lanes 0-31: chunk e, lanes 32-63: chunk e+1 of the same triple's row) instead of
      // 128 B of two different rows
      float p[4], qi[4], qj[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { p[e] = pr[e * G + gl]; qi[e] = ir[e * G + gl]; qj[e] = jr[e * G + gl]; }
      float x = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) x += p[e] * (qi[e] - qj[e]);
      x = gsum(x);
      const float w = 1.f / (1.f + __expf(x));
      acc += x;
      // rows of the wave's two triples, known to all lanes
      const int iA = __shfl(i, 0, 64), iB = __shfl(i, 32, 64), jA = __shfl(j, 0, 64), jB = __shfl(j, 32, 64);
      const bool actA = __shfl((int)act, 0, 64), actB = __shfl((int)act, 32, 64);
      float* rows[4] = {Q + (int64_t)iA * D, Q + (int64_t)jA * D, Q + (int64_t)iB * D, Q + (int64_t)jB * D};
      const bool acts[4] = {actA, actA, actB, actB};
      if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pr[e * G + gl] = p[e] + lr * (w * (qi[e] - qj[e]) - 0.01f * p[e]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // 2 instructions x 256 B per row
          if (acts[r]) aadd(rows[r] + h * 64 + lane, lr * (w * p[h] - 0.01f * qi[h]));
        }
      }
    } else if constexpr (V == V_ATOMS_SYS || V == V_ATOMS_ONLYQ) {
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (V == V_ATOMS_SYS) {
            __hip_atomic_fetch_add(pr + e * G + gl, lr * (w * (qi[e] - qj[e]) - 0.01f * p[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_add(ir + e * G + gl, lr * (w * p[e] - 0.01f * qi[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_add(jr + e * G + gl, lr * (-w * p[e] - 0.01f * qj[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          } else {
            pr[e * G + gl] = p[e] + lr * (w * (qi[e] - qj[e]) - 0.01f * p[e]);
            aadd(ir + e * G + gl, lr * (w * p[e] - 0.01f * qi[e]));
            aadd(jr + e * G + gl, lr * (-w * p[e] - 0.01f * qj[e]));
          }
        }
      }
    } else if constexpr (V == V_ATOMS) {
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          aadd(pr + e * G + gl, lr * (w * (qi[e] - qj[e]) - 0.01f * p[e]));
          aadd(ir + e * G + gl, lr * (w * p[e] - 0.01f * qi[e]));
          aadd(jr + e * G + gl, lr * (-w * p[e] - 0.01f * qj[e]));
        }
      }
    } else {
      float4 p, qi, qj;
      if constexpr (V == V_NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f a_ = __builtin_nontemporal_load(reinterpret_cast<v4f*>(pr) + gl);
        v4f b_ = __builtin_nontemporal_load(reinterpret_cast<v4f*>(ir) + gl);
        v4f c_ = __builtin_nontemporal_load(reinterpret_cast<v4f*>(jr) + gl);
        p = make_float4(a_.x, a_.y, a_.z, a_.w); qi = make_float4(b_.x, b_.y, b_.z, b_.w); qj = make_float4(c_.x, c_.y, c_.z, c_.w);
      } else {
        p = reinterpret_cast<float4*>(pr)[gl];
        qi = reinterpret_cast<float4*>(ir)[gl];
        qj = reinterpret_cast<float4*>(jr)[gl];
      }
      float x = p.x * (qi.x - qj.x) + p.y * (qi.y - qj.y) + p.z * (qi.z - qj.z) + p.w * (qi.w - qj.w);
      x = gsum(x);
      const float w = 1.f / (1.f + __expf(x));
      acc += x;
      float4 dp, di, dj;
      dp.x = lr * (w * (qi.x - qj.x) - 0.01f * p.x); dp.y = lr * (w * (qi.y - qj.y) - 0.01f * p.y);
      dp.z = lr * (w * (qi.z - qj.z) - 0.01f * p.z); dp.w = lr * (w * (qi.w - qj.w) - 0.01f * p.w);
      di.x = lr * (w * p.x - 0.01f * qi.x); di.y = lr * (w * p.y - 0.01f * qi.y);
      di.z = lr * (w * p.z - 0.01f * qi.z); di.w = lr * (w * p.w - 0.01f * qi.w);
      dj.x = lr * (-w * p.x - 0.01f * qj.x); dj.y = lr * (-w * p.y - 0.01f * qj.y);
      dj.z = lr * (-w * p.z - 0.01f * qj.z); dj.w = lr * (-w * p.w - 0.01f * qj.w);
      if constexpr (V == V_READ) {
        acc += dp.x + di.y + dj.z;
      } else if constexpr (V == V_RMW) {
        if (act) {
          reinterpret_cast<float4*>(pr)[gl] = make_float4(p.x + dp.x, p.y + dp.y, p.z + dp.z, p.w + dp.w);
          reinterpret_cast<float4*>(ir)[gl] = make_float4(qi.x + di.x, qi.y + di.y, qi.z + di.z, qi.w + di.w);
          reinterpret_cast<float4*>(jr)[gl] = make_float4(qj.x + dj.x, qj.y + dj.y, qj.z + dj.z, qj.w + dj.w);
        }
      } else if constexpr (V == V_ATOM4 || V == V_NT) {
        if (act) {
          aadd(pr + 4 * gl + 0, dp.x); aadd(pr + 4 * gl + 1, dp.y); aadd(pr + 4 * gl + 2, dp.z); aadd(pr + 4 * gl + 3, dp.w);
          aadd(ir + 4 * gl + 0, di.x); aadd(ir + 4 * gl + 1, di.y); aadd(ir + 4 * gl + 2, di.z); aadd(ir + 4 * gl + 3, di.w);
          aadd(jr + 4 * gl + 0, dj.x); aadd(jr + 4 * gl + 1, dj.y); aadd(jr + 4 * gl + 2, dj.z); aadd(jr + 4 * gl + 3, dj.w);
        }
      } else if constexpr (V == V_ATOM4_WG) {
        if (act) {
          aadd_wg(pr + 4 * gl + 0, dp.x); aadd_wg(pr + 4 * gl + 1, dp.y); aadd_wg(pr + 4 * gl + 2, dp.z); aadd_wg(pr + 4 * gl + 3, dp.w);
          aadd_wg(ir + 4 * gl + 0, di.x); aadd_wg(ir + 4 * gl + 1, di.y); aadd_wg(ir + 4 * gl + 2, di.z); aadd_wg(ir + 4 * gl + 3, di.w);
          aadd_wg(jr + 4 * gl + 0, dj.x); aadd_wg(jr + 4 * gl + 1, dj.y); aadd_wg(jr + 4 * gl + 2, dj.z); aadd_wg(jr + 4 * gl + 3, dj.w);
        }
      } else if constexpr (V == V_ATOMS_Q_RMW_P) {
        if (act) {
          reinterpret_cast<float4*>(pr)[gl] = make_float4(p.x + dp.x, p.y + dp.y, p.z + dp.z, p.w + dp.w);
          aadd(ir + 4 * gl + 0, di.x); aadd(ir + 4 * gl + 1, di.y); aadd(ir + 4 * gl + 2, di.z); aadd(ir + 4 * gl + 3, di.w);
          aadd(jr + 4 * gl + 0, dj.x); aadd(jr + 4 * gl + 1, dj.y); aadd(jr + 4 * gl + 2, dj.z); aadd(jr + 4 * gl + 3, dj.w);
        }
      } else if constexpr (V == V_ATOM_LDS) {
        // transpose through LDS so that each atomic instruction covers 128 contiguous bytes per group
        float* l = reinterpret_cast<float*>(lds) + (threadIdx.x & ~(G - 1)) * 4;  // group's 128 floats
        float4* l4 = reinterpret_cast<float4*>(l);
        float* rows[3] = {pr, ir, jr};
        float4 vals[3] = {dp, di, dj};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          l4[gl] = vals[r];
          __builtin_amdgcn_wave_barrier();
          float e0 = l[gl], e1 = l[G + gl], e2 = l[2 * G + gl], e3 = l[3 * G + gl];
          __builtin_amdgcn_wave_barrier();
          if (act) { aadd(rows[r] + gl, e0); aadd(rows[r] + G + gl, e1); aadd(rows[r] + 2 * G + gl, e2); aadd(rows[r] + 3 * G + gl, e3); }
        }
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}


// Replica experiment: the H hottest item rows take their updates in R delta replicas (writer picks
// one by wave id), readers add the replicas to the base row.  Same algebra, R-fold less same-line
// contention; the deltas are folded into Q after the launch (not timed here: H*R rows).
__global__ __launch_bounds__(256) void k_rep(float* P, float* Q, float* Dl, const int* slot, int H, int R,
                                             const int* us, const int* is, const int* js, int64_t n, float lr,
                                             float* out) {
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), gw = lane / G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  const int rep = (int)(wave % R);
  for (int64_t base = wave * 2; base < n; base += nw * 2) {
    const int64_t t = base + gw;
    const bool act = t < n;
    const int64_t tt = act ? t : n - 1;
    const int u = us[tt], i = is[tt], j = js[tt];
    const int si = slot[i], sj = slot[j];
    float* pr = P + (int64_t)u * D;
    float* ir = Q + (int64_t)i * D;
    float* jr = Q + (int64_t)j * D;
    float p[4], qi[4], qj[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[e] = pr[e * G + gl]; qi[e] = ir[e * G + gl]; qj[e] = jr[e * G + gl]; }
    if (si >= 0)
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) qi[e] += Dl[((int64_t)r * H + si) * D + e * G + gl];
    if (sj >= 0)
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) qj[e] += Dl[((int64_t)r * H + sj) * D + e * G + gl];
    float x = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) x += p[e] * (qi[e] - qj[e]);
    x = gsum(x);
    const float w = 1.f / (1.f + __expf(x));
    acc += x;
    float* iw = si >= 0 ? Dl + ((int64_t)rep * H + si) * D : ir;
    float* jw = sj >= 0 ? Dl + ((int64_t)rep * H + sj) * D : jr;
    if (act) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pr[e * G + gl] = p[e] + lr * (w * (qi[e] - qj[e]) - 0.01f * p[e]);
        aadd(iw + e * G + gl, lr * (w * p[e] - 0.01f * qi[e]));
        aadd(jw + e * G + gl, lr * (-w * p[e] - 0.01f * qj[e]));
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int V>
double run(const char* name, int blocks, float* P, float* Q, int* us, int* is, int* js, int64_t n, float* out, int iters = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, P, Q, us, is, js, n, 1e-6f, out);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int it = 0; it < iters; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, P, Q, us, is, js, n, 1e-6f, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  double ms = ts[ts.size() / 2];
  printf("%-18s blocks=%6d n=%9lld  %8.3f ms  %8.1f Mtriples/s  alg %7.1f GB/s\n", name, blocks, (long long)n, ms,
         n / ms * 1e-3, n * 3080.0 / ms * 1e-6);
  return ms;
}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 199168;
  int64_t U = argc > 2 ? atoll(argv[2]) : 136678, I = argc > 3 ? atoll(argv[3]) : 20109;
  int hot = argc > 4 ? atoi(argv[4]) : 1;  // 0 uniform items, 1 Zipf with rank r -> row r+1, 2 Zipf over permuted rows
  float *P, *Q, *out; int *us, *is, *js;
  CK(hipMalloc(&P, U * D * 4)); CK(hipMalloc(&Q, I * D * 4)); CK(hipMalloc(&out, 16));
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
    hi[t] = hot ? 1 + perm[std::min<int64_t>(I - 2, std::lower_bound(cdf.begin(), cdf.end(), ud(rng) * tot) - cdf.begin())] : 1 + rng() % (I - 1);
    hj[t] = 1 + rng() % (I - 1);
  }
  CK(hipMemcpy(us, hu.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(is, hi.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(js, hj.data(), n * 4, hipMemcpyHostToDevice));
  printf("n=%lld U=%lld I=%lld d=%d hot=%d\n", (long long)n, (long long)U, (long long)I, D, hot);
  for (int H : {64, 256, 1024, 4096}) {
    for (int R : {2, 4, 8}) {
      std::vector<int> hs(I, -1);
      for (int r = 0; r < H && r < I - 1; ++r) hs[1 + perm[r]] = r;  // rank r -> row 1+perm[r]
      int* slot; float* Dl;
      CK(hipMalloc(&slot, I * 4)); CK(hipMalloc(&Dl, (size_t)R * H * D * 4));
      CK(hipMemcpy(slot, hs.data(), I * 4, hipMemcpyHostToDevice));
      CK(hipMemset(Dl, 0, (size_t)R * H * D * 4));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      std::vector<float> ts;
      for (int it = 0; it < 23; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_rep, dim3(2048), dim3(256), 0, 0, P, Q, Dl, slot, H, R, us, is, js, n, 1e-6f, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      printf("replicas H=%5d R=%d  %8.3f ms  %8.1f Mtriples/s\n", H, R, ts[ts.size() / 2], n / ts[ts.size() / 2] * 1e-3);
      CK(hipFree(slot)); CK(hipFree(Dl));
    }
  }
  for (int blocks : {2048}) {
    run<V_READ>("read-only", blocks, P, Q, us, is, js, n, out);
    run<V_RMW>("rmw-store", blocks, P, Q, us, is, js, n, out);
    run<V_ATOM4>("atomic-strided4", blocks, P, Q, us, is, js, n, out);
    run<V_NT>("atomic4+nt-load", blocks, P, Q, us, is, js, n, out);
    run<V_ATOMS>("atomic-contig", blocks, P, Q, us, is, js, n, out);
    run<V_ATOM_LDS>("atomic-lds-T", blocks, P, Q, us, is, js, n, out);
    run<V_ATOMS_SYS>("atomic-contig-sys", blocks, P, Q, us, is, js, n, out);
    run<V_ATOMS_ONLYQ>("P-store+Q-contig", blocks, P, Q, us, is, js, n, out);
    run<V_ATOMS_ONLYQ_256>("P-store+Q-256B", blocks, P, Q, us, is, js, n, out);
    run<V_ATOM4_WG>("atomic4-wgscope", blocks, P, Q, us, is, js, n, out);
    run<V_ATOMS_Q_RMW_P>("P-rmw+Q-atomic4", blocks, P, Q, us, is, js, n, out);
  }
  return 0;
}
