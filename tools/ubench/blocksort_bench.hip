// blocksort_bench.hip — block-size / items-per-thread / radix-bits choices for the in-LDS column sort
// of bpr_adaptive_refresh (256 workgroups, 10,064 (key, uint16 id) pairs each, descending, stable).
//   hipcc -O3 --offload-arch=gfx950 blocksort_bench.hip -o blocksort_bench && ./blocksort_bench
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/block/block_sort.hpp>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int BLOCK, int ITEMS, int RADIX>
__global__ __launch_bounds__(BLOCK) void k_sort(const float* __restrict__ T, int cnt, float* __restrict__ ko, uint16_t* __restrict__ vo) {
  using Sort = rocprim::block_radix_sort<float, BLOCK, ITEMS, uint16_t, 1, 1, RADIX>;
  __shared__ typename Sort::storage_type sm;
  const float* row = T + (int64_t)blockIdx.x * cnt;
  float keys[ITEMS]; uint16_t vals[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int l = threadIdx.x * ITEMS + k;
    keys[k] = l < cnt ? row[l] : -__builtin_huge_valf();
    vals[k] = (uint16_t)l;
  }
  Sort().sort_desc_to_striped(keys, vals, sm);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int pos = k * BLOCK + threadIdx.x;
    if (pos < cnt) { ko[(int64_t)blockIdx.x * cnt + pos] = keys[k]; vo[(int64_t)blockIdx.x * cnt + pos] = vals[k]; }
  }
}

struct Desc { __device__ bool operator()(const float& a, const float& b) const { return a > b; } };
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_msort(const float* __restrict__ T, int cnt, float* __restrict__ ko, uint16_t* __restrict__ vo) {
  using Sort = rocprim::block_sort<float, BLOCK, ITEMS, uint16_t, rocprim::block_sort_algorithm::stable_merge_sort>;
  __shared__ typename Sort::storage_type sm;
  const float* row = T + (int64_t)blockIdx.x * cnt;
  float keys[ITEMS]; uint16_t vals[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int l = threadIdx.x * ITEMS + k;
    keys[k] = l < cnt ? row[l] : -__builtin_huge_valf();
    vals[k] = (uint16_t)l;
  }
  Sort().sort(keys, vals, sm, Desc());
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int pos = threadIdx.x * ITEMS + k;
    if (pos < cnt) { ko[(int64_t)blockIdx.x * cnt + pos] = keys[k]; vo[(int64_t)blockIdx.x * cnt + pos] = vals[k]; }
  }
}
template <int BLOCK, int ITEMS>
void runm(const float* T, int wgs, int cnt, float* ko, uint16_t* vo) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ts;
  for (int it = 0; it < 13; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_msort<BLOCK, ITEMS>), dim3(wgs), dim3(BLOCK), 0, 0, T, cnt, ko, vo);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("merge sort block %4d items %2d : %7.1f us\n", BLOCK, ITEMS, ts[ts.size() / 2] * 1e3);
}

template <int BLOCK, int ITEMS, int RADIX>
void run(const float* T, int wgs, int cnt, float* ko, uint16_t* vo) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ts;
  for (int it = 0; it < 13; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_sort<BLOCK, ITEMS, RADIX>), dim3(wgs), dim3(BLOCK), 0, 0, T, cnt, ko, vo);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("block %4d items %2d radix %2d : %7.1f us\n", BLOCK, ITEMS, RADIX, ts[ts.size() / 2] * 1e3);
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, cnt = argc > 2 ? atoi(argv[2]) : 10064;
  std::vector<float> h((size_t)wgs * cnt);
  std::mt19937 rng(5); std::normal_distribution<float> nd(0.f, 0.05f);
  for (auto& x : h) x = nd(rng);
  float *T, *ko; uint16_t* vo;
  CK(hipMalloc(&T, h.size() * 4)); CK(hipMalloc(&ko, h.size() * 4)); CK(hipMalloc(&vo, h.size() * 2));
  CK(hipMemcpy(T, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("wgs=%d cnt=%d\n", wgs, cnt);
  if (cnt <= 10240) {
    runm<1024, 8>(T, wgs, cnt > 8192 ? 8192 : cnt, ko, vo);
    runm<512, 16>(T, wgs, cnt > 8192 ? 8192 : cnt, ko, vo);
    runm<1024, 16>(T, wgs, cnt, ko, vo); run<1024, 8, 0>(T, wgs, cnt > 8192 ? 8192 : cnt, ko, vo);
    run<1024, 10, 0>(T, wgs, cnt, ko, vo);
    run<1024, 10, 4>(T, wgs, cnt, ko, vo);
    run<1024, 10, 6>(T, wgs, cnt, ko, vo);
    run<1024, 10, 8>(T, wgs, cnt, ko, vo);
    run<512, 20, 0>(T, wgs, cnt, ko, vo);
    run<512, 20, 8>(T, wgs, cnt, ko, vo);
    run<512, 20, 6>(T, wgs, cnt, ko, vo);
    run<256, 40, 0>(T, wgs, cnt, ko, vo);
    run<256, 40, 8>(T, wgs, cnt, ko, vo);
    run<256, 40, 6>(T, wgs, cnt, ko, vo);
    run<256, 40, 5>(T, wgs, cnt, ko, vo);
  }
  if (cnt <= 5120) {
    run<1024, 5, 0>(T, wgs, cnt, ko, vo);
    run<512, 10, 0>(T, wgs, cnt, ko, vo);
    run<256, 20, 0>(T, wgs, cnt, ko, vo);
    run<256, 20, 8>(T, wgs, cnt, ko, vo);
  }
  return 0;
}
