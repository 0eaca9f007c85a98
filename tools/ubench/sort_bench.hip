// sort_bench.hip — which library sort is fastest for the adaptive-sampler refresh (d segments of I keys)?
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_compose(const float* keys, uint64_t* out, int64_t n, int I) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    uint32_t b = __float_as_uint(keys[k]);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending-orderable
    b = ~b;                                            // descending
    out[k] = ((uint64_t)(k / I) << 32) | b;
  }
}

template <typename F> float timeit(F&& f, int iters = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  int d = argc > 1 ? atoi(argv[1]) : 128; int I = argc > 2 ? atoi(argv[2]) : 20109;
  int64_t n = (int64_t)d * I;
  std::vector<float> h(n); for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f);
  std::vector<int> hid(n), hoff(d + 1); for (int64_t k = 0; k < n; ++k) hid[k] = k % I; for (int f = 0; f <= d; ++f) hoff[f] = f * I;
  float *keys, *keys_out; int *ids, *ids_out, *offs; uint64_t *k64, *k64o;
  CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&keys_out, n * 4)); CK(hipMalloc(&ids, n * 4)); CK(hipMalloc(&ids_out, n * 4)); CK(hipMalloc(&offs, (d + 1) * 4));
  CK(hipMalloc(&k64, n * 8)); CK(hipMalloc(&k64o, n * 8));
  CK(hipMemcpy(keys, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ids, hid.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(offs, hoff.data(), (d + 1) * 4, hipMemcpyHostToDevice));
  void* tmp = nullptr; size_t bytes = 0, b2 = 0, b3 = 0;
  CK(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, bytes, keys, keys_out, ids, ids_out, (int)n, d, offs, offs + 1, 0, 32, 0));
  CK(hipcub::DeviceSegmentedSort::SortPairsDescending(nullptr, b2, keys, keys_out, ids, ids_out, (int)n, d, offs, offs + 1, 0));
  CK(hipcub::DeviceRadixSort::SortPairs(nullptr, b3, k64, k64o, ids, ids_out, (int)n, 0, 40, 0));
  size_t mx = std::max(bytes, std::max(b2, b3)); CK(hipMalloc(&tmp, mx));
  printf("d=%d I=%d n=%lld tmp bytes seg-radix %zu seg-sort %zu radix64 %zu\n", d, I, (long long)n, bytes, b2, b3);
  float t;
  t = timeit([&] { size_t b = bytes; hipcub::DeviceSegmentedRadixSort::SortPairsDescending(tmp, b, keys, keys_out, ids, ids_out, (int)n, d, offs, offs + 1, 0, 32, 0); });
  printf("DeviceSegmentedRadixSort      %8.3f ms\n", t);
  t = timeit([&] { size_t b = b2; hipcub::DeviceSegmentedSort::SortPairsDescending(tmp, b, keys, keys_out, ids, ids_out, (int)n, d, offs, offs + 1, 0); });
  printf("DeviceSegmentedSort (merge)   %8.3f ms\n", t);
  t = timeit([&] { size_t b = b2; hipcub::DeviceSegmentedSort::StableSortPairsDescending(tmp, b, keys, keys_out, ids, ids_out, (int)n, d, offs, offs + 1, 0); });
  printf("DeviceSegmentedSort stable    %8.3f ms\n", t);
  int fb = 32; while ((1 << (fb - 32)) < d) ++fb;
  t = timeit([&] { hipLaunchKernelGGL(k_compose, dim3(2048), dim3(256), 0, 0, keys, k64, n, I); size_t b = b3; hipcub::DeviceRadixSort::SortPairs(tmp, b, k64, k64o, ids, ids_out, (int)n, 0, fb, 0); });
  printf("compose + DeviceRadixSort u64 (%d bits) %8.3f ms\n", fb, t);
  t = timeit([&] { size_t b = b3; hipcub::DeviceRadixSort::SortPairs(tmp, b, k64, k64o, ids, ids_out, (int)n, 0, 32, 0); });
  printf("DeviceRadixSort u64 low 32 bits only  %8.3f ms (not a solution; pass-cost probe)\n", t);
  return 0;
}
