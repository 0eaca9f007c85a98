// cumask_probe.hip — which physical CUs does a hipExtStreamCreateWithCUMask bit range select?
//   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
// For masks with bits [first, first+count) set, launches a census kernel on the masked stream and
// prints, per XCD (HW_REG_XCC_ID), how many distinct (SE, SH, CU) ids ran blocks.  DESIGN.md §4.3
// relies on "a contiguous bit range takes the same number of CUs from every XCD".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <vector>

__global__ void census(uint32_t* out, int spin) {
  // HW_REG_XCC_ID = 20 (bits 3:0), HW_REG_HW_ID = 4
  const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
  const uint32_t hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
  float x = (float)threadIdx.x;
  for (int k = 0; k < spin; ++k) x = x * 1.0001f + 0.5f;  // keep the block resident for a while
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffffu) | (x == 12345.f ? 1u << 31 : 0u);
}

static void run(int first, int count, int total) {
  std::vector<uint32_t> mask((total + 31) / 32, 0u);
  for (int b = first; b < first + count; ++b) mask[b / 32] |= 1u << (b % 32);
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    printf("mask [%d,%d): create failed\n", first, first + count);
    return;
  }
  const int nb = 8192;
  uint32_t* d;
  hipMalloc(&d, nb * 4);
  hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, st, d, 20000);
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(nb);
  hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
  std::set<uint32_t> per_xcc[16];
  for (uint32_t v : h) per_xcc[(v >> 16) & 15].insert((v >> 8) & 0xffu);  // cu[11:8] sh[12] se[15:13]
  printf("mask bits [%3d,%3d):", first, first + count);
  int sum = 0;
  for (int x = 0; x < 8; ++x) {
    printf(" xcd%d=%zu", x, per_xcc[x].size());
    sum += (int)per_xcc[x].size();
  }
  printf("  total=%d\n", sum);
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int total = p.multiProcessorCount;
  printf("%s: %d CUs\n", p.name, total);
  run(0, total, total);
  for (int c : {8, 16, 32, 64, 128}) run(0, c, total);
  run(64, total - 64, total);
  run(32, total - 32, total);
  run(100, 17, total);
  return 0;
}
