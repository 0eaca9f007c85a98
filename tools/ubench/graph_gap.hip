// graph_gap.hip (r6) — what does a HIP graph buy for a chain of DEPENDENT kernels on gfx950?
// The STREAM step is [k_stream ~200 us] -> [epilogue/cut ~10 us] -> next k_stream, launch after launch on one stream;
// rocprofv3 puts ~5 us between two dependent kernels of that chain (profiles/r05d_timeline.txt: 19.5 us between two
// k_stream launches, 9.9 of them the cut kernel).  This probe runs the same shape — a long and a short spin kernel,
// alternating, N pairs — (a) launched on a stream, host far ahead of the device, (b) captured once into a hipGraph
// and replayed, (c) the graph replayed on a CU-MASKED stream (does the mask survive the graph?  kernel nodes carry
// none), and reports the time per pair beyond the kernels' own spin time.
//   hipcc -O2 --offload-arch=gfx950 -o graph_gap graph_gap.hip && ./graph_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void spin(long long ticks, unsigned* cu_seen) {
  const long long t0 = wall_clock64();
  if (cu_seen != nullptr && threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    atomicOr(&cu_seen[blockIdx.x & 1023], id);
  }
  while (wall_clock64() - t0 < ticks) {}
}

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  // wall_clock64 ticks at 100 MHz on gfx9: 10 ns per tick
  const long long LONG_T = 20000, SHORT_T = 1000;  // 200 us, 10 us
  const int N = 200;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto chain = [&](hipStream_t s) {
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(spin, dim3(cus), dim3(256), 0, s, LONG_T, (unsigned*)nullptr);
      hipLaunchKernelGGL(spin, dim3(cus), dim3(256), 0, s, SHORT_T, (unsigned*)nullptr);
    }
  };
  auto timed = [&](const char* label, auto&& fn, hipStream_t s) {
    fn();
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a, s));
      fn();
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    const double per_pair = best * 1e3 / N, spin_us = (LONG_T + SHORT_T) / 100.0;
    printf("%-52s %8.2f us per pair = %6.2f us beyond the kernels' %.0f us (two boundaries)\n", label, per_pair,
           per_pair - spin_us, spin_us);
  };
  timed("stream launches, host ahead of the device", [&] { chain(st); }, st);
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  chain(st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  timed("hipGraphLaunch of the captured chain", [&] { CK(hipGraphLaunch(ge, st)); }, st);
  // a CU-masked stream: 32 of the CUs.  Does a graph replayed on it stay on those CUs?
  std::vector<uint32_t> mask((cus + 31) / 32, 0u);
  mask[0] = 0xFFFFFFFFu;
  hipStream_t ms;
  CK(hipExtStreamCreateWithCUMask(&ms, (uint32_t)mask.size(), mask.data()));
  unsigned* seen;
  CK(hipMalloc(&seen, 1024 * sizeof(unsigned)));
  auto count_cus = [&](const char* label, auto&& launch) {
    CK(hipMemset(seen, 0, 1024 * sizeof(unsigned)));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(1024);
    CK(hipMemcpy(h.data(), seen, 1024 * sizeof(unsigned), hipMemcpyDeviceToHost));
    // HW_ID: cu_id bits 8..11, sh_id 12, se_id 13..15 (+ xcc elsewhere): count distinct (se, sh, cu) patterns OR-ed per slot
    int slots = 0;
    for (unsigned v : h) slots += v != 0;
    printf("%-52s %d of %d workgroup slots ran\n", label, slots, cus);
  };
  // a grid of `cus` blocks each spinning 50 us: on 32 CUs it takes ~ (cus / (32 * blocks per CU)) rounds
  auto masked_time = [&](const char* label, auto&& fn) {
    fn();
    CK(hipStreamSynchronize(ms));
    CK(hipEventRecord(a, ms));
    fn();
    CK(hipEventRecord(b, ms));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    printf("%-52s %8.2f us\n", label, t * 1e3);
  };
  const int big = cus * 8;  // 8 blocks per CU on the whole chip: one round unmasked, 8 rounds on 32 of 256 CUs
  masked_time("one 50-us kernel, grid 8 x CUs, unmasked stream", [&] { hipLaunchKernelGGL(spin, dim3(big), dim3(256), 0, st, 5000LL, (unsigned*)nullptr); CK(hipStreamSynchronize(st)); });
  masked_time("the same on the 32-CU masked stream", [&] { hipLaunchKernelGGL(spin, dim3(big), dim3(256), 0, ms, 5000LL, (unsigned*)nullptr); });
  hipGraph_t g2;
  hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(ms, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(spin, dim3(big), dim3(256), 0, ms, 5000LL, (unsigned*)nullptr);
  CK(hipStreamEndCapture(ms, &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  masked_time("the same captured on it and replayed as a graph", [&] { CK(hipGraphLaunch(ge2, ms)); });
  (void)count_cus;
  return 0;
}
