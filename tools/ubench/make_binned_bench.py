#!/usr/bin/env python
"""Builds tools/ubench/binned_bench: k_sort_binned (cut out of csrc/bpr_refresh.hip as it stands) timed alone on
the idle chip, with variants that leave phases out (WRONG orders: bounds on what each phase costs).
    python tools/ubench/make_binned_bench.py && tools/ubench/binned_bench"""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
src = (ROOT / "revisit-bpr_amd/csrc/bpr_refresh.hip").read_text()
body = src[src.index("constexpr int BIN_MAX = 64;"):src.index("// The binned sort with G workgroups per column")]
body = body[:body.rindex("// -----")]  # (the one-workgroup kernel only)


def variant(name, text):
    return "namespace " + name + " {\n" + text + "\n}\n"


no_rank = body.replace("      for (int j = lo; j < hi; j += 4) {", "      for (int j = lo; j < lo; j += 4) {", 1)
no_scatter = no_rank.replace("    if (l < n) {\n      s_key[at] = orderable_desc(keys[k]);", "    if (l < n && at < 0) {\n      s_key[at] = orderable_desc(keys[k]);")
assert no_rank != body and no_scatter != no_rank
onetrip = body.replace("      for (int j = lo; j < hi; j += 4) {", "      for (int j = lo; j < min(hi, lo + 1); j += 4) {", 1)
r1 = body.replace("  // ---- coarse histogram: 1,024 value-linear bins", "  return;\n  // ---- coarse histogram: 1,024 value-linear bins")
r2 = body.replace("  // ---- a key's bin from its interpolated rank", "  return;\n  // ---- a key's bin from its interpolated rank")
r3 = body.replace("  // ---- the bins' sizes -> first positions", "  return;\n  // ---- the bins' sizes -> first positions")
r4 = body.replace("  // ---- counting sort into LDS", "  return;\n  // ---- counting sort into LDS")
assert body not in (r1, r2, r3, r4)
harness = r'''
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>
''' + variant("full", body) + variant("norank", no_rank) + variant("noscatter", no_scatter) + variant("onetrip", onetrip) + variant("r1", r1) + variant("r2", r2) + variant("r3", r3) + variant("r4", r4) + r'''
template <typename K> float run(K k, const float* T, int64_t I, int d, int32_t* order, float* sigma, int32_t* meta) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(d), dim3(1024), 0, 0, T, I, order, sigma, meta);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(d), dim3(1024), 0, 0, T, I, order, sigma, meta);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}
int main() {
  const int64_t I = 20109; const int d = 128;
  std::vector<float> h(I * d); std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 0.05f);
  for (auto& v : h) v = nd(g);
  float *T, *sigma; int32_t *order, *meta;
  hipMalloc(&T, I * d * 4); hipMalloc(&order, I * d * 4); hipMalloc(&sigma, d * 4); hipMalloc(&meta, 2 * d * 4);
  hipMemcpy(T, h.data(), I * d * 4, hipMemcpyHostToDevice);
  printf("k_sort_binned<20>, %d columns of %lld keys, one per CU (128 of 256 CUs):\n", d, (long long)I);
  printf("  full                       %.1f us\n", run(full::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  no rank loop               %.1f us\n", run(norank::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  ... and no scatter         %.1f us\n", run(noscatter::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  rank loop capped at one trip %.1f us\n", run(onetrip::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  stops after sigma          %.1f us\n", run(r1::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  ... after the coarse histogram + scan %.1f us\n", run(r2::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  ... after the classification %.1f us\n", run(r3::k_sort_binned<20>, T, I, d, order, sigma, meta));
  printf("  ... after the bins' scan   %.1f us\n", run(r4::k_sort_binned<20>, T, I, d, order, sigma, meta));
  int32_t m[4]; hipMemcpy(m, meta, 16, hipMemcpyDeviceToHost); printf("  meta[0..3] = %d %d %d %d\n", m[0], m[1], m[2], m[3]);
  return 0;
}
'''
out = ROOT / "tools/ubench/binned_bench.hip"
out.write_text(harness)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                       str(out), "-o", str(ROOT / "tools/ubench/binned_bench")])
print("built tools/ubench/binned_bench")
