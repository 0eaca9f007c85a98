// pmc_calib.hip — known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE for the
// access pattern of k_stream (one dword per lane, 128-B runs) on gfx950.
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./pmc_calib     (and again with WRITE_SIZE)
// Each kernel moves exactly N bytes of a buffer larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void k_read_dword(const float* in, float* out, size_t n) {   // n floats read once
  float acc = 0.f;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) acc += in[k];
  if (acc == 1234.5f) out[0] = acc;
}
__global__ void k_read_dwordx4(const float4* in, float* out, size_t n4) {
  float acc = 0.f;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (size_t)gridDim.x * blockDim.x) { float4 v = in[k]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 1234.5f) out[0] = acc;
}
__global__ void k_write_dword(float* out, size_t n) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) out[k] = 1.0f;
}
__global__ void k_atomic_dword(float* out, size_t n) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
    __hip_atomic_fetch_add(out + k, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// random 512-B rows, dword per lane (the gather pattern of k_stream)
__global__ void k_gather_rows(const float* in, float* out, size_t rows, size_t n_gathers) {
  float acc = 0.f;
  const int lane = threadIdx.x & 31;
  for (size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; g < n_gathers; g += ((size_t)gridDim.x * blockDim.x) >> 5) {
    size_t r = (g * 2654435761ull) % rows;
    const float* row = in + r * 128;
    acc += row[lane] + row[32 + lane] + row[64 + lane] + row[96 + lane];
  }
  if (acc == 1234.5f) out[0] = acc;
}

int main() {
  const size_t n = (size_t)1 << 28;  // 1 GiB of floats
  float *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipLaunchKernelGGL(k_read_dword, dim3(4096), dim3(256), 0, 0, a, b, n);
  hipLaunchKernelGGL(k_read_dwordx4, dim3(4096), dim3(256), 0, 0, (const float4*)a, b, n / 4);
  hipLaunchKernelGGL(k_write_dword, dim3(4096), dim3(256), 0, 0, b, n);
  hipLaunchKernelGGL(k_atomic_dword, dim3(4096), dim3(256), 0, 0, b, n / 8);
  hipLaunchKernelGGL(k_gather_rows, dim3(4096), dim3(256), 0, 0, a, b, n / 128, (size_t)1 << 21);
  CK(hipDeviceSynchronize());
  printf("bytes: read_dword %zu read_dwordx4 %zu write_dword %zu atomic_dword %zu (n/8 floats) gather_rows %zu\n",
         n * 4, n * 4, n * 4, n / 8 * 4, ((size_t)1 << 21) * 512);
  return 0;
}
