// Launch-overhead microbenchmark behind DESIGN.md's "no scratch memory in the hot kernels" rule:
//   hipcc --offload-arch=gfx950 -O2 -o launch_overhead scripts/launch_overhead.hip && ./launch_overhead
// MI355X, ROCm 7.2 (200 back-to-back launches, HIP events): every empty kernel costs ~3 us whatever its
// grid / LDS size, but a kernel with a scratch (private memory) segment costs 15.5 us at 256 x 1024 threads
// and 78.8 us at 6250 x 256 threads -- ~12 ns per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void k_static(float* out) { __shared__ float s[LDS > 0 ? LDS : 1]; if (threadIdx.x == 0) s[0] = 1.f; __syncthreads(); if (out && s[0] == 2.f) out[0] = 1.f; }
__global__ void k_dyn(float* out) { extern __shared__ float s[]; if (threadIdx.x == 0) s[0] = 1.f; __syncthreads(); if (out && s[0] == 2.f) out[0] = 1.f; }
__global__ void k_scratch(float* out, int n) { float a[64]; for (int i = 0; i < 64; ++i) a[i] = i * 1.5f; float r = 0; for (int i = 0; i < n; ++i) r += a[(i * 7 + threadIdx.x) & 63]; if (out && r == -1.f) out[0] = r; }
template <typename F> float timeit(F f, int n = 200) { for (int i = 0; i < 20; ++i) f(); hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a); for (int i = 0; i < n; ++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / n; }
int main() {
  float* d; hipMalloc(&d, 1024);
  printf("256x256 noLDS      %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<0>, dim3(256), dim3(256), 0, 0, (float*)nullptr); }));
  printf("256x1024 noLDS     %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<0>, dim3(256), dim3(1024), 0, 0, (float*)nullptr); }));
  printf("256x1024 LDS 60KB  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<15000>, dim3(256), dim3(1024), 0, 0, (float*)nullptr); }));
  printf("1024x256 LDS 60KB  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<15000>, dim3(1024), dim3(256), 0, 0, (float*)nullptr); }));
  printf("1563x256 LDS 60KB  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<15000>, dim3(1563), dim3(256), 0, 0, (float*)nullptr); }));
  hipFuncSetAttribute((const void*)k_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
  printf("256x1024 dyn 140KB %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_dyn, dim3(256), dim3(1024), 140000, 0, (float*)nullptr); }));
  printf("256x1024 scratch   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_scratch, dim3(256), dim3(1024), 0, 0, (float*)nullptr, 1); }));
  printf("6250x256 scratch   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_scratch, dim3(6250), dim3(256), 0, 0, (float*)nullptr, 1); }));
  printf("6250x256 noLDS     %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<0>, dim3(6250), dim3(256), 0, 0, (float*)nullptr); }));
  printf("1x64               %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_static<0>, dim3(1), dim3(64), 0, 0, (float*)nullptr); }));
  return 0;
}
