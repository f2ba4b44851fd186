// VALU calibration: fp32 FMA throughput for (a) a rolled loop and (b) a long straight-line body
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC, int ITERS>
__global__ __launch_bounds__(192) void rolled(const float* __restrict__ w, float* __restrict__ out) {
  float k[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) k[i] = w[i * 96 + (threadIdx.x % 96)];
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (float)i;
  float v = (float)threadIdx.x;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 49; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = fmaf(v + (float)i, k[j], acc[i]);
    v += 1.0f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 192 + threadIdx.x] = s;
}

template <int NACC, int ITERS>
__global__ __launch_bounds__(192) void straight(const float* __restrict__ w, float* __restrict__ out) {
  float k[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) k[i] = w[i * 96 + (threadIdx.x % 96)];
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (float)i;
  float v = (float)threadIdx.x;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 49; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = fmaf(v + (float)(i + it), k[j], acc[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 192 + threadIdx.x] = s;
}

int main() {
  const int blocks = 14336;
  float *w, *out;
  CK(hipMalloc(&w, 49 * 96 * 4));
  CK(hipMalloc(&out, (size_t)blocks * 192 * 4));
  CK(hipMemset(w, 0, 49 * 96 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](auto kern, const char* name, double fma_per_thread) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(192), 0, 0, w, out);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(192), 0, 0, w, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0;
    const double tf = fma_per_thread * blocks * 192 * 2 / (us * 1e-6) / 1e12;
    printf("%-28s %8.1f us  %6.1f TFLOP/s fp32\n", name, us, tf);
  };
  time(rolled<32, 1>, "rolled 32acc x49 x1", 32.0 * 49);
  time(rolled<32, 4>, "rolled 32acc x49 x4", 32.0 * 49 * 4);
  time(rolled<16, 8>, "rolled 16acc x49 x8", 16.0 * 49 * 8);
  time(straight<32, 1>, "straight 32acc x49 x1", 32.0 * 49);
  time(straight<32, 2>, "straight 32acc x49 x2", 32.0 * 49 * 2);
  time(straight<32, 4>, "straight 32acc x49 x4", 32.0 * 49 * 4);
  return 0;
}
