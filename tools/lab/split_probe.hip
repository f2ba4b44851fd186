// Probe for the "exact" precision (development aid): how close is a split-operand MFMA product to fp32 / fp64?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/split_probe.hip -o tools/lab/split_probe && tools/lab/split_probe
// 1. does v_mfma_f32_16x16x32_f16 keep subnormal f16 inputs?  (a lo part of a small weight is subnormal)
// 2. error of C = A . B^T (16 x 16, K = 1536) against fp64 for: fp32 MFMA (16x16x4), f16 hi/lo 3-term, bf16 hi/lo 3-term,
//    plain bf16, plain f16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE>
__global__ void probe(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
      a[i] = A[(size_t)r * K + k0 + 8 * g + i];
      b[i] = B[(size_t)r * K + k0 + 8 * g + i];
    }
    if constexpr (MODE == 0) {  // fp32 MFMA
      for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
    } else if constexpr (MODE == 1 || MODE == 4) {  // f16 hi/lo (3 terms) | plain f16
      h8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)a[i];
        al[i] = (_Float16)(a[i] - (float)ah[i]);
        bh[i] = (_Float16)b[i];
        bl[i] = (_Float16)(b[i] - (float)bh[i]);
      }
      if (MODE == 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    } else {  // bf16 hi/lo (3 terms) | plain bf16
      b8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (__bf16)a[i];
        al[i] = (__bf16)(a[i] - (float)ah[i]);
        bh[i] = (__bf16)b[i];
        bl[i] = (__bf16)(b[i] - (float)bh[i]);
      }
      if (MODE == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  // D[i = 4 g + j][n = r]: rows of A operand... (A rows index the output ROW for operand 1; here both are 16 x K, C = A . B^T)
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = acc[j];
}

__global__ void denorm_probe(float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)9.5367431640625e-07f;  // 2^-20: subnormal in f16
    b[i] = (_Float16)1.0f;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];  // 32 * 2^-20 = 3.0517578125e-05 if subnormals are kept, 0 if flushed
}

int main() {
  const int K = 1536;
  std::vector<float> hA(16 * K), hB(16 * K);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (float)((s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
  };
  for (auto& v : hA) v = rnd() * 2.5f;                    // activations O(1)
  for (auto& v : hB) v = rnd() * 0.06f;                   // weights O(0.05)
  float *A, *B, *C, *D;
  hipMalloc(&A, hA.size() * 4);
  hipMalloc(&B, hB.size() * 4);
  hipMalloc(&C, 256 * 4);
  hipMalloc(&D, 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, D);
  float d = 0;
  hipMemcpy(&d, D, 4, hipMemcpyDeviceToHost);
  printf("f16 MFMA with subnormal inputs: got %.10g, kept = %.10g, flushed = 0\n", d, 32.0 * 9.5367431640625e-07);
  std::vector<double> ref(256);
  double mag = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double a = 0;
      for (int k = 0; k < K; ++k) a += (double)hA[m * K + k] * (double)hB[n * K + k];
      ref[m * 16 + n] = a;
      mag += fabs(a) / 256;
    }
  const char* names[5] = {"fp32 MFMA 16x16x4", "f16 hi/lo, 3 MFMAs", "bf16 hi/lo, 3 MFMAs", "bf16", "f16"};
  for (int mode = 0; mode < 5; ++mode) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, A, B, K, C); break;
      case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, A, B, K, C); break;
      case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, A, B, K, C); break;
      case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, A, B, K, C); break;
      case 4: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, A, B, K, C); break;
    }
    std::vector<float> h(256);
    hipMemcpy(h.data(), C, 1024, hipMemcpyDeviceToHost);
    double mx = 0, sm = 0;
    // output layout of this probe: C[(4 g + j) * 16 + r] = D[row 4 g + j of operand 1][col r of operand 2]
    for (int i = 0; i < 256; ++i) {
      const double e = fabs((double)h[i] - ref[i]);
      mx = e > mx ? e : mx;
      sm += e / 256;
    }
    printf("%-22s max |err| %.3e  mean |err| %.3e   (mean |C| %.3f)\n", names[mode], mx, sm, mag);
  }
  return 0;
}
