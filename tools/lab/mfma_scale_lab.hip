// What would block-scaled fp8 buy the fused MLP kernels?  (lab only, not part of the library.)
//
// The stage-2 fused MLP (mlp_rs16.h) runs its two GEMMs on v_mfma_f32_16x16x32_bf16; the MI355X guide lists the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands at twice the cycles for four times the K: 2x per clock.  But a kernel's
// rate is FLOP per cycle x the clock the chip HOLDS in that loop, and the shipped kernel's bare MFMA skeleton already runs at
// the 1.5-1.7 GHz of MFMA-dense bf16 loops (profiles/r05_notes.md section 3).  This lab measures, on random operands, all CUs,
// two waves per SIMD:
//   mode 0  bf16 16x16x32, operands in registers          mode 1  scaled fp8 16x16x128, operands in registers
//   mode 2  bf16 32x32x16, registers                      mode 3  scaled fp8 32x32x64, registers
//   mode 4  bf16 16x16x32, the A operand of every MFMA read from LDS (16 bytes per lane), as the fused kernels feed W1 / W2
//   mode 5  scaled fp8 16x16x128, the A operand of every MFMA from LDS (32 bytes per lane: the same bytes per FLOP x 1/2)
// and prints TFLOP/s and the implied MFMA issue rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/mfma_scale_lab.hip -o tools/lab/mfma_scale_lab && tools/lab/mfma_scale_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
constexpr int NACC = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k_mfma(const i32x8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) int s_a[64 * 8 * NACC];   // NACC fragments of 32 bytes per lane
  const int tid = threadIdx.x, lane = tid & 63;
  const i32x8 va = src[(blockIdx.x * 256 + tid) & 4095], vb = src[(blockIdx.x * 256 + tid + 77) & 4095];
  if (tid < 64)
    for (int j = 0; j < NACC; ++j) *(i32x8*)(s_a + (j * 64 + lane) * 8) = src[(tid + 13 * j) & 4095];
  __syncthreads();
  if constexpr (MODE == 0 || MODE == 4) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, i32x4{va[0], va[1], va[2], va[3]});
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, i32x4{vb[0], vb[1], vb[2], vb[3]});
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        bf16x8 a = a0;
        if constexpr (MODE == 4) a = __builtin_bit_cast(bf16x8, *(const volatile i32x4*)(s_a + (j * 64 + lane) * 8));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[j], 0, 0, 0);
      }
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + tid] = s;
  } else if constexpr (MODE == 1 || MODE == 5) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        i32x8 a = va;
        if constexpr (MODE == 5) {
          const i32x4 lo = *(const volatile i32x4*)(s_a + (j * 64 + lane) * 8), hi = *(const volatile i32x4*)(s_a + (j * 64 + lane) * 8 + 4);
          a = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, vb, acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + tid] = s;
  } else if constexpr (MODE == 2) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, i32x4{va[0], va[1], va[2], va[3]});
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, i32x4{vb[0], vb[1], vb[2], vb[3]});
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    out[blockIdx.x * 256 + tid] = s;
  }
}

template <int MODE> static void run(const i32x8* src, float* out, int blocks, const char* name, double flop_per_mfma, int mfma_per_iter) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mfma<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, 2000);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mfma<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double n_mfma = (double)blocks * 4 * iters * mfma_per_iter;   // wave-level MFMA instructions
  const double tf = n_mfma * flop_per_mfma / (best * 1e-3) / 1e12;
  // per SIMD: instructions per second; the chip has 1024 SIMDs
  printf("mode %d  %-46s %8.1f TFLOP/s   %6.2f G MFMA/s per SIMD   (%.2f ms)\n", MODE, name, tf, n_mfma / 1024 / (best * 1e-3) / 1e9, best);
}

int main() {
  std::vector<int> h(4096 * 8);
  unsigned s = 2463534242u;
  for (auto& w : h) {
    unsigned v = 0;
    for (int b = 0; b < 4; ++b) {
      s ^= s << 13; s ^= s >> 17; s ^= s << 5;
      unsigned byte = s & 0xff;
      if ((byte & 0x7f) == 0x7f) byte ^= 0x10;          // no e4m3 NaN; as bf16 halves: finite values of mixed magnitude
      if (b == 1 || b == 3) byte = (byte & 0x80) | 0x3c | (byte & 3);   // bf16 exponent bytes: values around 1
      v |= byte << (8 * b);
    }
    w = (int)v;
  }
  i32x8* src; float* out;
  CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, 512 * 256 * 4));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const int blocks = 512;   // 256 CUs x 2 blocks of 4 waves: two waves per SIMD
  run<0>(src, out, blocks, "bf16 16x16x32, registers", 2.0 * 16 * 16 * 32, NACC);
  run<1>(src, out, blocks, "scaled fp8 (e4m3) 16x16x128, registers", 2.0 * 16 * 16 * 128, NACC);
  run<2>(src, out, blocks, "bf16 32x32x16, registers", 2.0 * 32 * 32 * 16, 4);
  run<3>(src, out, blocks, "scaled fp8 (e4m3) 32x32x64, registers", 2.0 * 32 * 32 * 64, 4);
  run<4>(src, out, blocks, "bf16 16x16x32, A from LDS (16 B / lane)", 2.0 * 16 * 16 * 32, NACC);
  run<5>(src, out, blocks, "scaled fp8 16x16x128, A from LDS (32 B / lane)", 2.0 * 16 * 16 * 128, NACC);
  return 0;
}
