// VALU issue / throughput calibration on gfx950 (development aid): cycles per wave-instruction for the instruction
// kinds of the GELU epilogue at 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/lab/valu_lab.hip -o tools/lab/valu_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;

// KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_rcp_f32, 4 v_min_f32, 5 v_cvt_pk_bf16_f32, 6 v_pk_mul_f32, 7 mix (gelu-like),
// 8 v_dot2_f32_f16, 9 v_fma_mix_f32 (f16 src0), 10 v_perm_b32, 11 v_pk_fma_f16, 12 v_dot2c_f32_f16 (round 5: the depthwise conv's candidates)
template <int KIND>
__global__ void k(unsigned long long* out, float seed, int iters) {
  float a[8];
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 0.001f + i;
    p[i] = f32x2{a[i], a[i] * 0.5f};
  }
  const float c = seed * 0.999f, d = 0.001f;
  __syncthreads();
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (KIND == 8) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 9) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 11) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 12) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 13) {  // 3 dot2 : 1 fma_mix -- the row-paired depthwise conv (7 taps of a column in 4 instructions)
          if ((i & 3) == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(c), "v"(d));
          else asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(d));
        }
        if (KIND == 14) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(c), "v"(d));  // both sources f16
        if (KIND == 7) {  // 2 trans : 5 regular, as in the sigmoid GELU
          if ((i & 7) == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
          else if ((i & 7) == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        }
      }
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  if (s == 12345.678f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  unsigned long long* out;
  CK(hipMalloc(&out, 16));
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_min_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "mix 6 fma : 1 exp : 1 rcp",
                         "v_dot2_f32_f16", "v_fma_mix_f32", "v_perm_b32", "v_pk_fma_f16", "v_dot2c_f32_f16", "3 dot2 : 1 fma_mix", "v_fma_mix_f32 (f16 x f16)"};
  void (*ks[])(unsigned long long*, float, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>};
  for (int kind = 0; kind < 15; ++kind)
    for (int wps : {1, 2, 4, 5}) {
      const int iters = 4096;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(ks[kind], dim3(256), dim3(wps * 256), 0, 0, out, 1.0001f, iters);
      hipEventRecord(e0);
      hipLaunchKernelGGL(ks[kind], dim3(256), dim3(wps * 256), 0, 0, out, 1.0001f, iters);
      hipEventRecord(e1);
      CK(hipDeviceSynchronize());
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h;
      CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
      const double n = 64.0 * iters;
      printf("%-28s %d waves/SIMD: %6.2f ticks/instr/wave  %5.2f ticks/instr/SIMD | wall %7.1f us -> %5.3f ns per wave-instruction per SIMD (tick = %.3f ns)\n",
             names[kind], wps, h / n, h / n / wps, ms * 1e3, ms * 1e6 / (n * wps), ms * 1e6 / (double)h);
    }
  return 0;
}
