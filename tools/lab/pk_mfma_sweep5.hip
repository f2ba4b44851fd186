// SWEEP 5 of pk_mfma_probe.hip: the other cross-lane / conversion / transcendental instructions the library's kernels execute beside MFMAs
// (DPP row operations, ds_bpermute, v_permlane32_swap, packed conversions, v_exp / v_rcp, v_perm, v_med3), plus the faulty form as control.
// Minimal reproducer behind DESIGN.md's note on the log-mel corruption:
//   a VICTIM kernel whose waves repeat ONE packed-fp32 VALU instruction form on fixed per-lane inputs and compare every result
//   with the first one, next to a CO-RUNNER kernel on another stream that does one kind of work (MFMA, plain VALU, LDS, nothing).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/pk_mfma_probe tools/lab/pk_mfma_probe.hip && tools/lab/pk_mfma_probe
// Prints, per (victim form, co-runner), the number of wrong results and the lanes they were seen in.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int FORM> __device__ __forceinline__ f32x2 victim_op(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  if (FORM == 0) { float d0; asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d0) : "v"(a[0]), "v"(b[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 1) { float d0; asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d0) : "v"(a[0]), "v"(b[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 2) { float d0; asm volatile("v_add_f32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d0) : "v"(a[0]), "v"(b[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 3) { float d0; asm volatile("v_add_f32_dpp %0, %1, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d0) : "v"(a[0]), "v"(b[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 4) { float d0; d0 = 0.f; asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(d0) : "v"(a[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 5) { float d0; { int idx = ((threadIdx.x & 63) ^ 32) << 2; asm volatile("ds_bpermute_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(d0) : "v"(idx), "v"(a[0])); } d = f32x2{d0, 0.f}; }
  if (FORM == 6) { float x0 = a[0], x1 = b[0]; asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1)); d = f32x2{x0, x1}; }
  if (FORM == 7) { float d0; { unsigned u; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(a[0]), "v"(b[0])); d0 = __builtin_bit_cast(float, u); } d = f32x2{d0, 0.f}; }
  if (FORM == 8) { float d0; { unsigned u; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(a[0]), "v"(b[0])); d0 = __builtin_bit_cast(float, u); } d = f32x2{d0, 0.f}; }
  if (FORM == 9) { float d0; asm volatile("v_exp_f32 %0, %1" : "=v"(d0) : "v"(a[1])); d = f32x2{d0, 0.f}; }
  if (FORM == 10) { float d0; asm volatile("v_rcp_f32 %0, %1" : "=v"(d0) : "v"(a[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 11) { float d0; { unsigned u; asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u) : "v"(__builtin_bit_cast(unsigned, a[0])), "v"(__builtin_bit_cast(unsigned, b[0])), "v"(0x07060302u)); d0 = __builtin_bit_cast(float, u); } d = f32x2{d0, 0.f}; }
  if (FORM == 12) { float d0; asm volatile("v_fma_f32 %0, -%1, |%2|, %3" : "=v"(d0) : "v"(a[0]), "v"(b[0]), "v"(c[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 13) { float d0; asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(a[0]), "v"(b[0]), "v"(c[0])); d = f32x2{d0, 0.f}; }
  if (FORM == 14) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
#define NFORM 15
static const char* FORM_NAME[NFORM] = {
    "v_add_f32 dpp quad_perm:[1,0,3,2]",
    "v_add_f32 dpp row_shr:1",
    "v_add_f32 dpp row_half_mirror",
    "v_add_f32 dpp row_mirror",
    "v_mov_b32 dpp row_bcast:15",
    "ds_bpermute_b32 (lane ^ 32)",
    "v_permlane32_swap",
    "v_cvt_pk_bf16_f32",
    "v_cvt_pk_f16_f32",
    "v_exp_f32",
    "v_rcp_f32",
    "v_perm_b32",
    "v_fma_f32 neg/abs modifiers",
    "v_med3_f32",
    "v_pk_mul_f32 v,v op_sel:[0,1] (control: the faulty form)"};

template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(int iters, unsigned long long* res) {
  const int lane = threadIdx.x & 63;
  const f32x2 a = {1.25f + 0.001f * lane, -0.75f + 0.003f * lane}, b = {0.5f - 0.002f * lane, 1.5f + 0.001f * lane}, c = {0.125f, -0.25f * lane};
  const f32x2 ref = victim_op<FORM>(a, b, c);
  unsigned bad = 0;
  for (int i = 0; i < iters; ++i) {
    f32x2 aa = a, bb = b, cc = c;
    asm volatile("" : "+v"(aa), "+v"(bb), "+v"(cc));
    const f32x2 d = victim_op<FORM>(aa, bb, cc);
    bad += (__builtin_bit_cast(unsigned, d[0]) != __builtin_bit_cast(unsigned, ref[0])) | (__builtin_bit_cast(unsigned, d[1]) != __builtin_bit_cast(unsigned, ref[1]));
  }
  const unsigned long long m = __ballot(bad != 0);
  if (bad) atomicAdd(res, (unsigned long long)bad);
  if (lane == 0) {
    if (m) atomicOr(res + 1, m);
    atomicAdd(res + 3, (unsigned long long)iters * 64);
  }
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// co-runners: MFMA shapes
template <int KIND>
__global__ __launch_bounds__(256) void corun_kernel(int iters, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x16 a16;
  for (int i = 0; i < 16; ++i) a16[i] = 0.f;
  bf16x8 f;
  f16x8 h;
  for (int i = 0; i < 8; ++i) f[i] = (__bf16)(1.0f + 0.01f * threadIdx.x), h[i] = (_Float16)(1.0f + 0.01f * threadIdx.x);
  const long f8 = 0x3838383838383838l + threadIdx.x;
  if (KIND == 0) for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, acc, 0, 0, 0);
  if (KIND == 1) for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, acc, 0, 0, 0);
  if (KIND == 2) for (int i = 0; i < iters / 2; ++i) a16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, f, a16, 0, 0, 0);
  if (KIND == 3) for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(f8, f8, acc, 0, 0, 0);
  if (KIND == 4) for (int i = 0; i < iters / 2; ++i) a16 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(f8, f8, a16, 0, 0, 0);
  if (KIND == 5) for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f + threadIdx.x, 2.0f, acc, 0, 0, 0);
  if (acc[0] + acc[1] + acc[2] + acc[3] + a16[0] + a16[9] == 12345.678f) sink[threadIdx.x] = acc[0];
}
#define NCORUN 6
static const char* CORUN_NAME[NCORUN] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_fp8_fp8", "v_mfma_f32_32x32x16_fp8_fp8", "v_mfma_f32_16x16x4_f32"};

typedef void (*launch_t)(int, int, unsigned long long*, hipStream_t);
template <int FORM> static void launch_victim(int grid, int iters, unsigned long long* res, hipStream_t s) {
  hipLaunchKernelGGL(victim_kernel<FORM>, dim3(grid), dim3(256), 0, s, iters, res);
}
template <int N> struct Fill { static void go(launch_t* t) { t[N - 1] = launch_victim<N - 1>; Fill<N - 1>::go(t); } };
template <> struct Fill<0> { static void go(launch_t*) {} };
typedef void (*claunch_t)(int, int, float*, hipStream_t);
template <int K> static void launch_corun(int grid, int iters, float* sink, hipStream_t s) {
  hipLaunchKernelGGL(corun_kernel<K>, dim3(grid), dim3(256), 0, s, iters, sink);
}

int main(int argc, char** argv) {
  const int v_iters = argc > 1 ? atoi(argv[1]) : 60000, c_iters = argc > 2 ? atoi(argv[2]) : 120000, grid = argc > 3 ? atoi(argv[3]) : 512;
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa));
  CHECK(hipStreamCreate(&sb));
  unsigned long long* res;
  float* sink;
  CHECK(hipMalloc(&res, 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 4096));
  launch_t forms[NFORM];
  Fill<NFORM>::go(forms);
  claunch_t coruns[NCORUN] = {launch_corun<0>, launch_corun<1>, launch_corun<2>, launch_corun<3>, launch_corun<4>, launch_corun<5>};
  for (int kind = 0; kind < NCORUN; ++kind) {
    int n_bad = 0;
    for (int form = 0; form < NFORM; ++form) {
      CHECK(hipMemset(res, 0, 4 * sizeof(unsigned long long)));
      CHECK(hipDeviceSynchronize());
      coruns[kind](grid, c_iters, sink, sb);
      forms[form](grid, v_iters, res, sa);
      CHECK(hipDeviceSynchronize());
      unsigned long long h[4];
      CHECK(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
      if (h[0]) {
        ++n_bad;
        printf("beside %-28s | %-72s | wrong %10llu of %llu results (%.2f %%), lanes %016llx\n", CORUN_NAME[kind], FORM_NAME[form], h[0], h[3], 100.0 * h[0] / h[3], h[1]);
      }
      fflush(stdout);
    }
    printf("beside %-28s | %d of %d packed-fp32 forms returned wrong results\n", CORUN_NAME[kind], n_bad, NFORM);
  }
  return 0;
}
