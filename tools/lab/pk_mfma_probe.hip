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

// one application of the form under test: d = op(a, b, c)
template <int FORM> __device__ __forceinline__ f32x2 victim_op(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (FORM == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 4) {
    float d0;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
    d = f32x2{d0, 0.f};
  }
  if (FORM == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (FORM == 9) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
static const char* FORM_NAME[10] = {
    "v_pk_mul_f32 (no modifiers)", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_add_f32 (no modifiers)",
    "v_fma_f32 (scalar)", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]", "v_pk_mul_f32 neg_lo:[0,1]", "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_fma_f32 (no modifiers)",
    "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]"};

// res[0] = wrong results, res[1..2] = lanes (bit mask, 64 bits) in which a wrong result was seen, res[3] = results checked
template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(int iters, unsigned long long* res) {
  const int lane = threadIdx.x & 63;
  const f32x2 a = {1.25f + 0.001f * lane, -0.75f + 0.003f * lane}, b = {0.5f - 0.002f * lane, 1.5f + 0.001f * lane}, c = {0.125f, -0.25f * lane};
  const f32x2 ref = victim_op<FORM>(a, b, c);
  unsigned bad = 0;
  for (int i = 0; i < iters; ++i) {
    f32x2 aa = a, bb = b, cc = c;
    asm volatile("" : "+v"(aa), "+v"(bb), "+v"(cc));  // (opaque copies: the loop is not folded)
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

// co-runners: 0 nothing launched, 1 MFMA bf16 16x16x32, 2 MFMA bf16 32x32x16, 3 plain VALU, 4 MFMA f32 16x16x4, 5 LDS traffic
template <int KIND>
__global__ __launch_bounds__(256) void corun_kernel(int iters, float* sink) {
  __shared__ f32x4 s[1024];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (KIND == 1) {
    bf16x8 f;
    for (int i = 0; i < 8; ++i) f[i] = (__bf16)(1.0f + 0.01f * threadIdx.x);
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, acc, 0, 0, 0);
  }
  if (KIND == 2) {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    bf16x8 f;
    for (int i = 0; i < 8; ++i) f[i] = (__bf16)(1.0f + 0.01f * threadIdx.x);
    f32x16 a16;
    for (int i = 0; i < 16; ++i) a16[i] = 0.f;
    for (int i = 0; i < iters / 2; ++i) a16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, f, a16, 0, 0, 0);
    acc[0] = a16[0] + a16[7];
  }
  if (KIND == 3) {
    float x = 1.0f + threadIdx.x;
    for (int i = 0; i < iters * 4; ++i) x = fmaf(x, 1.0001f, 0.5f);
    acc[0] = x;
  }
  if (KIND == 4) {
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f + threadIdx.x, 2.0f, acc, 0, 0, 0);
  }
  if (KIND == 5) {
    for (int i = 0; i < iters / 4; ++i) {
      s[(threadIdx.x + 17 * i) & 1023] = acc + (float)i;
      __syncthreads();
      acc += s[(threadIdx.x * 5 + i) & 1023];
      __syncthreads();
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
static const char* CORUN_NAME[6] = {"nothing", "MFMA bf16 16x16x32", "MFMA bf16 32x32x16", "fp32 VALU", "MFMA f32 16x16x4", "LDS + barriers"};

template <int FORM> static void launch_victim(int grid, int iters, unsigned long long* res, hipStream_t s) {
  hipLaunchKernelGGL(victim_kernel<FORM>, dim3(grid), dim3(256), 0, s, iters, res);
}
static void launch_corun(int kind, int grid, int iters, float* sink, hipStream_t s) {
  switch (kind) {
    case 1: hipLaunchKernelGGL(corun_kernel<1>, dim3(grid), dim3(256), 0, s, iters, sink); break;
    case 2: hipLaunchKernelGGL(corun_kernel<2>, dim3(grid), dim3(256), 0, s, iters, sink); break;
    case 3: hipLaunchKernelGGL(corun_kernel<3>, dim3(grid), dim3(256), 0, s, iters, sink); break;
    case 4: hipLaunchKernelGGL(corun_kernel<4>, dim3(grid), dim3(256), 0, s, iters, sink); break;
    case 5: hipLaunchKernelGGL(corun_kernel<5>, dim3(grid), dim3(256), 0, s, iters, sink); break;
    default: break;
  }
}

int main(int argc, char** argv) {
  const int v_iters = argc > 1 ? atoi(argv[1]) : 200000, c_iters = argc > 2 ? atoi(argv[2]) : 400000, grid = argc > 3 ? atoi(argv[3]) : 512;
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa));
  CHECK(hipStreamCreate(&sb));
  unsigned long long* res;
  float* sink;
  CHECK(hipMalloc(&res, 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 4096));
  typedef void (*launch_t)(int, int, unsigned long long*, hipStream_t);
  launch_t forms[10] = {launch_victim<0>, launch_victim<1>, launch_victim<2>, launch_victim<3>, launch_victim<4>,
                        launch_victim<5>, launch_victim<6>, launch_victim<7>, launch_victim<8>, launch_victim<9>};
  for (int kind = 0; kind < 6; ++kind) {
    for (int form = 0; form < 10; ++form) {
      CHECK(hipMemset(res, 0, 4 * sizeof(unsigned long long)));
      CHECK(hipDeviceSynchronize());
      launch_corun(kind, grid, c_iters, sink, sb);
      forms[form](grid, v_iters, res, sa);
      CHECK(hipDeviceSynchronize());
      unsigned long long h[4];
      CHECK(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
      printf("co-runner %-20s | victim %-55s | wrong %10llu of %llu results, lanes %016llx\n", CORUN_NAME[kind], FORM_NAME[form], h[0], h[3], h[1]);
      fflush(stdout);
    }
  }
  return 0;
}
