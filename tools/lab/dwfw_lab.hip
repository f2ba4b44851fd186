// Timing lab of the SHIPPED full-width depthwise + LayerNorm kernel (cn_dwconv_ln_fw_kernel, extracted from csrc/encoder.hip by
// tools/lab/dwfw_lab.sh into fw_extract.h) at the pipeline's stage-2 / stage-3 shapes, with two phase cuts:
//   FW_ABL 0 = the kernel, 1 = convolution only (return before the LayerNorm), 2 = LayerNorm + store only (no convolution)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#ifndef FW_ABL
#define FW_ABL 0
#endif
#define CN_DW_DOT2 1
__device__ int g_cn_nonfinite;
__device__ __forceinline__ void cn_watch_stat(float v) {
  if (!(__builtin_fabsf(v) <= 3.0e38f)) atomicAdd(&g_cn_nonfinite, 1);
}
void cn_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
#include "fw_extract.h"
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
template <int C, int WW, int TH, int SPLIT> static void run(int B, int H) {
  const size_t n = (size_t)B * H * WW * C;
  half_t* x; bf16_t* y; unsigned* wp; float *dw, *db, *lw, *lb;
  CK(hipMalloc(&x, n * 2 + 4096)); CK(hipMalloc(&y, n * 2 + 4096)); CK(hipMalloc(&wp, 42 * C * 4)); CK(hipMalloc(&dw, 49 * C * 4));
  CK(hipMalloc(&db, C * 4)); CK(hipMalloc(&lw, C * 4)); CK(hipMalloc(&lb, C * 4));
  std::vector<unsigned short> hx(n);
  unsigned s = 12345;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = 0x3000 + ((s >> 16) & 0x7ff); }
  CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
  std::vector<unsigned> hw(42 * C, 0x2c002e00u);
  CK(hipMemcpy(wp, hw.data(), 42 * C * 4, hipMemcpyHostToDevice));
  std::vector<float> ones(49 * C, 0.1f);
  CK(hipMemcpy(dw, ones.data(), 49 * C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lw, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lb, ones.data(), C * 4, hipMemcpyHostToDevice));
  const int tiles_h = (H + TH - 1) / TH;
  const size_t smem = ((size_t)TH * WW * (C + 4) + 2 * TH * WW) * sizeof(float);
  auto kern = cn_dwconv_ln_fw_kernel<bf16_t, half_t, C, WW, TH, SPLIT>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int threads = C * SPLIT > 384 ? 768 : 384;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(B * tiles_h), dim3(threads), smem, 0, x, H, tiles_h, dw, wp, db, lw, lb, y);
  CK(hipDeviceSynchronize());
  const int it = 30;
  CK(hipEventRecord(e0));
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(B * tiles_h), dim3(threads), smem, 0, x, H, tiles_h, dw, wp, db, lw, lb, y);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("fw C=%d W=%d TH=%d SPLIT=%d ABL=%d lds=%zu B blocks=%d: %.1f us per launch\n", C, WW, TH, SPLIT, FW_ABL, smem, B * tiles_h, ms * 1000.f / it);
}
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 384;
  if (C == 384) run<384, 14, 4, 2>(64, 63); else run<768, 7, 4, 1>(64, 31);
  return 0;
}
