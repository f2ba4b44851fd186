// Timing lab of the LDS-staged depthwise + LayerNorm kernel (tools/lab/dw_lds.h: an experiment, not part of the library) at the pipeline's shapes, with phase ablations:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I conette-audio-captioning_amd/csrc -I tools/lab [-DCN_DWL_ABL=n] [-DLAB_S=..] [-DLAB_TH=..] tools/lab/dwl_lab.hip -o tools/lab/dwl_lab_n
//   tools/lab/dwl_lab_n <C: 96|192>     (ABL bits: 1 no global loads, 2 one pair-row of the convolution only, 4 no LayerNorm / store)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dw_lds.h"
void cn_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
#ifndef LAB_S
#define LAB_S 2
#endif
#ifndef LAB_TH
#define LAB_TH 12
#endif
template <int C> static void run(int B, int H, int W) {
  const size_t n = (size_t)B * H * W * C;
  half_t* x; bf16_t* y; unsigned* wp; float *db, *lw, *lb;
  CK(hipMalloc(&x, n * 2 + 4096)); CK(hipMalloc(&y, n * 2 + 4096)); CK(hipMalloc(&wp, 42 * C * 4)); CK(hipMalloc(&db, C * 4)); CK(hipMalloc(&lw, C * 4)); CK(hipMalloc(&lb, C * 4));
  std::vector<unsigned short> hx(n);
  unsigned s = 12345;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = 0x3000 + ((s >> 16) & 0x7ff); }
  CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
  std::vector<unsigned> hw(42 * C, 0x2c002e00u);
  CK(hipMemcpy(wp, hw.data(), 42 * C * 4, hipMemcpyHostToDevice));
  std::vector<float> ones(C, 1.0f);
  CK(hipMemcpy(db, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lw, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lb, ones.data(), C * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch_dwconv_lds<bf16_t, C, LAB_S, LAB_TH>(x, B, H, W, wp, db, lw, lb, y, 0);
  CK(hipDeviceSynchronize());
  const int it = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < it; ++i) launch_dwconv_lds<bf16_t, C, LAB_S, LAB_TH>(x, B, H, W, wp, db, lw, lb, y, 0);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("C=%d S=%d TH=%d ABL=%d lds=%zu B: %.1f us per launch\n", C, LAB_S, LAB_TH, CN_DWL_ABL, DwLds<C, LAB_S, LAB_TH>::BYTES, ms * 1000.f / it);
}
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 96;
  if (C == 96) run<96>(64, 252, 56); else run<192>(64, 126, 28);
  return 0;
}
