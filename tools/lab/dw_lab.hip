// Kernel lab for the depthwise-conv + LayerNorm kernels (tools/lab/dwconv_slide.h, an experiment that is NOT part of the library): stand-alone check against a naive GPU
// reference (fp32 output, tolerance 2e-5 + 2e-5 |ref|) and timing at the pipeline's shapes (B = 64, 10 s clips).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I conette-audio-captioning_amd/csrc -I tools/lab tools/lab/dw_lab.hip -o tools/lab/dw_lab
//   tools/lab/dw_lab <C: 96|192|384|768> [batch] [iters]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "dwconv_slide.h"

void cn_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }

#define CK(e)                                                                         \
  do {                                                                                \
    hipError_t _e = (e);                                                              \
    if (_e != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

static unsigned g_seed = 12345;
static float frand() {
  g_seed = g_seed * 1664525u + 1013904223u;
  return ((g_seed >> 8) & 0xffff) / 32768.0f - 1.0f;
}

// one thread per position: conv for all channels into a scratch row, then two-pass LayerNorm
__global__ void ref_dwconv_ln(const float* x, int B, int H, int W, int C, const float* dw_w, const float* dw_b, const float* ln_w,
                              const float* ln_b, float* conv, float* y) {
  const long pos = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (long)B * H * W) return;
  const int w = pos % W, h = (pos / W) % H, b = pos / ((long)W * H);
  float* cv = conv + pos * C;
  double sum = 0;
  for (int c = 0; c < C; ++c) {
    float a = dw_b[c];
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) {
        const int hh = h + i - 3, ww = w + j - 3;
        if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
        a = fmaf(x[(((long)b * H + hh) * W + ww) * C + c], dw_w[(i * 7 + j) * C + c], a);
      }
    cv[c] = a;
    sum += a;
  }
  const float mean = (float)(sum / C);
  double sq = 0;
  for (int c = 0; c < C; ++c) sq += (double)(cv[c] - mean) * (cv[c] - mean);
  const float rstd = 1.0f / sqrtf((float)(sq / C) + 1e-6f);
  for (int c = 0; c < C; ++c) y[pos * C + c] = (cv[c] - mean) * rstd * ln_w[c] + ln_b[c];
}

template <typename T> static T* dalloc(size_t n) {
  T* p;
  CK(hipMalloc(&p, n * sizeof(T)));
  return p;
}

template <typename T, int C> static int launch(const float* x, int B, int H, int W, int HL, const float* dw_w, const float* dw_b,
                                                const float* ln_w, const float* ln_b, T* y) {
  if constexpr (C == 96) return cn_launch_dwconv_slide<T, C, 2, 7>(x, B, H, W, HL, dw_w, dw_b, ln_w, ln_b, y, 0);
  else if constexpr (C == 192) return cn_launch_dwconv_slide<T, C, 2, 4>(x, B, H, W, HL, dw_w, dw_b, ln_w, ln_b, y, 0);
  else if constexpr (C == 384) return cn_launch_dwconv_slide<T, C, 2, 2>(x, B, H, W, HL, dw_w, dw_b, ln_w, ln_b, y, 0);
  else return cn_launch_dwconv_slide<T, C, 2, 1>(x, B, H, W, HL, dw_w, dw_b, ln_w, ln_b, y, 0);
}

template <int C> static int run(int batch, int iters, int HLarg) {
  const int H = C == 96 ? 252 : C == 192 ? 126 : C == 384 ? 63 : 32, W = 5376 / C;
  const int Bc = 2, Hc = 37;  // check shape: two clips, ragged last row tile
  const size_t n = (size_t)batch * H * W * C, nc = (size_t)Bc * Hc * W * C;
  printf("== C = %d, B = %d, H = %d, W = %d (%.1f M outputs); check B = %d, H = %d\n", C, batch, H, W, n * 1e-6, Bc, Hc);
  std::vector<float> hx(n + 8 * C), hk(49 * C), hb(C), hg(C), hbe(C);
  for (auto& v : hx) v = frand();
  for (auto& v : hk) v = frand() * 0.3f;
  for (auto& v : hb) v = frand() * 0.5f;
  for (auto& v : hg) v = 1.0f + 0.3f * frand();
  for (auto& v : hbe) v = 0.2f * frand();
  float* xall = dalloc<float>(hx.size());
  float* x = xall + 4 * C;  // readable slack on both sides
  float *k = dalloc<float>(49 * C), *bb = dalloc<float>(C), *g = dalloc<float>(C), *be = dalloc<float>(C);
  CK(hipMemcpy(xall, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(k, hk.data(), 49 * C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bb, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hg.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(be, hbe.data(), C * 4, hipMemcpyHostToDevice));
  float *yref = dalloc<float>(nc), *conv = dalloc<float>(nc), *yf = dalloc<float>(nc + 64);
  bf16_t* yb = dalloc<bf16_t>(n);
  hipLaunchKernelGGL(ref_dwconv_ln, dim3((unsigned)((Bc * Hc * W + 63) / 64)), dim3(64), 0, 0, x, Bc, Hc, W, C, k, bb, g, be, conv, yref);
  CK(hipDeviceSynchronize());
  std::vector<float> href(nc), hgot(nc + 64);
  CK(hipMemcpy(href.data(), yref, nc * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int HL : {16, 5, 64}) {
    CK(hipMemset(yf, 0xff, (nc + 64) * 4));
    if (launch<float, C>(x, Bc, Hc, W, HL, k, bb, g, be, yf) != CN_OK) return 1;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hgot.data(), yf, (nc + 64) * 4, hipMemcpyDeviceToHost));
    double max_err = 0;
    size_t n_bad = 0, touched = 0;
    for (size_t i = 0; i < nc; ++i) {
      const double e = fabs((double)hgot[i] - href[i]);
      if (!(e <= 2e-5 + 2e-5 * fabs(href[i]))) ++n_bad;
      if (e == e) max_err = std::max(max_err, e);
    }
    for (size_t i = nc; i < nc + 64; ++i) {
      unsigned u;
      memcpy(&u, &hgot[i], 4);
      touched += u != 0xffffffffu;
    }
    printf("  check HL = %2d  max|err| %.3e  out-of-tol %zu  beyond-end touched %zu  %s\n", HL, max_err, n_bad, touched,
           (n_bad == 0 && touched == 0) ? "OK" : "FAIL");
    bad += (n_bad != 0 || touched != 0);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<int> hls = HLarg > 0 ? std::vector<int>{HLarg} : std::vector<int>{H, (H + 1) / 2, (H + 3) / 4, (H + 7) / 8, 16, 8};
  for (int HL : hls) {
    std::vector<float> ts;
    for (int round = 0; round < 5; ++round) {
      launch<bf16_t, C>(x, batch, H, W, HL, k, bb, g, be, yb);
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) launch<bf16_t, C>(x, batch, H, W, HL, k, bb, g, be, yb);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1000.0f / iters);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)n * 6.0;
    printf("  time  slide HL = %3d (%4d blocks)  median %8.1f us  min %8.1f us  -> %6.2f TB/s algorithmic, %6.1f GFMA/s\n", HL,
           batch * ((H + HL - 1) / HL), ts[2], ts[0], bytes / ts[2] * 1e-6, (double)n * 49 / ts[2] * 1e-3);
  }
  return bad;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 96;
  const int batch = argc > 2 ? atoi(argv[2]) : 64;
  const int iters = argc > 3 ? atoi(argv[3]) : 10;
  const int HL = argc > 4 ? atoi(argv[4]) : 0;
  if (C == 96) return run<96>(batch, iters, HL);
  if (C == 192) return run<192>(batch, iters, HL);
  if (C == 384) return run<384>(batch, iters, HL);
  if (C == 768) return run<768>(batch, iters, HL);
  fprintf(stderr, "C must be 96, 192, 384 or 768\n");
  return 2;
}
