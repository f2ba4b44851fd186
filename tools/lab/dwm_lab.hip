// Kernel lab for the matrix-core depthwise conv + LayerNorm (csrc/dwconv_mfma.h): check against a naive GPU reference (fp16 input, weights
// rounded to fp16 like the kernel's Toeplitz fragments, fp32 arithmetic) and timing at the pipeline's shapes (B = 64, 10 s clips).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I conette-audio-captioning_amd/csrc -I tools/lab tools/lab/dwm_lab.hip -o tools/lab/dwm_lab
//   tools/lab/dwm_lab <C: 96|192> [batch] [iters]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "dwconv_mfma.h"

void cn_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

static unsigned g_seed = 12345;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffff) / 32768.0f - 1.0f; }

__global__ void ref_dwconv_ln(const half_t* x, int B, int H, int W, int C, const float* dw_w, const float* dw_b, const float* ln_w,
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
        a = fmaf((float)x[(((long)b * H + hh) * W + ww) * C + c], (float)(half_t)dw_w[(i * 7 + j) * C + c], a);
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
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

struct Variant { std::string name; int (*runf)(const half_t*, int, int, int, const half_t*, const float*, const float*, const float*, float*); int (*runb)(const half_t*, int, int, int, const half_t*, const float*, const float*, const float*, bf16_t*); };
#define VA(C, TH, NW, NP, A) Variant{ (A ? "ABL" #A " mfma<" : "mfma<") + std::string(#C ",TH" #TH ",NW" #NW ",NPASS" #NP ">"), \
  [](const half_t* x, int B, int H, int W, const half_t* fr, const float* b, const float* g, const float* be, float* y) { return cn_launch_dwconv_mfma<float, C, TH, 16, NW, NP, A>(x, B, H, W, fr, b, g, be, y, 0); }, \
  [](const half_t* x, int B, int H, int W, const half_t* fr, const float* b, const float* g, const float* be, bf16_t* y) { return cn_launch_dwconv_mfma<bf16_t, C, TH, 16, NW, NP, A>(x, B, H, W, fr, b, g, be, y, 0); } }
#define V(C, TH, NW, NP) VA(C, TH, NW, NP, 0)

template <int C> static std::vector<Variant> variants();
template <> std::vector<Variant> variants<96>() { return {V(96, 8, 6, 2), V(96, 4, 3, 2), VA(96, 4, 3, 2, 1), VA(96, 4, 3, 2, 2), VA(96, 4, 3, 2, 4), VA(96, 4, 3, 2, 8), VA(96, 4, 3, 2, 9), VA(96, 4, 3, 2, 11), VA(96, 4, 3, 2, 15)}; }
template <> std::vector<Variant> variants<192>() { return {V(192, 4, 6, 2), V(192, 8, 12, 2), V(192, 4, 12, 1), V(192, 8, 6, 4), V(192, 4, 3, 4)}; }

template <int C> static int run(int batch, int iters) {
  const int H = C == 96 ? 252 : 126, W = 5376 / C;
  const int Bc = 2, Hc = 37;
  const size_t n = (size_t)batch * H * W * C, nc = (size_t)Bc * Hc * W * C;
  printf("== C = %d, B = %d, H = %d, W = %d (%.1f M outputs); check B = %d, H = %d\n", C, batch, H, W, n * 1e-6, Bc, Hc);
  std::vector<half_t> hx(n + 16 * C);
  std::vector<float> hk(49 * C), hb(C), hg(C), hbe(C);
  for (auto& v : hx) v = (half_t)frand();
  for (auto& v : hk) v = frand() * 0.3f;
  for (auto& v : hb) v = frand() * 0.5f;
  for (auto& v : hg) v = 1.0f + 0.3f * frand();
  for (auto& v : hbe) v = 0.2f * frand();
  half_t* xall = dalloc<half_t>(hx.size());
  half_t* x = xall + 8 * C;
  float *k = dalloc<float>(49 * C), *bb = dalloc<float>(C), *g = dalloc<float>(C), *be = dalloc<float>(C);
  CK(hipMemcpy(xall, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(k, hk.data(), 49 * C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bb, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hg.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(be, hbe.data(), C * 4, hipMemcpyHostToDevice));
  half_t* fr = dalloc<half_t>(cn_dw_toeplitz_bytes(C) / 2);
  hipLaunchKernelGGL(pk_dw_toeplitz, dim3(((C / 16) * 21 * 64 + 255) / 256), dim3(256), 0, 0, k, C, fr);
  float *yref = dalloc<float>(nc), *conv = dalloc<float>(nc), *yf = dalloc<float>(nc + 64);
  bf16_t* yb = dalloc<bf16_t>(n);
  hipLaunchKernelGGL(ref_dwconv_ln, dim3((unsigned)((Bc * Hc * W + 63) / 64)), dim3(64), 0, 0, x, Bc, Hc, W, C, k, bb, g, be, conv, yref);
  CK(hipDeviceSynchronize());
  std::vector<float> href(nc), hgot(nc + 64);
  CK(hipMemcpy(href.data(), yref, nc * 4, hipMemcpyDeviceToHost));
  auto vs = variants<C>();
  int bad = 0;
  for (auto& v : vs) {
    CK(hipMemset(yf, 0xff, (nc + 64) * 4));
    if (v.runf(x, Bc, Hc, W, fr, bb, g, be, yf) != CN_OK) return 1;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hgot.data(), yf, (nc + 64) * 4, hipMemcpyDeviceToHost));
    double max_err = 0; size_t n_bad = 0, touched = 0;
    for (size_t i = 0; i < nc; ++i) {
      const double e = fabs((double)hgot[i] - href[i]);
      if (!(e <= 5e-5 + 5e-5 * fabs(href[i]))) ++n_bad;
      if (e == e) max_err = std::max(max_err, e);
    }
    for (size_t i = nc; i < nc + 64; ++i) { unsigned u; memcpy(&u, &hgot[i], 4); touched += u != 0xffffffffu; }
    if (v.name.rfind("ABL", 0) == 0) continue;
    printf("  check %-28s max|err| %.3e  out-of-tol %zu  beyond-end touched %zu  %s\n", v.name.c_str(), max_err, n_bad, touched, (n_bad == 0 && touched == 0) ? "OK" : "FAIL");
    bad += (n_bad != 0 || touched != 0);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vs) {
    std::vector<float> ts;
    for (int round = 0; round < 5; ++round) {
      v.runb(x, batch, H, W, fr, bb, g, be, yb);
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) v.runb(x, batch, H, W, fr, bb, g, be, yb);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1000.0f / iters);
    }
    std::sort(ts.begin(), ts.end());
    printf("  time  %-28s median %8.1f us  min %8.1f us  -> %6.2f TB/s algorithmic (4 C bytes per position), %6.1f GFMA/s\n", v.name.c_str(), ts[2], ts[0],
           (double)n * 4.0 / ts[2] * 1e-6, (double)n * 49 / ts[2] * 1e-3);
  }
  return bad;
}
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 96, batch = argc > 2 ? atoi(argv[2]) : 64, iters = argc > 3 ? atoi(argv[3]) : 10;
  if (C == 96) return run<96>(batch, iters);
  if (C == 192) return run<192>(batch, iters);
  fprintf(stderr, "C must be 96 or 192\n");
  return 2;
}
