// Context creation (weight packing), error reporting and the resampler of the C ABI.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "ctx.h"
#include <algorithm>
#include "mlp_rc2.h"
#include "mlp_rs.h"
#include "mlp_rs16.h"
#include "mlp_sp.h"
#include "down_fused.h"

static thread_local char g_err[512] = "";

void cn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* conette_last_error(void) { return g_err; }

// ---- runtime state (C++ members kept out of the POD context) -------------------------------------
struct CnProfRec {
  int cls;
  hipEvent_t start, stop;
};
struct CnRuntime {
  std::vector<CnProfRec> recs;       // recorded launches (in order)
  std::vector<hipEvent_t> free_ev;   // recycled events
  int open_cls = -1;
  hipEvent_t open_start = nullptr;
};

static hipEvent_t rt_event(CnRuntime* rt) {
  if (!rt->free_ev.empty()) {
    hipEvent_t e = rt->free_ev.back();
    rt->free_ev.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void cn_prof_begin(conette_ctx* ctx, int cls, hipStream_t s) {
  CnRuntime* rt = ctx->rt;
  rt->open_cls = cls;
  rt->open_start = rt_event(rt);
  (void)hipEventRecord(rt->open_start, s);
}
void cn_prof_end(conette_ctx* ctx, int cls, hipStream_t s) {
  CnRuntime* rt = ctx->rt;
  if (rt->open_cls != cls || !rt->open_start) return;
  hipEvent_t stop = rt_event(rt);
  (void)hipEventRecord(stop, s);
  rt->recs.push_back(CnProfRec{cls, rt->open_start, stop});
  rt->open_cls = -1;
  rt->open_start = nullptr;
}

extern "C" int conette_profile_enable(conette_ctx* ctx, uint32_t class_mask) {
  if (!ctx) return CN_ERR_ARG;
  ctx->prof_mask = class_mask;
  return CN_OK;
}

extern "C" int conette_profile_read(conette_ctx* ctx, float* ms, int32_t* counts) {
  if (!ctx || !ms || !counts) {
    cn_set_error("profile_read: bad argument");
    return CN_ERR_ARG;
  }
  CnRuntime* rt = ctx->rt;
  for (auto& r : rt->recs) {
    CN_HIP(hipEventSynchronize(r.stop));
    float t = 0.f;
    CN_HIP(hipEventElapsedTime(&t, r.start, r.stop));
    if (r.cls >= 0 && r.cls < CONETTE_PROF_NCLASS) {
      ms[r.cls] += t;
      counts[r.cls] += 1;
    }
    rt->free_ev.push_back(r.start);
    rt->free_ev.push_back(r.stop);
  }
  rt->recs.clear();
  return CN_OK;
}
extern "C" int conette_abi_version(void) { return CONETTE_ABI_VERSION; }

// ---- packing kernels --------------------------------------------------------------------------
template <typename T>
__global__ void pk_copy(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = cn_from_f32<T>(src[i]);
}
// (C, 1, KH, KW) -> [KH*KW][C]
__global__ void pk_taps_last(const float* __restrict__ src, float* __restrict__ dst, int C, int taps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * taps) dst[(i % taps) * C + i / taps] = src[i];
}
// depthwise (C, 1, 7, 7) -> [42][C] fp16 pairs of consecutive kernel rows (encoder.hip, CN_DW_DOT2): slot a * 7 + j = (k[2a][j], k[2a+1][j]),
// slot 21 + a * 7 + j = (k[2a+1][j], k[2a+2][j]), a < 3; the low half is the first of the pair
__global__ void pk_dw_pairs(const float* __restrict__ src, unsigned* __restrict__ dst, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 42) return;
  const int c = i % C, slot = i / C;
  const int odd = slot >= 21, a = (slot % 21) / 7, j = slot % 7;
  const int kh = 2 * a + odd;
  const half_t lo = (half_t)src[c * 49 + kh * 7 + j], hi = (half_t)src[c * 49 + (kh + 1) * 7 + j];
  dst[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}
// (N, C, 2, 2) -> [N][(kh*2+kw)*C + c]
template <typename T>
__global__ void pk_down(const float* __restrict__ src, T* __restrict__ dst, int N, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * C * 4) return;
  const int kk = (int)(i % 4);
  const size_t nc = i / 4;
  const int c = (int)(nc % C);
  const size_t n = nc / C;
  dst[(n * 4 + kk) * C + c] = cn_from_f32<T>(src[i]);
}
// decoder block stream (dec_block.h): 16-byte unit u = ((((m*4 + w)*4 + qd)*4 + a)*2 + kk)*64 + lane holds
// W_m[64w + 16a + (lane & 15)][32(2qd + kk) + 8(lane >> 4) .. +8]
struct PkBlockSrc { const float* W[6]; };
template <typename HT>
__global__ void pk_block_stream(PkBlockSrc src, HT* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= 6 * 4 * 4 * 4 * 2 * 64) return;
  const int lane = u & 63, kk = (u >> 6) & 1, a = (u >> 7) & 3, qd = (u >> 9) & 3, w = (u >> 11) & 3, m = u >> 13;
  const float* s = src.W[m] + (size_t)(64 * w + 16 * a + (lane & 15)) * 256 + 32 * (2 * qd + kk) + 8 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = cn_from_f32<HT>(s[i]);
}
// decoder FFN stream (dec_ffn.h): unit u = (((((c*2 + t)*4 + w)*4 + qd)*4 + a)*2 + kk)*64 + lane;
// t = 0: W1[c*256 + 64w + 16a + (lane & 15)][32(2qd + kk) + 8(lane >> 4) .. +8]        (W1: [dff][256])
// t = 1: W2[64w + 16a + (lane & 15)][c*256 + 32(2qd + kk) + 8(lane >> 4) .. +8]        (W2: [256][dff])
template <typename HT>
__global__ void pk_ffn_stream(const float* __restrict__ W1, const float* __restrict__ W2, int dff, HT* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (dff / 256) * 2 * 8192) return;
  const int lane = u & 63, kk = (u >> 6) & 1, a = (u >> 7) & 3, qd = (u >> 9) & 3, w = (u >> 11) & 3, t = (u >> 13) & 1,
            c = u >> 14;
  const int r = 64 * w + 16 * a + (lane & 15), k = 32 * (2 * qd + kk) + 8 * (lane >> 4);
  const float* s = t == 0 ? W1 + (size_t)(c * 256 + r) * 256 + k : W2 + (size_t)r * dff + c * 256 + k;
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = cn_from_f32<HT>(s[i]);
}
// the same two streams for the exact precision (sp16): every matrix / tile twice, its lo halves then its hi halves
// (hi = rn16(x), lo = rn16(x - hi): common.h), each pass in the fp16 fragment order above.
// block stream: pass p = 2 m + part at unit p * 8192 + (w, qd, a, kk, lane); FFN stream: pass (c * 2 + t) * 2 + part
__global__ void pk_block_stream_sp(PkBlockSrc src, half_t* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= 12 * 8192) return;
  const int lane = u & 63, kk = (u >> 6) & 1, a = (u >> 7) & 3, qd = (u >> 9) & 3, w = (u >> 11) & 3, part = (u >> 13) & 1, m = u >> 14;
  const float* s = src.W[m] + (size_t)(64 * w + 16 * a + (lane & 15)) * 256 + 32 * (2 * qd + kk) + 8 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned b = cn_sp16_bits(s[i]);
    dst[(size_t)u * 8 + i] = __builtin_bit_cast(half_t, (unsigned short)(part ? (b & 0xffffu) : (b >> 16)));
  }
}
__global__ void pk_ffn_stream_sp(const float* __restrict__ W1, const float* __restrict__ W2, int dff, half_t* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (dff / 256) * 4 * 8192) return;
  const int lane = u & 63, kk = (u >> 6) & 1, a = (u >> 7) & 3, qd = (u >> 9) & 3, w = (u >> 11) & 3, part = (u >> 13) & 1,
            t = (u >> 14) & 1, c = u >> 15;
  const int r = 64 * w + 16 * a + (lane & 15), k = 32 * (2 * qd + kk) + 8 * (lane >> 4);
  const float* s = t == 0 ? W1 + (size_t)(c * 256 + r) * 256 + k : W2 + (size_t)r * dff + c * 256 + k;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned b = cn_sp16_bits(s[i]);
    dst[(size_t)u * 8 + i] = __builtin_bit_cast(half_t, (unsigned short)(part ? (b & 0xffffu) : (b >> 16)));
  }
}
__global__ void pk_bn(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift,
                      int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float sc = w[i] / sqrtf(var[i] + 1e-5f);
    scale[i] = sc;
    shift[i] = b[i] - mean[i] * sc;
  }
}

// ---- host-side builder --------------------------------------------------------------------------
struct Builder {
  conette_ctx* ctx;
  std::unordered_map<std::string, int> index;
  const void* const* tensors;
  const int64_t* numel;
  int err = CN_OK;

  const float* find(const std::string& name, int64_t expect) {
    auto it = index.find(name);
    if (it == index.end()) {
      if (err == CN_OK) cn_set_error("create: missing tensor '%s'", name.c_str());
      err = CN_ERR_WEIGHT;
      return nullptr;
    }
    if (expect >= 0 && numel[it->second] != expect) {
      if (err == CN_OK)
        cn_set_error("create: tensor '%s' has %lld elements, expected %lld", name.c_str(),
                     (long long)numel[it->second], (long long)expect);
      err = CN_ERR_WEIGHT;
      return nullptr;
    }
    return (const float*)tensors[it->second];
  }
  int64_t count(const std::string& name) {
    auto it = index.find(name);
    return it == index.end() ? -1 : numel[it->second];
  }
  void* alloc(size_t bytes) {
    size_t off = cn_align(ctx->arena_used);
    if (off + bytes > ctx->arena_bytes) {
      if (err == CN_OK) cn_set_error("create: arena overflow");
      err = CN_ERR_WEIGHT;
      return ctx->arena;  // keep going, error reported at the end
    }
    ctx->arena_used = off + bytes;
    return ctx->arena + off;
  }
  // fp32 copy
  const float* f32(const std::string& name, int64_t n) {
    const float* src = find(name, n);
    if (n < 0) n = 0;
    float* dst = (float*)alloc((size_t)n * 4);
    if (src) hipLaunchKernelGGL((pk_copy<float>), dim3(256), dim3(256), 0, 0, src, dst, (size_t)n);
    return dst;
  }
  // operand-type copy of a sub-range [first, first + n) of a tensor
  const void* operand(const std::string& name, int64_t total, int64_t first, int64_t n) {
    const float* src = find(name, total);
    void* dst = alloc((size_t)n * ctx->esize);
    if (src) {
      copy_operand(src + first, dst, (size_t)n);
    }
    return dst;
  }
  // fp32 -> the context's operand type (bf16 | fp16 | sp16 = fp16 hi/lo pairs | fp32)
  void copy_operand(const float* src, void* dst, size_t n) {
    if (ctx->esize == 2) CN_H16_CALL(ctx, hipLaunchKernelGGL((pk_copy<HT>), dim3(256), dim3(256), 0, 0, src, (HT*)dst, n));
    else if (ctx->sp16) hipLaunchKernelGGL((pk_copy<sp16_t>), dim3(256), dim3(256), 0, 0, src, (sp16_t*)dst, n);
    else hipLaunchKernelGGL((pk_copy<float>), dim3(256), dim3(256), 0, 0, src, (float*)dst, n);
  }
};

extern "C" int conette_create(const conette_config* cfg, int32_t n_tensors, const char* const* names,
                              const void* const* tensors, const int64_t* numel, conette_ctx** out) {
  if (!cfg || !names || !tensors || !numel || !out || n_tensors <= 0) {
    cn_set_error("create: bad argument");
    return CN_ERR_ARG;
  }
  if (cfg->precision != CONETTE_PREC_F32 && cfg->precision != CONETTE_PREC_BF16 && cfg->precision != CONETTE_PREC_F16X2 &&
      cfg->precision != CONETTE_PREC_F16) {
    cn_set_error(cfg->precision == 3 ? "create: precision 3 (the experimental fp8 mode of ABI 2) was withdrawn in ABI 3: e4m3 operands cost the "
                                       "frame embeddings 4 %% whatever the scaling (profiles/r06_notes.md section 3)"
                                     : "create: unknown precision %d", cfg->precision);
    return CN_ERR_ARG;
  }
  if (cfg->n_layers < 1 || cfg->n_layers > CN_MAX_LAYERS || cfg->d_model != 256 || cfg->nhead != 8 ||
      cfg->d_ff % 32 != 0 || cfg->vocab_size < 4) {
    cn_set_error("create: unsupported decoder geometry (d_model=%d nhead=%d layers=%d d_ff=%d)", cfg->d_model,
                 cfg->nhead, cfg->n_layers, cfg->d_ff);
    return CN_ERR_ARG;
  }
  conette_ctx* ctx = new conette_ctx();
  memset(ctx, 0, sizeof(*ctx));
  ctx->cfg = *cfg;
  ctx->rt = new CnRuntime();
  ctx->esize = (cfg->precision == CONETTE_PREC_BF16 || cfg->precision == CONETTE_PREC_F16) ? 2 : 4;
  ctx->f16 = cfg->precision == CONETTE_PREC_F16 ? 1 : 0;
  ctx->sp16 = cfg->precision == CONETTE_PREC_F16X2 ? 1 : 0;
  {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n_cu <= 0)
      n_cu = 256;
    ctx->n_cu = n_cu;
  }
  size_t total = (1 << 20) + (size_t)CN_N_BINS * CN_N_MELS * 4;  // (+ the band-compact mel matrix, at most a dense copy)
  for (int i = 0; i < n_tensors; ++i) total += cn_align((size_t)numel[i] * 4) + 256;
  if (cfg->precision == CONETTE_PREC_F16X2) {
    total += 8u << 20;   // the fp16 hi / lo streams of the 6 fused ConvNeXt blocks (~3.4 MB)
    // + per decoder layer the hi / lo fragment streams of dec_block.h (12 passes of 128 KB) and dec_ffn.h (4 per hidden chunk)
    total += (size_t)cfg->n_layers * ((size_t)12 * 131072 + (size_t)(cfg->d_ff / 256 + 1) * 4 * 131072 + 4096);
  }
  ctx->arena_bytes = total;
  hipError_t e = hipMalloc((void**)&ctx->arena, total);
  if (e != hipSuccess) {
    cn_set_error("create: hipMalloc(%zu) -> %s", total, hipGetErrorString(e));
    delete ctx->rt;
    delete ctx;
    return CN_ERR_HIP;
  }
  Builder B;
  B.ctx = ctx;
  B.tensors = tensors;
  B.numel = numel;
  for (int i = 0; i < n_tensors; ++i) B.index[names[i]] = i;
  const int d = cfg->d_model, V = cfg->vocab_size, dff = cfg->d_ff;
  const std::string E = "preprocessor.encoder.";
  // A BaselinePLM-layout checkpoint (pl_modules/baseline.py:84-140: FrameIdentEncoder + projection + decoder) has no audio
  // encoder: its input is precomputed frame embeddings.  A tensor list without any "preprocessor.encoder." entry creates a
  // DECODER-ONLY context -- conette_decode / conette_greedy / conette_forcing work, conette_encode and
  // conette_frontend_logmel return CN_ERR_ARG.
  bool has_encoder = false;
  for (int i = 0; i < n_tensors && !has_encoder; ++i) has_encoder = strncmp(names[i], E.c_str(), E.size()) == 0;
  ctx->no_encoder = has_encoder ? 0 : 1;

  if (has_encoder) {
  // ---- frontend tables ----
  {
    const float* cr = B.find(E + "spectrogram_extractor.stft.conv_real.weight", (int64_t)CN_N_BINS * CN_N_FFT);
    float* win = (float*)B.alloc(CN_N_FFT * 4);
    if (cr) hipMemcpy(win, cr, CN_N_FFT * 4, hipMemcpyDeviceToDevice);  // row k = 0: cos(0) * window
    ctx->window = win;
    // The frontend replaces the checkpoint's DFT-as-conv1d tensors by an FFT with the window taken from row 0.  That is
    // only the same transform if the tensors ARE the windowed DFT (torchlibrosa.stft.STFT: W[k][n] = win[n] e^{-2 pi i k n / N}):
    // spot-check rows of both tensors against the closed form and refuse anything else instead of silently diverging.
    const float* ci = B.find(E + "spectrogram_extractor.stft.conv_imag.weight", (int64_t)CN_N_BINS * CN_N_FFT);
    if (cr && ci && B.err == CN_OK) {
      const int rows[5] = {1, 7, 100, 333, 512};
      std::vector<float> hw(CN_N_FFT), hr(CN_N_FFT), hi(CN_N_FFT);
      hipMemcpy(hw.data(), cr, CN_N_FFT * 4, hipMemcpyDeviceToHost);
      float wmax = 0.f;
      for (float v : hw) wmax = fmaxf(wmax, fabsf(v));
      double worst = 0.0;
      for (int k : rows) {
        hipMemcpy(hr.data(), cr + (size_t)k * CN_N_FFT, CN_N_FFT * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hi.data(), ci + (size_t)k * CN_N_FFT, CN_N_FFT * 4, hipMemcpyDeviceToHost);
        for (int n = 0; n < CN_N_FFT; n += 13) {
          const double a = -2.0 * M_PI * (double)((long)k * n % CN_N_FFT) / CN_N_FFT;
          worst = fmax(worst, fabs(hr[n] - hw[n] * cos(a)));
          worst = fmax(worst, fabs(hi[n] - hw[n] * sin(a)));
        }
      }
      if (!(wmax > 0.f) || worst > 1e-4 * wmax) {
        cn_set_error("create: spectrogram_extractor.stft.conv_real / conv_imag are not a windowed DFT (max deviation %.3g of "
                     "window peak %.3g): the FFT frontend cannot reproduce this checkpoint", worst, (double)wmax);
        B.err = CN_ERR_WEIGHT;
      }
    }
    std::vector<float2> t512(512), t1024(513);
    for (int j = 0; j < 512; ++j) {
      const double a = -2.0 * M_PI * j / 512.0;
      t512[j] = float2{(float)cos(a), (float)sin(a)};
    }
    for (int j = 0; j <= 512; ++j) {
      const double a = -2.0 * M_PI * j / 1024.0;
      t1024[j] = float2{(float)cos(a), (float)sin(a)};
    }
    float2* d512 = (float2*)B.alloc(512 * 8);
    float2* d1024 = (float2*)B.alloc(513 * 8);
    hipMemcpy(d512, t512.data(), 512 * 8, hipMemcpyHostToDevice);
    hipMemcpy(d1024, t1024.data(), 513 * 8, hipMemcpyHostToDevice);
    ctx->tw512 = d512;
    ctx->tw1024 = d1024;
    const float* mw = B.find(E + "logmel_extractor.melW", (int64_t)CN_N_BINS * CN_N_MELS);
    std::vector<int> band(2 * CN_N_MELS, 0);
    if (mw) {
      std::vector<float> h((size_t)CN_N_BINS * CN_N_MELS);
      hipMemcpy(h.data(), mw, h.size() * 4, hipMemcpyDeviceToHost);
      for (int m = 0; m < CN_N_MELS; ++m) {
        int lo = CN_N_BINS, hi = 0;
        for (int k = 0; k < CN_N_BINS; ++k)
          if (h[(size_t)k * CN_N_MELS + m] != 0.0f) {
            lo = k < lo ? k : lo;
            hi = k + 1;
          }
        if (hi == 0) lo = 0;
        band[2 * m] = lo;
        band[2 * m + 1] = hi;
      }
    }
    {  // band-compact copy of melW for the kernel: row i holds, for every mel bin m, the weight of FFT bin lo(m) + i
      int maxw = 1;
      for (int m = 0; m < CN_N_MELS; ++m) maxw = std::max(maxw, band[2 * m + 1] - band[2 * m]);
      std::vector<float> mc((size_t)maxw * CN_N_MELS, 0.f);
      if (mw) {
        std::vector<float> h((size_t)CN_N_BINS * CN_N_MELS);
        hipMemcpy(h.data(), mw, h.size() * 4, hipMemcpyDeviceToHost);
        for (int m = 0; m < CN_N_MELS; ++m)
          for (int k = band[2 * m]; k < band[2 * m + 1]; ++k) mc[(size_t)(k - band[2 * m]) * CN_N_MELS + m] = h[(size_t)k * CN_N_MELS + m];
      }
      float* dmc = (float*)B.alloc(mc.size() * 4);
      hipMemcpy(dmc, mc.data(), mc.size() * 4, hipMemcpyHostToDevice);
      ctx->melC = dmc;
      ctx->mel_rows = maxw;
    }
    int* dband = (int*)B.alloc(band.size() * 4);
    hipMemcpy(dband, band.data(), band.size() * 4, hipMemcpyHostToDevice);
    ctx->band = dband;
    float* sc = (float*)B.alloc(CN_N_MELS * 4);
    float* sh = (float*)B.alloc(CN_N_MELS * 4);
    const float *bw = B.find(E + "bn0.weight", CN_N_MELS), *bb = B.find(E + "bn0.bias", CN_N_MELS),
                *bm = B.find(E + "bn0.running_mean", CN_N_MELS), *bv = B.find(E + "bn0.running_var", CN_N_MELS);
    if (bw && bb && bm && bv) hipLaunchKernelGGL(pk_bn, dim3(1), dim3(256), 0, 0, bw, bb, bm, bv, sc, sh, CN_N_MELS);
    ctx->bn_scale = sc;
    ctx->bn_shift = sh;
  }
  // ---- stem ----
  {
    const float* w = B.find(E + "downsample_layers.0.0.weight", 96 * 16);
    float* dst = (float*)B.alloc(96 * 16 * 4);
    if (w) hipLaunchKernelGGL(pk_taps_last, dim3(6), dim3(256), 0, 0, w, dst, 96, 16);
    ctx->stem_w = dst;
    ctx->stem_b = B.f32(E + "downsample_layers.0.0.bias", 96);
    ctx->stem_ln_w = B.f32(E + "downsample_layers.0.1.weight", 96);
    ctx->stem_ln_b = B.f32(E + "downsample_layers.0.1.bias", 96);
  }
  // ---- downsample layers 1..3 ----
  for (int i = 0; i < 3; ++i) {
    const int C = CN_DIMS[i], C2 = CN_DIMS[i + 1];
    const std::string p = E + "downsample_layers." + std::to_string(i + 1) + ".";
    ctx->down[i].ln_w = B.f32(p + "0.weight", C);
    ctx->down[i].ln_b = B.f32(p + "0.bias", C);
    const float* w = B.find(p + "1.weight", (int64_t)C2 * C * 4);
    void* dst = B.alloc((size_t)C2 * C * 4 * ctx->esize);
    if (w) {
      const unsigned blocks = (unsigned)(((size_t)C2 * C * 4 + 255) / 256);
      if (ctx->esize == 2) CN_H16_CALL(ctx, hipLaunchKernelGGL((pk_down<HT>), dim3(blocks), dim3(256), 0, 0, w, (HT*)dst, C2, C));
      else if (ctx->sp16) hipLaunchKernelGGL((pk_down<sp16_t>), dim3(blocks), dim3(256), 0, 0, w, (sp16_t*)dst, C2, C);
      else hipLaunchKernelGGL((pk_down<float>), dim3(blocks), dim3(256), 0, 0, w, (float*)dst, C2, C);
    }
    ctx->down[i].w = dst;
    ctx->down[i].bias = B.f32(p + "1.bias", C2);
    ctx->down[i].fused = nullptr;
    if (i <= 1 && ctx->esize == 2) {  // fused LayerNorm + patch GEMM (down_fused.h): LN affine folded into the packed weights
      const float *g = B.find(p + "0.weight", C), *bt = B.find(p + "0.bias", C), *cb = B.find(p + "1.bias", C2);
      void* fs = B.alloc(i == 0 ? DownGeom<96>::TOTAL_BYTES : DownGeom<192>::TOTAL_BYTES);
      if (w && g && bt && cb) {
        const int n = (4 * C / 16) * (C2 / 32) * 64;
        CN_H16_CALL(ctx, hipLaunchKernelGGL(pk_down_fused<HT>, dim3((n + 255) / 256), dim3(256), 0, 0, w, g, bt, cb, C2, C, (HT*)fs));
        ctx->down[i].fused = fs;
      }
    }
  }
  // ---- ConvNeXt blocks ----
  {
    int blk = 0;
    for (int s = 0; s < 4; ++s) {
      const int C = CN_DIMS[s];
      for (int b = 0; b < CN_DEPTHS[s]; ++b, ++blk) {
        const std::string p = E + "stages." + std::to_string(s) + "." + std::to_string(b) + ".";
        CnBlockW& bw = ctx->blocks[blk];
        // old checkpoints call the layer scale "gamma" (convnext.py:76-102)
        const std::string sname = B.count(p + "scale_layer") >= 0 ? p + "scale_layer" : p + "gamma";
        bw.scale = B.f32(sname, C);
        const float* w = B.find(p + "dwconv.weight", (int64_t)C * 49);
        float* dst = (float*)B.alloc((size_t)C * 49 * 4);
        if (w) hipLaunchKernelGGL(pk_taps_last, dim3((C * 49 + 255) / 256), dim3(256), 0, 0, w, dst, C, 49);
        bw.dw_w = dst;
        bw.dw_wp = nullptr;
        if (ctx->esize == 2) {   // the precisions whose encoder stream is fp16 (encoder.hip: XT = half_t)
          unsigned* wp = (unsigned*)B.alloc((size_t)C * 42 * 4);
          if (w) hipLaunchKernelGGL(pk_dw_pairs, dim3((C * 42 + 255) / 256), dim3(256), 0, 0, w, wp, C);
          bw.dw_wp = wp;
        }
        bw.dw_b = B.f32(p + "dwconv.bias", C);
        bw.ln_w = B.f32(p + "norm.weight", C);
        bw.ln_b = B.f32(p + "norm.bias", C);
        bw.w1 = B.operand(p + "pwconv1.weight", (int64_t)4 * C * C, 0, (int64_t)4 * C * C);
        bw.b1 = B.f32(p + "pwconv1.bias", 4 * C);
        bw.w2 = B.operand(p + "pwconv2.weight", (int64_t)4 * C * C, 0, (int64_t)4 * C * C);
        bw.b2 = B.f32(p + "pwconv2.bias", C);
        bw.mlp_stream = nullptr;
        bw.mlp_sp = nullptr;
        if (ctx->sp16 && C <= 192) {
          void* ms = B.alloc(C == 96 ? SpGeom<96>::TOTAL_BYTES : SpGeom<192>::TOTAL_BYTES);
          const float* w1 = B.find(p + "pwconv1.weight", (int64_t)4 * C * C);
          const float* w2 = B.find(p + "pwconv2.weight", (int64_t)4 * C * C);
          if (w1 && w2 && bw.b1 && bw.b2 && bw.scale && cn_pack_mlp_sp(w1, bw.b1, w2, bw.b2, bw.scale, C, ms, 0) == CN_OK) bw.mlp_sp = ms;
        }
        if (ctx->esize == 2 && C <= 384) {
          // Rc2Geom<C, 1>::TOTAL_BYTES; the role-split streams of C = 384 are [C/8][C/8 + 1] KB (mlp_rs.h) or [C/8][C/8 + 2] KB
          // (mlp_rs16.h) + C fp32
          const size_t bytes = std::max((size_t)(C / 8) * (C / 8 + 1) * 1024 + (size_t)(C / 32) * 1024, (size_t)(C / 8) * (C / 8 + 2) * 1024 + (size_t)C * 4);
          void* ms = B.alloc(bytes);
          const float* w1 = B.find(p + "pwconv1.weight", (int64_t)4 * C * C);
          const float* w2 = B.find(p + "pwconv2.weight", (int64_t)4 * C * C);
          if (w1 && w2 && bw.b1 && bw.b2 && bw.scale) {
            const int units = ((C / 8) * (C / 8 + 1) + C / 32) * 64;  // one thread per 16-byte fragment piece, the bias fragments included
            // C = 384 runs the role-split kernel (mlp_rs.h): same fragments, entry e = [W1 of chunk e | W2 of chunk e - 2]
#ifndef CN_NO_RS
            // (with the 16-bit residual stream of the bf16 / f16 precisions: mlp_rs16.h, the same pipeline on 16x16x32 MFMAs)
            if (C == 384 && CN_RS16) {
              const int u16 = (C / 8) * (C / 8 + 2) * 64;
              CN_H16_CALL(ctx, hipLaunchKernelGGL(pk_mlp_rs16<HT>, dim3((u16 + 255) / 256), dim3(256), 0, 0, w1, bw.b1, w2, bw.b2, bw.scale, C, (HT*)ms));
            } else if (C == 384) CN_H16_CALL(ctx, hipLaunchKernelGGL(pk_mlp_rs<HT>, dim3((units + 255) / 256), dim3(256), 0, 0, w1, bw.b1, w2, bw.b2, bw.scale, C, (HT*)ms));
            else
#endif
            CN_H16_CALL(ctx, hipLaunchKernelGGL(pk_mlp_rc2<HT>, dim3((units + 255) / 256), dim3(256), 0, 0, w1, bw.b1, w2, bw.b2, bw.scale, C, CN_RC2_NCK(C), (HT*)ms));
            bw.mlp_stream = ms;
          }
        }
      }
    }
  }
  ctx->norm_w = B.f32(E + "norm.weight", CN_FEAT);
  ctx->norm_b = B.f32(E + "norm.bias", CN_FEAT);
  ctx->head_w = B.operand(E + "head_audioset.weight", (int64_t)CN_N_TAGS * CN_FEAT, 0, (int64_t)CN_N_TAGS * CN_FEAT);
  ctx->head_b = B.f32(E + "head_audioset.bias", CN_N_TAGS);
  }  // has_encoder

  // ---- decoder ----
  ctx->proj_w = B.operand("model.projection.2.weight", (int64_t)d * CN_FEAT, 0, (int64_t)d * CN_FEAT);
  ctx->proj_b = B.f32("model.projection.2.bias", d);
  const std::string D = "model.decoder.";
  {
    char* kvw = (char*)B.alloc((size_t)cfg->n_layers * 2 * d * d * ctx->esize);
    float* kvb = (float*)B.alloc((size_t)cfg->n_layers * 2 * d * 4);
    ctx->kv_w = kvw;
    ctx->kv_b = kvb;
    for (int l = 0; l < cfg->n_layers; ++l) {
      const std::string p = D + "layers." + std::to_string(l) + ".";
      CnLayerW& lw = ctx->layers[l];
      lw.sa_in_w = B.operand(p + "self_attn.in_proj_weight", (int64_t)3 * d * d, 0, (int64_t)3 * d * d);
      lw.sa_in_b = B.f32(p + "self_attn.in_proj_bias", 3 * d);
      lw.sa_out_w = B.operand(p + "self_attn.out_proj.weight", (int64_t)d * d, 0, (int64_t)d * d);
      lw.sa_out_b = B.f32(p + "self_attn.out_proj.bias", d);
      lw.ca_q_w = B.operand(p + "multihead_attn.in_proj_weight", (int64_t)3 * d * d, 0, (int64_t)d * d);
      const float* cab = B.find(p + "multihead_attn.in_proj_bias", 3 * d);
      float* qb = (float*)B.alloc((size_t)d * 4);
      if (cab) {
        hipMemcpy(qb, cab, (size_t)d * 4, hipMemcpyDeviceToDevice);
        hipMemcpy(kvb + (size_t)l * 2 * d, cab + d, (size_t)2 * d * 4, hipMemcpyDeviceToDevice);
      }
      lw.ca_q_b = qb;
      const float* caw = B.find(p + "multihead_attn.in_proj_weight", (int64_t)3 * d * d);
      if (caw) {
        const size_t n = (size_t)2 * d * d;
        B.copy_operand(caw + (size_t)d * d, kvw + (size_t)l * n * ctx->esize, n);
      }
      lw.ca_out_w = B.operand(p + "multihead_attn.out_proj.weight", (int64_t)d * d, 0, (int64_t)d * d);
      lw.ca_out_b = B.f32(p + "multihead_attn.out_proj.bias", d);
      lw.ff1_w = B.operand(p + "linear1.weight", (int64_t)dff * d, 0, (int64_t)dff * d);
      lw.ff1_b = B.f32(p + "linear1.bias", dff);
      lw.ff2_w = B.operand(p + "linear2.weight", (int64_t)dff * d, 0, (int64_t)dff * d);
      lw.ff2_b = B.f32(p + "linear2.bias", d);
      lw.n1w = B.f32(p + "norm1.weight", d);
      lw.n1b = B.f32(p + "norm1.bias", d);
      lw.n2w = B.f32(p + "norm2.weight", d);
      lw.n2b = B.f32(p + "norm2.bias", d);
      lw.n3w = B.f32(p + "norm3.weight", d);
      lw.n3b = B.f32(p + "norm3.bias", d);
      lw.blk_w = nullptr;
      lw.blk_p = nullptr;
      lw.ffn_w = nullptr;
      if (ctx->esize == 2 && dff % 256 == 0 && dff <= 2048) {
        void* fwp = B.alloc((size_t)2 * dff * d * 2);
        const float* w1 = B.find(p + "linear1.weight", (int64_t)dff * d);
        const float* w2 = B.find(p + "linear2.weight", (int64_t)dff * d);
        if (w1 && w2) {
          const int units = (dff / 256) * 2 * 8192;
          CN_H16_CALL(ctx, hipLaunchKernelGGL(pk_ffn_stream<HT>, dim3(units / 256), dim3(256), 0, 0, w1, w2, dff, (HT*)fwp));
          lw.ffn_w = fwp;
        }
      }
      if (ctx->sp16) {  // exact precision: the same two fused kernels on hi / lo half-tiles (dec_block.h, dec_ffn.h)
        const float* ipw = B.find(p + "self_attn.in_proj_weight", (int64_t)3 * d * d);
        const float* sow = B.find(p + "self_attn.out_proj.weight", (int64_t)d * d);
        const float* cow = B.find(p + "multihead_attn.out_proj.weight", (int64_t)d * d);
        void* bw = B.alloc((size_t)12 * d * d * 2);
        if (ipw && sow && caw && cow) {
          PkBlockSrc ps;
          ps.W[0] = ipw, ps.W[1] = ipw + (size_t)d * d, ps.W[2] = ipw + (size_t)2 * d * d, ps.W[3] = sow, ps.W[4] = caw, ps.W[5] = cow;
          hipLaunchKernelGGL(pk_block_stream_sp, dim3(12 * 8192 / 256), dim3(256), 0, 0, ps, (half_t*)bw);
          lw.blk_w = bw;
        }
        if (dff % 256 == 0 && dff <= 2048) {
          void* fwp = B.alloc((size_t)4 * dff * d * 2);
          const float* w1 = B.find(p + "linear1.weight", (int64_t)dff * d);
          const float* w2 = B.find(p + "linear2.weight", (int64_t)dff * d);
          if (w1 && w2) {
            hipLaunchKernelGGL(pk_ffn_stream_sp, dim3((dff / 256) * 4 * 8192 / 256), dim3(256), 0, 0, w1, w2, dff, (half_t*)fwp);
            lw.ffn_w = fwp;
          }
        }
      }
      if (ctx->esize == 2) {
        void* bw = B.alloc((size_t)6 * d * d * 2);
        const float* ipw = B.find(p + "self_attn.in_proj_weight", (int64_t)3 * d * d);
        const float* sow = B.find(p + "self_attn.out_proj.weight", (int64_t)d * d);
        const float* cow = B.find(p + "multihead_attn.out_proj.weight", (int64_t)d * d);
        if (ipw && sow && caw && cow) {
          PkBlockSrc ps;
          ps.W[0] = ipw, ps.W[1] = ipw + (size_t)d * d, ps.W[2] = ipw + (size_t)2 * d * d, ps.W[3] = sow, ps.W[4] = caw, ps.W[5] = cow;
          CN_H16_CALL(ctx, hipLaunchKernelGGL(pk_block_stream<HT>, dim3(6 * 4 * 4 * 4 * 2 * 64 / 256), dim3(256), 0, 0, ps, (HT*)bw));
        }
        lw.blk_w = bw;
      }
      {
        float* bp = (float*)B.alloc(2560 * 4);
        const float* parts[8] = {lw.sa_in_b, lw.sa_out_b, lw.ca_q_b, lw.ca_out_b, lw.n1w, lw.n1b, lw.n2w, lw.n2b};
        size_t off = 0;
        for (int i = 0; i < 8; ++i) {
          const size_t n = i == 0 ? 768 : 256;
          if (parts[i]) hipMemcpy(bp + off, parts[i], n * 4, hipMemcpyDeviceToDevice);
          off += n;
        }
        lw.blk_p = bp;
      }
    }
  }
  ctx->emb = B.f32(D + "emb_layer.weight", (int64_t)V * d);
  {
    const int64_t n = B.count(D + "pos_encoding.pos_embedding");
    ctx->pe_len = n > 0 ? (int)(n / d) : 0;
    ctx->pe = B.f32(D + "pos_encoding.pos_embedding", n);
  }
  ctx->cls_w = B.operand(D + "classifier.weight", (int64_t)V * d, 0, (int64_t)V * d);
  ctx->cls_b = B.f32(D + "classifier.bias", V);

  hipError_t se = hipDeviceSynchronize();
  if (B.err == CN_OK && se != hipSuccess) {
    cn_set_error("create: packing failed -> %s", hipGetErrorString(se));
    B.err = CN_ERR_HIP;
  }
  if (B.err != CN_OK) {
    (void)hipFree(ctx->arena);
    delete ctx->rt;
    delete ctx;
    return B.err;
  }
  *out = ctx;
  return CN_OK;
}

void cn_decode_graphs_free(conette_ctx* ctx);

extern "C" void conette_destroy(conette_ctx* ctx) {
  if (!ctx) return;
  cn_decode_graphs_free(ctx);
  if (ctx->rt) {
    for (auto& r : ctx->rt->recs) {
      (void)hipEventDestroy(r.start);
      (void)hipEventDestroy(r.stop);
    }
    for (auto e : ctx->rt->free_ev) (void)hipEventDestroy(e);
    delete ctx->rt;
  }
  if (ctx->arena) (void)hipFree(ctx->arena);
  delete ctx;
}

// ---- CU-partitioned streams ------------------------------------------------------------------------
extern "C" int conette_stream_create_masked(const uint32_t* mask_words, int32_t n_words, void** out_stream) {
  if (!mask_words || n_words <= 0 || !out_stream) {
    cn_set_error("stream_create_masked: bad argument");
    return CN_ERR_ARG;
  }
  hipStream_t st = nullptr;
  CN_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words));
  *out_stream = (void*)st;
  return CN_OK;
}
extern "C" int conette_stream_destroy(void* stream) {
  if (stream) CN_HIP(hipStreamDestroy((hipStream_t)stream));
  return CN_OK;
}

// ---- resampler (row a1): torchaudio.functional.resample, sinc_interpolation, width 6, rolloff 0.99
static int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }

extern "C" int32_t conette_resample_len(int32_t n_in, int32_t orig_sr, int32_t new_sr) {
  const int g = gcd_i(orig_sr, new_sr);
  const long o = orig_sr / g, n = new_sr / g;
  return (int32_t)((n * (long)n_in + o - 1) / o);
}

__global__ void cn_resample_kernel(const float* __restrict__ in, int n_in, int n_out, int o, int n, int width, int K,
                                   const float* __restrict__ kern, float* __restrict__ out) {
  const int row = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int j = i % n, m = i / n;
  const int start = m * o - width;
  const float* x = in + (size_t)row * n_in;
  const float* kk = kern + (size_t)j * K;
  float acc = 0.f;
  for (int t = 0; t < K; ++t) {
    const int q = start + t;
    const float v = (q >= 0 && q < n_in) ? x[q] : 0.f;
    acc = fmaf(kk[t], v, acc);
  }
  out[(size_t)row * n_out + i] = acc;
}

struct ResampleTable {
  int o, n, width, K;
  float* dev;
};
static std::unordered_map<long, ResampleTable> g_resample;  // per (device, orig_sr, new_sr): the table lives in device memory

extern "C" int conette_resample(const float* in, int32_t rows, int32_t n_in, int32_t orig_sr, int32_t new_sr,
                                float* out, void* stream) {
  if (!in || !out || rows <= 0 || n_in <= 0 || orig_sr <= 0 || new_sr <= 0) {
    cn_set_error("resample: bad argument");
    return CN_ERR_ARG;
  }
  int dev_id = 0;
  CN_HIP(hipGetDevice(&dev_id));
  const long key = ((long)orig_sr * 1000003L + new_sr) * 64 + dev_id;
  auto it = g_resample.find(key);
  if (it == g_resample.end()) {
    // kernel table in fp32 arithmetic, as torchaudio 0.13.1 builds it in the waveform dtype
    const int g = gcd_i(orig_sr, new_sr);
    ResampleTable t;
    t.o = orig_sr / g;
    t.n = new_sr / g;
    const float lpw = 6.0f;
    float base = (float)(t.o < t.n ? t.o : t.n);
    base *= 0.99f;
    t.width = (int)ceil((double)lpw * t.o / (double)base);
    t.K = 2 * t.width + t.o;
    std::vector<float> h((size_t)t.n * t.K);
    const float scale = base / (float)t.o;
    for (int j = 0; j < t.n; ++j)
      for (int c = 0; c < t.K; ++c) {
        const float idx = (float)(c - t.width) / (float)t.o;
        float tt = (float)(-j) / (float)t.n + idx;
        tt *= base;
        tt = fminf(fmaxf(tt, -lpw), lpw);
        const float cw = cosf(tt * (float)M_PI / lpw / 2.0f);
        const float window = cw * cw;
        tt *= (float)M_PI;
        const float sinc = tt == 0.0f ? 1.0f : sinf(tt) / tt;
        h[(size_t)j * t.K + c] = sinc * window * scale;
      }
    CN_HIP(hipMalloc((void**)&t.dev, h.size() * 4));
    CN_HIP(hipMemcpy(t.dev, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    it = g_resample.emplace(key, t).first;
  }
  const ResampleTable& t = it->second;
  const int n_out = conette_resample_len(n_in, orig_sr, new_sr);
  hipLaunchKernelGGL(cn_resample_kernel, dim3((unsigned)((n_out + 255) / 256), (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, in, n_in, n_out, t.o, t.n, t.width, t.K, t.dev, out);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
