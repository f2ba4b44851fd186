// ConvNeXt-tiny audio encoder for gfx950 (rows a3-a7 of SURVEY.md section 8a).
//
// Reference: nn/encoders/convnext.py:61-74 (block), :207-217 (stem / downsample), :264-336
// (forward), nn/modules/norm.py:31-40 (LayerNorm).  Activations are channels-last (B, H, W, C)
// with H = time, W = frequency; the residual stream is of type XT (common.h: fp16 in the 16-bit
// precisions, fp32 otherwise), GEMM operands are the context's operand type.  Depthwise 7x7 + LayerNorm run on the VALU, the pointwise and
// downsample contractions on MFMA through cn_gemm (north_star).
#include <stddef.h>

#include "ctx.h"
#include <type_traits>

// Overflow watch of the fp16 residual stream (round 6).  A stream value beyond 65504 is inf where the fused MLP stored it; the
// saturating conversions further down the encoder then turn the NaNs of the next LayerNorm back into finite garbage, so the
// frame embeddings cannot be trusted to show it.  Every LayerNorm that reads the stream -- through the 7 x 7 windows of the next
// block's depthwise conv, in the downsample layers, and the frame-mean head -- counts the positions whose statistics are not
// finite in this per-device counter; conette_encode_nonfinite() reads and resets it.  (Rare path: one compare per position.)
__device__ int g_cn_nonfinite;
__device__ __forceinline__ void cn_watch_stat(float v) {
  if (!(__builtin_fabsf(v) <= 3.0e38f)) atomicAdd(&g_cn_nonfinite, 1);
}

#include "gemm2.h"
#include "mlp_rc2.h"
#include "mlp_rs.h"
#include "mlp_rs16.h"
#include "mlp_sp.h"
#include "down_fused.h"

// fp16 stream: depthwise conv on row pairs through v_dot2_f32_f16 (0: A/B builds, one v_fma_mix_f32 per tap as in the first half of round 5)
#ifndef CN_DW_DOT2
#define CN_DW_DOT2 1
#endif

// ---------------------------------------------------------------------------------------------
// stem: Conv2d(1 -> 96, k 4x4, s 4x4, pad (4, 0)) + LayerNorm(channels_first, eps 1e-6)
// in: logmel (B, F, 224) fp32; out: (B, H0, 56, 96) fp32.  4 lanes per output position (24 channels each, weights
// read from LDS 16 bytes at a time); the 64 positions of a block are contiguous in the output, so the normalised
// values go through an LDS tile and leave as whole 1 KB wave stores (16-byte pieces at a 96-byte lane stride made
// the kernel store bound: 172 us for 403 MB).
// ---------------------------------------------------------------------------------------------
template <int P, typename XT>  // P positions per lane group: the weights read from LDS (6 KB per position otherwise) are shared by P positions
__global__ __launch_bounds__(256) void cn_stem_kernel(const float* __restrict__ in, int F, int H0, long n_pos,
                                                      const float* __restrict__ w /*[16][96]*/,
                                                      const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                      const float* __restrict__ ln_b, XT* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_w[16 * 96];
  __shared__ __attribute__((aligned(16))) float s_o[P * 64 * 96];
  for (int i = threadIdx.x; i < 16 * 96; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int q = threadIdx.x & 3, lp = threadIdx.x >> 2;
  const long pos0 = (long)blockIdx.x * (64 * P);
  float xin[P][16];
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const long pos = pos0 + 64 * u + lp;
    const long p = pos < n_pos ? pos : n_pos - 1;
    const int wq = (int)(p % 56);
    const long t = p / 56;
    const int h = (int)(t % H0);
    const int b = (int)(t / H0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * h - 4 + i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r >= 0 && r < F) v = *(const f32x4*)(in + ((size_t)b * F + r) * CN_N_MELS + 4 * wq);
      xin[u][4 * i] = v[0], xin[u][4 * i + 1] = v[1], xin[u][4 * i + 2] = v[2], xin[u][4 * i + 3] = v[3];
    }
  }
  f32x4 acc[P][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const f32x4 b4 = *(const f32x4*)(bias + q * 24 + 4 * j);
#pragma unroll
    for (int u = 0; u < P; ++u) acc[u][j] = b4;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f32x4 w4 = *(const f32x4*)(s_w + i * 96 + q * 24 + 4 * j);
#pragma unroll
      for (int u = 0; u < P; ++u) acc[u][j] += xin[u][i] * w4;
    }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const f32x4 s4 = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]) + (acc[u][4] + acc[u][5]);
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s * (1.0f / 96.0f);
    f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      acc[u][j] -= mean;
      q4 += acc[u][j] * acc[u][j];
    }
    float v2 = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    v2 += __shfl_xor(v2, 1);
    v2 += __shfl_xor(v2, 2);
    const float rstd = 1.0f / sqrtf(v2 * (1.0f / 96.0f) + 1e-6f);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ch = q * 24 + 4 * j;
      *(f32x4*)(s_o + (64 * u + lp) * 96 + ch) = acc[u][j] * rstd * *(const f32x4*)(ln_w + ch) + *(const f32x4*)(ln_b + ch);
    }
  }
  __syncthreads();
  if constexpr (sizeof(XT) == 2) {
    const long n8 = min((long)(64 * P), n_pos - pos0) * 12;  // 16-byte pieces (8 values) of this block's positions
    cn_h8<XT>* o8 = (cn_h8<XT>*)(out + (size_t)pos0 * 96);
#pragma unroll
    for (int k = 0; k < 3 * P; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < n8) {
        const f32x4 a = ((const f32x4*)s_o)[2 * i], b = ((const f32x4*)s_o)[2 * i + 1];
        o8[i] = cn_h8<XT>{cn_from_f32<XT>(a[0]), cn_from_f32<XT>(a[1]), cn_from_f32<XT>(a[2]), cn_from_f32<XT>(a[3]),
                          cn_from_f32<XT>(b[0]), cn_from_f32<XT>(b[1]), cn_from_f32<XT>(b[2]), cn_from_f32<XT>(b[3])};
      }
    }
  } else {
  const long n4 = min((long)(64 * P), n_pos - pos0) * 24;  // 16-byte pieces of this block's positions
  f32x4* o4 = (f32x4*)(out + (size_t)pos0 * 96);
#pragma unroll
  for (int k = 0; k < 6 * P; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < n4) o4[i] = ((const f32x4*)s_o)[i];
  }
  }
}

// Stem on the fp32 matrix cores (round 5).  The VALU kernel above reads its 16 x 96 weights from LDS once per position (6 KB, 69 us
// of LDS time per launch at B = 64: it is LDS bound); here the product runs transposed as O^T = W^T (96 x 16) . patches^T (16 x 32
// positions) on v_mfma_f32_32x32x2_f32 -- exact fp32 products and sums -- with W^T as 24 A-operand registers per lane, loaded once
// per wave: no weights in LDS at all.  Lane (j = l & 31, h = l >> 5) is position tile * 32 + j; k-step s of the product takes
// patch row s >> 1, columns 2 h + (s & 1): one 8-byte load per patch row and lane.  The rows of W^T are permuted (cn_rc2_chan) so that
// accumulator register r of channel tile t is channel 32 t + 16 (r >> 3) + 8 h + (r & 7): a lane owns 48 channels of ITS position in
// runs of eight, the LayerNorm statistics are a lane sum plus one exchange between the halves, the stores are 16 bytes.
template <typename XT>
__global__ __launch_bounds__(256) void cn_stem_mfma_kernel(const float* __restrict__ in, int F, int H0, long n_pos,
                                                           const float* __restrict__ w /*[16][96]*/, const float* __restrict__ bias,
                                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                           XT* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_p[3 * 96];  // bias, LN weight, LN bias
  for (int i = threadIdx.x; i < 3 * 96; i += 256) s_p[i] = i < 96 ? bias[i] : i < 192 ? ln_w[i - 96] : ln_b[i - 192];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  float a[3][8];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s) a[t][s] = w[(4 * (s >> 1) + 2 * h + (s & 1)) * 96 + 32 * t + cn_rc2_chan(j)];
  __syncthreads();
  const long n_tiles = (n_pos + 31) >> 5;
  const long n_waves = (long)gridDim.x * 4;
  for (long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
    const long pos = tile * 32 + j;
    const long p = pos < n_pos ? pos : n_pos - 1;
    const int wq = (int)(p % 56);
    const long tt = p / 56;
    const int hh = (int)(tt % H0);
    const int b = (int)(tt / H0);
    f32x2 xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * hh - 4 + i;
      xv[i] = f32x2{0.f, 0.f};
      if (r >= 0 && r < F) xv[i] = *(const f32x2*)(in + ((size_t)b * F + r) * CN_N_MELS + 4 * wq + 2 * h);
    }
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 b0 = *(const f32x4*)(s_p + 32 * t + 16 * u + 8 * h), b1 = *(const f32x4*)(s_p + 32 * t + 16 * u + 8 * h + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][8 * u + e] = b0[e], acc[t][8 * u + 4 + e] = b1[e];
      }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], xv[s >> 1][s & 1], acc[t], 0, 0, 0);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[t][r];
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / 96.0f);
    float v2 = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[t][r] -= mean;
        v2 = fmaf(acc[t][r], acc[t][r], v2);
      }
    v2 += __shfl_xor(v2, 32);
    const float rstd = 1.0f / sqrtf(v2 * (1.0f / 96.0f) + 1e-6f);
    if (pos < n_pos) {
      XT* o = out + (size_t)pos * 96 + 8 * h;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ch = 32 * t + 16 * u + 8 * h;
          float v[8];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 lw = *(const f32x4*)(s_p + 96 + ch + 4 * q), lb = *(const f32x4*)(s_p + 192 + ch + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = acc[t][8 * u + 4 * q + e] * rstd * lw[e] + lb[e];
          }
          if constexpr (sizeof(XT) == 2) {
            *(cn_h8<XT>*)(o + 32 * t + 16 * u) = cn_h8<XT>{cn_from_f32<XT>(v[0]), cn_from_f32<XT>(v[1]), cn_from_f32<XT>(v[2]), cn_from_f32<XT>(v[3]),
                                                          cn_from_f32<XT>(v[4]), cn_from_f32<XT>(v[5]), cn_from_f32<XT>(v[6]), cn_from_f32<XT>(v[7])};
          } else {
            *(f32x4*)(o + 32 * t + 16 * u) = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(o + 32 * t + 16 * u + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// depthwise 7x7 (pad 3) + LayerNorm over C (eps 1e-6): x XT (B,H,W,C) -> y T (B,H,W,C)
// (XT = half_t: 2-byte loads feed v_fma_mix_f32 directly -- the fp16 input is an operand of the fp32 fma, no conversion
// instruction, the same products and sums as from an fp32 stream holding the same values)
// One thread = one channel x a TH(h) x 4(w) output patch: (TH+6) x 10 loads feed TH*4*49 FMAs.
// Loads are branch-free (clamped address, zero-masked value) and issued a whole input row at a
// time so that many are in flight; lanes run over channels, so every load instruction is one
// coalesced 256-byte line.  The block is C x S threads covering a TH x (4S) tile for all
// channels, so the LayerNorm over C is local to the block: conv results go through an LDS tile
// [pos][C] and one wave normalises a position (two-pass mean / variance, wave-shuffle sums).
// ---------------------------------------------------------------------------------------------
__device__ unsigned long long g_dw_prof[8];
#define DW_STAMP(i)                                                   \
  if (dbg) {                                                          \
    const unsigned long long t_ = clock64();                          \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_dw_prof[i], t_ - t_prev); \
    t_prev = t_;                                                      \
  }

// LDS tile geometry of cn_dwconv_ln_kernel: rows of PITCH4 16-byte chunks, PITCH4 odd.  Round 4 (rocprof r03_h: 34 % / 23 % of
// the LDS cycles of the C = 96 / 192 launches were bank conflicts, 2.7-3x on the statistics pass by the simulator of
// profiles/r04_notes.md): the statistics pass now runs with lane = POSITION -- the PARTS threads of a position sit in
// different waves, a wave's 16-byte reads go to the same chunk of 64 (or 2 x 32) different rows, and an odd pitch puts
// consecutive rows on different bank quads: conflict free, with every thread's sum in the order it always had.  The store pass keeps its 8 consecutive channels per lane (one 16-byte store): a
// layout that also frees it of its 2-way conflict (two half-row runs per lane, pitch = 12 mod 16) needs two 8-byte stores per
// item and measured SLOWER (205 against 194 us at C = 96: the pass is store-issue bound, not LDS bound).
template <int C, int S, int TH> struct DwTile {
  static constexpr int NCHUNK = C / 4;
  static constexpr int PITCH4 = NCHUNK | 1;
  static constexpr int PITCH = PITCH4 * 4;
};

template <typename T, typename XT, int C, int S, int TH>
__global__ __launch_bounds__((C > 384 ? 384 : C) * S) void cn_dwconv_ln_kernel(const XT* __restrict__ x, int H, int W, int tiles_h,
                                                            int tiles_w, const float* __restrict__ dw_w /*[49][C]*/,
                                                            const unsigned* __restrict__ dw_wp /*[42][C] fp16 row pairs (fp16 stream)*/,
                                                            const float* __restrict__ dw_b,
                                                            const float* __restrict__ ln_w,
                                                            const float* __restrict__ ln_b, T* __restrict__ y,
                                                            int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_v = (float*)smem_raw;  // [TH*4*S][C]
  unsigned long long t_prev = dbg ? clock64() : 0;
  constexpr int NP = TH * 4;      // positions per thread
  constexpr int CT = C > 384 ? 384 : C;  // threads along the channel axis (C = 768: two passes)
  constexpr int PITCH = DwTile<C, S, TH>::PITCH;  // LDS tile pitch (words): 16-byte aligned rows, see DwTile
  const int tid = threadIdx.x;
  const int c0 = tid % CT, sidx = tid / CT;
  int bid = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % tiles_w;
  bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, w0 = tw * (4 * S) + sidx * 4;

  int wcl[10];
  bool wok[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    const int ww = w0 - 3 + q;
    wok[q] = (ww >= 0) && (ww < W);
    wcl[q] = min(max(ww, 0), W - 1) * C;
  }
#pragma unroll 1
  for (int c = c0; c < C; c += CT) {
  float acc[TH][4];
  const float bias = dw_b[c];
#pragma unroll
  for (int a = 0; a < TH; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[a][e] = bias;
  const XT* xb = x + (size_t)b * H * W * C + c;
  // 80 % of the tiles are interior: no clamping / masking, and every load of an input row is
  // `row base + compile-time offset` (q * C floats fits the 13-bit immediate), which removes the
  // per-load 64-bit address arithmetic that dominated the VALU instruction count (rocprof:
  // 4250 VALU wave-instructions per 32-output patch against 1568 FMAs).
  const bool interior = (h0 >= 3) && (h0 + TH + 3 <= H) && (w0 >= 3) && (w0 + 4 + 3 <= W);
  if constexpr (sizeof(XT) == 2 && CN_DW_DOT2) {
    // ---- fp16 stream, round 5: input rows in PAIRS.  p[q] = (x[r][q], x[r + 1][q]) for even r is one register, the weights come as
    // fp16 pairs of consecutive kernel rows -- ke[a][j] = (k[2a][j], k[2a + 1][j]), ko[a][j] = (k[2a + 1][j], k[2a + 2][j]): whichever
    // parity r - oh has, the row pair meets a weight pair -- and `v_dot2_f32_f16` (exact fp16 products, fp32 sum) adds TWO taps per
    // instruction: 28 instructions per output (21 dot2 + 7 single taps through v_fma_mix_f32: k[0][j] / k[6][j] are halves of ke[0][j] /
    // ko[2][j]) against 49, + one v_perm per loaded pair.  The VALU-issue floor of the layer drops by 40 %; the only change in the
    // arithmetic is the depthwise weight as an fp16 operand (2^-12 relative, listed in oracle/bf16_ref.py) and the order of an
    // output's 49 terms.
    static_assert(TH % 2 == 0, "row pairs");
    cn_h2 ke[3][7], ko[3][7];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        ke[a][j] = __builtin_bit_cast(cn_h2, dw_wp[(a * 7 + j) * C + c]);
        ko[a][j] = __builtin_bit_cast(cn_h2, dw_wp[(21 + a * 7 + j) * C + c]);
      }
    DW_STAMP(0)
    auto fma_pair = [&](int r, const cn_h2 (&p)[10]) {   // rows r (even) and r + 1 of the halo
#pragma unroll
      for (int oh = 0; oh < TH; ++oh) {
        const int i = r - oh;   // kernel row of halo row r for output row oh
        if (i < -1 || i > 6) continue;
#pragma unroll
        for (int q = 0; q < 10; ++q)
#pragma unroll
          for (int ow = 0; ow < 4; ++ow) {
            const int j = q - ow;
            if (j < 0 || j > 6) continue;
            if (i == -1) acc[oh][ow] = fmaf((float)p[q][1], (float)ke[0][j][0], acc[oh][ow]);
            else if (i == 6) acc[oh][ow] = fmaf((float)p[q][0], (float)ko[2][j][1], acc[oh][ow]);
            else if ((i & 1) == 0) acc[oh][ow] = __builtin_amdgcn_fdot2(p[q], ke[i >> 1][j], acc[oh][ow], false);
            else acc[oh][ow] = __builtin_amdgcn_fdot2(p[q], ko[i >> 1][j], acc[oh][ow], false);
          }
      }
    };
    constexpr int NPR = (TH + 6) / 2;
    if (interior) {
      const XT* base = xb + ((size_t)(h0 - 3) * W + (w0 - 3)) * C;
      auto load_pair = [&](int r, cn_h2 (&p)[10]) {
        const XT* x0 = base + (size_t)r * W * C;
        const XT* x1 = x0 + (size_t)W * C;
#pragma unroll
        for (int q = 0; q < 10; ++q) p[q] = cn_h2{x0[q * C], x1[q * C]};
      };
      cn_h2 pa[10], pb[10];
      load_pair(0, pa);
#pragma unroll
      for (int u = 0; u < NPR; u += 2) {   // two-pair software pipeline: the loads of pair u + 1 are in flight under the products of pair u
        if (u + 1 < NPR) load_pair(2 * (u + 1), pb);
        fma_pair(2 * u, pa);
        if (u + 2 < NPR) load_pair(2 * (u + 2), pa);
        if (u + 1 < NPR) fma_pair(2 * (u + 1), pb);
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPR; ++u) {
        const int hh0 = h0 - 3 + 2 * u, hh1 = hh0 + 1;
        const bool ok0 = (hh0 >= 0) && (hh0 < H), ok1 = (hh1 >= 0) && (hh1 < H);
        const XT* x0 = xb + (size_t)min(max(hh0, 0), H - 1) * W * C;
        const XT* x1 = xb + (size_t)min(max(hh1, 0), H - 1) * W * C;
        XT v0[10], v1[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) v0[q] = x0[wcl[q]], v1[q] = x1[wcl[q]];
        cn_h2 p[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) p[q] = cn_h2{(ok0 && wok[q]) ? v0[q] : (XT)0.f, (ok1 && wok[q]) ? v1[q] : (XT)0.f};
        fma_pair(2 * u, p);
      }
    }
  } else {
  float k[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) k[i] = dw_w[i * C + c];
  DW_STAMP(0)
  auto fma_row = [&](int r, const XT (&v)[10]) {
#pragma unroll
    for (int oh = 0; oh < TH; ++oh) {
      const int i = r - oh;
      if (i < 0 || i > 6) continue;
#pragma unroll
      for (int q = 0; q < 10; ++q)
#pragma unroll
        for (int ow = 0; ow < 4; ++ow) {
          const int j = q - ow;
          if (j < 0 || j > 6) continue;
          acc[oh][ow] = fmaf((float)v[q], k[i * 7 + j], acc[oh][ow]);
        }
    }
  };
  if (interior) {
    const XT* base = xb + ((size_t)(h0 - 3) * W + (w0 - 3)) * C;
    // explicit two-row software pipeline: the loads of row r+1 are in flight under the FMAs of row r
    XT va[10], vb[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) va[q] = base[q * C];
#pragma unroll
    for (int r = 0; r < TH + 6; r += 2) {
      if (r + 1 < TH + 6) {
        const XT* xr = base + (size_t)(r + 1) * W * C;
#pragma unroll
        for (int q = 0; q < 10; ++q) vb[q] = xr[q * C];
      }
      fma_row(r, va);
      if (r + 2 < TH + 6) {
        const XT* xr = base + (size_t)(r + 2) * W * C;
#pragma unroll
        for (int q = 0; q < 10; ++q) va[q] = xr[q * C];
      }
      if (r + 1 < TH + 6) fma_row(r + 1, vb);
    }
  } else {
#pragma unroll
    for (int r = 0; r < TH + 6; ++r) {
      const int hh = h0 - 3 + r;
      const bool hok = (hh >= 0) && (hh < H);
      const XT* xr = xb + (size_t)min(max(hh, 0), H - 1) * W * C;
      XT v[10];
#pragma unroll
      for (int q = 0; q < 10; ++q) v[q] = xr[wcl[q]];
#pragma unroll
      for (int q = 0; q < 10; ++q) v[q] = (hok && wok[q]) ? v[q] : (XT)0.f;
      fma_row(r, v);
    }
  }
  }  // fp32 stream / CN_DW_DOT2 = 0
#pragma unroll
  for (int oh = 0; oh < TH; ++oh)
#pragma unroll
    for (int ow = 0; ow < 4; ++ow) s_v[((sidx * NP) + oh * 4 + ow) * PITCH + c] = acc[oh][ow];
  DW_STAMP(1)
  }
  __syncthreads();
  DW_STAMP(2)

  // ---- LayerNorm over C.  Phase A: statistics, PARTS threads per position (two-pass mean / variance,
  // partial sums through LDS).  Phase B: normalise + store with the conv mapping (lanes = channels,
  // coalesced).  (One wave per position with shuffle reductions was a serial latency chain that took
  // longer than the convolution itself.)
  // A thread sums the 16-byte chunks part, part + PARTS, ... of its position's row: the PARTS threads of a position read
  // consecutive chunks and, with PITCH / 4 = PARTS (mod 16), consecutive positions continue the sequence, so a
  // ds_read_b128 wave access is bank-conflict free (rocprof r02_a: SQ_LDS_BANK_CONFLICT was 57-75 % of the LDS cycles and the
  // LDS active for 59 % of the kernel at C = 192 with contiguous per-thread segments and 4-byte reads).
  constexpr int NPOS = NP * S, NT = CT * S, PARTS = NT / NPOS, NCHUNK = C / 4, CPT = NCHUNK / PARTS;
  static_assert(NT % NPOS == 0 && NCHUNK % PARTS == 0, "LayerNorm thread mapping");
  float* s_ps = s_v + NPOS * PITCH;       // [PARTS][NPOS]
  float* s_mean = s_ps + NPOS * PARTS;    // [NPOS]
  float* s_rstd = s_mean + NPOS;          // [NPOS]
  {
    // lane = position (see DwTile); a thread still sums the chunks part, part + PARTS, ... of its row in that order, so the
    // statistics are bit for bit those of rounds 2-3 (only which lane reads which row changed)
    const int pos = tid % NPOS, part = tid / NPOS;
    const float* row = s_v + pos * PITCH + part * 4;
    f32x4 seg[CPT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      seg[i] = *(const f32x4*)(row + i * PARTS * 4);
      sum += (seg[i][0] + seg[i][1]) + (seg[i][2] + seg[i][3]);
    }
    s_ps[part * NPOS + pos] = sum;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int j = 0; j < PARTS; ++j) mean += s_ps[j * NPOS + pos];
    mean *= (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = seg[i][e] - mean;
        sq = fmaf(d, d, sq);
      }
    __syncthreads();
    s_ps[part * NPOS + pos] = sq;
    __syncthreads();
    if (part == 0) {
      float var = 0.f;
#pragma unroll
      for (int j = 0; j < PARTS; ++j) var += s_ps[j * NPOS + pos];
      s_mean[pos] = mean;
      s_rstd[pos] = 1.0f / sqrtf(var * (1.0f / C) + 1e-6f);
      cn_watch_stat(var);
    }
    __syncthreads();
  }
  DW_STAMP(3)
  // Phase B: normalise + store, 8 consecutive channels per item (one 16-byte bf16 store; 2-byte
  // stores per lane made this phase as long as the convolution itself -- store-issue bound)
  constexpr int C8 = C / 8;
  static_assert(NT % C8 == 0, "a thread keeps its 8 channels over all its items: the LN affine is loaded once");
  const int c8 = (tid % C8) * 8;
  const f32x4 lw0 = *(const f32x4*)(ln_w + c8), lw1 = *(const f32x4*)(ln_w + c8 + 4);
  const f32x4 lb0 = *(const f32x4*)(ln_b + c8), lb1 = *(const f32x4*)(ln_b + c8 + 4);
  for (int item = tid; item < NPOS * C8; item += NT) {
    const int pos = item / C8;
    const int ps = pos / NP, oh = (pos % NP) >> 2, ow = pos & 3;
    const int h = h0 + oh, w = tw * (4 * S) + ps * 4 + ow;
    if (h >= H || w >= W) continue;
    const float mean = s_mean[pos], rstd = s_rstd[pos];
    const f32x4 v0 = *(const f32x4*)(s_v + pos * PITCH + c8), v1 = *(const f32x4*)(s_v + pos * PITCH + c8 + 4);
    T* dst = y + (((size_t)b * H + h) * W + w) * C + c8;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = ((i < 4 ? v0[i] : v1[i - 4]) - mean) * rstd * (i < 4 ? lw0[i] : lw1[i - 4]) + (i < 4 ? lb0[i] : lb1[i - 4]);
    cn_store8(dst, o);
  }
  DW_STAMP(4)
}

extern "C" int conette_debug_g2prof(unsigned long long* out16, int reset) {
  if (out16) CN_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_g2_prof), 128));
  if (reset) {
    unsigned long long z[16] = {0};
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_g2_prof), z, 128));
  }
  return CN_OK;
}

extern "C" int conette_debug_dwprof(unsigned long long* out8, int reset) {
  if (out8) CN_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dw_prof), 64));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dw_prof), z, 64));
  }
  return CN_OK;
}

template <typename T, typename XT, int C, int S, int TH>
static int launch_dwconv(const XT* x, int B, int H, int W, const CnBlockW& bw, T* y, hipStream_t s) {
  const int tiles_h = cn_cdiv(H, TH), tiles_w = cn_cdiv(W, 4 * S);
  constexpr int NPOS_ = TH * 4 * S, NT_ = (C > 384 ? 384 : C) * S;
  const size_t smem = ((size_t)NPOS_ * DwTile<C, S, TH>::PITCH + NPOS_ * (NT_ / NPOS_) + 2 * NPOS_) * sizeof(float);
  CN_TRY(cn_configure_lds((const void*)cn_dwconv_ln_kernel<T, XT, C, S, TH>, (int)smem));
  hipLaunchKernelGGL((cn_dwconv_ln_kernel<T, XT, C, S, TH>), dim3((unsigned)(B * tiles_h * tiles_w)),
                     dim3((C > 384 ? 384 : C) * S), smem, s, x, H, W, tiles_h, tiles_w, bw.dw_w, bw.dw_wp, bw.dw_b, bw.ln_w, bw.ln_b, y,
#ifdef CN_G2_PROF
                     getenv("CN_DW_DEBUG") ? atoi(getenv("CN_DW_DEBUG")) : 0);
#else
                     0);
#endif
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---------------------------------------------------------------------------------------------
// Full-width variant for the deep stages (W = 14 at C = 384, W = 7 at C = 768: the frequency axis, a constant
// of the architecture whatever the clip length).  With 4-wide tiles almost no tile of such a narrow map is
// "interior", so every load went through the clamped / masked slow path and each input element was fetched
// ~6x (10 x 10 halo per 4 x 4 outputs): rocprof 113 us per launch at 1.1 TB/s of HBM traffic, 4x the VALU
// floor.  Here a block owns TH rows x the WHOLE width for all channels: a thread (= channel) keeps TH x WW
// accumulators, every input row is loaded once (WW coalesced loads), the left / right zero padding is resolved
// at compile time (taps that fall outside simply do not exist) and rows above / below the map are skipped by a
// block-uniform branch.  LayerNorm: conv results through an LDS tile [pos][C], one wave per position with DPP
// sums, two positions in flight; 16-byte bf16 stores as in the tiled kernel.
// ---------------------------------------------------------------------------------------------
// The convolution of one channel for output columns [OW0, OW0 + NOW) of TH rows: the thread loads the NQ input columns those
// outputs can see and keeps TH x NOW accumulators.  SPLIT = 2 (C = 384) gives each half of the width to its own thread, so a
// block has twice the waves over the same LDS tile (12 per CU instead of 6) and a thread half the accumulators; the taps of
// an output are still added in the same (kh, kw) order, so the result does not depend on SPLIT.
template <int C, int WW, int TH, int OW0, int NOW, typename XT>
__device__ __forceinline__ void cn_fw_conv(const XT* __restrict__ xb /* + c */, int H, int h0,
                                           const float* __restrict__ dw_w, const unsigned* __restrict__ dw_wp, float bias, int c,
                                           float* __restrict__ s_v, int PITCH) {
  constexpr int Q0 = OW0 - 3 < 0 ? 0 : OW0 - 3, Q1 = OW0 + NOW + 3 > WW ? WW : OW0 + NOW + 3, NQ = Q1 - Q0;
  if constexpr (sizeof(XT) == 2 && CN_DW_DOT2) {
    // fp16 stream: input rows in pairs, two taps per v_dot2_f32_f16 (see cn_dwconv_ln_kernel)
    static_assert(TH % 2 == 0, "row pairs");
    cn_h2 ke[3][7], ko[3][7];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        ke[a][j] = __builtin_bit_cast(cn_h2, dw_wp[(a * 7 + j) * C + c]);
        ko[a][j] = __builtin_bit_cast(cn_h2, dw_wp[(21 + a * 7 + j) * C + c]);
      }
    float acc[TH][NOW];
#pragma unroll
    for (int a = 0; a < TH; ++a)
#pragma unroll
      for (int e = 0; e < NOW; ++e) acc[a][e] = bias;
    auto load_pair = [&](int r, cn_h2 (&p)[NQ]) {   // halo rows r (even) and r + 1; rows above / below the map are zeros (block-uniform)
      const int hh0 = h0 - 3 + r, hh1 = hh0 + 1;
      XT v0[NQ], v1[NQ];
      if (hh0 >= 0 && hh0 < H) {
        const XT* xr = xb + (size_t)hh0 * WW * C;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v0[q] = xr[(Q0 + q) * C];
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v0[q] = (XT)0.f;
      }
      if (hh1 >= 0 && hh1 < H) {
        const XT* xr = xb + (size_t)hh1 * WW * C;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v1[q] = xr[(Q0 + q) * C];
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v1[q] = (XT)0.f;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) p[q] = cn_h2{v0[q], v1[q]};
    };
    auto fma_pair = [&](int r, const cn_h2 (&p)[NQ]) {
#pragma unroll
      for (int oh = 0; oh < TH; ++oh) {
        const int i = r - oh;
        if (i < -1 || i > 6) continue;
#pragma unroll
        for (int ow = 0; ow < NOW; ++ow)
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            const int q = OW0 + ow + j - 3;
            if (q < 0 || q >= WW) continue;  // zero padding left / right of the map: the tap does not exist
            const cn_h2 pv = p[q - Q0];
            if (i == -1) acc[oh][ow] = fmaf((float)pv[1], (float)ke[0][j][0], acc[oh][ow]);
            else if (i == 6) acc[oh][ow] = fmaf((float)pv[0], (float)ko[2][j][1], acc[oh][ow]);
            else if ((i & 1) == 0) acc[oh][ow] = __builtin_amdgcn_fdot2(pv, ke[i >> 1][j], acc[oh][ow], false);
            else acc[oh][ow] = __builtin_amdgcn_fdot2(pv, ko[i >> 1][j], acc[oh][ow], false);
          }
      }
    };
    constexpr int NPR = (TH + 6) / 2;
    cn_h2 pa[NQ], pb[NQ];
    load_pair(0, pa);
#pragma unroll
    for (int u = 0; u < NPR; u += 2) {
      if (u + 1 < NPR) load_pair(2 * (u + 1), pb);
      fma_pair(2 * u, pa);
      if (u + 2 < NPR) load_pair(2 * (u + 2), pa);
      if (u + 1 < NPR) fma_pair(2 * (u + 1), pb);
    }
#pragma unroll
    for (int oh = 0; oh < TH; ++oh)
#pragma unroll
      for (int ow = 0; ow < NOW; ++ow) s_v[(oh * WW + OW0 + ow) * PITCH + c] = acc[oh][ow];
    return;
  }
  float k[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) k[i] = dw_w[i * C + c];
  float acc[TH][NOW];
#pragma unroll
  for (int a = 0; a < TH; ++a)
#pragma unroll
    for (int e = 0; e < NOW; ++e) acc[a][e] = bias;
  auto load_row = [&](int r, XT (&v)[NQ]) {
    const int hh = h0 - 3 + r;
    if (hh >= 0 && hh < H) {  // block-uniform
      const XT* xr = xb + (size_t)hh * WW * C;
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[q] = xr[(Q0 + q) * C];
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[q] = (XT)0.f;
    }
  };
  auto fma_row = [&](int r, const XT (&v)[NQ]) {
#pragma unroll
    for (int oh = 0; oh < TH; ++oh) {
      const int i = r - oh;
      if (i < 0 || i > 6) continue;
#pragma unroll
      for (int ow = 0; ow < NOW; ++ow)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int q = OW0 + ow + j - 3;
          if (q < 0 || q >= WW) continue;  // zero padding left / right of the map: the tap does not exist
          acc[oh][ow] = fmaf((float)v[q - Q0], k[i * 7 + j], acc[oh][ow]);
        }
    }
  };
  XT va[NQ], vb[NQ];
  load_row(0, va);
#pragma unroll
  for (int r = 0; r < TH + 6; r += 2) {
    if (r + 1 < TH + 6) load_row(r + 1, vb);
    fma_row(r, va);
    if (r + 2 < TH + 6) load_row(r + 2, va);
    if (r + 1 < TH + 6) fma_row(r + 1, vb);
  }
#pragma unroll
  for (int oh = 0; oh < TH; ++oh)
#pragma unroll
    for (int ow = 0; ow < NOW; ++ow) s_v[(oh * WW + OW0 + ow) * PITCH + c] = acc[oh][ow];
}

template <typename T, typename XT, int C, int WW, int TH, int SPLIT>
__global__ __launch_bounds__(C* SPLIT > 384 ? 768 : 384) void cn_dwconv_ln_fw_kernel(const XT* __restrict__ x, int H, int tiles_h,
                                                              const float* __restrict__ dw_w /*[49][C]*/, const unsigned* __restrict__ dw_wp,
                                                              const float* __restrict__ dw_b,
                                                              const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, T* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_v = (float*)smem_raw;  // [TH*WW][C + 4]
  constexpr int CT = C * SPLIT > 384 ? 768 : 384, NPOS = TH * WW, PITCH = C + 4, NW = CT / 64;  // 16-byte aligned rows, odd chunk count
  static_assert(SPLIT == 1 || (SPLIT == 2 && C == 384 && WW % 2 == 0), "the width is split in two only at C = 384");
  float* s_mean = s_v + NPOS * PITCH;
  float* s_rstd = s_mean + NPOS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH;
  const XT* xb0 = x + (size_t)b * H * WW * C;

  if constexpr (SPLIT == 1) {
#pragma unroll 1
    for (int c = tid; c < C; c += CT) cn_fw_conv<C, WW, TH, 0, WW, XT>(xb0 + c, H, h0, dw_w, dw_wp, dw_b[c], c, s_v, PITCH);
  } else {
    const int c = tid < C ? tid : tid - C;  // waves 0-5: left half, waves 6-11: right half (wave-uniform)
    if (tid < C)
      cn_fw_conv<C, WW, TH, 0, WW / 2, XT>(xb0 + c, H, h0, dw_w, dw_wp, dw_b[c], c, s_v, PITCH);
    else
      cn_fw_conv<C, WW, TH, WW / 2, WW / 2, XT>(xb0 + c, H, h0, dw_w, dw_wp, dw_b[c], c, s_v, PITCH);
  }
  __syncthreads();
  // LayerNorm statistics: wave per position, C / 64 values per lane, two positions per iteration
  constexpr int VPL = C / 64;
  for (int p0 = wave * 2; p0 < NPOS; p0 += NW * 2) {
    const int p1 = min(p0 + 1, NPOS - 1);
    float a0[VPL], a1[VPL], s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      a0[i] = s_v[p0 * PITCH + i * 64 + lane];
      a1[i] = s_v[p1 * PITCH + i * 64 + lane];
      s0 += a0[i];
      s1 += a1[i];
    }
    const float m0 = cn_wave_sum_dpp(s0) * (1.0f / C), m1 = cn_wave_sum_dpp(s1) * (1.0f / C);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      q0 = fmaf(a0[i] - m0, a0[i] - m0, q0);
      q1 = fmaf(a1[i] - m1, a1[i] - m1, q1);
    }
    q0 = cn_wave_sum_dpp(q0);
    q1 = cn_wave_sum_dpp(q1);
    if (lane == 0) {
      s_mean[p0] = m0;
      s_rstd[p0] = 1.0f / sqrtf(q0 * (1.0f / C) + 1e-6f);
      s_mean[p1] = m1;
      s_rstd[p1] = 1.0f / sqrtf(q1 * (1.0f / C) + 1e-6f);
      cn_watch_stat(q0 + q1);
    }
  }
  __syncthreads();
  constexpr int C8 = C / 8;
  static_assert(CT % C8 == 0, "a thread keeps its 8 channels over all its items: the LN affine is loaded once");
  const int c8 = (tid % C8) * 8;
  const f32x4 lw0 = *(const f32x4*)(ln_w + c8), lw1 = *(const f32x4*)(ln_w + c8 + 4);
  const f32x4 lb0 = *(const f32x4*)(ln_b + c8), lb1 = *(const f32x4*)(ln_b + c8 + 4);
  for (int item = tid; item < NPOS * C8; item += CT) {
    const int pos = item / C8;
    const int h = h0 + pos / WW, w = pos % WW;
    if (h >= H) continue;
    const float mean = s_mean[pos], rstd = s_rstd[pos];
    const f32x4 v0 = *(const f32x4*)(s_v + pos * PITCH + c8), v1 = *(const f32x4*)(s_v + pos * PITCH + c8 + 4);
    T* dst = y + (((size_t)b * H + h) * WW + w) * C + c8;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = ((i < 4 ? v0[i] : v1[i - 4]) - mean) * rstd * (i < 4 ? lw0[i] : lw1[i - 4]) + (i < 4 ? lb0[i] : lb1[i - 4]);
    cn_store8(dst, o);
  }
}

// tile of the depthwise kernel of stages 0 / 1: 4 S columns x TH rows per block of C x S threads (A/B builds: CN_DW96_S ...)
#ifndef CN_DW96_S
#define CN_DW96_S 2
#endif
#ifndef CN_DW96_TH
#define CN_DW96_TH 12   // (8 until the v_dot2 form: with 40 % fewer VALU instructions the loads bind, and 18 halo rows per 12 cost less than 14 per 8)
#endif
#ifndef CN_DW192_S
#define CN_DW192_S 1
#endif
#ifndef CN_DW192_TH
#define CN_DW192_TH 12
#endif
#ifndef CN_STEM_MFMA
#define CN_STEM_MFMA 1   // 0 (A/B builds): the VALU stem kernel of rounds 1-4
#endif
#ifndef CN_FW_SPLIT
#define CN_FW_SPLIT 2
#endif
#ifndef CN_RS_PRIO
#define CN_RS_PRIO 8   // mlp_rs.h ABL bits 8 / 16: static wave priority for the B / the A role (0: none)
#endif
#ifndef CN_FW_TH
#define CN_FW_TH 4   // output rows per block of the full-width kernel at C = 384 (A/B builds: 2 = half the LN tile, two blocks per CU)
#endif
template <typename T, typename XT, int C, int WW, int TH, int SPLIT = 1>
static int launch_dwconv_fw(const XT* x, int B, int H, const CnBlockW& bw, T* y, hipStream_t s) {
  const int tiles_h = cn_cdiv(H, TH);
  const size_t smem = ((size_t)TH * WW * (C + 4) + 2 * TH * WW) * sizeof(float);
  CN_TRY(cn_configure_lds((const void*)cn_dwconv_ln_fw_kernel<T, XT, C, WW, TH, SPLIT>, (int)smem));
  hipLaunchKernelGGL((cn_dwconv_ln_fw_kernel<T, XT, C, WW, TH, SPLIT>), dim3((unsigned)(B * tiles_h)), dim3(C * SPLIT > 384 ? 768 : 384), smem, s, x, H,
                     tiles_h, bw.dw_w, bw.dw_wp, bw.dw_b, bw.ln_w, bw.ln_b, y);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---------------------------------------------------------------------------------------------
// downsample input: LayerNorm(channels_first == per-position over C, eps 1e-6) + 2x2/2 patchify
// x fp32 (B,H,W,C) -> p T (B, H/2, W/2, (kh, kw, C)); rows/cols beyond 2*floor() are dropped.
// One wave per input position.
// ---------------------------------------------------------------------------------------------
template <typename T, typename XT, int C>
__global__ __launch_bounds__(256) void cn_ln_patchify_kernel(const XT* __restrict__ x, int H, int W, long n_pos,
                                                             const float* __restrict__ ln_w,
                                                             const float* __restrict__ ln_b, T* __restrict__ p) {
  // A lane owns 4 consecutive channels (16-byte loads, 8-byte bf16 stores: one 2-byte store per lane made the
  // kernel store-issue bound, 2.4 TB/s); C = 96 packs two positions into a wave (one per 32-lane half).  All loads of
  // the PPW position groups of a wave are issued before any reduction (a group at a time was a pure latency chain).
  constexpr int PER = C > 192 ? 2 : 1;          // 4-channel chunks per lane
  constexpr int LANES = C / 4 / PER;            // active lanes per position: 24 / 48 / 48
  constexpr int SLOTS = LANES <= 32 ? 2 : 1;    // positions side by side in a wave
  constexpr int PPW = 4;
  const int lane = threadIdx.x & 63;
  const int sl = SLOTS == 2 ? lane >> 5 : 0, ll = SLOTS == 2 ? lane & 31 : lane;
  const bool act = ll < LANES;
  const long pos0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (PPW * SLOTS);
  if (pos0 >= n_pos) return;
  const int H2 = H / 2, W2 = W / 2;
  f32x4 v[PPW][PER];
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    const long pos = min(pos0 + u * SLOTS + sl, n_pos - 1);
#pragma unroll
    for (int i = 0; i < PER; ++i)
      v[u][i] = act ? cn_ld4(x + (size_t)pos * C + 4 * (ll + LANES * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 gw[PER], gb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    gw[i] = act ? *(const f32x4*)(ln_w + 4 * (ll + LANES * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
    gb[i] = act ? *(const f32x4*)(ln_b + 4 * (ll + LANES * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto group_sum = [&](float s) {  // over the position's lanes: a 32-lane half or the whole wave
    s = cn_sum8_dpp(s);
    s += cn_dpp<0x140>(s);
    s += __shfl_xor(s, 16);
    if (SLOTS == 1) s += __shfl_xor(s, 32);
    return s;
  };
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    const long pos = pos0 + u * SLOTS + sl;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += (v[u][i][0] + v[u][i][1]) + (v[u][i][2] + v[u][i][3]);
    const float mean = group_sum(s) * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = act ? v[u][i][e] - mean : 0.f;
        s2 = fmaf(d, d, s2);
      }
    const float rstd = 1.0f / sqrtf(group_sum(s2) * (1.0f / C) + 1e-6f);
    if (pos >= n_pos || !act) continue;
    if (ll == 0) cn_watch_stat(rstd);
    const int w = (int)(pos % W);
    const long t = pos / W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    if (h >= 2 * H2 || w >= 2 * W2) continue;  // odd trailing row / column is dropped by the stride-2 conv
    T* o = p + ((((size_t)b * H2 + (h >> 1)) * W2 + (w >> 1)) * 4 + ((h & 1) * 2 + (w & 1))) * C;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cc = 4 * (ll + LANES * i);
      cn_store4(o + cc, (v[u][i][0] - mean) * rstd * gw[i][0] + gb[i][0], (v[u][i][1] - mean) * rstd * gw[i][1] + gb[i][1],
                (v[u][i][2] - mean) * rstd * gw[i][2] + gb[i][2], (v[u][i][3] - mean) * rstd * gw[i][3] + gb[i][3]);
    }
  }
}

// frame_embs[b][t][c] = mean over the W freq positions (convnext.py:306); also an operand-type copy
// + the last post of the overflow watch (g_cn_nonfinite above): the output of the encoder's last block has no LayerNorm behind it
template <typename T, typename XT>
__global__ __launch_bounds__(256) void cn_frame_mean_kernel(const XT* __restrict__ x, int W, int C,
                                                            float* __restrict__ fe, T* __restrict__ fe_t) {
  const size_t bt = blockIdx.x;
  bool bad = false;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int w = 0; w < W; ++w) s += cn_ld1(x + (bt * W + w) * C + c);
    const float m = s / (float)W;
    bad |= !(__builtin_fabsf(m) <= 3.0e38f);
    fe[bt * C + c] = m;
    if (fe_t) fe_t[bt * C + c] = cn_from_f32<T>(m);
  }
  if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) atomicAdd(&g_cn_nonfinite, 1);
}

// clip head input: max_t + mean_t -> nn.LayerNorm(768, eps 1e-6) (convnext.py:324-330)
template <typename T>
__global__ __launch_bounds__(256) void cn_clip_pool_ln_kernel(const float* __restrict__ fe, int Tn,
                                                              const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, T* __restrict__ out) {
  __shared__ float s_red[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  float v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = tid + 256 * i;
    float mx = -INFINITY, sm = 0.f;
    const float* col = fe + (size_t)b * Tn * CN_FEAT + c;
    int t = 0;
    for (; t + 8 <= Tn; t += 8) {  // eight loads in flight (one at a time this loop was a latency chain); same order of the sum
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = col[(size_t)(t + u) * CN_FEAT];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        mx = fmaxf(mx, a[u]);
        sm += a[u];
      }
    }
    for (; t < Tn; ++t) {
      const float a = col[(size_t)t * CN_FEAT];
      mx = fmaxf(mx, a);
      sm += a;
    }
    v[i] = mx + sm / (float)Tn;
    s += v[i];
  }
  s = cn_wave_sum(s);
  if ((tid & 63) == 0) s_red[tid >> 6] = s;
  __syncthreads();
  const float mean = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) * (1.0f / CN_FEAT);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float d = v[i] - mean;
    s2 = fmaf(d, d, s2);
  }
  s2 = cn_wave_sum(s2);
  if ((tid & 63) == 0) s_red[4 + (tid >> 6)] = s2;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((s_red[4] + s_red[5] + s_red[6] + s_red[7]) * (1.0f / CN_FEAT) + 1e-6f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = tid + 256 * i;
    out[(size_t)b * CN_FEAT + c] = cn_from_f32<T>((v[i] - mean) * rstd * ln_w[c] + ln_b[c]);
  }
}

// ---------------------------------------------------------------------------------------------
// orchestration
// ---------------------------------------------------------------------------------------------
#ifndef CN_SP_NW96
#define CN_SP_NW96 12  // waves per block / ring depth of the exact-precision fused MLP at C = 96
#define CN_SP_NST96 4
#endif

struct EncGeom {
  int F, H[4], W[4];
};
static EncGeom enc_geom(int L) {
  EncGeom g;
  g.F = L / CN_HOP + 1;
  g.H[0] = (g.F + 8 - 4) / 4 + 1;
  g.W[0] = 56;
  for (int i = 1; i < 4; ++i) {
    g.H[i] = g.H[i - 1] / 2;
    g.W[i] = g.W[i - 1] / 2;
  }
  return g;
}

extern "C" int32_t conette_num_frames(int32_t n_samples) { return n_samples / CN_HOP + 1; }
extern "C" int32_t conette_num_audio_frames(int32_t n_samples) { return enc_geom(n_samples).H[3]; }

struct EncWs {
  float* logmel;
  void* x;   // the residual stream (XT)
  void* y;
  void* h;
  void* fe_t;
  void* clip_t;
  size_t total;
};
static EncWs enc_ws(const conette_ctx* ctx, int B, int L, char* base) {
  const EncGeom g = enc_geom(L);
  const size_t es = ctx->esize;
  const size_t n0 = (size_t)B * g.H[0] * g.W[0] * 96;
  EncWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += cn_align(bytes);
    return p;
  };
  w.logmel = (float*)take((size_t)B * g.F * CN_N_MELS * 4);
  // (+ 32 rows: the fused MLP reads whole 32-position tiles; rows past the last position are read, never written)
  w.x = take((n0 + 32 * 96) * 4);  // (sized for fp32 whatever the stream's type)
  w.y = take((n0 + 32 * 96) * es);
  w.h = take(n0 * 4 * es);
  w.fe_t = take((size_t)B * g.H[3] * CN_FEAT * es);
  w.clip_t = take((size_t)B * CN_FEAT * es);
  w.total = off;
  return w;
}

extern "C" size_t conette_encode_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t n_samples) {
  return enc_ws(ctx, batch, n_samples, nullptr).total;
}

template <typename T, typename XT>
static int dwconv_dispatch(int C, const XT* x, int B, int H, int W, const CnBlockW& bw, T* y, hipStream_t s) {
  switch (C) {
    // (the fp32 stream of the exact / fp32 precisions keeps 8-row tiles: its conv is bound by 49 fp32 multiply-adds per output)
    case 96: return launch_dwconv<T, XT, 96, CN_DW96_S, (sizeof(XT) == 2 && CN_DW_DOT2) ? CN_DW96_TH : 8>(x, B, H, W, bw, y, s);
    case 192: return launch_dwconv<T, XT, 192, CN_DW192_S, (sizeof(XT) == 2 && CN_DW_DOT2) ? CN_DW192_TH : 8>(x, B, H, W, bw, y, s);
    case 384:
      if (W == 14) return launch_dwconv_fw<T, XT, 384, 14, CN_FW_TH, CN_FW_SPLIT>(x, B, H, bw, y, s);
      return launch_dwconv<T, XT, 384, 1, 4>(x, B, H, W, bw, y, s);
    case 768:
      if (W == 7) return launch_dwconv_fw<T, XT, 768, 7, 4>(x, B, H, bw, y, s);
      return launch_dwconv<T, XT, 768, 1, 4>(x, B, H, W, bw, y, s);
  }
  cn_set_error("dwconv: unsupported C=%d", C);
  return CN_ERR_ARG;
}

static int tap_copy(float* dst, const float* src, size_t n, hipStream_t s) {
  if (dst) CN_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return CN_OK;
}
// (taps are fp32 at the C ABI whatever the residual stream's type: an fp16 stream is widened, exactly)
static __global__ void cn_tap_widen_kernel(const half_t* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
static int tap_copy(float* dst, const half_t* src, size_t n, hipStream_t s) {
  if (dst) {
    hipLaunchKernelGGL(cn_tap_widen_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, src, dst, n);
    CN_LAUNCH_CHECK();
  }
  return CN_OK;
}

// T: the GEMM operand type of the precision; XT: the residual stream's type (common.h)
template <typename T, typename XT>
static int encode_impl(conette_ctx* ctx, const float* wave, int B, int L, float* frame_embs, float* clip_probs,
                       const conette_encode_taps* taps, char* wsp, hipStream_t s) {
  const EncGeom g = enc_geom(L);
  EncWs ws = enc_ws(ctx, B, L, wsp);
  T* y = (T*)ws.y;
  T* hbuf = (T*)ws.h;
  XT* const wsx = (XT*)ws.x;
  {
    CnProfScope ps(ctx, CONETTE_PROF_FRONTEND, s);
    CN_TRY(cn_frontend(ctx, wave, B, L, ws.logmel, s));
  }
  if (taps) CN_TRY(tap_copy(taps->logmel, ws.logmel, (size_t)B * g.F * CN_N_MELS, s));
  {
    const long n_pos = (long)B * g.H[0] * g.W[0];
    CnProfScope ps(ctx, CONETTE_PROF_STEM, s);
#if CN_STEM_MFMA
    {
      const long n_tiles = (n_pos + 31) / 32;
      const unsigned grid = (unsigned)std::min<long>((n_tiles + 3) / 4, 8L * ctx->n_cu);
      hipLaunchKernelGGL((cn_stem_mfma_kernel<XT>), dim3(grid), dim3(256), 0, s, ws.logmel, g.F, g.H[0], n_pos, ctx->stem_w, ctx->stem_b,
                         ctx->stem_ln_w, ctx->stem_ln_b, wsx);
    }
#else
    hipLaunchKernelGGL((cn_stem_kernel<1, XT>), dim3((unsigned)((n_pos + 63) / 64)), dim3(256), 0, s, ws.logmel, g.F, g.H[0],
                       n_pos, ctx->stem_w, ctx->stem_b, ctx->stem_ln_w, ctx->stem_ln_b, wsx);
#endif
    CN_LAUNCH_CHECK();
    if (taps) CN_TRY(tap_copy(taps->stem, wsx, (size_t)n_pos * 96, s));
  }
  XT* xc = wsx;  // the residual stream: ws.x, or ws.h after the fused stage-1 downsample (which cannot run in place)
  int blk = 0;
  for (int st = 0; st < 4; ++st) {
    const int C = CN_DIMS[st], H = g.H[st], W = g.W[st];
    const long P = (long)B * H * W;
    if (st > 0) {
      const int Cp = CN_DIMS[st - 1], Hp = g.H[st - 1], Wp = g.W[st - 1];
      const long n_in = (long)B * Hp * Wp;
      const CnDownW& dw = ctx->down[st - 1];
      CnProfScope ps(ctx, CONETTE_PROF_DOWNSAMPLE, s);
      bool fused_down = false;
      if constexpr (CnIsH16<T>::value) if (Cp <= 192 && dw.fused != nullptr) {
        // LayerNorm + patch GEMM in one kernel (down_fused.h).  It cannot run in place: the residual stream moves to the
        // other of ws.x / ws.h (stages 0-2 run the fused MLP in bf16, so the hidden buffer is free until stage 3).
        XT* xo = xc == wsx ? (XT*)ws.h : wsx;
        const int nb = ctx->n_cu - ctx->enc_reserved_cus;
        if (Cp == 96) CN_TRY((cn_launch_down_fused<96, 8, T, XT>(xc, B, Hp, Wp, dw.fused, xo, nb, s)));
        else CN_TRY((cn_launch_down_fused_ring<192, 4, 4, 3, T, XT>(xc, B, Hp, Wp, dw.fused, xo, nb, s)));
        xc = xo;
        if (taps) CN_TRY(tap_copy(taps->down[st], xc, (size_t)P * C, s));
        fused_down = true;
      }
      if (!fused_down) {
      const int ppb = Cp == 96 ? 32 : 16;           // positions per block: 4 waves x 4 groups x (2 | 1) positions
      const dim3 pg((unsigned)((n_in + ppb - 1) / ppb));
      if (Cp == 96)
        hipLaunchKernelGGL((cn_ln_patchify_kernel<T, XT, 96>), pg, dim3(256), 0, s, xc, Hp, Wp, n_in, dw.ln_w, dw.ln_b, y);
      else if (Cp == 192)
        hipLaunchKernelGGL((cn_ln_patchify_kernel<T, XT, 192>), pg, dim3(256), 0, s, xc, Hp, Wp, n_in, dw.ln_w, dw.ln_b, y);
      else
        hipLaunchKernelGGL((cn_ln_patchify_kernel<T, XT, 384>), pg, dim3(256), 0, s, xc, Hp, Wp, n_in, dw.ln_w, dw.ln_b, y);
      CN_LAUNCH_CHECK();
      EpiBiasAct<XT, ACT_NONE> epi{dw.bias, wsx, C, ACT_NONE};
      CN_TRY(cn_mm(y, 4 * Cp, (const T*)dw.w, 4 * Cp, (int)P, C, 4 * Cp, epi, s));
      xc = wsx;
      if (taps) CN_TRY(tap_copy(taps->down[st], xc, (size_t)P * C, s));
      }
    }
    for (int b = 0; b < CN_DEPTHS[st]; ++b, ++blk) {
      const CnBlockW& bw = ctx->blocks[blk];
      {
        CnProfScope ps(ctx, CONETTE_PROF_DWCONV_LN, s);
        CN_TRY((dwconv_dispatch<T, XT>(C, xc, B, H, W, bw, y, s)));
      }
      bool fused = false;
      if constexpr (std::is_same<T, sp16_t>::value) {
        // exact precision, stages 0-1: the same fused block with fp16 hi / lo operand pairs (mlp_sp.h); timed under PW1
        if (bw.mlp_sp != nullptr && C <= 192) {
          CnProfScope ps(ctx, CONETTE_PROF_PW1_GEMM, s);
          const int nb = ctx->n_cu - ctx->enc_reserved_cus;
          if (C == 96) CN_TRY((cn_launch_mlp_sp_ring<96, CN_SP_NW96, CN_SP_NST96>(y, bw.mlp_sp, xc, (int)P, nb, s)));
          else CN_TRY((cn_launch_mlp_sp_ring<192, 8, 3>(y, bw.mlp_sp, xc, (int)P, nb, s)));
          fused = true;
        }
      }
      if constexpr (CnIsH16<T>::value) if (!fused) {
        // stages 0-2: register-chained fused MLP (mlp_rc2.h): the 4C hidden never leaves the registers; timed under PW1
        if (bw.mlp_stream != nullptr && C <= 384) {
          CnProfScope ps(ctx, CONETTE_PROF_PW1_GEMM, s);
          const T* wsm = (const T*)bw.mlp_stream;
          if (C == 96) CN_TRY((cn_launch_mlp_rc2_resident<96, 12, 1>(y, wsm, xc, (int)P, ctx->n_cu - ctx->enc_reserved_cus, s)));
          else if (C == 192) CN_TRY((cn_launch_mlp_rc2_ring<192, 8, CN_RC2_NCK(192), CN_RC2_NCK(192) == 2 ? 3 : 5>(y, wsm, xc, (int)P, ctx->n_cu - ctx->enc_reserved_cus, s)));
#ifdef CN_NO_RS  // A/B builds only (tools/lab/ab.sh): round 2's chained kernel at stage 2
          else CN_TRY((cn_launch_mlp_rc2_ring<384, 4, 1, 3>(y, wsm, xc, (int)P, ctx->n_cu - ctx->enc_reserved_cus, s)));
#else
          // (8 = the B waves -- GEMM2 + ring refill, the younger half of the block -- run at s_setprio 1: 143 against 148 us in the
          // lab with the fp16 residual stream, profiles/r05_notes.md; the MI355X guide's "static priority for the younger half")
          // 16-bit residual stream: the same pipeline on 16x16x32 MFMAs (mlp_rs16.h; 125 against 132 us, profiles/r05_notes.md section 8);
          // the stream was packed for the kernel that runs (api.hip)
          else if constexpr (sizeof(XT) == 2 && CN_RS16) CN_TRY((cn_launch_mlp_rs16<384, 4, 3, CN_RS_PRIO>(y, wsm, xc, (int)P, ctx->n_cu - ctx->enc_reserved_cus, s)));
          else CN_TRY((cn_launch_mlp_rs<384, 4, 3, CN_RS_PRIO>(y, wsm, xc, (int)P, ctx->n_cu - ctx->enc_reserved_cus, s)));
#endif
          fused = true;
        }
      }
      if (!fused) {
        {
          CnProfScope ps(ctx, CONETTE_PROF_PW1_GEMM, s);
#ifdef CN_G2_NOACT  // timing experiment only (wrong results): how much of the epilogue is the activation?
          constexpr int kAct = ACT_NONE;
#else
          constexpr int kAct = CnGeluAct<T>::value;
#endif
          EpiBiasAct<T, kAct> e1{bw.b1, hbuf, 4 * C, kAct};
          CN_TRY(cn_mm(y, C, (const T*)bw.w1, C, (int)P, 4 * C, C, e1, s));
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_PW2_GEMM, s);
          EpiResidT<XT> e2{bw.b2, bw.scale, xc, xc, C};
          CN_TRY(cn_mm(hbuf, 4 * C, (const T*)bw.w2, 4 * C, (int)P, C, 4 * C, e2, s));
        }
      }
      if (taps && b == 0) CN_TRY(tap_copy(taps->stage_block0[st], xc, (size_t)P * C, s));
      if (taps && taps->struct_bytes >= offsetof(conette_encode_taps, block) + (size_t)(blk + 1) * sizeof(float*))
        CN_TRY(tap_copy(taps->block[blk], xc, (size_t)P * C, s));
    }
    if (taps) CN_TRY(tap_copy(taps->stage[st], xc, (size_t)P * C, s));
  }
  const int Tn = g.H[3];
  CnProfScope ps_heads(ctx, CONETTE_PROF_HEADS, s);
  hipLaunchKernelGGL((cn_frame_mean_kernel<T, XT>), dim3((unsigned)(B * Tn)), dim3(256), 0, s, xc, g.W[3], CN_FEAT,
                     frame_embs, (T*)nullptr);
  CN_LAUNCH_CHECK();
  if (clip_probs) {
    hipLaunchKernelGGL((cn_clip_pool_ln_kernel<T>), dim3((unsigned)B), dim3(256), 0, s, frame_embs, Tn, ctx->norm_w,
                       ctx->norm_b, (T*)ws.clip_t);
    CN_LAUNCH_CHECK();
    EpiBiasAct<float> eh{ctx->head_b, clip_probs, CN_N_TAGS, ACT_SIGMOID};
    CN_TRY(cn_mm((const T*)ws.clip_t, CN_FEAT, (const T*)ctx->head_w, CN_FEAT, B, CN_N_TAGS, CN_FEAT, eh, s));
  }
  return CN_OK;
}

extern "C" int conette_encode_nonfinite(conette_ctx* ctx, void* stream, int32_t* count) {
  if (!ctx || !count) {
    cn_set_error("encode_nonfinite: bad argument");
    return CN_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  int32_t host = 0;
  CN_HIP(hipMemcpyFromSymbolAsync(&host, HIP_SYMBOL(g_cn_nonfinite), sizeof(host), 0, hipMemcpyDeviceToHost, s));
  CN_HIP(hipStreamSynchronize(s));
  if (host != 0) {
    const int32_t zero = 0;
    CN_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_cn_nonfinite), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, s));
    CN_HIP(hipStreamSynchronize(s));
  }
  *count = host;
  return CN_OK;
}

extern "C" int conette_frontend_logmel(conette_ctx* ctx, const float* wave, int32_t batch, int32_t n_samples,
                                       float* out, void* stream) {
  if (!ctx || !wave || !out || batch <= 0) {
    cn_set_error("frontend_logmel: bad argument");
    return CN_ERR_ARG;
  }
  if (ctx->no_encoder) {
    cn_set_error("frontend_logmel: decoder-only context (created without preprocessor.encoder.* tensors)");
    return CN_ERR_ARG;
  }
  return cn_frontend(ctx, wave, batch, n_samples, out, (hipStream_t)stream);
}

extern "C" int conette_encode(conette_ctx* ctx, const float* wave, int32_t batch, int32_t n_samples,
                              float* frame_embs, float* clip_probs, const conette_encode_taps* taps, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (conette_num_audio_frames(n_samples) < 1) {  // (checked first: the caller's (B, 0, 768) output has no storage)
    cn_set_error("encode: n_samples=%d is too short: the encoder's three 2x downsamplings need >= 7680 samples "
                 "(0.24 s at 32 kHz) to leave one audio frame", n_samples);
    return CN_ERR_ARG;
  }
  if (!ctx || !wave || !frame_embs || !workspace || batch <= 0) {
    cn_set_error("encode: bad argument");
    return CN_ERR_ARG;
  }
  if (ctx->no_encoder) {
    cn_set_error("encode: decoder-only context (created without preprocessor.encoder.* tensors: a BaselinePLM-layout checkpoint "
                 "takes precomputed frame embeddings)");
    return CN_ERR_ARG;
  }
  if (taps && taps->struct_bytes < offsetof(conette_encode_taps, block)) {
    cn_set_error("encode: conette_encode_taps.struct_bytes = %zu is smaller than the version-1 members (%zu): set it to "
                 "sizeof(conette_encode_taps)", taps->struct_bytes, offsetof(conette_encode_taps, block));
    return CN_ERR_ARG;
  }
  const size_t need = conette_encode_workspace_bytes(ctx, batch, n_samples);
  if (workspace_bytes < need) {
    cn_set_error("encode: workspace %zu < %zu", workspace_bytes, need);
    return CN_ERR_WORKSPACE;
  }
  // the residual stream's type: fp16 in the 16-bit precisions (round 5), fp32 in the others
  switch (ctx->cfg.precision) {
    case CONETTE_PREC_BF16:
      return encode_impl<bf16_t, half_t>(ctx, wave, batch, n_samples, frame_embs, clip_probs, taps, (char*)workspace, (hipStream_t)stream);
    case CONETTE_PREC_F16:
      return encode_impl<half_t, half_t>(ctx, wave, batch, n_samples, frame_embs, clip_probs, taps, (char*)workspace, (hipStream_t)stream);
    case CONETTE_PREC_F16X2:
      return encode_impl<sp16_t, float>(ctx, wave, batch, n_samples, frame_embs, clip_probs, taps, (char*)workspace, (hipStream_t)stream);
    default:
      return encode_impl<float, float>(ctx, wave, batch, n_samples, frame_embs, clip_probs, taps, (char*)workspace, (hipStream_t)stream);
  }
}
