// EXPERIMENT (kernel lab only, not part of the library): see profiles/r02_notes.md.
// Depthwise 7x7 convolution (pad 3) + LayerNorm over C (eps 1e-6), sliding-window form:
//     x fp32 (B, H, W, C)  ->  y T (B, H, W, C)                     (reference convnext.py:62-66, Block.dwconv + Block.norm)
//
// A block owns HL output rows x the whole width x all channels of one clip; a thread owns ONE channel of an 8-column strip
// and walks down the rows.  Per input row it loads the 14 columns of its window once (lanes = channels: every load is a
// coalesced line) and feeds 7 x 8 x 7 multiply-adds into SEVEN rows of accumulators -- slot s holds the output row that
// still needs kernel rows s+1 .. 6 -- so every input element is loaded 14 / 8 times instead of the 10x (tiled kernel) or
// 2.5x (full-width kernel) of the patch kernels, and the arithmetic is packed: two neighbouring output columns per
// v_pk_fma_f32 with the tap weight broadcast (98 + 98 instead of 392 FMA instructions per row).  The slots shift for free:
// the first multiply-add into slot s+1 takes slot s as its addend.
// A finished row goes into an LDS tile [row][column][C] (double buffered: ONE barrier per R rows); every R rows the block
// normalises them: a group of C / 12 lanes (8 / 16 / 32 / 64) owns a position, three 16-byte chunks per lane, two-pass mean /
// variance with DPP sums inside the group, normalised values stored straight from the registers (8-byte bf16 stores).
//
// Left / right zero padding: per-lane column masks (loop invariant); rows above / below the map: block-uniform skip.
// x must be readable 3 C floats before its first and after its last element (masked lanes still issue their loads).
#pragma once
#include "common.h"

// d = a * k + c on two columns at once, k = the low (hi = 0) or high (hi = 1) half of kk for BOTH columns.  (Written as
// f32x2{k, k} the compiler materialises every broadcast in a register pair: 98 registers of weights instead of 50.)
static __device__ __forceinline__ f32x2 cn_pk_fma_bk(f32x2 a, f32x2 kk, f32x2 c, int hi) {
  f32x2 d;
  if (hi) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(kk), "v"(c));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(kk), "v"(c));
  return d;
}

template <int C, int R, int W8> struct DwSlide {  // W8 = number of 8-column strips
  static constexpr int NT = C * W8;       // threads
  static constexpr int WP = W8 * 8;       // columns of the LDS tile
  static constexpr int LPP = C / 12;      // lanes per position in the LayerNorm phase (12 channels = 3 chunks of 4 per lane)
  static constexpr int NU = NT / LPP;     // position units of the block
  static constexpr size_t SMEM = ((size_t)2 * R * WP * C + 2 * C) * sizeof(float);
  static_assert(LPP == 8 || LPP == 16 || LPP == 32 || LPP == 64, "C = 96 / 192 / 384 / 768");
};

// sum over each aligned group of LPP lanes (result in every lane of the group)
template <int LPP> static __device__ __forceinline__ float cn_group_sum(float v) {
  v += cn_dpp<0xB1>(v);                       // xor 1
  v += cn_dpp<0x4E>(v);                       // xor 2
  v += cn_dpp<0x141>(v);                      // row_half_mirror: pairs with the other 4 of each 8
  if constexpr (LPP >= 16) v += cn_dpp<0x140>(v);  // row_mirror: the other 8 of each 16
  if constexpr (LPP >= 32) v += __shfl_xor(v, 16);
  if constexpr (LPP >= 64) v += __shfl_xor(v, 32);
  return v;
}

template <typename T, int C, int R, int W8>
__global__ __launch_bounds__(C * W8) void cn_dwconv_ln_slide_kernel(const float* __restrict__ x, int H, int W, int HL, int tiles_h,
                                                                    const float* __restrict__ dw_w /*[49][C]*/,
                                                                    const float* __restrict__ dw_b, const float* __restrict__ ln_w,
                                                                    const float* __restrict__ ln_b, T* __restrict__ y) {
  typedef DwSlide<C, R, W8> G;
  constexpr int NT = G::NT, WP = G::WP, LPP = G::LPP, NU = G::NU;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_v = (float*)smem_raw;          // [2][R][WP][C]
  float* s_g = s_v + 2 * R * WP * C;      // LayerNorm weight, bias
  const int tid = threadIdx.x;
  const int c = tid % C, strip = tid / C;
  int bid = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int th = bid % tiles_h, b = bid / tiles_h;
  const int h0 = th * HL, nrow = min(HL, H - h0);  // output rows h0 .. h0 + nrow - 1
  for (int i = tid; i < 2 * C; i += NT) s_g[i] = i < C ? ln_w[i] : ln_b[i - C];

  f32x2 kk[25];  // taps 2m, 2m + 1 (tap t = 7 i + j); the multiply-adds pick their half with op_sel
#pragma unroll
  for (int m = 0; m < 25; ++m) kk[m] = f32x2{dw_w[(2 * m) * C + c], m < 24 ? dw_w[(2 * m + 1) * C + c] : 0.f};
  const float bias = dw_b[c];
  const int w0 = strip * 8 - 3;  // column of window element 0
  bool ok[14];
#pragma unroll
  for (int q = 0; q < 14; ++q) ok[q] = (w0 + q >= 0) && (w0 + q < W);

  const float* xb = x + (size_t)b * H * W * C + (long)w0 * C + c;  // window element 0 of row 0 (this lane)
  auto load_row = [&](int hh, float (&v)[14]) {
#ifdef CN_DW_ABL_NOLOAD
    if (hh == -12345) {
#else
    if (hh >= 0 && hh < H) {  // block-uniform
#endif
      const float* xr = xb + (size_t)hh * W * C;
#pragma unroll
      for (int q = 0; q < 14; ++q) v[q] = xr[q * C];
    } else {
#pragma unroll
      for (int q = 0; q < 14; ++q) v[q] = 0.f;
    }
  };

  f32x2 acc[7][4];  // slot s, column pair p: output row (current input row - s), columns 2p, 2p + 1
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[s][p] = f32x2{0.f, 0.f};

  // LayerNorm phase: unit = group of LPP lanes, lane ul of it owns the 16-byte chunks ul, ul + LPP, ul + 2 LPP of a position
  const int unit = tid / LPP, ul = tid % LPP;

  float nxt[14];
  load_row(h0 - 3, nxt);
  const int n_in = nrow + 6;
#pragma unroll 1
  for (int r = 0; r < n_in; ++r) {
    // window as register pairs: ve[m] = columns (2m, 2m+1), vo[m] = columns (2m+1, 2m+2)
    f32x2 ve[7], vo[6];
#pragma unroll
    for (int m = 0; m < 7; ++m) ve[m] = f32x2{ok[2 * m] ? nxt[2 * m] : 0.f, ok[2 * m + 1] ? nxt[2 * m + 1] : 0.f};
#pragma unroll
    for (int m = 0; m < 6; ++m) vo[m] = f32x2{ve[m][1], ve[m + 1][0]};
    if (r + 1 < n_in) load_row(h0 - 3 + r + 1, nxt);
    // slots, oldest first: slot s+1 <- slot s + kernel row s+1 (x) this input row; slot 0 <- bias + kernel row 0
#ifdef CN_DW_ABL_NOFMA
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[6][p] += ve[p] + vo[p] + ve[p + 3];
    if (false)
#endif
#pragma unroll
    for (int s = 6; s >= 0; --s) {
      f32x2 a[4];  // the four column pairs are independent chains
#pragma unroll
      for (int p = 0; p < 4; ++p) a[p] = s == 0 ? f32x2{bias, bias} : acc[s - 1][p];
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int q = 2 * p + j, t = s * 7 + j;
          a[p] = cn_pk_fma_bk(q & 1 ? vo[q >> 1] : ve[q >> 1], kk[t >> 1], a[p], t & 1);
        }
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[s][p] = a[p];
    }
    const int o = r - 6;  // output row completed by this input row (slot 6)
    if (o >= 0) {
      const int round = o / R, rr_w = o - round * R;
      float* tile = s_v + (round & 1) * (R * WP * C);
      float* dst = tile + (rr_w * WP + strip * 8) * C + c;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        dst[(2 * p) * C] = acc[6][p][0];
        dst[(2 * p + 1) * C] = acc[6][p][1];
      }
#ifdef CN_DW_ABL_NOLN
      if (false) {
#else
      if (rr_w == R - 1 || o + 1 == nrow) {
#endif
        const int npos = (rr_w + 1) * W, o0 = o - rr_w;  // rows o0 .. o of this round, valid columns only
        __syncthreads();  // (the only barrier of the round: the tile is double buffered)
        for (int p = unit; p < npos; p += NU) {
          const int rr = p / W, w = p - rr * W;
          const f32x4* row = (const f32x4*)(tile + (rr * WP + w) * C);
          f32x4 v[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) v[i] = row[ul + i * LPP];
          const f32x4 t4 = v[0] + v[1] + v[2];
          const float mean = cn_group_sum<LPP>((t4[0] + t4[1]) + (t4[2] + t4[3])) * (1.0f / C);
          f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            v[i] -= mean;
            q4 += v[i] * v[i];
          }
          const float var = cn_group_sum<LPP>((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / C);
          const float rstd = 1.0f / sqrtf(var + 1e-6f);
          T* dsty = y + (((size_t)b * H + (h0 + o0 + rr)) * W + w) * C;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int ch = (ul + i * LPP) * 4;
            const f32x4 g4 = *(const f32x4*)(s_g + ch), b4 = *(const f32x4*)(s_g + C + ch);
            const f32x4 ov = v[i] * rstd * g4 + b4;
            cn_store4(dsty + ch, ov[0], ov[1], ov[2], ov[3]);
          }
        }
      }
    }
  }
}

template <typename T, int C, int R, int W8>
static int cn_launch_dwconv_slide(const float* x, int B, int H, int W, int HL, const float* dw_w, const float* dw_b, const float* ln_w,
                                  const float* ln_b, T* y, hipStream_t s) {
  typedef DwSlide<C, R, W8> G;
  const int tiles_h = cn_cdiv(H, HL);
  CN_TRY(cn_configure_lds((const void*)cn_dwconv_ln_slide_kernel<T, C, R, W8>, (int)G::SMEM));
  hipLaunchKernelGGL((cn_dwconv_ln_slide_kernel<T, C, R, W8>), dim3((unsigned)(B * tiles_h)), dim3(G::NT), G::SMEM, s, x, H, W, HL, tiles_h,
                     dw_w, dw_b, ln_w, ln_b, y);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
