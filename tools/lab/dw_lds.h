// EXPERIMENT, NOT PART OF THE LIBRARY (round 6; measured slower than the shipped kernel: profiles/r06_notes.md section 2).
// Depthwise 7x7 + LayerNorm for the fp16 residual stream of stages 0-1 (C = 96 / 192), halo tile staged through LDS.
//
// cn_dwconv_ln_kernel (encoder.hip) fetches its input straight from global memory, one 2-byte load per lane, tap column and halo
// row: 3.75 loads per output with 12 x 4 patches, each a wave instruction that moves 128 bytes -- after the v_dot2 form took 40 %
// of the VALU instructions away the kernel sat on those loads (0.26 of the HBM rate at stage 0, neither HBM
// nor LDS busy).  Here a block first copies the halo tile of its TH x 4S outputs -- (TH + 6) x (4S + 6) positions x C channels --
// with 16-byte loads (8 channels per lane: an eighth of the load instructions, every one a full 1 KB wave access) and writes it
// to LDS ALREADY IN THE OPERAND FORM of the convolution: word [u][q][c] = (x[2u][q][c], x[2u + 1][q][c]), the fp16 row pair that
// v_dot2_f32_f16 multiplies with a pair of kernel rows (one v_perm per word, where the global-load form spent one per pair as
// well).  Rows and columns outside the map are written as zeros, so the convolution has no edge path.  The convolution then reads
// one dword per pair and tap column from LDS (lanes = consecutive channels = consecutive words: conflict free, immediate
// offsets), and runs the same 21 dot2 + 7 single taps per output in the same order: the results are bit for bit those of
// cn_dwconv_ln_kernel.  The LayerNorm phases are that kernel's (same LDS tile, same order of every sum), on an LDS tile that
// reuses the halo's space behind a barrier.
#pragma once
#include "common.h"   // (-I conette-audio-captioning_amd/csrc)

#ifndef CN_DW_LDS
#define CN_DW_LDS 1   // 0 (A/B builds): stages 0-1 of the fp16 stream on cn_dwconv_ln_kernel
#endif
// tiles (A/B builds: -DCN_DWL96_S=.. etc.): S = quads of columns, TH = rows per block of C x S threads
#ifndef CN_DWL96_S
#define CN_DWL96_S 2
#endif
#ifndef CN_DWL96_TH
#define CN_DWL96_TH 12
#endif
#ifndef CN_DWL192_S
#define CN_DWL192_S 2
#endif
#ifndef CN_DWL192_TH
#define CN_DWL192_TH 6
#endif

#ifndef CN_DWL_ABL
#define CN_DWL_ABL 0   // lab builds (tools/lab/dwl_lab.hip): 1 = no global loads, 2 = no convolution, 4 = no LayerNorm / store,
                       // 8 = no statistics (mean 0, rstd 1), 16 = no store pass, 32 = no accumulator -> LDS writes, 64 = no global stores (LDS reads + arithmetic kept)
#endif

template <int C, int S, int TH> struct DwLds {
  static constexpr int NT = C * S, TW = 4 * S, HQ = TW + 6, HR = TH + 6, NPR = HR / 2, C8 = C / 8;
  static constexpr int NP = TH * 4, NPOS = NP * S, PARTS = NT / NPOS, NCHUNK = C / 4, CPT = NCHUNK / PARTS;
  static constexpr int PITCH = ((C / 4) | 1) * 4;   // the LayerNorm tile of cn_dwconv_ln_kernel (DwTile)
  static constexpr size_t HALO_BYTES = (size_t)NPR * HQ * C * 4;
  static constexpr size_t LN_BYTES = ((size_t)NPOS * PITCH + (size_t)NPOS * PARTS + 2 * NPOS) * 4;
  static constexpr size_t BYTES = HALO_BYTES > LN_BYTES ? HALO_BYTES : LN_BYTES;
  static_assert(HR % 2 == 0 && C % 8 == 0 && C <= 384, "row pairs, 16-byte pieces, one channel pass");
  static_assert(NT % NPOS == 0 && NCHUNK % PARTS == 0 && NT % C8 == 0, "LayerNorm thread mapping");
};

template <typename T, int C, int S, int TH>
__global__ __launch_bounds__(C * S) void cn_dwconv_ln_lds_kernel(const half_t* __restrict__ x, int H, int W, int tiles_h, int tiles_w,
                                                                  const unsigned* __restrict__ dw_wp /*[42][C] fp16 row pairs*/,
                                                                  const float* __restrict__ dw_b, const float* __restrict__ ln_w,
                                                                  const float* __restrict__ ln_b, T* __restrict__ y) {
  using L = DwLds<C, S, TH>;
  constexpr int NT = L::NT, TW = L::TW, HQ = L::HQ, NPR = L::NPR, C8 = L::C8, NP = L::NP, PITCH = L::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned* s_h = (unsigned*)smem_raw;   // [NPR][HQ][C] row-pair words
  float* s_v = (float*)smem_raw;         // [NPOS][PITCH] conv results (after the halo has been consumed)
  const int tid = threadIdx.x;
  const int c = tid % C, sidx = tid / C;
  int bid = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % tiles_w;
  bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, wb = tw * TW;
  const half_t* xb = x + (size_t)b * H * W * C;

  // the thread's 42 weight pairs and its bias: requested first, needed after the staging barrier
  cn_h2 ke[3][7], ko[3][7];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      ke[a][j] = __builtin_bit_cast(cn_h2, dw_wp[(a * 7 + j) * C + c]);
      ko[a][j] = __builtin_bit_cast(cn_h2, dw_wp[(21 + a * 7 + j) * C + c]);
    }
  const float bias = dw_b[c];

  // ---- stage the halo: item = (pair row u, column q, 8 channels): two 16-byte loads -> 8 pair words -> two 16-byte LDS writes
  constexpr int NITEM = NPR * HQ * C8, NIT = (NITEM + NT - 1) / NT;
  u32x4 ra[NIT], rb[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {   // all loads first: branch free (clamped address, value masked below)
    const int it = tid + k * NT;
    const int c8 = it % C8, q = (it / C8) % HQ, u = it / (C8 * HQ);
    const int ww = min(max(wb - 3 + q, 0), W - 1);
    const int hh0 = min(max(h0 - 3 + 2 * u, 0), H - 1), hh1 = min(max(h0 - 2 + 2 * u, 0), H - 1);
    if constexpr (CN_DWL_ABL & 1) {
      ra[k] = u32x4{(unsigned)it, 1u, 2u, 3u}, rb[k] = u32x4{4u, 5u, (unsigned)hh0, (unsigned)ww};
    } else {
      ra[k] = *(const u32x4*)(xb + ((size_t)hh0 * W + ww) * C + c8 * 8);
      rb[k] = *(const u32x4*)(xb + ((size_t)hh1 * W + ww) * C + c8 * 8);
    }
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * NT;
    if (NITEM % NT != 0 && it >= NITEM) break;
    const int c8 = it % C8, q = (it / C8) % HQ, u = it / (C8 * HQ);
    const int ww = wb - 3 + q, hh0 = h0 - 3 + 2 * u, hh1 = hh0 + 1;
    const bool wok = ww >= 0 && ww < W;
    const bool ok0 = wok && hh0 >= 0 && hh0 < H, ok1 = wok && hh1 >= 0 && hh1 < H;
    u32x4 a = ra[k], bq = rb[k];
    if (!ok0) a = u32x4{0u, 0u, 0u, 0u};
    if (!ok1) bq = u32x4{0u, 0u, 0u, 0u};
    u32x4 o0, o1;
    o0[0] = __builtin_amdgcn_perm(bq[0], a[0], 0x05040100u);   // (row 2u + 1, row 2u) of channel c8 * 8 + 0
    o0[1] = __builtin_amdgcn_perm(bq[0], a[0], 0x07060302u);   //                                     + 1
    o0[2] = __builtin_amdgcn_perm(bq[1], a[1], 0x05040100u);
    o0[3] = __builtin_amdgcn_perm(bq[1], a[1], 0x07060302u);
    o1[0] = __builtin_amdgcn_perm(bq[2], a[2], 0x05040100u);
    o1[1] = __builtin_amdgcn_perm(bq[2], a[2], 0x07060302u);
    o1[2] = __builtin_amdgcn_perm(bq[3], a[3], 0x05040100u);
    o1[3] = __builtin_amdgcn_perm(bq[3], a[3], 0x07060302u);
    unsigned* dst = s_h + (u * HQ + q) * C + c8 * 8;
    *(u32x4*)dst = o0;
    *(u32x4*)(dst + 4) = o1;
  }
  __syncthreads();

  // ---- the convolution of channel c for the TH x 4 patch at columns sidx * 4: rows in pairs, two taps per v_dot2_f32_f16
  float acc[TH][4];
#pragma unroll
  for (int a = 0; a < TH; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[a][e] = bias;
  const unsigned* hp = s_h + (sidx * 4) * C + c;
#pragma unroll
  for (int u = 0; u < ((CN_DWL_ABL & 2) ? 1 : NPR); ++u) {
    cn_h2 p[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) p[q] = __builtin_bit_cast(cn_h2, hp[(u * HQ + q) * C]);
    const int r = 2 * u;   // halo rows r (even) and r + 1
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
  }
  __syncthreads();   // every thread has read its last halo word: the space becomes the LayerNorm tile
  if constexpr (!(CN_DWL_ABL & 32)) {
#pragma unroll
    for (int oh = 0; oh < TH; ++oh)
#pragma unroll
      for (int ow = 0; ow < 4; ++ow) s_v[((sidx * NP) + oh * 4 + ow) * PITCH + c] = acc[oh][ow];
  } else {
    float t = 0.f;
#pragma unroll
    for (int oh = 0; oh < TH; ++oh)
#pragma unroll
      for (int ow = 0; ow < 4; ++ow) t += acc[oh][ow];
    if (t == 123.456f) s_v[tid] = t;
  }
  __syncthreads();

  // ---- LayerNorm over C + store: the phases of cn_dwconv_ln_kernel (lane = position statistics, 8 channels per store item)
  constexpr int NPOS = L::NPOS, PARTS = L::PARTS, CPT = L::CPT;
  if constexpr (CN_DWL_ABL & 4) {
    if (s_v[tid] == 123.456f) y[tid] = (T)1.0f;
    return;
  }
  float* s_ps = s_v + NPOS * PITCH;       // [PARTS][NPOS]
  float* s_mean = s_ps + NPOS * PARTS;    // [NPOS]
  float* s_rstd = s_mean + NPOS;          // [NPOS]
  if constexpr (CN_DWL_ABL & 8) {
    if (tid < NPOS) s_mean[tid] = 0.f, s_rstd[tid] = 1.f;
    __syncthreads();
  } else {
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
    }
    __syncthreads();
  }
  const int c8s = (tid % C8) * 8;
  const f32x4 lw0 = *(const f32x4*)(ln_w + c8s), lw1 = *(const f32x4*)(ln_w + c8s + 4);
  const f32x4 lb0 = *(const f32x4*)(ln_b + c8s), lb1 = *(const f32x4*)(ln_b + c8s + 4);
  if constexpr (CN_DWL_ABL & 16) return;
  for (int item = tid; item < NPOS * C8; item += NT) {
    const int pos = item / C8;
    const int ps = pos / NP, oh = (pos % NP) >> 2, ow = pos & 3;
    const int h = h0 + oh, w = wb + ps * 4 + ow;
    if (h >= H || w >= W) continue;
    const float mean = s_mean[pos], rstd = s_rstd[pos];
    const f32x4 v0 = *(const f32x4*)(s_v + pos * PITCH + c8s), v1 = *(const f32x4*)(s_v + pos * PITCH + c8s + 4);
    T* dst = y + (((size_t)b * H + h) * W + w) * C + c8s;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = ((i < 4 ? v0[i] : v1[i - 4]) - mean) * rstd * (i < 4 ? lw0[i] : lw1[i - 4]) + (i < 4 ? lb0[i] : lb1[i - 4]);
    if constexpr (CN_DWL_ABL & 64) {
      if (o[0] + o[7] == 123.456f) cn_store8(dst, o);
    } else {
      cn_store8(dst, o);
    }
  }
}

template <typename T, int C, int S, int TH>
static int launch_dwconv_lds(const half_t* x, int B, int H, int W, const unsigned* dw_wp, const float* dw_b, const float* ln_w,
                             const float* ln_b, T* y, hipStream_t s) {
  using L = DwLds<C, S, TH>;
  const int tiles_h = cn_cdiv(H, TH), tiles_w = cn_cdiv(W, L::TW);
  CN_TRY(cn_configure_lds((const void*)cn_dwconv_ln_lds_kernel<T, C, S, TH>, (int)L::BYTES));
  hipLaunchKernelGGL((cn_dwconv_ln_lds_kernel<T, C, S, TH>), dim3((unsigned)(B * tiles_h * tiles_w)), dim3(L::NT), L::BYTES, s, x, H, W,
                     tiles_h, tiles_w, dw_wp, dw_b, ln_w, ln_b, y);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
