// EXPERIMENT (round 5; not part of the library -- tools/lab/dwm_lab.hip is its harness; result: profiles/r05_notes.md section 4).
// Depthwise 7x7 convolution on the MATRIX cores + LayerNorm over C (stages 0-1 of the 16-bit precisions, whose residual
// stream is fp16):   y = LN_C( dwconv7x7(x) )                                    (convnext.py:61-66; pad 3, eps 1e-6)
//
// A depthwise convolution has no contraction over channels, so a dense MFMA cannot carry it -- but v_mfma_f32_4x4x4_16B_f16 is
// SIXTEEN independent 4 x 4 x 4 products per instruction (block b = lanes 4 b .. 4 b + 3; tools/lab/mfma4_probe.hip: A lane
// (b, i) holds A_b[i][0..3], B lane (b, j) holds B_b[0..3][j], D lane (b, j) register i = D_b[i][j]; 11 cycles back to back,
// 40 dependent).  With block = CHANNEL the 7 x 7 stencil becomes, per kernel row kh, a banded (Toeplitz) product along w:
//     out[h0 + i][w0 + j][c] += sum_k  x[h0 + i + kh - 3][w0 - 3 + 4 q + k][c]  *  w[c][kh][4 q + k - j]        q = 0, 1, 2
// -- A = 4 rows x 4 consecutive input columns of one channel, B = a 4 x 4 Toeplitz slice of the kernel row (zero where
// 4 q + k - j is outside 0..6), D = a 4 x 4 output block.  21 MFMAs (7 kh x 3 q) give 16 outputs x 16 channels: 0.9 matrix-pipe
// cycles per output against 49 v_fma_f32 = 3.06 issue cycles on the VALU (28 of the 48 weight slots of a Toeplitz slice triple
// are non-zero: the matrix pipe does 1.7x the arithmetic and is still 3.4x faster).
//
// The A operand wants 4 consecutive w of ONE channel per lane, the stream is channels-last: a block first TRANSPOSES its
// halo tile ((TH + 6) x (TW + 6) positions x CP channels of a pass) into channel planes in LDS -- 16-byte global loads of two
// neighbouring positions, eight v_perm_b32, eight ds_write_b32 of (x[w][c], x[w + 1][c]) pairs -- and the MFMA phase reads
// one ds_read_b64 per A operand.  Plane / row pitches (16 dwords per row, rows x 16 + 2 per plane) make the reads conflict
// free and the writes 2-way (= their register-transfer time); tools/lab/lds_sim_dwconv.py is the search.
//
// A wave owns one UNIT per pass: 16 channels x 4 rows x TW columns (TW / 4 accumulators, independent MFMA chains); the B
// fragments of its channel group (21 x 8 bytes per lane, packed at create time: pk_dw_toeplitz) sit in 42 registers.  After the
// last pass the accumulators go to an fp32 [position][C] tile that ALIASES the planes, and LayerNorm + the 16-byte stores of y
// run as in cn_dwconv_ln_kernel (encoder.hip).  fp16 inputs are exact MFMA operands, the weights are rounded to fp16 (the
// only difference from the VALU kernel's arithmetic: |dw| 2^-12 relative, listed among the rounding points of oracle/bf16_ref.py),
// products and sums are fp32.
#pragma once
#include "common.h"

typedef _Float16 cn_f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// Toeplitz fragments: dst[group g = c / 16][kh 7][q 3][lane 64][4 halves]: lane (b = l >> 2, j = l & 3), element k = w[c = 16 g + b][kh][4 q + k - j]
static __global__ void pk_dw_toeplitz(const float* __restrict__ dw_w /*[49][C]*/, int C, half_t* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (C / 16) * 21 * 64) return;
  const int l = u & 63, f = (u >> 6) % 21, g = (u >> 6) / 21;
  const int kh = f / 3, q = f % 3, c = 16 * g + (l >> 2), j = l & 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int kw = 4 * q + k - j;
    dst[(size_t)u * 4 + k] = (half_t)((kw >= 0 && kw <= 6) ? dw_w[(kh * 7 + kw) * C + c] : 0.0f);
  }
}
static inline size_t cn_dw_toeplitz_bytes(int C) { return (size_t)(C / 16) * 21 * 64 * 8; }

template <int C, int TH, int TW, int NPASS> struct DwmGeom {
  static_assert(TW == 16 && TH % 4 == 0 && C % (16 * NPASS) == 0, "tile geometry");
  static constexpr int R = TH + 6;                 // halo rows
  static constexpr int PP = (TW + 8) / 2;          // column pairs written per row (columns TW + 6, TW + 7 are zero fill)
  static constexpr int PWD = 16;                   // dwords per plane row (12 used): the conflict-free pitch of the ds_read_b64 pattern
  static constexpr int CHS = R * PWD + 2;          // dwords per channel plane
  static constexpr int CP = C / NPASS, CGP = CP / 16, RB = TH / 4, NBW = TW / 4, UNITS = CGP * RB;
  static constexpr int NPOS = TH * TW;
  static constexpr int NCHUNK = C / 4, PITCH = (NCHUNK | 1) * 4;   // LayerNorm tile pitch (words): DwTile of encoder.hip
};

// ABL (kernel lab only, wrong results): 1 = no global loads, 2 = no MFMA phase, 4 = no LayerNorm / store, 8 = no transposition stores
template <typename T, int C, int TH, int TW, int NW, int NPASS, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void cn_dwconv_mfma_ln_kernel(const half_t* __restrict__ x, int H, int W, int tiles_h, int tiles_w,
                                                                    const half_t* __restrict__ frag, const float* __restrict__ dw_b,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                                    T* __restrict__ y) {
  typedef DwmGeom<C, TH, TW, NPASS> G;
  static_assert(G::UNITS == NW, "one unit (16 channels x 4 rows x TW columns) per wave and pass");
  constexpr int NT = NW * 64, R = G::R, PP = G::PP, PWD = G::PWD, CHS = G::CHS, CP = G::CP, CGP = G::CGP, NBW = G::NBW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned* xs = (unsigned*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % tiles_w;
  bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, w0 = tw * TW;
  const half_t* xb = x + (size_t)b * H * W * C;

  const int gl = wave % CGP, rb = wave / CGP;       // this wave's unit: channel group within the pass, row band
  f32x4 acc[NPASS][NBW];

#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    // ---- the unit's Toeplitz fragments (global, L2 resident) and bias: requested first, used after the transposition ----
    u32x2 bf[21];
    {
      const u32x2* fp = (const u32x2*)frag + ((size_t)(pass * CGP + gl) * 21) * 64 + lane;
#pragma unroll
      for (int f = 0; f < 21; ++f) bf[f] = fp[f * 64];
    }
    const float bias = dw_b[pass * CP + 16 * gl + (lane >> 2)];
    // ---- phase 1: halo tile of the pass's CP channels -> channel planes (pairs of columns per dword) -------------------------
    if (pass > 0) __syncthreads();                 // every wave has finished reading the previous pass's planes
    {
      constexpr int OP = CP / 8, OL = OP % 4 == 0 ? 4 : 2, OH = OP / OL, ITEMS = R * OH * PP * OL;   // OL octets = 64 (32) contiguous bytes per position
      static_assert(OP % OL == 0, "octets per pass");
      // every load of the thread is in flight before the first permute (a load -> permute -> store loop paid the memory latency
      // once per item: 4 round trips per block)
      constexpr int NI = (ITEMS + NT - 1) / NT;
      u32x4 va[NI], vb[NI];
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int t = tid + it * NT;
        const int ol = t % OL, pp = (t / OL) % PP, oh = (t / (OL * PP)) % OH, r = t / (OL * PP * OH);
        const int o = oh * OL + ol;
        const int h = h0 - 3 + r, cc = 2 * pp, wa = w0 - 3 + cc;
        va[it] = u32x4{0u, 0u, 0u, 0u};
        vb[it] = u32x4{0u, 0u, 0u, 0u};
        if (!(ABL & 1) && t < ITEMS && h >= 0 && h < H) {
          const half_t* src = xb + ((size_t)h * W + wa) * C + pass * CP + 8 * o;
          if (cc < TW + 6 && wa >= 0 && wa < W) va[it] = *(const u32x4*)src;
          if (cc + 1 < TW + 6 && wa + 1 >= 0 && wa + 1 < W) vb[it] = *(const u32x4*)(src + C);
        }
      }
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int t = tid + it * NT;
        if (!(ABL & 8) && t < ITEMS) {
          const int ol = t % OL, pp = (t / OL) % PP, oh = (t / (OL * PP)) % OH, r = t / (OL * PP * OH);
          unsigned* dst = xs + (8 * (oh * OL + ol)) * CHS + r * PWD + pp;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            dst[e * CHS] = __builtin_amdgcn_perm(vb[it][e >> 1], va[it][e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
        }
      }
    }
    __syncthreads();
    // ---- phase 2: 7 kernel rows x (NBW + 2) column chunks; chunk m feeds output blocks m, m - 1, m - 2 (q = 0, 1, 2) ---------
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[pass][nb] = f32x4{bias, bias, bias, bias};
    const unsigned* ap = xs + (16 * gl + (lane >> 2)) * CHS + (4 * rb + (lane & 3)) * PWD;
    if constexpr (!(ABL & 2))
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
      for (int m = 0; m < NBW + 2; ++m) {
        const u32x2 av = *(const u32x2*)(ap + kh * PWD + 2 * m);
        const cn_f16x4 a = __builtin_bit_cast(cn_f16x4, av);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int nb = m - q;
          if (nb >= 0 && nb < NBW)
            acc[pass][nb] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, __builtin_bit_cast(cn_f16x4, bf[kh * 3 + q]), acc[pass][nb], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();   // the planes are dead: the LayerNorm tile takes their place

  // ---- conv results -> fp32 tile [position][C]; position = (row, column) of the TH x TW tile --------------------------------
  constexpr int NPOS = G::NPOS, PITCH = G::PITCH;
  float* s_v = (float*)smem_raw;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int c = pass * CP + 16 * gl + (lane >> 2);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) s_v[((4 * rb + i) * TW + 4 * nb + (lane & 3)) * PITCH + c] = acc[pass][nb][i];
  }
  __syncthreads();

  if constexpr (ABL & 4) {
    if (s_v[tid] == 123.456f) y[tid] = cn_from_f32<T>(s_v[tid + 1]);
    return;
  }
  // ---- LayerNorm over C: statistics with lane = position (PARTS threads per position, two-pass), then normalise + store
  // 8 consecutive channels per item (cn_dwconv_ln_kernel of encoder.hip: same tile layout, same order of the sums) ------------
  constexpr int PARTS = NT / NPOS, NCHUNK = C / 4, CPT = NCHUNK / PARTS;
  static_assert(NT % NPOS == 0 && NCHUNK % PARTS == 0, "LayerNorm thread mapping");
  float* s_ps = s_v + NPOS * PITCH;       // [PARTS][NPOS]
  float* s_mean = s_ps + NPOS * PARTS;    // [NPOS]
  float* s_rstd = s_mean + NPOS;          // [NPOS]
  {
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
  constexpr int C8 = C / 8;
  static_assert(NT % C8 == 0, "a thread keeps its 8 channels over all its items: the LN affine is loaded once");
  const int c8 = (tid % C8) * 8;
  const f32x4 lw0 = *(const f32x4*)(ln_w + c8), lw1 = *(const f32x4*)(ln_w + c8 + 4);
  const f32x4 lb0 = *(const f32x4*)(ln_b + c8), lb1 = *(const f32x4*)(ln_b + c8 + 4);
  for (int item = tid; item < NPOS * C8; item += NT) {
    const int pos = item / C8;
    const int h = h0 + pos / TW, w = w0 + pos % TW;
    if (h >= H || w >= W) continue;
    const float mean = s_mean[pos], rstd = s_rstd[pos];
    const f32x4 v0 = *(const f32x4*)(s_v + pos * PITCH + c8), v1 = *(const f32x4*)(s_v + pos * PITCH + c8 + 4);
    T* dst = y + (((size_t)b * H + h) * W + w) * C + c8;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = ((i < 4 ? v0[i] : v1[i - 4]) - mean) * rstd * (i < 4 ? lw0[i] : lw1[i - 4]) + (i < 4 ? lb0[i] : lb1[i - 4]);
    cn_store8(dst, o);
  }
}

template <typename T, int C, int TH, int TW, int NW, int NPASS, int ABL = 0>
static int cn_launch_dwconv_mfma(const half_t* x, int B, int H, int W, const half_t* frag, const float* dw_b, const float* ln_w,
                                 const float* ln_b, T* y, hipStream_t s) {
  typedef DwmGeom<C, TH, TW, NPASS> G;
  const int tiles_h = cn_cdiv(H, TH), tiles_w = cn_cdiv(W, TW);
  constexpr size_t planes = (size_t)G::CP * G::CHS * 4;
  constexpr size_t tile = ((size_t)G::NPOS * G::PITCH + (size_t)G::NPOS * ((NW * 64) / G::NPOS) + 2 * G::NPOS) * 4;
  constexpr size_t smem = planes > tile ? planes : tile;
  static_assert(smem <= 160 * 1024, "LDS");
  CN_TRY(cn_configure_lds((const void*)cn_dwconv_mfma_ln_kernel<T, C, TH, TW, NW, NPASS, ABL>, (int)smem));
  hipLaunchKernelGGL((cn_dwconv_mfma_ln_kernel<T, C, TH, TW, NW, NPASS, ABL>), dim3((unsigned)(B * tiles_h * tiles_w)), dim3(NW * 64), smem, s, x,
                     H, W, tiles_h, tiles_w, frag, dw_b, ln_w, ln_b, y);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
