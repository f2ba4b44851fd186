// Lab only (not part of the library): measured NEUTRAL -- 20.1 + 12.6 us per layer against 7.6 + 4.9 + 7.6 and 7.6 + 4.9 us for the
// launches it replaces (a block of 16 rows streams 256 KB per product through one CU: ~5 us of L1 time); profiles/r03_notes.md.
// Row-complete fused step kernels for the exact precision's decoder (sp16 operands, d_model 256):
//
//   K = 1:  x = LN(x + a Wo^T + bo)                               -> x (fp32), xt (sp16)       [out-proj + residual + LayerNorm]
//   K = 2:  ... and then  q = xt Wq^T + bq                         -> q (fp32)                  [+ the cross-attention query projection]
//
// i.e. the (GEMM, LayerNorm[, GEMM]) launches that follow each attention of the one-launch-per-sub-layer decoder path
// (decoder.hip; torch's post-norm TransformerDecoderLayer, aac_tfmer.py:46-58) as ONE launch: the exact precision has no
// fused block kernel (dec_block.h keeps a whole bf16 matrix in 128 registers per wave; hi / lo pairs would need 256), its
// decode is a chain of ~70 launches of ~8.6 us per step, and these two fusions take 18 of them away.
//
// A block owns 16 rows (one MFMA M tile) and ALL 256 output columns, so the LayerNorm is block local: wave w owns columns
// 64 w .. 64 w + 63.  Nothing is staged through LDS for the products -- a wave is the only reader of its 64 weight rows, and
// the 16 activation rows are 16 KB that the four waves read straight from L1 / L2: fragments go global -> registers, split
// into hi / lo vectors by v_perm_b32, three v_mfma_f32_16x16x32_f16 per fragment pair (lo.hi + hi.lo + hi.hi).  LDS carries
// the row statistics across the waves and, for K = 2, the normalised rows as the second product's operand.
#pragma once
#include "gemm2.h"

#define RB_ROWS 16
#define RB_PITCH (256 * 4 + 32)  // bytes per row of the LDS operand tile (padded: consecutive rows start 8 banks apart)

// 8 sp16 elements at p (32 bytes) -> hi / lo fp16 vectors
__device__ __forceinline__ void rb_load_split(const sp16_t* p, f16x8& hi, f16x8& lo) {
  cn_sp_split(*(const u32x4*)p, *(const u32x4*)(p + 4), hi, lo);
}

// acc[a] (output column 64 w + 16 a + 4 (lane >> 4) + j, row lane & 15) = rows . W[64 w .. 64 w + 63]^T over K = 256.
// All 16 loads of a 16-column tile (8 k-steps x 32 bytes per lane) are in flight before its MFMAs, and the next tile's are
// issued before this tile's products: a fragment loaded right in front of its MFMAs made every k-step a round trip to L2.
struct RbTile {
  u32x4 c[8][2];
};
__device__ __forceinline__ void rb_tile_load(RbTile& t, const sp16_t* p) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    t.c[ks][0] = *(const u32x4*)(p + 32 * ks);
    t.c[ks][1] = *(const u32x4*)(p + 32 * ks + 4);
  }
}
__device__ __forceinline__ f32x4 rb_tile_mma(const RbTile& t, const f16x8 (&ah)[8], const f16x8 (&al)[8]) {
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    f16x8 wh, wl;
    cn_sp_split(t.c[ks][0], t.c[ks][1], wh, wl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[ks], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[ks], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[ks], c, 0, 0, 0);
  }
  return c;
}
// (the four tiles of a product are requested together by rb_gemm_begin, as early as the caller knows the address: one memory
// round trip per product; 256 registers of raw fragments, affordable at one wave per SIMD)
struct RbTiles {
  RbTile t[4];
};
__device__ __forceinline__ void rb_gemm_begin(const sp16_t* __restrict__ W, int wave, int lane, RbTiles& w) {
  const sp16_t* wrow = W + (size_t)(64 * wave + (lane & 15)) * 256 + 8 * (lane >> 4);
#pragma unroll
  for (int a = 0; a < 4; ++a) rb_tile_load(w.t[a], wrow + (size_t)a * 16 * 256);
}
__device__ __forceinline__ void rb_gemm(const RbTiles& w, const f16x8 (&ah)[8], const f16x8 (&al)[8], f32x4 (&acc)[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = rb_tile_mma(w.t[a], ah, al);
}

template <int K>
__global__ __launch_bounds__(256) void cn_dec_rowblock_sp_kernel(const sp16_t* __restrict__ A, const sp16_t* __restrict__ Wo,
                                                                  const float* __restrict__ bo, const float* __restrict__ lnw,
                                                                  const float* __restrict__ lnb, int R, float* __restrict__ x,
                                                                  sp16_t* __restrict__ xt, const sp16_t* __restrict__ Wq,
                                                                  const float* __restrict__ bq, float* __restrict__ q,
                                                                  const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;  // every hypothesis has finished (beam.py:192-194 stops here)
  __shared__ float s_part[2][4][RB_ROWS];
  __shared__ __attribute__((aligned(16))) char s_tile[K == 2 ? RB_ROWS * RB_PITCH : 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * RB_ROWS + m;
  const int rrow = row < R ? row : R - 1;  // padding rows of the last block recompute the last row and store nothing
  const bool live = row < R;

  // everything this block will read from memory for the first product and its epilogue is requested up front: the weight
  // tiles, the activation rows, bias + residual, the LayerNorm affine -- one memory latency instead of four in a row
  RbTiles wt;
  rb_gemm_begin(Wo, wave, lane, wt);
  f16x8 ah[8], al[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) rb_load_split(A + (size_t)rrow * 256 + 32 * ks + 8 * g, ah[ks], al[ks]);
  f32x4 ebias[4], eres[4], eg[4], eb[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = 64 * wave + 16 * a + 4 * g;
    ebias[a] = *(const f32x4*)(bo + n);
    eres[a] = *(const f32x4*)(x + (size_t)rrow * 256 + n);
    eg[a] = *(const f32x4*)(lnw + n);
    eb[a] = *(const f32x4*)(lnb + n);
  }
  f32x4 acc[4];
  rb_gemm(wt, ah, al, acc);
  if constexpr (K == 2) rb_gemm_begin(Wq, wave, lane, wt);  // the second product's weights fly during the LayerNorm

  // + bias + residual, then LayerNorm (eps 1e-5) over the row's 256 columns: this lane holds 16 of them
  f32x4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[a][j] = acc[a][j] + ebias[a][j] + eres[a][j];
      sum += v[a][j];
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if (g == 0) s_part[0][wave][m] = sum;
  __syncthreads();
  const float mean = (s_part[0][0][m] + s_part[0][1][m] + s_part[0][2][m] + s_part[0][3][m]) * (1.0f / 256.0f);
  float sq = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) sq = fmaf(v[a][j] - mean, v[a][j] - mean, sq);
  sq += __shfl_xor(sq, 16);
  sq += __shfl_xor(sq, 32);
  if (g == 0) s_part[1][wave][m] = sq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((s_part[1][0][m] + s_part[1][1][m] + s_part[1][2][m] + s_part[1][3][m]) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = 64 * wave + 16 * a + 4 * g;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[a][j] - mean) * rstd * eg[a][j] + eb[a][j];
    if (live) {
      *(f32x4*)(x + (size_t)row * 256 + n) = o;
      cn_store4(xt + (size_t)row * 256 + n, o[0], o[1], o[2], o[3]);
    }
    if constexpr (K == 2) cn_store4((sp16_t*)(s_tile + m * RB_PITCH) + n, o[0], o[1], o[2], o[3]);
  }
  if constexpr (K == 2) {
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) rb_load_split((const sp16_t*)(s_tile + m * RB_PITCH) + 32 * ks + 8 * g, ah[ks], al[ks]);
    rb_gemm(wt, ah, al, acc);
    if (live)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int n = 64 * wave + 16 * a + 4 * g;
        const f32x4 bb = *(const f32x4*)(bq + n);
        *(f32x4*)(q + (size_t)row * 256 + n) = f32x4{acc[a][0] + bb[0], acc[a][1] + bb[1], acc[a][2] + bb[2], acc[a][3] + bb[3]};
      }
  }
}
