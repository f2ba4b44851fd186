// Fused decoder "attention block" for one new token per row (bf16 operands, d_model 256, 8 heads):
//
//   a  = SelfAttn(qkv row, cached K/V of its ancestors)          (writes this step's K/V)
//   x1 = LN1(x + a Wo^T + bo)
//   c  = CrossAttn(x1 Wq^T + bq, audio K/V of the row's clip, frame mask)
//   x2 = LN2(x1 + c Wo2^T + bo2)                                   -> x (fp32) and xt (bf16)
//
// i.e. torch's post-norm TransformerDecoderLayer up to the feed-forward (aac_tfmer.py:46-58,
// 108-115).  Everything here is row-local, so a block owns DB_ROWS rows (padded to one MFMA M tile of
// 16) and keeps them on chip across all seven sub-steps; the unfused path needs 7 dependent launches for the same
// work and the decode phase is bound by per-launch latency (~5 us each, rocprof), not by bytes.
//
//   * GEMMs (16 x 256 x 256): activations are the MFMA B operand from a swizzled 8 KB LDS tile,
//     weights go straight from L2 to registers (each weight byte is used by exactly one wave of
//     one block -- no LDS reuse to exploit at M = 16; cdna guide "GEMV / M <= 16" row);
//     wave w owns output columns 64w .. 64w+63.
//   * attention: one wave per row, lane = 4 dims of one head, keys in batches of 8 (all loads of a
//     batch in flight), wave-shuffle dot products, online softmax in fp32.
//   * LayerNorm: per-lane partials -> xor-16/32 shuffles -> 4-wave LDS reduction; two-pass.
#pragma once
#include "gemm2.h"

#define DB_ROWS 4   // rows (= waves) per block: many small blocks spread the K/V and weight streams over more CUs

__device__ __forceinline__ f32x4 db_cvt4(const bf16x8& v, int hi) {
  return f32x4{(float)v[4 * hi], (float)v[4 * hi + 1], (float)v[4 * hi + 2], (float)v[4 * hi + 3]};
}

// Weight fragments of one 256 x 256 matrix for this wave's 64 output columns, straight from L2 to
// registers: fw[a][ks] = W[64w + 16a + (lane & 15)][32ks + 8(lane >> 4) .. +8].  Issued one phase
// EARLY (weights do not depend on data) so their latency hides under the attention / LayerNorm work;
// at one wave per SIMD the 512-entry register file holds them (128 VGPRs per matrix).
__device__ __forceinline__ void db_wload(const bf16_t* __restrict__ W, int wave, int lane, bf16x8 (&fw)[4][8]) {
  const bf16_t* wrow = W + (size_t)(64 * wave + (lane & 15)) * 256 + 8 * (lane >> 4);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) fw[a][ks] = *(const bf16x8*)(wrow + (size_t)a * 16 * 256 + ks * 32);
}

// out(n, m) tile-set: acc[a] holds columns n = 64*wave + 16a + 4*(lane>>4) + j of row m = lane & 15
__device__ __forceinline__ void db_gemm16(const bf16x8 (&fw)[4][8], const char* sA, int lane, f32x4 (&acc)[4]) {
  typedef G2Geom<256> G;
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const bf16x8 fa = *(const bf16x8*)(sA + lr * G::RBY + (((lq + 4 * ks) ^ (lr & G::SWM)) * 16));
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[a][ks], fa, acc[a], 0, 0, 0);
  }
}

// v[a][j] (column n, row m = lane & 15) -> LayerNorm over the 256 columns of each row (eps 1e-5).
// s_red: [2][4][16] floats.  Returns normalised * g + b in place.
__device__ __forceinline__ void db_layernorm(f32x4 (&v)[4], const float* __restrict__ g, const float* __restrict__ b,
                                             float* s_red, int wave, int lane) {
  const int lr = lane & 15, lq = lane >> 4;
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += v[a][0] + v[a][1] + v[a][2] + v[a][3];
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if (lq == 0) s_red[wave * 16 + lr] = s;
  __syncthreads();
  const float mean = (s_red[lr] + s_red[16 + lr] + s_red[32 + lr] + s_red[48 + lr]) * (1.0f / 256.0f);
  float q = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) q = fmaf(v[a][j] - mean, v[a][j] - mean, q);
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  if (lq == 0) s_red[64 + wave * 16 + lr] = q;
  __syncthreads();
  const float rstd =
      1.0f / sqrtf((s_red[64 + lr] + s_red[80 + lr] + s_red[96 + lr] + s_red[112 + lr]) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = 64 * wave + 16 * a + 4 * lq;
    const f32x4 gg = *(const f32x4*)(g + n), bb = *(const f32x4*)(b + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[a][j] = (v[a][j] - mean) * rstd * gg[j] + bb[j];
  }
}

// write v (column-n layout) as bf16 into the swizzled A tile: columns n..n+3 of row m
__device__ __forceinline__ void db_store_tile(char* sA, const f32x4 (&v)[4], int wave, int lane) {
  typedef G2Geom<256> G;
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = 64 * wave + 16 * a + 4 * lq;  // chunk n/8, half (n%8)/4
    cn_store4((bf16_t*)(sA + lr * G::RBY + ((((n >> 3)) ^ (lr & G::SWM)) * 16) + ((n >> 2) & 1) * 8), v[a][0], v[a][1],
              v[a][2], v[a][3]);
  }
}

// Wave-per-row attention: lane l owns dims 4l..4l+3 (head l >> 3); keys in batches of NB with all
// loads of a batch in flight; fp32 online softmax.  kp(s) / vp(s): this lane's 4 bf16 of key / value s.
template <int NB, class KeyPtr, class ValPtr>
__device__ __forceinline__ void db_attend(const f32x4& q, int n_keys, KeyPtr kp, ValPtr vp, float& m, float& l,
                                          f32x4& o) {
  for (int s0 = 0; s0 < n_keys; s0 += NB) {
    bf16x4 kr[NB], vr[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int s = min(s0 + u, n_keys - 1);
      kr[u] = *(const bf16x4*)kp(s);
      vr[u] = *(const bf16x4*)vp(s);
    }
    float sc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      float d = q[0] * (float)kr[u][0] + q[1] * (float)kr[u][1] + q[2] * (float)kr[u][2] + q[3] * (float)kr[u][3];
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      sc[u] = (s0 + u < n_keys) ? d : -INFINITY;
    }
    float mb = sc[0];
#pragma unroll
    for (int u = 1; u < NB; ++u) mb = fmaxf(mb, sc[u]);
    const float mn = fmaxf(m, mb);
    const float corr = __expf(m - mn);
    l *= corr;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= corr;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const float p = __expf(sc[u] - mn);
      l += p;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = fmaf(p, (float)vr[u][i], o[i]);
    }
    m = mn;
  }
}

// attention output of row `wave` (lane's 4 dims) -> bf16 -> swizzled A tile row `wave`
__device__ __forceinline__ void db_store_row(char* sA, int wave, int lane, const f32x4& o, float inv) {
  typedef G2Geom<256> G;
  cn_store4((bf16_t*)(sA + wave * G::RBY + (((lane >> 1) ^ (wave & G::SWM)) * 16) + (lane & 1) * 8), o[0] * inv,
            o[1] * inv, o[2] * inv, o[3] * inv);
}

__global__ __launch_bounds__(256, 1) void cn_dec_block_kernel(
    const float* __restrict__ qkv,                       // (R, 768) fp32: this step's q | k | v
    bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,    // self K/V cache of this layer [step][R][256]
    const int* __restrict__ anc, int step, int R, int beam, int maxp,
    const bf16_t* __restrict__ kvx, int kv_ld, int kv_off, const int* __restrict__ lens, int Ta,  // cross K/V
    const bf16_t* __restrict__ Wo, const float* __restrict__ bo, const float* __restrict__ g1,
    const float* __restrict__ b1, const bf16_t* __restrict__ Wq, const float* __restrict__ bq,
    const bf16_t* __restrict__ Wo2, const float* __restrict__ bo2, const float* __restrict__ g2,
    const float* __restrict__ b2, float* __restrict__ x /* in: residual, out: x2 */, bf16_t* __restrict__ xt,
    float scale) {
  typedef G2Geom<256> G;
  __shared__ __attribute__((aligned(16))) char sA[16 * 512];   // MFMA M tile: rows >= DB_ROWS are padding
  __shared__ __attribute__((aligned(16))) float sQ[DB_ROWS * 256];
  __shared__ float s_red[128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * DB_ROWS;
  const int lr = lane & 15, lq = lane >> 4;
  const int tr = min(r0 + wave, R - 1);  // the row this wave attends for

  for (int i = tid; i < (16 - DB_ROWS) * 512 / 16; i += 256) ((uint4*)(sA + DB_ROWS * 512))[i] = uint4{0, 0, 0, 0};
  bf16x8 fw[4][8];
  db_wload(Wo, wave, lane, fw);  // in flight during P1

  // ---- P1: self-attention -------------------------------------------------------------------------
  {
    const float* row = qkv + (size_t)tr * 768 + 4 * lane;
    f32x4 q = *(const f32x4*)row;
    f32x4 kn = *(const f32x4*)(row + 256), vn = *(const f32x4*)(row + 512);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      q[i] *= scale;
      kn[i] = (float)(bf16_t)kn[i];  // cache precision
      vn[i] = (float)(bf16_t)vn[i];
    }
    if (r0 + wave < R) {
      cn_store4(kc + ((size_t)step * R + tr) * 256 + 4 * lane, kn[0], kn[1], kn[2], kn[3]);
      cn_store4(vc + ((size_t)step * R + tr) * 256 + 4 * lane, vn[0], vn[1], vn[2], vn[3]);
    }
    const int rb = (tr / beam) * beam;
    const int* arow = anc + (size_t)tr * maxp;
    float m = -INFINITY, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    db_attend<8>(
        q, step, [&](int s) { return kc + ((size_t)s * R + rb + arow[s]) * 256 + 4 * lane; },
        [&](int s) { return vc + ((size_t)s * R + rb + arow[s]) * 256 + 4 * lane; }, m, l, o);
    {  // own key / value
      float d = q[0] * kn[0] + q[1] * kn[1] + q[2] * kn[2] + q[3] * kn[3];
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      const float mn = fmaxf(m, d);
      const float corr = __expf(m - mn), p = __expf(d - mn);
      l = l * corr + p;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = o[i] * corr + p * vn[i];
    }
    db_store_row(sA, wave, lane, o, 1.0f / l);
  }
  __syncthreads();

  // ---- P2: out-proj + residual + LN1 -----------------------------------------------------------------
  const int mrow = min(r0 + min(lr, DB_ROWS - 1), R - 1);
  f32x4 v[4];
  db_gemm16(fw, sA, lane, v);
  db_wload(Wq, wave, lane, fw);  // in flight during the residual add + LN1
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = 64 * wave + 16 * a + 4 * lq;
    const f32x4 bb = *(const f32x4*)(bo + n), rs = *(const f32x4*)(x + (size_t)mrow * 256 + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[a][j] += bb[j] + rs[j];
  }
  db_layernorm(v, g1, b1, s_red, wave, lane);   // v = x1 (kept in registers as the next residual)
  __syncthreads();
  if (lr < DB_ROWS) db_store_tile(sA, v, wave, lane);
  __syncthreads();

  // ---- P3: cross-attention query ------------------------------------------------------------------------
  {
    f32x4 qv[4];
    db_gemm16(fw, sA, lane, qv);
    db_wload(Wo2, wave, lane, fw);  // in flight during the cross-attention
    if (lr < DB_ROWS) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int n = 64 * wave + 16 * a + 4 * lq;
        const f32x4 bb = *(const f32x4*)(bq + n);
        *(f32x4*)(sQ + lr * 256 + n) = f32x4{(qv[a][0] + bb[0]) * scale, (qv[a][1] + bb[1]) * scale,
                                             (qv[a][2] + bb[2]) * scale, (qv[a][3] + bb[3]) * scale};
      }
    }
  }
  __syncthreads();

  // ---- P4: cross-attention over the clip's audio memory --------------------------------------------------
  {
    const f32x4 q = *(const f32x4*)(sQ + wave * 256 + 4 * lane);
    const int clip = tr / beam;
    int n = lens[clip];
    n = n < 1 ? 1 : (n > Ta ? Ta : n);
    const bf16_t* base = kvx + (size_t)clip * Ta * kv_ld + kv_off + 4 * lane;
    float m = -INFINITY, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    db_attend<8>(
        q, n, [&](int t) { return base + (size_t)t * kv_ld; }, [&](int t) { return base + (size_t)t * kv_ld + 256; }, m,
        l, o);
    db_store_row(sA, wave, lane, o, 1.0f / l);
  }
  __syncthreads();

  // ---- P5: out-proj + residual (x1) + LN2 -> x, xt -----------------------------------------------------------
  {
    f32x4 y[4];
    db_gemm16(fw, sA, lane, y);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = 64 * wave + 16 * a + 4 * lq;
      const f32x4 bb = *(const f32x4*)(bo2 + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[a][j] += bb[j] + v[a][j];
    }
    db_layernorm(y, g2, b2, s_red, wave, lane);
    if (lr < DB_ROWS && r0 + lr < R) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int n = 64 * wave + 16 * a + 4 * lq;
        *(f32x4*)(x + (size_t)(r0 + lr) * 256 + n) = y[a];
        cn_store4(xt + (size_t)(r0 + lr) * 256 + n, y[a][0], y[a][1], y[a][2], y[a][3]);
      }
    }
  }
}
