// Register-chained fused ConvNeXt MLP (bf16), stages 0-2:
//
//     x[m][:] += scale * ( W2 . gelu( W1 . y[m][:] + b1 ) + b2 )          (convnext.py:66-74)
//
// A wave owns 32 positions.  Per chunk of 32 hidden units:
//
//   GEMM1   X[32 hid][32 pos]  = W1c[32 hid][C] . y[32 pos][C]^T (+ b1c)     v_mfma_f32_32x32x16_bf16, A = weights (LDS), B = y (registers)
//   GELU    on the 16 accumulator registers of X, packed pairwise to bf16
//   GEMM2   O[32 pos][C]      += gelu(X)^T[32 pos][32 hid] . W2c[C][32 hid]^T   A = the converted accumulator, B = weights (LDS)
//
// The accumulator of a 32x32 MFMA has its column (position) on the lane and its rows (hidden units) in the 16
// registers, so registers 8s .. 8s+7 converted to bf16 ARE the A fragment of k-step s of a product that sums over
// the rows (cdna guide 3, "an accumulator tile as the next MFMA's operand"): the hidden activations never touch LDS,
// no barrier separates the two GEMMs and the waves of a block are independent of each other.  The k order inside
// such a fragment is permuted (element j of lane half h = row 16s + 8(j>>2) + 4h + (j&3)); the packed W2 fragments
// (pk_mlp_rc) carry the same permutation.  The bias b1 enters GEMM1 as one extra k-step (hi + lo bf16 split of the
// fp32 bias against a constant "ones" fragment), so the accumulators need no per-register initialisation.
//
// Weights are stored as the exact sequence of 1 KB MFMA fragments the loop consumes (lane l reads 16 bytes at
// l * 16 of a fragment: conflict-free ds_read_b128, and a global_load_lds_dwordx4 piece IS a fragment):
//   chunk j = [ W1 fragments s = 0 .. C/16-1 | bias fragment | W2 fragments (t, s), t = 0 .. C/32-1, s = 0, 1 ]
//
// Output orientation: O = X^T . B has channels on the lanes and positions in the registers, so every residual load /
// store instruction covers two whole 128-byte rows pieces (32 channels x 4 B): full-line HBM traffic without staging.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int C> struct RcGeom {
  static constexpr int KS1 = C / 16;                 // k-steps of GEMM1
  static constexpr int NT2 = C / 32;                 // 32-channel tiles of GEMM2
  static constexpr int NCH = C / 8;                  // hidden chunks of 32 (4C / 32)
  static constexpr int PIECES = KS1 + 1 + 2 * NT2;   // 1 KB fragments per chunk
  static constexpr int CHUNK_BYTES = PIECES * 1024;
};

// GELU as x * sigmoid(x * (a + b x^2 + c x^4)): minimax fit of the logit of the normal CDF on [-8, 8],
// max |error| against the exact erf form 2.5e-5 (the tanh form: 4.7e-4).  7 VALU + 2 transcendental per element
// (A&S 7.1.26 erf: 12 + 2).  x^2 is clamped at 64, beyond which the quartic term would bend the logit back.
__device__ __forceinline__ float cn_gelu_sig(float x) {
  constexpr float L2E = 1.4426950408889634f;
  const float x2 = fminf(x * x, 64.0f);
  float p = fmaf(x2, 0.0007030350670982541f * L2E, -0.07401130190658815f * L2E);
  p = fmaf(p, x2, -1.5950157568571721f * L2E);
  const float e = __builtin_amdgcn_exp2f(x * p);  // exp(-x * logit(x))
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ bf16x8 cn_pack8(float a, float b, float c, float d, float e, float f, float g, float h) {
  return bf16x8{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d, (bf16_t)e, (bf16_t)f, (bf16_t)g, (bf16_t)h};
}

// ---- weight packing: fp32 nn.Linear layouts -> fragment stream --------------------------------------------------
// W1 [4C][C], b1 [4C], W2 [C][4C]; one thread = one 16-byte unit (chunk j, piece q, lane l)
__global__ void pk_mlp_rc(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2, int C,
                          bf16_t* __restrict__ dst) {
  const int KS1 = C / 16, NT2 = C / 32, NCH = C / 8, PIECES = KS1 + 1 + 2 * NT2;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= NCH * PIECES * 64) return;
  const int l = u & 63, q = (u >> 6) % PIECES, j = (u >> 6) / PIECES;
  const int r = l & 31, h = l >> 5;
  float v[8];
  if (q < KS1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = W1[(size_t)(32 * j + r) * C + 16 * q + 8 * h + i];
  } else if (q == KS1) {
    const float b = b1[32 * j + r];
    const float hi = (float)(bf16_t)b;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (h == 0) {
      v[0] = hi;
      v[1] = b - hi;
    }
  } else {
    const int t = (q - KS1 - 1) >> 1, s = (q - KS1 - 1) & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      v[i] = W2[(size_t)(32 * t + r) * (4 * C) + 32 * j + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = (bf16_t)v[i];
}

// ---- per-wave building blocks ------------------------------------------------------------------------------------
template <int C> struct RcWave {
  typedef RcGeom<C> G;

  // one hidden chunk whose fragments start at wc (LDS, already offset by lane * 16)
  // ABL (kernel lab only): 1 = no GELU, 2 = GELU without its two transcendentals, 3 = weight fragments not read from LDS
  template <int ABL = 0>
  static __device__ __forceinline__ void chunk(const char* wc, const bf16x8 (&fy)[G::KS1], const bf16x8 ones,
                                               f32x16 (&O)[G::NT2]) {
    f32x16 X;
#pragma unroll
    for (int i = 0; i < 16; ++i) X[i] = 0.f;
    if constexpr (ABL == 3) {
      bf16x8 w = ones;
#pragma unroll
      for (int s = 0; s <= G::KS1; ++s) {
        asm volatile("" : "+v"(w));
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, s < G::KS1 ? fy[s] : ones, X, 0, 0, 0);
      }
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = cn_gelu_sig(X[i]);
      const bf16x8 h0 = cn_pack8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
      const bf16x8 h1 = cn_pack8(g[8], g[9], g[10], g[11], g[12], g[13], g[14], g[15]);
#pragma unroll
      for (int t = 0; t < G::NT2; ++t) {
        asm volatile("" : "+v"(w));
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, w, O[t], 0, 0, 0);
        asm volatile("" : "+v"(w));
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, w, O[t], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < G::KS1; ++s)
      X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wc + s * 1024), fy[s], X, 0, 0, 0);
    X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wc + G::KS1 * 1024), ones, X, 0, 0, 0);
    float g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (ABL == 1) g[i] = X[i];
      else if constexpr (ABL == 2) {
        const float x = X[i], x2 = fminf(x * x, 64.0f);
        float p = fmaf(x2, 0.001f, -0.1f);
        p = fmaf(p, x2, -2.3f);
        g[i] = x * (1.0f + x * p);
      } else g[i] = cn_gelu_sig(X[i]);
    }
    const bf16x8 h0 = cn_pack8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
    const bf16x8 h1 = cn_pack8(g[8], g[9], g[10], g[11], g[12], g[13], g[14], g[15]);
#pragma unroll
    for (int t = 0; t < G::NT2; ++t) {
      O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, *(const bf16x8*)(wc + (G::KS1 + 1 + 2 * t) * 1024), O[t], 0, 0, 0);
      O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, *(const bf16x8*)(wc + (G::KS1 + 2 + 2 * t) * 1024), O[t], 0, 0, 0);
    }
  }

  // y rows of a 32-position tile as B fragments: fy[s] = y[m0 + (l & 31)][16 s + 8 (l >> 5) .. + 8]
  static __device__ __forceinline__ void load_y(const bf16_t* __restrict__ Y, int m0, int M, int lane, bf16x8 (&fy)[G::KS1]) {
    const int m = min(m0 + (lane & 31), M - 1);
    const bf16_t* p = Y + (size_t)m * C + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < G::KS1; ++s) fy[s] = *(const bf16x8*)(p + 16 * s);
  }

  // x[m0 + p][c] += sc[c] * (O + b2[c]); lane = channel (32 t + (l & 31)), register r = position (r&3) + 8 (r>>2) + 4 (l>>5)
  static __device__ __forceinline__ void epilogue(float* __restrict__ X, int m0, int M, int lane, const f32x16 (&O)[G::NT2],
                                                  const float (&sc)[G::NT2], const float (&bb)[G::NT2]) {
    const int cl = lane & 31, ph = 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < G::NT2; ++t) {
      float* xp = X + (size_t)m0 * C + 32 * t + cl;
      float xr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = (r & 3) + 8 * (r >> 2) + ph;
        xr[r] = xp[(size_t)min(p, M - 1 - m0) * C];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = (r & 3) + 8 * (r >> 2) + ph;
        if (m0 + p < M) xp[(size_t)p * C] = fmaf(sc[t], O[t][r] + bb[t], xr[r]);
      }
    }
  }
};

// ---- resident variant: the whole packed weight stream fits in LDS (C = 96: 156 KB) ----------------------------------
// Persistent blocks, one per CU; after the one-time fill there is no barrier and no DMA: every wave walks its own
// position tiles at its own pace, so one wave's GELU (VALU) runs under its SIMD partner's MFMAs.
template <int C, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void cn_mlp_rc_resident_kernel(const bf16_t* __restrict__ Y,
                                                                     const bf16_t* __restrict__ WS,
                                                                     const float* __restrict__ b2,
                                                                     const float* __restrict__ scale, float* __restrict__ X,
                                                                     int M) {
  typedef RcGeom<C> G;
  typedef RcWave<C> W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int TOTAL = G::NCH * G::PIECES;
  for (int i = wave; i < TOTAL; i += NW)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)WS + (size_t)i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
  // tile range of this block, then of this wave (contiguous, balanced to within one tile)
  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  float sc[G::NT2], bb[G::NT2];
#pragma unroll
  for (int t = 0; t < G::NT2; ++t) {
    sc[t] = scale[32 * t + (lane & 31)];
    bb[t] = b2[32 * t + (lane & 31)];
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)((lane < 32 && i < 2) ? 1.0f : 0.0f);
  bf16x8 fy[G::KS1];
  int tile = t_lo + wave;
  if (tile < t_hi) W::load_y(Y, tile * 32, M, lane, fy);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* wl = smem + lane * 16;
  for (; tile < t_hi; tile += NW) {
    f32x16 O[G::NT2];
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[t][i] = 0.f;
#pragma unroll 2
    for (int j = 0; j < G::NCH; ++j) W::template chunk<ABL>(wl + j * G::CHUNK_BYTES, fy, ones, O);
    if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, M, lane, fy);  // in flight under the epilogue
    W::epilogue(X, tile * 32, M, lane, O, sc, bb);
  }
}

template <int C, int NW, int ABL = 0>
static int cn_launch_mlp_rc_resident(const bf16_t* Y, const bf16_t* WS, const float* b2, const float* scale, float* X, int M,
                                     int n_blocks, hipStream_t s) {
  constexpr int SMEM = RcGeom<C>::NCH * RcGeom<C>::CHUNK_BYTES;
  static_assert(SMEM <= 160 * 1024, "resident variant: the weight stream must fit in LDS");
  static bool configured = false;
  if (!configured) {
    CN_HIP(hipFuncSetAttribute((const void*)cn_mlp_rc_resident_kernel<C, NW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    configured = true;
  }
  const int n_tiles = (M + 31) / 32;
  const int grid = n_blocks < cn_cdiv(n_tiles, NW) ? n_blocks : cn_cdiv(n_tiles, NW);
  hipLaunchKernelGGL((cn_mlp_rc_resident_kernel<C, NW, ABL>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, WS, b2, scale, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
