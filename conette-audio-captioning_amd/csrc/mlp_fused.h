// Fused ConvNeXt MLP for stages 0-2 (C = 96, 192, 384), bf16:
//
//     x[m][:] += scale * ( W2 . gelu( W1 . y[m][:] + b1 ) + b2 )          (convnext.py:66-74)
//
// The reference (and the unfused path) materialises the (P x 4C) hidden in memory: at C = 96 that
// is 8x the bytes of the block's input and makes pw1 / pw2 HBM-bound.  Here
// a block owns BM = 32*TM positions and walks the hidden dimension in chunks of 32 units:
//
//   GEMM1  Hc[BM][32]  = y[BM][C] . W1c[32][C]^T      A operand of y kept in REGISTERS for all chunks
//   epi1   Hc = gelu(Hc + b1c) -> bf16 -> LDS (swizzled [BM][32] tile, 64-byte rows)
//   GEMM2  O[BM][C]   += Hc[BM][32] . W2c[C][32]^T    fp32 accumulators live across all chunks
//
// so the hidden never leaves the CU.  W1c / W2c chunks (C*128 bytes) stream L2 -> LDS with
// global_load_lds_dwordx4 into a 3-deep ring (chunks j+1, j+2 in flight during chunk j).  The weights
// are stored a second time as the exact LDS image of that ring (api.hip pk_mlp_stream, XOR swizzle
// for conflict-free ds_read_b128 included), so every DMA instruction moves 1 KB of consecutive bytes;
// 64-byte row pieces gathered from the nn.Linear layout streamed at half the per-CU rate, which is
// what bounds the kernel once C is large (2.4 MB of weights per block at C = 384).
// 4 waves as 2 (rows) x 2 (hidden / channel halves); two barriers per chunk.
#pragma once
#include <stdlib.h>

#include "gemm.h"

// Phase stamps: profiling build only (CN_G2_PROF=1 python build.py --force), accumulated in registers -- an atomic
// per stamp would sit in the vmcnt queue the chunk loop waits on.  CN_MLP_DEBUG = C selects the instantiation.
__device__ unsigned long long g_mlp_prof[8];
#ifndef CN_G2_PROF
#define MLP_STAMP(i)
#else
#define MLP_STAMP(i)                              \
  if (dbg) {                                      \
    const unsigned long long t_ = wall_clock64(); \
    t_acc[i] += t_ - t_prev;                      \
    t_prev = t_;                                  \
  }
#endif

// NWM = wave rows (2 or 4): 2 * NWM waves per block, each owning TM 16-row tiles and one half of the hidden chunk /
// of the channels.  C = 384 runs 8 waves (2 per SIMD, 250 registers each): with 4 waves of TM = 4 every LDS
// fragment read in front of its MFMAs was exposed (one wave per SIMD, no registers left to prefetch into) and the
// kernel was no faster than the two GEMMs it replaces.
template <int C, int TM, int NWM, int NST = 3, int PIPE = 0>
__global__ __launch_bounds__(NWM * 128) void cn_mlp_fused_kernel(const bf16_t* __restrict__ Y,
                                                           const bf16_t* __restrict__ WS /* packed ring image */,
                                                           const float* __restrict__ b1,
                                                           const float* __restrict__ b2,
                                                           const float* __restrict__ scale, float* __restrict__ X,
                                                           int M, int dbg) {
#ifdef CN_G2_PROF
  unsigned long long t_prev = dbg ? wall_clock64() : 0;
  unsigned long long t_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  constexpr int BM = NWM * 16 * TM;     // rows per block
  constexpr int NT = NWM * 128;         // threads
  constexpr int NW = NWM * 2;           // waves
  constexpr int KS1 = C / 32;           // k-steps of GEMM1
  constexpr int TN2 = C / 32;           // 16-channel tiles per wave in GEMM2 (wave owns C/2 channels)
  constexpr int NCH = 4 * C / 32;       // hidden chunks
  constexpr int W1C_BYTES = 32 * C * 2; // [KS1][32 rows][64 B]
  constexpr int W2C_BYTES = C * 64;     // [C rows][64 B]
  constexpr int BUF = W1C_BYTES + W2C_BYTES;
  constexpr int N_DMA = BUF / 1024;     // per chunk
  constexpr int DPW = N_DMA / NW;
  static_assert(N_DMA % NW == 0, "DMA pieces must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // NST = weight-chunk ring depth: chunks j+1 .. j+NST-1 in flight during chunk j
  static_assert(PIPE != 1 || NST >= 4, "the pipelined schedule reads two ring slots per iteration");
  constexpr int RING_BYTES = PIPE == 2 ? 2 * BUF : NST * BUF;  // PIPE 2: separate two-slot rings for W1 and W2 chunks
  char* sH = smem + RING_BYTES;         // [BM][64 B] (x 2 when PIPE)
  float* sB1 = (float*)(sH + (PIPE ? 2 : 1) * BM * 64);  // [4C] pwconv1 bias (no ordinary global load may sit inside the loop:
                                        // with LDS-DMA in flight hipcc would wait vmcnt(0) for it every chunk)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = cn_xcd_remap(blockIdx.x, gridDim.x) * BM;
  const int lr = lane & 15, lq = lane >> 4;
  const int sw = (lr >> 2) & 3;         // swizzle term of a 64-byte-row tile for row (16k + lr)

  // ---- A operand (y rows) straight to registers: frag[b][ks] = y[m][32ks + 8lq .. +8] ----------
  bf16x8 fa[TM][KS1];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = min(m0 + wm * (16 * TM) + b * 16 + lr, M - 1);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) fa[b][ks] = *(const bf16x8*)(Y + (size_t)m * C + ks * 32 + lq * 8);
  }

  // ---- weight ring: chunk j, piece inst = 1 KB at WS + (j * N_DMA + inst) * 1024 -----------------------------
  const char* wsrc = (const char*)WS + (size_t)wave * DPW * 1024 + lane * 16;
  auto stage = [&](int buf, int j) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int inst = wave * DPW + i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)j * BUF + i * 1024),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + inst * 1024), 16, 0,
                                       0);
    }
  };

  // fp32 accumulators of GEMM2.  For the narrow stages they START from the residual x (requested here, together
  // with the A fragments, so its HBM latency hides under the whole chunk loop; the final epilogue was 28 % of a
  // C = 96 block, all of it waiting for these rows); at C = 384 the registers are needed for the operands.
  constexpr bool kResidEarly = TN2 * TM <= 12;
  f32x4 acc2[TN2][TM];
  f32x4 res[kResidEarly ? TN2 : 1][kResidEarly ? TM : 1];
#pragma unroll
  for (int a = 0; a < TN2; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (kResidEarly)
        res[a][b] = *(const f32x4*)(X + (size_t)min(m0 + wm * (16 * TM) + b * 16 + lr, M - 1) * C + wn * (C / 2) + a * 16 + 4 * lq);
    }

  for (int i = tid; i < 4 * C; i += NT) sB1[i] = b1[i];
  // PIPE 2 (C = 384, where a four-slot ring of whole chunks does not fit): the W1 and W2 halves of a chunk live in
  // separate two-slot rings, because the pipelined loop needs W1 of chunk j+1 together with W2 of chunk j
  constexpr int W1P = W1C_BYTES / 1024, W2P = W2C_BYTES / 1024;
  auto stage_w1 = [&](int slot, int j) {
#pragma unroll
    for (int i = 0; i < W1P / NW; ++i) {
      const int inst = wave * (W1P / NW) + i;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)((const char*)WS + (size_t)j * BUF + inst * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(smem + slot * W1C_BYTES + inst * 1024), 16, 0, 0);
    }
  };
  auto stage_w2 = [&](int slot, int j) {
#pragma unroll
    for (int i = 0; i < W2P / NW; ++i) {
      const int inst = wave * (W2P / NW) + i;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)((const char*)WS + (size_t)j * BUF + W1C_BYTES + inst * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(smem + 2 * W1C_BYTES + slot * W2C_BYTES + inst * 1024), 16, 0, 0);
    }
  };
  if constexpr (PIPE == 2) {
    static_assert(W1P % NW == 0 && W2P % NW == 0, "split rings: pieces must divide over the waves");
    stage_w1(0, 0);
    stage_w2(0, 0);
    stage_w1(1, 1);
  } else {
    stage(0, 0);
    if (NST >= 3) stage(1, 1);
    if (PIPE == 1) {
      stage(2, 2);
      stage(3, 3);
    }
  }
  // retire the ordinary loads (y fragments, bias) HERE, once: touching the registers makes the compiler
  // place its vmcnt wait before the loop instead of a vmcnt(0) in front of the first MFMA of every chunk
#pragma unroll
  for (int b = 0; b < TM; ++b)
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(fa[b][ks]));
  if constexpr (kResidEarly) {
#pragma unroll
    for (int a = 0; a < TN2; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) asm volatile("" : "+v"(res[a][b]));
  }
  MLP_STAMP(0)
  if constexpr (PIPE != 0) {
    // Software-pipelined schedule: iteration j runs GEMM 1 of chunk j+1, GEMM 2 of chunk j and the GELU epilogue of
    // chunk j+1 between ONE pair of barriers (double-buffered hidden tile).  The epilogue's VALU work is independent
    // of GEMM 2's MFMAs, so hipcc interleaves them: at one wave per SIMD (256-row tiles, chosen so that decode kernels
    // fit beside this one) nothing else would hide the 32 GELUs per lane and chunk.
    auto gemm1 = [&](const char* sW1, f32x4 (&acc1)[TM]) {
#pragma unroll
      for (int b = 0; b < TM; ++b) acc1[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const bf16x8 fw = *(const bf16x8*)(sW1 + ks * 2048 + (wn * 16 + lr) * 64 + ((lq ^ sw) * 16));
#pragma unroll
        for (int b = 0; b < TM; ++b) acc1[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fa[b][ks], acc1[b], 0, 0, 0);
      }
    };
    auto epi1 = [&](int jj, const f32x4 (&acc1)[TM], char* sHb) {
      const f32x4 bb = *(const f32x4*)(sB1 + jj * 32 + wn * 16 + 4 * lq);
      const int chunk = wn * 2 + (lq >> 1);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int ml = wm * (16 * TM) + b * 16 + lr;
        bf16_t* dst = (bf16_t*)(sHb + ml * 64 + ((chunk ^ sw) * 16) + (lq & 1) * 8);
        const f32x4 g = cn_gelu_fast4(f32x4{acc1[b][0] + bb[0], acc1[b][1] + bb[1], acc1[b][2] + bb[2], acc1[b][3] + bb[3]});
        cn_store4(dst, g[0], g[1], g[2], g[3]);
      }
    };
    auto gemm2 = [&](const char* sW2, const char* sHb) {
      bf16x8 fh[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) fh[b] = *(const bf16x8*)(sHb + (wm * (16 * TM) + b * 16 + lr) * 64 + ((lq ^ sw) * 16));
#pragma unroll
      for (int a = 0; a < TN2; ++a) {
        const bf16x8 fw2 = *(const bf16x8*)(sW2 + (wn * (C / 2) + a * 16 + lr) * 64 + ((lq ^ sw) * 16));
#pragma unroll
        for (int b = 0; b < TM; ++b) acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2, fh[b], acc2[a][b], 0, 0, 0);
      }
    };
    f32x4 acc1[TM];
    if constexpr (PIPE == 1) {
      // chunk 0: GEMM 1 + epilogue (chunks 1..3 stay in flight)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DPW) : "memory");
      __builtin_amdgcn_s_barrier();
      gemm1(smem, acc1);
      epi1(0, acc1, sH);
      for (int j = 0; j < NCH; ++j) {
        // chunk j+1 landed (chunks j+2 [, j+3 at j = 0] may stay in flight); the barrier publishes H(j) and every wave's
        // DMA pieces, and proves chunk j-1's slot and H(j-1)'s buffer are no longer read
        if (j == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * DPW) : "memory");
        else if (j + 2 < NCH) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j >= 1 && j + 3 < NCH) stage((j + 3) % NST, j + 3);
        char* sHj = sH + (j & 1) * (BM * 64);
        char* sHn = sH + ((j + 1) & 1) * (BM * 64);
        if (j + 1 < NCH) gemm1(smem + ((j + 1) % NST) * BUF, acc1);
        gemm2(smem + (j % NST) * BUF + W1C_BYTES, sHj);
        if (j + 1 < NCH) epi1(j + 1, acc1, sHn);
      }
    } else {
      // split rings: W1 of chunk j+1 and W2 of chunk j are resident in iteration j; the loads issued in iteration j
      // (W1 of chunk j+2, W2 of chunk j+1) have one whole iteration to land
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      gemm1(smem, acc1);
      epi1(0, acc1, sH);
      for (int j = 0; j < NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + 2 < NCH) stage_w1(j & 1, j + 2);         // slot of W1(j): consumed in iteration j-1
        if (j + 1 < NCH) stage_w2((j + 1) & 1, j + 1);   // slot of W2(j-1): consumed in iteration j-1
        char* sHj = sH + (j & 1) * (BM * 64);
        char* sHn = sH + ((j + 1) & 1) * (BM * 64);
        if (j + 1 < NCH) gemm1(smem + ((j + 1) & 1) * W1C_BYTES, acc1);
        gemm2(smem + 2 * W1C_BYTES + (j & 1) * W2C_BYTES, sHj);
        if (j + 1 < NCH) epi1(j + 1, acc1, sHn);
      }
    }
  } else
  for (int j = 0; j < NCH; ++j) {
    const int buf = j % NST;
    // chunk j landed for this wave when at most the newer chunk's DPW pieces are outstanding; the raw
    // barrier (no vmcnt drain) then publishes every wave's pieces and proves chunk j-1 (buffers + H) is
    // no longer read, so its ring slot is refilled with chunk j+2
    if (NST >= 3 && j + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    MLP_STAMP(1)
    if (j + NST - 1 < NCH) stage((j + NST - 1) % NST, j + NST - 1);
    const char* sW1 = smem + buf * BUF;
    const char* sW2 = sW1 + W1C_BYTES;

    // GEMM1: this wave -> rows wm*(BM/2).. (TM tiles) x hidden wn*16..+16 of the chunk
    f32x4 acc1[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) acc1[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const bf16x8 fw = *(const bf16x8*)(sW1 + ks * 2048 + (wn * 16 + lr) * 64 + ((lq ^ sw) * 16));
#pragma unroll
      for (int b = 0; b < TM; ++b) acc1[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fa[b][ks], acc1[b], 0, 0, 0);
    }
    MLP_STAMP(2)
    // epilogue 1: + b1, GELU, -> bf16 -> H tile; lane holds hidden 4lq..4lq+3 (of this wave's 16) of row lr
    {
      const f32x4 bb = *(const f32x4*)(sB1 + j * 32 + wn * 16 + 4 * lq);
      const int chunk = wn * 2 + (lq >> 1);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int ml = wm * (16 * TM) + b * 16 + lr;
        bf16_t* dst = (bf16_t*)(sH + ml * 64 + ((chunk ^ sw) * 16) + (lq & 1) * 8);
        const f32x4 g = cn_gelu_fast4(f32x4{acc1[b][0] + bb[0], acc1[b][1] + bb[1], acc1[b][2] + bb[2], acc1[b][3] + bb[3]});
        cn_store4(dst, g[0], g[1], g[2], g[3]);
      }
    }
    MLP_STAMP(3)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my H writes are in LDS ...
    __builtin_amdgcn_s_barrier();                        // ... and so are everyone's (raw: keeps the DMA in flight)
    MLP_STAMP(4)
    // GEMM2: rows wm*(BM/2).. (TM tiles) x channels wn*(C/2).. (TN2 tiles), K = 32
    bf16x8 fh[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) fh[b] = *(const bf16x8*)(sH + (wm * (16 * TM) + b * 16 + lr) * 64 + ((lq ^ sw) * 16));
#pragma unroll
    for (int a = 0; a < TN2; ++a) {
      const bf16x8 fw2 = *(const bf16x8*)(sW2 + (wn * (C / 2) + a * 16 + lr) * 64 + ((lq ^ sw) * 16));
#pragma unroll
      for (int b = 0; b < TM; ++b) acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2, fh[b], acc2[a][b], 0, 0, 0);
    }
    MLP_STAMP(5)
  }

  // ---- final epilogue: x += scale * (acc + b2), 16 bytes per lane (4 consecutive channels) ---------
  // X is read and written through the same pointer, so hipcc keeps every tile's load behind the previous tile's
  // store: TN2 * TM dependent HBM round trips per block.  All residual loads of a channel tile `a` are issued
  // before its stores, and tile a + 1's before tile a's stores (double-buffered registers).
  if constexpr (kResidEarly) {
    const int mrow = m0 + wm * (16 * TM) + lr;
#pragma unroll
    for (int a = 0; a < TN2; ++a) {
      const int n = wn * (C / 2) + a * 16 + 4 * lq;
      const f32x4 bb = *(const f32x4*)(b2 + n), sc = *(const f32x4*)(scale + n);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = mrow + b * 16;
        if (m < M) {
          const f32x4 v = acc2[a][b], rr = res[a][b];
          *(f32x4*)(X + (size_t)m * C + n) = f32x4{rr[0] + sc[0] * (v[0] + bb[0]), rr[1] + sc[1] * (v[1] + bb[1]),
                                                     rr[2] + sc[2] * (v[2] + bb[2]), rr[3] + sc[3] * (v[3] + bb[3])};
        }
      }
    }
  } else {
    f32x4 r[2][TM];
    const int mrow = m0 + wm * (16 * TM) + lr;
#pragma unroll
    for (int b = 0; b < TM; ++b)
      r[0][b] = *(const f32x4*)(X + (size_t)min(mrow + b * 16, M - 1) * C + wn * (C / 2) + 4 * lq);
#pragma unroll
    for (int a = 0; a < TN2; ++a) {
      const int n = wn * (C / 2) + a * 16 + 4 * lq;
      const f32x4 bb = *(const f32x4*)(b2 + n), sc = *(const f32x4*)(scale + n);
      if (a + 1 < TN2) {
#pragma unroll
        for (int b = 0; b < TM; ++b)
          r[(a + 1) & 1][b] = *(const f32x4*)(X + (size_t)min(mrow + b * 16, M - 1) * C + n + 16);
      }
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = mrow + b * 16;
        if (m < M) {
          const f32x4 v = acc2[a][b], rr = r[a & 1][b];
          *(f32x4*)(X + (size_t)m * C + n) = f32x4{rr[0] + sc[0] * (v[0] + bb[0]), rr[1] + sc[1] * (v[1] + bb[1]),
                                                     rr[2] + sc[2] * (v[2] + bb[2]), rr[3] + sc[3] * (v[3] + bb[3])};
        }
      }
    }
  }
  MLP_STAMP(6)
#ifdef CN_G2_PROF
  if (dbg && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) atomicAdd(&g_mlp_prof[i], t_acc[i]);
    atomicAdd(&g_mlp_prof[7], 1ull);
  }
#endif
}

template <int C, int TM, int NWM, int NST = 3, int PIPE = 0>
static int cn_launch_mlp_fused(const bf16_t* Y, const bf16_t* WS, const float* b1, const float* b2,
                               const float* scale, float* X, int M, hipStream_t s) {
  constexpr int SMEM = (PIPE == 2 ? 2 : NST) * (32 * C * 2 + C * 64) + (PIPE ? 2 : 1) * NWM * 16 * TM * 64 + 4 * C * 4;
  static bool configured = false;
  if (!configured) {
    CN_HIP(hipFuncSetAttribute((const void*)cn_mlp_fused_kernel<C, TM, NWM, NST, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               SMEM));
    configured = true;
  }
  hipLaunchKernelGGL((cn_mlp_fused_kernel<C, TM, NWM, NST, PIPE>), dim3((unsigned)cn_cdiv(M, NWM * 16 * TM)), dim3(NWM * 128), SMEM, s, Y, WS, b1,
                     b2, scale, X, M, getenv("CN_MLP_DEBUG") && atoi(getenv("CN_MLP_DEBUG")) == C ? 1 : 0);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
