// Register-chained fused MLP, streamed-weight variant (C = 192: 8 waves, C = 384: 4 waves of 32 positions).
//
// The packed fragment stream of mlp_rc.h (C/8 chunks of C/8 + 1 KB) does not fit in LDS, so the chunks flow
// L2 -> LDS through an NST-deep ring filled by global_load_lds_dwordx4 (a 1 KB piece = one MFMA fragment; piece i of a
// chunk is issued by wave i % NW).  The chunk stream is cyclic: step g of a block consumes chunk g % NCH, whatever
// tile a wave is in, and chunk g + NST - 1 is requested at the start of step g.  One raw s_barrier per step publishes
// every wave's pieces of chunk g and proves chunk g - 1 is no longer read (its slot is the one refilled).  vmcnt is
// counted by hand (the DMA must stay in flight across the barrier): at most (NST - 2) newer chunks of this wave's
// pieces may be outstanding when chunk g is needed.  At a tile boundary (every NCH steps) the wave drains its queue
// once (vmcnt(0): residual rows and the next tile's y fragments have to arrive anyway), which also means the next
// NST - 1 chunks are known to be resident and their steps need no wait.
//
// Epilogue: the 32 x 32 output tiles go through a small per-wave LDS staging tile so that every global access is a
// 16-byte piece of a 128-byte row segment (8 lanes per row, whole lines): x += scale * (O + b2).
#pragma once
#include "mlp_rc.h"

template <int C, int NW, int NST, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void cn_mlp_rc_ring_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ WS,
                                                                 const float* __restrict__ b2, const float* __restrict__ scale,
                                                                 float* __restrict__ X, int M) {
  typedef RcGeom<C> G;
  typedef RcWave<C> W;
  constexpr int PIECES = G::PIECES, CB = G::CHUNK_BYTES;
  constexpr int DPW_LO = PIECES / NW, N_HI = PIECES % NW;  // waves < N_HI issue DPW_LO + 1 pieces per chunk
  constexpr int SR = NW >= 8 ? 32 : 16;                    // staging rows per wave (LDS budget)
  constexpr int STG = SR * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* stg = smem + NST * CB + wave * STG;

  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NW - 1) / NW;  // block-uniform: every wave runs the same number of steps

  const char* wsrc = (const char*)WS + lane * 16;
  auto stage = [&](int g) {  // chunk g % NCH -> slot g % NST (this wave's pieces)
    const char* src = wsrc + (size_t)(g % G::NCH) * CB;
    char* dst = ring + (g % NST) * CB;
#pragma unroll
    for (int i = 0; i < DPW_LO + 1; ++i) {
      const int piece = wave + i * NW;
      if (i < DPW_LO || wave < N_HI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };

  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)((lane < 32 && i < 2) ? 1.0f : 0.0f);

#pragma unroll
  for (int g = 0; g < NST - 1; ++g) stage(g);
  bf16x8 fy[G::KS1];
  if (t_lo + wave < t_hi) W::load_y(Y, (t_lo + wave) * 32, M, lane, fy);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const char* wl = ring + lane * 16;
  int g = 0;
  for (int it = 0; it < max_it; ++it) {
    const int tile = t_lo + wave + it * NW;
    const bool valid = tile < t_hi;
    f32x16 O[G::NT2];
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[t][i] = 0.f;
    for (int j = 0; j < G::NCH; ++j, ++g) {
      if (j >= NST - 1) {  // chunks 0 .. NST-2 of a tile were drained at the boundary
        if (N_HI > 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (DPW_LO + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW_LO) : "memory");
      }
      if (ABL != 4) {
        __builtin_amdgcn_s_barrier();
        stage(g + NST - 1);
      }
      if (valid) W::template chunk<(ABL == 4 ? 0 : ABL)>(wl + (g % NST) * CB, fy, ones, O);
    }
    // ---- tile boundary ------------------------------------------------------------------------------------------
    const int m0 = tile * 32;
    if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, M, lane, fy);
    if (valid) {
      // lane = channel 32 t + (l & 31), register r = position (r&3) + 8 (r>>2) + 4 (l>>5)  ->  staging [pos][32 ch] fp32
      const int cl = lane & 31, ph = 4 * (lane >> 5);
      const int row8 = lane >> 3, col4 = (lane & 7) * 4;
#pragma unroll
      for (int t = 0; t < G::NT2; ++t) {
#pragma unroll
        for (int half = 0; half < 32 / SR; ++half) {
          constexpr int RPH = SR / 2;  // registers per half: SR = 32 -> 16, SR = 16 -> 8
          f32x4 xr[SR / 8];
          // epilogue lanes own 4 consecutive channels (lane & 7) * 4 of tile t (L1-resident vectors, not worth registers)
          const f32x4 sc4 = *(const f32x4*)(scale + 32 * t + col4), bb4 = *(const f32x4*)(b2 + 32 * t + col4);
#pragma unroll
          for (int i = 0; i < SR / 8; ++i) {
            const int p = half * SR + row8 + 8 * i;
            xr[i] = *(const f32x4*)(X + (size_t)min(m0 + p, M - 1) * C + 32 * t + col4);
          }
          // the one drain per tile: every DMA piece issued so far, the next tile's y fragments and these rows
          if (t == 0 && half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int r = 0; r < RPH; ++r) {
            const int rr = half * RPH + r;
            const int p = (rr & 3) + 8 * (rr >> 2) + ph - half * SR;
            *(float*)(stg + (p * 32 + cl) * 4) = O[t][rr];
          }
          asm volatile("" ::: "memory");  // (same wave: LDS executes in order; keep the compiler from reordering)
#pragma unroll
          for (int i = 0; i < SR / 8; ++i) {
            const int pl = row8 + 8 * i;
            const f32x4 o = *(const f32x4*)(stg + (pl * 32 + col4) * 4);
            const int p = half * SR + pl;
            if (m0 + p < M) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaf(sc4[e], o[e] + bb4[e], xr[i][e]);
              *(f32x4*)(X + (size_t)(m0 + p) * C + 32 * t + col4) = v;
            }
          }
          asm volatile("" ::: "memory");
        }
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 chunks past the end
}

template <int C, int NW, int NST, int ABL = 0>
static int cn_launch_mlp_rc_ring(const bf16_t* Y, const bf16_t* WS, const float* b2, const float* scale, float* X, int M,
                                 int n_blocks, hipStream_t s) {
  constexpr int SMEM = NST * RcGeom<C>::CHUNK_BYTES + NW * (NW >= 8 ? 32 : 16) * 128;
  static_assert(SMEM <= 160 * 1024, "ring + staging must fit in LDS");
  static bool configured = false;
  if (!configured) {
    CN_HIP(hipFuncSetAttribute((const void*)cn_mlp_rc_ring_kernel<C, NW, NST, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    configured = true;
  }
  const int n_tiles = (M + 31) / 32;
  const int grid = n_blocks < cn_cdiv(n_tiles, NW) ? n_blocks : cn_cdiv(n_tiles, NW);
  hipLaunchKernelGGL((cn_mlp_rc_ring_kernel<C, NW, NST, ABL>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, WS, b2, scale, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
