// Lab only (not part of the library): the exact precision's fused MLP at C = 384 with HALF entries in the ring.
// Correct (parity tests green with it in the product), and not faster than the two unfused sp16 GEMMs: 553 us per launch
// against 311 + 217 (profiles/r03_notes.md).  Include after mlp_sp.h.
#pragma once
#include "mlp_sp.h"

// Split-entry variant (C = 384): a chunk's entry is 97 KB -- W1 hi / lo + bias 49 KB, W2' hi / lo 48 KB -- and only one of
// those fits in LDS, so the ring holds HALF entries (three slots of 49 KB) and a step is two half-steps with a barrier each:
// half 2 g consumes the W1 part of chunk g (GEMM1, GELU, hi / lo split; the converted chunk stays in registers), half
// 2 g + 1 its W2 part (GEMM2).  One wave per SIMD: y pairs (192) + O (192) + the chunk are 430 of the 512 registers.
template <int C, int NW>
__global__ __launch_bounds__(NW * 64) void cn_mlp_sp_split_kernel(const sp16_t* __restrict__ Y, const _Float16* __restrict__ WS,
                                                                  float* __restrict__ X, int M) {
  typedef SpGeom<C> G;
  typedef SpWave<C> SW;
  typedef Rc2Wave<C, 1> W;
  constexpr int NST = 3, F1 = G::F1, F2 = G::F2, SLOT = (F1 > F2 ? F1 : F2) * 1024;
  constexpr int P1 = F1 / NW, H1 = F1 % NW, P2 = F2 / NW, H2 = F2 % NW;  // pieces per wave of a W1 / W2 half (+ 1 for waves < H)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* aux = (const float*)((const char*)WS + G::STREAM_BYTES);

  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NW - 1) / NW;  // block-uniform

  const unsigned voff = lane * 16;
  const unsigned lds0 = cn_lds_addr(smem);
  auto stage = [&](int g2) {  // half g2 (chunk g2 / 2 of the stream, part g2 & 1) -> slot g2 % NST: this wave's pieces
    const int h = g2 & 1;
    const char* src = (const char*)WS + (size_t)((g2 >> 1) % G::NSTEP) * G::STEP_BYTES + (h ? F1 * 1024 : 0);  // wave-uniform
    const unsigned dst = lds0 + (unsigned)((g2 % NST) * SLOT);
    const int n = h ? P2 + (wave < H2 ? 1 : 0) : P1 + (wave < H1 ? 1 : 0);
#pragma unroll
    for (int i = 0; i < (P1 > P2 ? P1 : P2) + 1; ++i)
      if (i < n) cn_dma16_s(src + (wave + i * NW) * 1024, voff, dst + (wave + i * NW) * 1024);
  };
  // own pieces of the half about to be consumed have landed: one younger half (of the OTHER kind) may still fly
  auto ring_wait = [&](int g2) {
    const int younger = (g2 & 1) ? P1 + (wave < H1 ? 1 : 0) : P2 + (wave < H2 ? 1 : 0);
    if (younger == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(P1 == 12 && P2 == 12 && H1 <= 1 && H2 == 0, "ring_wait's immediates are written for C = 384, NW = 4");
  f16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (_Float16)((lane < 32 && i < 2) ? 1.0f : 0.0f);

  stage(0);
  stage(1);
  f16x8 yh[G::KS1], yl[G::KS1];
  if (t_lo + wave < t_hi) SW::load_y(Y, (t_lo + wave) * 32, lane, yh, yl);
  const char* wl = smem + lane * 16;
  int g2 = 0;
  for (int it = 0; it < max_it; ++it) {
    const int tile = t_lo + wave + it * NW;
    const bool valid = tile < t_hi;
    f32x16 O[G::NT2];
    if (valid) W::init_o(X, tile * 32, lane, O);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): residual + y, once per tile, on every path into the loop
    for (int j = 0; j < G::NSTEP; ++j) {
      f16x8 Gh[2], Gl[2];
      ring_wait(g2);
      __builtin_amdgcn_s_barrier();
      stage(g2 + NST - 1);
      if (valid) SW::step_a(wl + (g2 % NST) * SLOT, yh, yl, ones, Gh, Gl);
      ++g2;
      ring_wait(g2);
      __builtin_amdgcn_s_barrier();
      stage(g2 + NST - 1);
      if (valid) SW::step_b(wl + (g2 % NST) * SLOT, Gh, Gl, O);
      ++g2;
    }
    if (tile + NW < t_hi) SW::load_y(Y, (tile + NW) * 32, lane, yh, yl);
    if (valid) {
      const float* bbv = aux;
      asm volatile("" : "+s"(bbv));
      W::store_o(X, bbv, tile * 32, M, lane, O);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled two halves past the end
}

template <int C, int NW>
static int cn_launch_mlp_sp_split(const sp16_t* Y, const void* WS, float* X, int M, int n_blocks, hipStream_t s) {
  typedef SpGeom<C> G;
  constexpr int SMEM = 3 * (G::F1 > G::F2 ? G::F1 : G::F2) * 1024;
  static_assert(SMEM <= 160 * 1024, "ring must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_sp_split_kernel<C, NW>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_sp_split_kernel<C, NW>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, (const _Float16*)WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

