// Lab only (not part of the library): the C = 384 fused-MLP ring kernel with the second k-half of a chunk's GEMM2 run one step
// late, so that the first half of the NEXT chunk's GELU -- exposed between the two GEMMs in the shipped order -- rides under it.
// Bit-identical to the shipped kernel, and slower: 195 us with packed GELU math, 186.5 us with scalar math beside the MFMAs,
// against 181 us (same box).  profiles/r02_notes.md.  Include after mlp_rc2.h.
#pragma once
#include "mlp_rc2.h"

// skew (NCK = 1 only): a step's entry is [W1 of chunk st | W2 k-half 1 of chunk st - 1 (cyclic) | W2 k-half 0 of chunk st]:
// the second half of a chunk's GEMM2 runs one step late, under it the first half of the NEXT chunk's GELU (Rc2Skew).
static __global__ void pk_mlp_rc2_skew(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                           const float* __restrict__ b2, const float* __restrict__ scale, int C, int NCK,
                           bf16_t* __restrict__ dst, int skew = 1) {
  const int KS1 = C / 16, NT2 = C / 32, NCH = C / 8, F1 = KS1 + 1, F2 = 2 * NT2, FRAGS = NCK * (F1 + F2), NSTEP = NCH / NCK;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < C) ((float*)((char*)dst + (size_t)NSTEP * FRAGS * 1024))[u] = scale[u] * b2[u];  // bb behind the stream
  if (u >= NSTEP * FRAGS * 64) return;
  const int l = u & 63, q = (u >> 6) % FRAGS, st = (u >> 6) / FRAGS;
  const int r = l & 31, h = l >> 5;
  float v[8];
  if (q < NCK * F1) {
    const int j = st * NCK + q / F1, s = q % F1;
    if (s < KS1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = W1[(size_t)(32 * j + r) * C + 16 * s + 8 * h + i];
    } else {
      const float b = b1[32 * j + r];
      const float hi = (float)(bf16_t)b;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
      if (h == 0) {
        v[0] = hi;
        v[1] = b - hi;
      }
    }
  } else {
    const int q2 = q - NCK * F1;
    int j = st * NCK + q2 / F2, s = (q2 % F2) / NT2;
    const int t = (q2 % F2) % NT2;
    if (skew) {
      s = q2 < NT2 ? 1 : 0;
      j = q2 < NT2 ? (st + NSTEP - 1) % NSTEP : st;
    }
    const int c = 32 * t + r;
    const float sc = scale[c];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sc * W2[(size_t)c * (4 * C) + 32 * j + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = (bf16_t)v[i];
}


// ---- skewed variant (C = 384: one wave per SIMD, so nothing but the wave's own MFMAs can cover its GELU) ----------------------
// Step j of a tile:   M1(j)  |  M2(j-1, k-half 1) + GELU lo(j)  |  M2(j, k-half 0) + GELU hi(j)
// The first half of a chunk's GELU, which the plain order leaves exposed between the two GEMMs, rides under the second half of
// the PREVIOUS chunk's GEMM2; only the converted upper half of the hidden chunk (4 registers) is carried over.  At a tile
// boundary the last chunk's second half is still owed when the next tile's first step starts: it is paid there, between
// that step's M1 and its own GEMM2 -- the tile's store and the next tile's residual load sit in the middle of step 0.
template <int C> struct Rc2Skew {
  typedef Rc2Geom<C, 1> G;
  typedef Rc2Wave<C, 1> W;
  static constexpr int KS1 = G::KS1, NT2 = G::NT2, F1 = G::F1;
  static constexpr int NM = F1 + 2 * NT2, QA = F1 + NT2;  // MFMAs per step; [0, QA) = phase A, [QA, NM) = phase B
  static constexpr int PRE = 4, R = PRE + 1;
  struct State {
    bf16x8 F[R];
    f32x16 X;
    float g[16];
    bf16x8 Hlo;
  };
  static constexpr int gelu_at(int e) { return e < 8 ? F1 + e * NT2 / 8 : QA + (e - 8) * NT2 / 8; }
  template <int Q, int E>
  static __device__ __forceinline__ void gelu_slices(State& st, bf16x8& Hhi) {
#ifdef CN_SKEW_SCALAR
    if constexpr (gelu_at(E) == Q) st.g[E] = cn_gelu_sig2(st.X[E]);
    if constexpr (gelu_at(E) == Q && (E & 7) == 7) {
#else
    if constexpr ((E & 1) == 0 && gelu_at(E) == Q) {
      const f32x2 r = cn_gelu_sig2_pk(f32x2{st.X[E], st.X[E + 1]});
      st.g[E] = r[0];
      st.g[E + 1] = r[1];
    }
    if constexpr (gelu_at(E & ~1) == Q && (E & 7) == 7) {
#endif
      constexpr int o = E - 7;
      const bf16x8 h = bf16x8{(bf16_t)st.g[o],     (bf16_t)st.g[o + 1], (bf16_t)st.g[o + 2], (bf16_t)st.g[o + 3],
                              (bf16_t)st.g[o + 4], (bf16_t)st.g[o + 5], (bf16_t)st.g[o + 6], (bf16_t)st.g[o + 7]};
      if constexpr (E == 7) st.Hlo = h;
      else Hhi = h;  // (its last reader of this step, MFMA QA - 1, is behind us)
    }
    if constexpr (E + 1 < 16) gelu_slices<Q, E + 1>(st, Hhi);
  }
  template <int Q, int QEND>
  static __device__ __forceinline__ void mstep(const char* wc, const bf16x8 (&fy)[KS1], const bf16x8 ones, f32x16 (&O)[NT2],
                                               bf16x8& Hhi, State& st) {
    if constexpr (Q + PRE < NM) st.F[(Q + PRE) % R] = W::frag(wc, Q + PRE);
    if constexpr (Q < F1) {
      if constexpr (Q == 0) st.X = W::zero16();
      if constexpr (Q < KS1) st.X = W::mma(st.F[Q % R], fy[Q], st.X);
      else st.X = W::mma(st.F[Q % R], ones, st.X);
    } else if constexpr (Q < QA) {
      O[Q - F1] = W::mma(Hhi, st.F[Q % R], O[Q - F1]);
    } else {
      O[Q - QA] = W::mma(st.Hlo, st.F[Q % R], O[Q - QA]);
    }
    gelu_slices<Q, 0>(st, Hhi);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < QEND) mstep<Q + 1, QEND>(wc, fy, ones, O, Hhi, st);
  }
  template <int Q>
  static __device__ __forceinline__ void prefetch(const char* wc, State& st) {
    st.F[Q % R] = W::frag(wc, Q);
    if constexpr (Q + 1 < PRE) prefetch<Q + 1>(wc, st);
  }
  static __device__ __forceinline__ void phase_a(const char* wc, const bf16x8 (&fy)[KS1], const bf16x8 ones, f32x16 (&O)[NT2],
                                                 bf16x8& Hhi, State& st) {
    prefetch<0>(wc, st);
    __builtin_amdgcn_sched_barrier(0);
    mstep<0, QA>(wc, fy, ones, O, Hhi, st);
  }
  static __device__ __forceinline__ void phase_b(const char* wc, const bf16x8 (&fy)[KS1], const bf16x8 ones, f32x16 (&O)[NT2],
                                                 bf16x8& Hhi, State& st) {
    mstep<QA, NM>(wc, fy, ones, O, Hhi, st);
  }
  // the owed second half alone (a wave whose next tile does not exist)
  static __device__ __forceinline__ void tail(const char* wc, f32x16 (&O)[NT2], const bf16x8 Hhi) {
#pragma unroll
    for (int t = 0; t < NT2; ++t) O[t] = W::mma(Hhi, W::frag(wc, F1 + t), O[t]);
  }
};

template <int C, int NW, int NST>
__global__ __launch_bounds__(NW * 64) void cn_mlp_rc2_skew_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ WS,
                                                                  float* __restrict__ X, int M) {
  typedef Rc2Geom<C, 1> G;
  typedef Rc2Wave<C, 1> W;
  typedef Rc2Skew<C> K;
  constexpr int FR = G::FRAGS, SB = G::STEP_BYTES;
  constexpr int DPW_LO = FR / NW, N_HI = FR % NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* aux = (const float*)((const char*)WS + G::STREAM_BYTES);
  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NW - 1) / NW;
  const unsigned voff = lane * 16;
  const unsigned lds0 = cn_lds_addr(smem);
  auto stage = [&](int g) {
    const char* src = (const char*)WS + (size_t)(g % G::NSTEP) * SB;
    const unsigned dst = lds0 + (unsigned)((g % NST) * SB);
#pragma unroll
    for (int i = 0; i < DPW_LO + 1; ++i) {
      const int piece = wave + i * NW;
      if (i < DPW_LO || wave < N_HI) cn_dma16_s(src + piece * 1024, voff, dst + piece * 1024);
    }
  };
  auto ring_wait = [&]() {
    if (N_HI > 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (DPW_LO + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW_LO) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)((lane < 32 && i < 2) ? 1.0f : 0.0f);
#pragma unroll
  for (int g = 0; g < NST - 1; ++g) stage(g);
  bf16x8 fy[G::KS1];
  if (t_lo + wave < t_hi) W::load_y(Y, (t_lo + wave) * 32, lane, fy);
  const char* wl = smem + lane * 16;
  f32x16 O[G::NT2];
#pragma unroll
  for (int t = 0; t < G::NT2; ++t) O[t] = W::zero16();
  bf16x8 Hhi;
#pragma unroll
  for (int i = 0; i < 8; ++i) Hhi[i] = (bf16_t)0.0f;
  bool pend = false;
  int pend_tile = 0;
  int g = 0;
  for (int it = 0; it < max_it; ++it) {
    const int tile = t_lo + wave + it * NW;
    const bool valid = tile < t_hi;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // this tile's y fragments (a wait the compiler sees, on every path into the loop)
    {  // step 0, with the tile boundary in its middle (kept out of the loop below: its loads, stores and temporaries would
       // otherwise weigh on the register allocation of every step)
      ring_wait();
      stage(g + NST - 1);
      const char* wc = wl + (g % NST) * SB;
      typename K::State st;
      if (valid) K::phase_a(wc, fy, ones, O, Hhi, st);
      else if (pend) K::tail(wc, O, Hhi);
      if (pend) {  // the previous tile is complete now
        const float* bbv = aux;
        asm volatile("" : "+s"(bbv));
        W::store_o(X, bbv, pend_tile * 32, M, lane, O);
      }
      if (valid) W::init_o(X, tile * 32, lane, O);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // the residual lands here, once per tile (see the ring kernel)
      pend = valid;
      pend_tile = tile;
      if (valid) K::phase_b(wc, fy, ones, O, Hhi, st);
      ++g;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (spill reloads of the boundary: nothing may be pending on entry to the step loop)
    for (int j = 1; j < G::NSTEP; ++j, ++g) {
      ring_wait();
      stage(g + NST - 1);
      if (valid) {
        const char* wc = wl + (g % NST) * SB;
        typename K::State st;
        K::phase_a(wc, fy, ones, O, Hhi, st);
        K::phase_b(wc, fy, ones, O, Hhi, st);
      }
    }
    if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, lane, fy);
  }
  ring_wait();  // entry g (chunk 0's) holds the k-half the last tile still owes
  if (pend) {
    K::tail(wl + (g % NST) * SB, O, Hhi);
    W::store_o(X, aux, pend_tile * 32, M, lane, O);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C, int NW, int NST>
static int cn_launch_mlp_rc2_skew(const bf16_t* Y, const bf16_t* WS, float* X, int M, int n_blocks, hipStream_t s) {
  constexpr int SMEM = NST * Rc2Geom<C, 1>::STEP_BYTES;
  static_assert(SMEM <= 160 * 1024, "ring must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_rc2_skew_kernel<C, NW, NST>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_rc2_skew_kernel<C, NW, NST>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
