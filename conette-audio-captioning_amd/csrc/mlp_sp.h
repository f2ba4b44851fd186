// Register-chained fused ConvNeXt MLP for the "exact" precision (CONETTE_PREC_F16X2), stages 0-1 (C = 96, 192):
//
//     x[m][:] += scale * ( W2 . gelu( W1 . y[m][:] + b1 ) + b2 )          (convnext.py:66-74)
//
// The wave program of mlp_rc2.h -- a wave owns 32 positions; per 32 hidden units GEMM1 -> GELU -> GEMM2 with the GEMM1
// accumulator converted in place into the A operand of GEMM2, O starting from the residual -- with every operand an fp16
// hi + lo pair (common.h sp16) and every product three v_mfma_f32_32x32x16_f16 (lo.hi + hi.lo + hi.hi, small terms first):
//   * y arrives as sp16 rows (4 bytes per element, written by the depthwise-conv kernel); a lane's 32 bytes per k-step are
//     split into a hi and a lo fragment once per tile (v_perm_b32);
//   * W1, W2' = s W2 (LayerScale folded in fp32, then split) stream through the LDS ring as separate hi and lo fragments;
//   * b1 as one more k-step: (hi16(b1), lo16(b1)) against a "ones" fragment -- exact to 2^-22;
//   * GELU on the fp32 accumulator: A&S erfc without cancellation (common.h cn_gelu_as2, 1.5e-7), then hi / lo split.
// Unfused, the exact mode moves the 4C hidden through HBM at 4 bytes per element -- 1.39 GB per stage-0 block at B = 64,
// written by pw1 and read back by pw2: 1.22 ms per block (profiles/r03_exact_encoder_by_grid.csv); here it stays in the
// registers.  Stage 2 (C = 384: y pairs + O = 384 registers) and stage 3 keep the unfused sp16 GEMMs of gemm2.h: a one-wave-
// per-SIMD version with half entries in the ring (tools/lab/mlp_sp_split.h) took 553 us per launch against 311 + 217 unfused.
//
// Ring entry of hidden chunk j (1 KB fragments, lane l reads 16 B at 16 l; packed by pk_mlp_sp):
//   [W1 hi k-step 0, W1 lo k-step 0, W1 hi 1, W1 lo 1, ...] [bias] [W2' hi (k 0, t 0), lo (k 0, t 0), hi (k 0, t 1), ...]
// followed by bb = s b2 (fp32, C).
#pragma once
#include "mlp_rc2.h"

template <int C> struct SpGeom {
  static constexpr int KS1 = C / 16, NT2 = C / 32, NSTEP = C / 8;
  static constexpr int F1 = 2 * KS1 + 1, F2 = 4 * NT2, FRAGS = F1 + F2;
  static constexpr int STEP_BYTES = FRAGS * 1024;
  static constexpr size_t STREAM_BYTES = (size_t)NSTEP * STEP_BYTES;
  static constexpr size_t TOTAL_BYTES = STREAM_BYTES + C * 4;
};

static __global__ void pk_mlp_sp(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                 const float* __restrict__ b2, const float* __restrict__ scale, int C, _Float16* __restrict__ dst) {
  const int KS1 = C / 16, NT2 = C / 32, NCH = C / 8, F1 = 2 * KS1 + 1, F2 = 4 * NT2, FRAGS = F1 + F2;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < C) ((float*)((char*)dst + (size_t)NCH * FRAGS * 1024))[u] = scale[u] * b2[u];  // bb behind the stream
  if (u >= NCH * FRAGS * 64) return;
  const int l = u & 63, q = (u >> 6) % FRAGS, j = (u >> 6) / FRAGS;
  const int r = l & 31, h = l >> 5;
  float v[8];
  bool lo;
  if (q < 2 * KS1) {
    const int s = q >> 1;
    lo = q & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = W1[(size_t)(32 * j + r) * C + 16 * s + 8 * h + i];
  } else if (q == 2 * KS1) {  // bias k-step: values 0 / 1 of the lanes < 32 = hi16(b1), lo16(b1)
    const float b = b1[32 * j + r];
    const float hi = (float)(_Float16)b;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (h == 0) {
      v[0] = hi;
      v[1] = b - hi;
    }
    lo = false;  // (both already fp16-representable parts: stored as they are)
  } else {
    const int q2 = q - F1, kt = q2 >> 1, k = kt / NT2, t = kt % NT2, c = 32 * t + r;
    lo = q2 & 1;
    const float sc = scale[c];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sc * W2[(size_t)c * (4 * C) + 32 * j + 16 * k + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 hi = (_Float16)v[i];
    dst[(size_t)u * 8 + i] = lo ? (_Float16)(v[i] - (float)hi) : hi;
  }
}

template <int C> struct SpWave {
  typedef SpGeom<C> G;
  typedef Rc2Wave<C, 1> W;  // residual I/O (init_o / store_o): the accumulator layout is the bf16 kernel's
  static constexpr int KS1 = G::KS1, NT2 = G::NT2, F1 = G::F1, F2 = G::F2;

  static __device__ __forceinline__ f16x8 frag(const char* wc, int f) { return *(const f16x8*)(wc + f * 1024); }
  static __device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f16x8 pack16(const float (&g)[8], bool lo) {
    f16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const _Float16 hi = (_Float16)g[i];
      r[i] = lo ? (_Float16)(g[i] - (float)hi) : hi;
    }
    return r;
  }

  // y rows of a 32-position tile as hi / lo B fragments: (yh, yl)[s] = split(y[m0 + (l & 31)][16 s + 8 (l >> 5) .. + 8])
  static __device__ __forceinline__ void load_y(const sp16_t* __restrict__ Y, int m0, int lane, f16x8 (&yh)[KS1], f16x8 (&yl)[KS1]) {
    const sp16_t* base = Y + (size_t)m0 * C;                 // scalar
    const int voff = (lane & 31) * C + 8 * (lane >> 5);      // elements
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const u32x4 c0 = *(const u32x4*)(base + voff + 16 * s), c1 = *(const u32x4*)(base + voff + 16 * s + 4);
      u32x4 hh, ll;
      hh[0] = __builtin_amdgcn_perm(c0[1], c0[0], 0x05040100u);
      hh[1] = __builtin_amdgcn_perm(c0[3], c0[2], 0x05040100u);
      hh[2] = __builtin_amdgcn_perm(c1[1], c1[0], 0x05040100u);
      hh[3] = __builtin_amdgcn_perm(c1[3], c1[2], 0x05040100u);
      ll[0] = __builtin_amdgcn_perm(c0[1], c0[0], 0x07060302u);
      ll[1] = __builtin_amdgcn_perm(c0[3], c0[2], 0x07060302u);
      ll[2] = __builtin_amdgcn_perm(c1[1], c1[0], 0x07060302u);
      ll[3] = __builtin_amdgcn_perm(c1[3], c1[2], 0x07060302u);
      yh[s] = __builtin_bit_cast(f16x8, hh);
      yl[s] = __builtin_bit_cast(f16x8, ll);
    }
  }

  // one hidden chunk in two halves (the split-entry kernel puts a barrier between them; the ring kernel calls both):
  // step_a: GEMM1 + bias + GELU + hi / lo split from the W1 fragments at wc; step_b: GEMM2 from the W2 fragments at w2.
  // Fragment pairs (hi, lo) are read PRE pairs ahead of the MFMAs that consume them (a pair feeds three MFMAs = 96 cycles;
  // an LDS read takes longer than that to come back).
  static constexpr int PRE = 3, RB = PRE + 1;
  template <int S>
  static __device__ __forceinline__ void a_k(const char* wc, const f16x8 (&yh)[KS1], const f16x8 (&yl)[KS1], f16x8 (&Fh)[RB],
                                             f16x8 (&Fl)[RB], f32x16& X) {
    if constexpr (S + PRE < KS1) {
      Fh[(S + PRE) % RB] = frag(wc, 2 * (S + PRE));
      Fl[(S + PRE) % RB] = frag(wc, 2 * (S + PRE) + 1);
    } else if constexpr (S + PRE == KS1) {
      Fh[(S + PRE) % RB] = frag(wc, 2 * KS1);  // the bias fragment
    }
    X = mma(Fh[S % RB], yl[S], X);
    X = mma(Fl[S % RB], yh[S], X);
    X = mma(Fh[S % RB], yh[S], X);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < KS1) a_k<S + 1>(wc, yh, yl, Fh, Fl, X);
  }
  static __device__ __forceinline__ void step_a(const char* wc, const f16x8 (&yh)[KS1], const f16x8 (&yl)[KS1], const f16x8 ones,
                                                f16x8 (&Gh)[2], f16x8 (&Gl)[2]) {
    static_assert(KS1 > PRE, "prefetch depth");
    f32x16 X = W::zero16();
    f16x8 Fh[RB], Fl[RB];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      Fh[i] = frag(wc, 2 * i);
      Fl[i] = frag(wc, 2 * i + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    a_k<0>(wc, yh, yl, Fh, Fl, X);
    X = mma(Fh[KS1 % RB], ones, X);
    // GELU (A&S erfc, cancellation free) and the hi / lo split of the chunk: registers 8 k .. 8 k + 7 are k-step k of GEMM2
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float g[8];
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const f32x2 r = cn_gelu_as2(f32x2{X[8 * k + i], X[8 * k + i + 1]});
        g[i] = r[0];
        g[i + 1] = r[1];
      }
      Gh[k] = pack16(g, false);
      Gl[k] = pack16(g, true);
    }
  }
  template <int KT>
  static __device__ __forceinline__ void b_k(const char* w2, const f16x8 (&Gh)[2], const f16x8 (&Gl)[2], f16x8 (&Fh)[RB],
                                             f16x8 (&Fl)[RB], f32x16 (&O)[NT2]) {
    if constexpr (KT + PRE < 2 * NT2) {
      Fh[(KT + PRE) % RB] = frag(w2, 2 * (KT + PRE));
      Fl[(KT + PRE) % RB] = frag(w2, 2 * (KT + PRE) + 1);
    }
    constexpr int k = KT / NT2, t = KT % NT2;
    O[t] = mma(Gl[k], Fh[KT % RB], O[t]);
    O[t] = mma(Gh[k], Fl[KT % RB], O[t]);
    O[t] = mma(Gh[k], Fh[KT % RB], O[t]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KT + 1 < 2 * NT2) b_k<KT + 1>(w2, Gh, Gl, Fh, Fl, O);
  }
  static __device__ __forceinline__ void step_b(const char* w2, const f16x8 (&Gh)[2], const f16x8 (&Gl)[2], f32x16 (&O)[NT2]) {
    static_assert(2 * NT2 > PRE, "prefetch depth");
    f16x8 Fh[RB], Fl[RB];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      Fh[i] = frag(w2, 2 * i);
      Fl[i] = frag(w2, 2 * i + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    b_k<0>(w2, Gh, Gl, Fh, Fl, O);
  }
  static __device__ __forceinline__ void step(const char* wc, const f16x8 (&yh)[KS1], const f16x8 (&yl)[KS1], const f16x8 ones,
                                              f32x16 (&O)[NT2]) {
    f16x8 Gh[2], Gl[2];
    step_a(wc, yh, yl, ones, Gh, Gl);
    __builtin_amdgcn_sched_barrier(0);
    step_b(wc + F1 * 1024, Gh, Gl, O);
  }
};

// Ring kernel: the protocol of cn_mlp_rc2_ring_kernel (mlp_rc2.h) -- entry g + NST - 1 requested at the start of step g,
// counted vmcnt per wave, one raw barrier per step, residual / y landing in ONE wait per tile.
template <int C, int NW, int NST>
__global__ __launch_bounds__(NW * 64) void cn_mlp_sp_ring_kernel(const sp16_t* __restrict__ Y, const _Float16* __restrict__ WS,
                                                                 float* __restrict__ X, int M) {
  typedef SpGeom<C> G;
  typedef SpWave<C> SW;
  typedef Rc2Wave<C, 1> W;
  constexpr int FR = G::FRAGS, SB = G::STEP_BYTES;
  constexpr int DPW_LO = FR / NW, N_HI = FR % NW;  // waves < N_HI issue DPW_LO + 1 pieces per entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* aux = (const float*)((const char*)WS + G::STREAM_BYTES);

  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NW - 1) / NW;  // block-uniform: every wave runs the same number of steps

  const unsigned voff = lane * 16;
  const unsigned lds0 = cn_lds_addr(smem);
  auto stage = [&](int g) {  // stream step g % NSTEP -> slot g % NST (this wave's pieces)
    const char* src = (const char*)WS + (size_t)(g % G::NSTEP) * SB;  // wave-uniform
    const unsigned dst = lds0 + (unsigned)((g % NST) * SB);
#pragma unroll
    for (int i = 0; i < DPW_LO + 1; ++i) {
      const int piece = wave + i * NW;
      if (i < DPW_LO || wave < N_HI) cn_dma16_s(src + piece * 1024, voff, dst + piece * 1024);
    }
  };
  f16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (_Float16)((lane < 32 && i < 2) ? 1.0f : 0.0f);

#pragma unroll
  for (int g = 0; g < NST - 1; ++g) stage(g);
  f16x8 yh[G::KS1], yl[G::KS1];
  if (t_lo + wave < t_hi) SW::load_y(Y, (t_lo + wave) * 32, lane, yh, yl);
  const char* wl = smem + lane * 16;
  int g = 0;
  for (int it = 0; it < max_it; ++it) {
    const int tile = t_lo + wave + it * NW;
    const bool valid = tile < t_hi;
    f32x16 O[G::NT2];
    if (valid) Rc2IoCl<C>::init_o(X, tile * 32, lane, O);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): residual + y, once per tile, on every path into the loop
    for (int j = 0; j < G::NSTEP; ++j, ++g) {
      if (N_HI > 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (DPW_LO + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW_LO) : "memory");
      __builtin_amdgcn_s_barrier();
      stage(g + NST - 1);
      if (valid) SW::step(wl + (g % NST) * SB, yh, yl, ones, O);
    }
    if (tile + NW < t_hi) SW::load_y(Y, (tile + NW) * 32, lane, yh, yl);
    if (valid) {
      const float* bbv = aux;
      asm volatile("" : "+s"(bbv));  // re-read per tile (mlp_rc2.h)
      Rc2IoCl<C>::store_o(X, bbv, tile * 32, M, lane, O);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 entries past the end
}

template <int C, int NW, int NST>
static int cn_launch_mlp_sp_ring(const sp16_t* Y, const void* WS, float* X, int M, int n_blocks, hipStream_t s) {
  constexpr int SMEM = NST * SpGeom<C>::STEP_BYTES;
  static_assert(SMEM <= 160 * 1024, "ring must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_sp_ring_kernel<C, NW, NST>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_sp_ring_kernel<C, NW, NST>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, (const _Float16*)WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

static int cn_pack_mlp_sp(const float* W1, const float* b1, const float* W2, const float* b2, const float* ls, int C, void* dst,
                          hipStream_t s) {
  const int units = (C / 8) * (2 * (C / 16) + 1 + 4 * (C / 32)) * 64;
  hipLaunchKernelGGL(pk_mlp_sp, dim3((units + 255) / 256), dim3(256), 0, s, W1, b1, W2, b2, ls, C, (_Float16*)dst);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
