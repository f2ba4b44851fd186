// Fused downsample layer (16-bit operands HT = bf16_t | half_t), stage 0 -> 1:   LayerNorm(channels_first, eps 1e-6) + Conv2d(C -> 2C, k 2x2, s 2)
//     x XT (B, H, W, C)  ->  out XT (B, H/2, W/2, 2C), XT = the residual stream's type (common.h)     (reference convnext.py:207-217)
//
// The two-kernel form (cn_ln_patchify_kernel writes the bf16 patch matrix, cn_gemm2 reads it back) moves the patch matrix
// through HBM twice; here a wave owns 32 OUTPUT positions and builds its GEMM operand in registers:
//   * the K = 4C row of an output position is (kh, kw, c) = its four input positions back to back, so the MFMA fragment of
//     k-step s (8 consecutive k per lane half) is 8 consecutive channels of ONE input position: lane l loads the channels
//     16 j + 8 (l >> 5) .. + 8, j < C/16, of each of its four input positions (fp32), the LayerNorm statistics of an input
//     position are the sums of lanes l and l ^ 32, and the normalised values become the fragment registers directly;
//   * the LayerNorm affine is folded into the packed weights at create time: W'[n][k] = bf16(W[n][k] g[c(k)]),
//     bias'[n] = bias[n] + sum_k W[n][k] b[c(k)] (fp32), so the operand is bf16((x - mean) rstd);
//   * the whole packed weight matrix (K x N bf16 = 144 KB at C = 96) is resident in LDS; persistent blocks, no barrier
//     after the fill; the product runs transposed since round 5 (C^T[n][pos] = W' . a^T, the weight fragment as the A operand):
//     a lane owns one output position and two runs of 8 consecutive channels per 32-channel tile, stored as 16-byte pieces
//     like the fused MLP's epilogue (mlp_rc2.h).
// Packed stream: fragment (s, t), s < K/16, t < N/32: lane l holds W'[32 t + cn_rc2_chan(l & 31)][16 s + 8 (l >> 5) .. + 8]; then bias'[N] fp32.
#pragma once
__device__ void cn_watch_stat(float v);   // the fp16 stream's overflow watch: defined in encoder.hip, the only file that instantiates these kernels
#include "mlp_rc2.h"

template <int CP> struct DownGeom {
  static constexpr int K = 4 * CP, N = 2 * CP, KS = K / 16, NT = N / 32, CH = CP / 16;
  static constexpr size_t STREAM_BYTES = (size_t)KS * NT * 1024;
  static constexpr size_t TOTAL_BYTES = STREAM_BYTES + N * 4;
};

// src: Conv2d weight (N, C, 2, 2) fp32; g, b: LayerNorm weight / bias (C); bias (N)
template <typename HT>
static __global__ void pk_down_fused(const float* __restrict__ src, const float* __restrict__ g, const float* __restrict__ b,
                                     const float* __restrict__ bias, int N, int C, HT* __restrict__ dst) {
  const int K = 4 * C, KS = K / 16, NT = N / 32;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < N) {  // bias' behind the stream
    float a = bias[u];
    for (int k = 0; k < K; ++k) a = fmaf(src[((size_t)u * C + k % C) * 4 + k / C], b[k % C], a);
    ((float*)((char*)dst + (size_t)KS * NT * 1024))[u] = a;
  }
  if (u >= KS * NT * 64) return;
  const int l = u & 63, t = (u >> 6) % NT, s = (u >> 6) / NT;
  const int n = 32 * t + cn_rc2_chan(l & 31);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = 16 * s + 8 * (l >> 5) + i, kk = k / C, c = k % C;
    dst[(size_t)u * 8 + i] = (HT)(src[((size_t)n * C + c) * 4 + kk] * g[c]);
  }
}


// The GEMM operand of a 32-position tile: a[(ip - IP0) * CH + j] = bf16((x - mean) rstd) of channels 16 j + 8 (lane >> 5) .. + 8 of
// input position ip (IP0 <= ip < IP0 + NIP; ip = 2 kh + kw) of output position `tile * 32 + (lane & 31)` (clamped to P - 1).
template <int CP, int IP0, int NIP, typename HT, typename XT>
static __device__ __forceinline__ void cn_down_operand(const XT* __restrict__ X, int H, int W, long P, long tile, int lane,
                                                       cn_h8<HT> (&a)[NIP * DownGeom<CP>::CH]) {
  constexpr int CH = DownGeom<CP>::CH;
  const int H2 = H / 2, W2 = W / 2, hh = lane >> 5;
    // ---- operand: 4 input positions x CH fragments ----------------------------------------------------------------
    const long p = min(tile * 32 + (lane & 31), P - 1);
    const int w2 = (int)(p % W2);
    const long tq = p / W2;
    const int h2 = (int)(tq % H2);
    const long b = tq / H2;
    const XT* x00 = X + (((size_t)b * H + 2 * h2) * W + 2 * w2) * CP + 8 * hh;
    f32x4 v[2][CH][2];
    auto load_ip = [&](int ip, f32x4 (&d)[CH][2]) {
      const XT* src = x00 + ((size_t)(ip >> 1) * W + (ip & 1)) * CP;
      if constexpr (sizeof(XT) == 2) {
        cn_h8<XT> h[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = *(const cn_h8<XT>*)(src + 16 * j);  // one 16-byte piece = the lane's 8 channels
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          d[j][0] = f32x4{(float)h[j][0], (float)h[j][1], (float)h[j][2], (float)h[j][3]};
          d[j][1] = f32x4{(float)h[j][4], (float)h[j][5], (float)h[j][6], (float)h[j][7]};
        }
      } else {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          d[j][0] = *(const f32x4*)(src + 16 * j);
          d[j][1] = *(const f32x4*)(src + 16 * j + 4);
        }
      }
    };
    load_ip(IP0, v[0]);
#pragma unroll
    for (int ii = 0; ii < NIP; ++ii) {
      const int ip = IP0 + ii;
      if (ii + 1 < NIP) load_ip(ip + 1, v[(ii + 1) & 1]);
      f32x4(&d)[CH][2] = v[ii & 1];
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < CH; ++j) s4 += d[j][0] + d[j][1];
      float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / CP);
      f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        d[j][0] -= mean;
        d[j][1] -= mean;
        q4 += d[j][0] * d[j][0] + d[j][1] * d[j][1];
      }
      float q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q * (1.0f / CP) + 1e-6f);
      if (lane < 32) cn_watch_stat(q);   // (the fp16 stream's overflow watch, encoder.hip)
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const f32x4 lo = d[j][0] * rstd, hi = d[j][1] * rstd;
        a[ii * CH + j] = cn_h8<HT>{(HT)lo[0], (HT)lo[1], (HT)lo[2], (HT)lo[3],
                                (HT)hi[0], (HT)hi[1], (HT)hi[2], (HT)hi[3]};
      }
    }
}

// out[tile * 32 + (l & 31)][:] = acc + bias': lane = output position, register r of tile t = channel 32 t + 16 (r >> 3) + 8 (l >> 5) + (r & 7)
template <int CP, typename XT>
static __device__ __forceinline__ void cn_down_store(XT* __restrict__ OUT, const float* __restrict__ biasp, long P, long tile, int lane,
                                                     const f32x16 (&acc)[DownGeom<CP>::NT]) {
  constexpr int NT = DownGeom<CP>::NT, N = DownGeom<CP>::N;
  const int hh = lane >> 5;
  const long m = tile * 32 + (lane & 31);
  if (m >= P) return;
  XT* row = OUT + (size_t)m * N + 8 * hh;
  const float* bl = biasp + 8 * hh;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float o[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const f32x4*)(bl + 32 * t + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * q + e] = acc[t][4 * q + e] + b[e];
    }
    if constexpr (sizeof(XT) == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *(cn_h8<XT>*)(row + 32 * t + 16 * h) = cn_pack8<XT>(o[8 * h], o[8 * h + 1], o[8 * h + 2], o[8 * h + 3], o[8 * h + 4],
                                                            o[8 * h + 5], o[8 * h + 6], o[8 * h + 7]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(row + 32 * t + 16 * (q >> 1) + 4 * (q & 1)) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
    }
  }
}

// MFMA q = s NT + t of a tile consumes fragment q of the packed stream; a rolling window of PRE fragments is in flight.
// (Left to the scheduler, all 144 LDS reads are hoisted to the top and ~500 registers spill.)
template <int CP, typename HT> struct DownMma {
  typedef DownGeom<CP> G;
  static constexpr int NM = G::KS * G::NT, PRE = 6, R = PRE + 1;
  template <int Q>
  static __device__ __forceinline__ void step(const char* wl, const cn_h8<HT> (&a)[G::KS], f32x16 (&acc)[G::NT], cn_h8<HT> (&f)[R]) {
    if constexpr (Q + PRE < NM) f[(Q + PRE) % R] = *(const cn_h8<HT>*)(wl + (Q + PRE) * 1024);
    constexpr int s = Q / G::NT, t = Q % G::NT;
    acc[t] = cn_mma32(f[Q % R], a[s], acc[t]);  // C^T += W' . a^T
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < NM) step<Q + 1>(wl, a, acc, f);
  }
  template <int Q>
  static __device__ __forceinline__ void run(const char* wl, const cn_h8<HT> (&a)[G::KS], f32x16 (&acc)[G::NT]) {
    cn_h8<HT> f[R];
#pragma unroll
    for (int q = 0; q < PRE; ++q) f[q] = *(const cn_h8<HT>*)(wl + q * 1024);
    __builtin_amdgcn_sched_barrier(0);
    step<0>(wl, a, acc, f);
  }
};

template <int CP, int NW, typename HT, typename XT>
__global__ __launch_bounds__(NW * 64) void cn_down_fused_kernel(const XT* __restrict__ X, int H, int W, long P,
                                                                const HT* __restrict__ WS, XT* __restrict__ OUT) {
  typedef DownGeom<CP> G;
  constexpr int KS = G::KS, NT = G::NT, CH = G::CH, N = G::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PIECES = (int)(G::STREAM_BYTES / 1024);
  for (int i = wave; i < PIECES; i += NW)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)WS + (size_t)i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
  const float* biasp = (const float*)((const char*)WS + G::STREAM_BYTES);
  const long n_tiles = (P + 31) >> 5;
  const long t_lo = (long)blockIdx.x * n_tiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * n_tiles / gridDim.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* wl = smem + lane * 16;
  for (long tile = t_lo + wave; tile < t_hi; tile += NW) {
    cn_h8<HT> a[KS];
    cn_down_operand<CP, 0, 4, HT, XT>(X, H, W, P, tile, lane, a);
    // ---- C[pos][n] = sum_k a[pos][k] W'[n][k]: positions in the registers' rows, channels on the lanes ------------------
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    DownMma<CP, HT>::template run<0>(wl, a, acc);  // hand-ordered: fragment q + PRE is requested before MFMA q (see mlp_rc2.h)
    cn_down_store<CP, XT>(OUT, biasp, P, tile, lane, acc);
  }
}

template <int CP, int NW, typename HT, typename XT>
static int cn_launch_down_fused(const XT* X, int B, int H, int W, const void* WS, XT* OUT, int n_blocks, hipStream_t s) {
  typedef DownGeom<CP> G;
  constexpr int SMEM = (int)G::STREAM_BYTES;
  static_assert(SMEM <= 160 * 1024, "the packed weights must fit in LDS");
  const long P = (long)B * (H / 2) * (W / 2);
  CN_TRY(cn_configure_lds((const void*)cn_down_fused_kernel<CP, NW, HT, XT>, SMEM));
  const int grid = cn_rc2_grid((int)((P + 31) / 32), NW, n_blocks);
  hipLaunchKernelGGL((cn_down_fused_kernel<CP, NW, HT, XT>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, X, H, W, P, (const HT*)WS, OUT);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---- ring variant (C = 192: K = 768, N = 384, 576 KB of weights): the stream goes L2 -> LDS in steps of KSTEP k-steps --
// One wave per SIMD (accumulators 192 registers + the operand of HALF the K range, 96); a tile is NSTEP = KS / KSTEP ring steps, each
// wait (counted vmcnt) -> barrier -> request entry g + NST - 1 -> KSTEP NT MFMAs, fully unrolled per tile because the
// operand fragment a[s] must be a compile-time register choice.  NSTEP is a multiple of NST, so the slot of step J of a
// tile is J % NST.  The vmcnt discipline is mlp_rc2.h's: the operand loads and the stores of a tile sit in the same
// in-order queue and only make a wait stricter.
template <int CP, int NW, int KSTEP, int NST, typename HT> struct DownRing {
  typedef DownGeom<CP> G;
  static constexpr int NT = G::NT, KS = G::KS, NSTEP = KS / KSTEP, FR = KSTEP * NT, SB = FR * 1024;
  static constexpr int DPW = FR / NW, PRE = 6, R = PRE + 1;
  static_assert(KS % KSTEP == 0 && NSTEP % NST == 0 && FR % NW == 0, "ring geometry");
  static constexpr int SMEM = NST * SB;

  static __device__ __forceinline__ void stage(const char* wsrc, char* smem, int wave, int g) {
    const char* src = wsrc + (size_t)(g % NSTEP) * SB;
    char* dst = smem + (g % NST) * SB;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int piece = wave + i * NW;
      cn_dma16_v(src + piece * 1024, cn_lds_addr(dst + piece * 1024));
    }
  }
  static constexpr int HS = NSTEP / 2;  // steps per half tile (input positions 0, 1 | 2, 3)
  static_assert(NSTEP % 2 == 0 && (KS / 2) % KSTEP == 0, "a half tile is a whole number of steps");
  template <int J, int Q>
  static __device__ __forceinline__ void mma(const char* wl, const cn_h8<HT> (&a)[KS / 2], f32x16 (&acc)[NT], cn_h8<HT> (&f)[R]) {
    if constexpr (Q + PRE < FR) f[(Q + PRE) % R] = *(const cn_h8<HT>*)(wl + (Q + PRE) * 1024);
    constexpr int s = (J % HS) * KSTEP + Q / NT, t = Q % NT;
    acc[t] = cn_mma32(f[Q % R], a[s], acc[t]);  // C^T += W' . a^T
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < FR) mma<J, Q + 1>(wl, a, acc, f);
  }
  // steps J .. JE - 1 of a tile (one half)
  template <int J, int JE>
  static __device__ __forceinline__ void steps(const char* wsrc, char* smem, int wave, int lane, int g0, bool valid,
                                               const cn_h8<HT> (&a)[KS / 2], f32x16 (&acc)[NT]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(wsrc, smem, wave, g0 + J + NST - 1);
    if (valid) {
      const char* wl = smem + (J % NST) * SB + lane * 16;
      cn_h8<HT> f[R];
#pragma unroll
      for (int q = 0; q < PRE; ++q) f[q] = *(const cn_h8<HT>*)(wl + q * 1024);
      __builtin_amdgcn_sched_barrier(0);
      mma<J, 0>(wl, a, acc, f);
    }
    if constexpr (J + 1 < JE) steps<J + 1, JE>(wsrc, smem, wave, lane, g0, valid, a, acc);
  }
};

template <int CP, int NW, int KSTEP, int NST, typename HT, typename XT>
__global__ __launch_bounds__(NW * 64) void cn_down_fused_ring_kernel(const XT* __restrict__ X, int H, int W, long P,
                                                                     const HT* __restrict__ WS, XT* __restrict__ OUT) {
  typedef DownGeom<CP> G;
  typedef DownRing<CP, NW, KSTEP, NST, HT> K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* biasp = (const float*)((const char*)WS + G::STREAM_BYTES);
  const char* wsrc = (const char*)WS + lane * 16;
  const long n_tiles = (P + 31) >> 5;
  const long t_lo = (long)blockIdx.x * n_tiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * n_tiles / gridDim.x;
  const int max_it = (int)((t_hi - t_lo + NW - 1) / NW);  // block-uniform: every wave runs the same number of steps
#pragma unroll
  for (int g = 0; g < NST - 1; ++g) K::stage(wsrc, smem, wave, g);
  for (int it = 0; it < max_it; ++it) {
    const long tile = t_lo + wave + (long)it * NW;
    const bool valid = tile < t_hi;
    const long ltile = valid ? tile : t_hi - 1;
    f32x16 acc[G::NT];
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    {  // K in two halves: the operand of two input positions (96 registers) at a time
      cn_h8<HT> a[G::KS / 2];
      cn_down_operand<CP, 0, 2, HT, XT>(X, H, W, P, ltile, lane, a);
      K::template steps<0, K::HS>(wsrc, smem, wave, lane, it * K::NSTEP, valid, a, acc);
      cn_down_operand<CP, 2, 2, HT, XT>(X, H, W, P, ltile, lane, a);
      K::template steps<K::HS, K::NSTEP>(wsrc, smem, wave, lane, it * K::NSTEP, valid, a, acc);
    }
    if (valid) cn_down_store<CP, XT>(OUT, biasp, P, tile, lane, acc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 entries past the end
}

template <int CP, int NW, int KSTEP, int NST, typename HT, typename XT>
static int cn_launch_down_fused_ring(const XT* X, int B, int H, int W, const void* WS, XT* OUT, int n_blocks, hipStream_t s) {
  typedef DownRing<CP, NW, KSTEP, NST, HT> K;
  static_assert(K::SMEM <= 160 * 1024, "ring must fit in LDS");
  const long P = (long)B * (H / 2) * (W / 2);
  CN_TRY(cn_configure_lds((const void*)cn_down_fused_ring_kernel<CP, NW, KSTEP, NST, HT, XT>, K::SMEM));
  const int grid = cn_rc2_grid((int)((P + 31) / 32), NW, n_blocks);
  hipLaunchKernelGGL((cn_down_fused_ring_kernel<CP, NW, KSTEP, NST, HT, XT>), dim3((unsigned)grid), dim3(NW * 64), K::SMEM, s, X, H, W, P,
                     (const HT*)WS, OUT);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
