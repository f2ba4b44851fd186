// EXPERIMENT, NOT PART OF THE LIBRARY since round 6 (the fp8 precision was withdrawn: profiles/r06_notes.md section 3); kept here so that
// tools/lab/mlp_lab.hip -- the harness of all fused MLP kernels -- still builds its fp8 variants (-I tools/lab).
// Register-chained fused ConvNeXt MLP with FP8 (OCP e4m3) operands -- CONETTE_PREC_FP8's variant of mlp_rc2.h for the
// pointwise convolutions of stages 0-2 (BASELINE.json configs[4]: "fp8 MFMA pointwise GEMMs"; everything else of that
// precision runs the bf16 kernels).  Accuracy: 4.5-4.8 % of a block's mean |update| off the bf16 kernel per block (lab,
// profiles/r02_notes.md "FP8 operands"); tests/test_gpu_fp8.py pins it against oracle/fp8_ref.py, which quantises at the
// same points.
//
//     x[m][:] += scale * ( W2 . gelu( W1 . y[m][:] + b1 ) + b2 )          (convnext.py:66-74)
//
// Same wave program as mlp_rc2.h (32 positions per wave, hidden chunks of 32, GEMM1 accumulator -> GELU -> A operand of
// GEMM2, weights streamed through an LDS ring or resident in LDS), with v_mfma_f32_32x32x16_fp8_fp8: fragments are 8 bytes
// per lane, so the stream, the LDS reads and the matrix-core time are all half of the bf16 kernel's.  Quantisation:
//   * y arrives as e4m3 (the depthwise-conv + LayerNorm kernel stores it that way, scale 1);
//   * W1q = e4m3(s1 W1) with ONE scale s1 = 448 / max|W1| (removed in the GELU argument: x = acc / s1 + b1, one fma
//     per element; b1 stays fp32, read from LDS in accumulator order -- an e4m3 bias k-step would be far too coarse);
//   * h = e4m3(gelu(x)), scale 1;
//   * W2q[c][:] = e4m3(ls[c] W2[c][:] / sc[c]) with a scale PER OUTPUT CHANNEL sc[c] = max_k |ls[c] W2[c][k]| / 448, which also
//     absorbs the LayerScale (its rows differ by orders of magnitude): the accumulators start from x / sc[c] and the epilogue
//     is x' = sc[c] O + ls[c] b2[c].
// Packed stream per step (hidden chunk j): [W1q fragments k-step 0 .. C/16-1] [W2q fragments (k 0, tile t) t < C/32, (k 1, t)],
// 512 bytes each (lane l reads 8 bytes at 8 l); behind the stream, fp32: b1 in accumulator order [chunk][lane >> 5][16],
// sc[C], 1/sc[C], bb[C] = ls b2, 1/s1.
#pragma once
#include "mlp_rc2.h"

template <int C> struct Rc2F8Geom {
  static constexpr int KS1 = C / 16, NT2 = C / 32, NSTEP = C / 8;
  static constexpr int F1 = KS1, F2 = 2 * NT2, FRAGS = F1 + F2;
  static constexpr int STEP_BYTES = FRAGS * 512;
  static constexpr size_t STREAM_BYTES = (size_t)NSTEP * STEP_BYTES;
  static constexpr int AUX_B1 = 0, AUX_SC = 4 * C, AUX_ISC = 5 * C, AUX_BB = 6 * C, AUX_IS1 = 7 * C, AUX_FLOATS = 7 * C + 4;
  static constexpr size_t TOTAL_BYTES = STREAM_BYTES + (size_t)AUX_FLOATS * 4;
  static_assert(FRAGS % 2 == 0, "the ring moves 1 KB pieces (two fragments)");
};

// ---- packing ---------------------------------------------------------------------------------------------------------
// one block: scales, bias vectors (aux = dst + STREAM_BYTES)
static __global__ void pk_mlp_f8_scales(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                        const float* __restrict__ b2, const float* __restrict__ ls, int C, float* __restrict__ aux) {
  __shared__ float red[256];
  float m = 0.f;
  for (long i = threadIdx.x; i < 4L * C * C; i += blockDim.x) m = fmaxf(m, fabsf(W1[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float amax = red[0] > 0.f ? red[0] : 1.f;
  if (threadIdx.x == 0) aux[7 * C] = amax / 448.0f;  // 1 / s1
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float r = 0.f;
    for (int k = 0; k < 4 * C; ++k) r = fmaxf(r, fabsf(ls[c] * W2[(size_t)c * 4 * C + k]));
    const float sc = r > 0.f ? r / 448.0f : 1.f;
    aux[4 * C + c] = sc;
    aux[5 * C + c] = 1.0f / sc;
    aux[6 * C + c] = ls[c] * b2[c];
  }
  for (int u = threadIdx.x; u < 4 * C; u += blockDim.x) {  // b1 in accumulator order: [chunk j][h][r] = b1[32 j + (r&3) + 8 (r>>2) + 4 h]
    const int r = u & 15, h = (u >> 4) & 1, j = u >> 5;
    aux[u] = b1[32 * j + (r & 3) + 8 * (r >> 2) + 4 * h];
  }
}

static __device__ __forceinline__ long cn_pack_fp8x8(const float (&v)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

static __global__ void pk_mlp_f8(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ ls, int C,
                                 const float* __restrict__ aux, long* __restrict__ dst) {
  const int KS1 = C / 16, NT2 = C / 32, NSTEP = C / 8, F1 = KS1, FRAGS = F1 + 2 * NT2;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= NSTEP * FRAGS * 64) return;
  const int l = u & 63, q = (u >> 6) % FRAGS, j = (u >> 6) / FRAGS;
  const int r = l & 31, h = l >> 5;
  float v[8];
  if (q < F1) {
    const float s1 = 1.0f / aux[7 * C];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s1 * W1[(size_t)(32 * j + r) * C + 16 * q + 8 * h + i];
  } else {
    const int q2 = q - F1, k = q2 / NT2, t = q2 % NT2, c = 32 * t + r;
    const float f = ls[c] * aux[5 * C + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f * W2[(size_t)c * (4 * C) + 32 * j + 16 * k + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
  dst[u] = cn_pack_fp8x8(v);
}

static int cn_pack_mlp_f8(const float* W1, const float* b1, const float* W2, const float* b2, const float* ls, int C, void* dst,
                          hipStream_t s) {
  const size_t stream = (size_t)(C / 8) * (C / 16 + C / 16) * 512;
  float* aux = (float*)((char*)dst + stream);
  hipLaunchKernelGGL(pk_mlp_f8_scales, dim3(1), dim3(256), 0, s, W1, b1, W2, b2, ls, C, aux);
  const int n = (C / 8) * (C / 8) * 64;
  hipLaunchKernelGGL(pk_mlp_f8, dim3((n + 255) / 256), dim3(256), 0, s, W1, W2, ls, C, aux, (long*)dst);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---- the wave program ------------------------------------------------------------------------------------------------
template <int C> struct Rc2F8Wave {
  typedef Rc2F8Geom<C> G;
  static constexpr int KS1 = G::KS1, NT2 = G::NT2, F1 = G::F1, F2 = G::F2;
  static constexpr int PRE = 4, NM = F1 + F2, R = PRE + 1;

  static __device__ __forceinline__ long frag(const char* wc, int f) { return *(const long*)(wc + f * 512); }
  static __device__ __forceinline__ f32x16 mma(long a, long b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
  }
  struct State {
    long F[R];
    f32x16 X;
    f32x4 B[4];  // b1 of this chunk in accumulator order
    float g[16];
    long H[2];
  };
  // GELU element pair (e, e + 1) rides behind MFMA gelu_at(e): the first 8 behind the last GEMM1 MFMA, the rest under GEMM2 (k 0)
  static constexpr int gelu_at(int e) { return e < 8 ? F1 - 1 : F1 + (e - 8) * NT2 / 8; }
  template <int Q, int E>
  static __device__ __forceinline__ void gelu_slices(State& st, float is1) {
    if constexpr (gelu_at(E) == Q) {
      const f32x2 xin = f32x2{st.X[E], st.X[E + 1]} * is1 + f32x2{st.B[E >> 2][E & 3], st.B[(E + 1) >> 2][(E + 1) & 3]};
      const f32x2 r = cn_gelu_sig2_pk(xin);
      st.g[E] = r[0];
      st.g[E + 1] = r[1];
      if constexpr ((E & 7) == 6) {
        constexpr int o = E - 6;
        const float v[8] = {st.g[o], st.g[o + 1], st.g[o + 2], st.g[o + 3], st.g[o + 4], st.g[o + 5], st.g[o + 6], st.g[o + 7]};
        st.H[E >> 3] = cn_pack_fp8x8(v);
      }
    }
    if constexpr (E + 2 < 16) gelu_slices<Q, E + 2>(st, is1);
  }
  template <int Q>
  static __device__ __forceinline__ void mstep(const char* wc, const long (&fy)[KS1], f32x16 (&O)[NT2], State& st, float is1) {
    if constexpr (Q + PRE < NM) st.F[(Q + PRE) % R] = frag(wc, Q + PRE);
    if constexpr (Q < F1) {
      if constexpr (Q == 0) st.X = zero16();
      st.X = mma(st.F[Q % R], fy[Q], st.X);
    } else {
      constexpr int q2 = Q - F1, k = q2 / NT2, t = q2 % NT2;
      O[t] = mma(st.H[k], st.F[Q % R], O[t]);
    }
    gelu_slices<Q, 0>(st, is1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < NM) mstep<Q + 1>(wc, fy, O, st, is1);
  }
  template <int Q>
  static __device__ __forceinline__ void prefetch(const char* wc, State& st) {
    st.F[Q % R] = frag(wc, Q);
    if constexpr (Q + 1 < PRE) prefetch<Q + 1>(wc, st);
  }
  // wc: this step's fragments (LDS, + lane * 8); bj: this chunk's biases (LDS, + (lane >> 5) * 16 floats)
  static __device__ __forceinline__ void step(const char* wc, const float* bj, const long (&fy)[KS1], f32x16 (&O)[NT2], float is1) {
    State st;
    prefetch<0>(wc, st);
#pragma unroll
    for (int i = 0; i < 4; ++i) st.B[i] = *(const f32x4*)(bj + 4 * i);
    __builtin_amdgcn_sched_barrier(0);
    mstep<0>(wc, fy, O, st, is1);
  }

  // y rows of a 32-position tile as B fragments: fy[s] = y8[m0 + (l & 31)][16 s + 8 (l >> 5) .. + 8]
  static __device__ __forceinline__ void load_y(const unsigned char* __restrict__ Y, int m0, int lane, long (&fy)[KS1]) {
    const unsigned char* base = Y + (size_t)m0 * C;          // scalar
    const int voff = (lane & 31) * C + 8 * (lane >> 5);      // bytes
#pragma unroll
    for (int s = 0; s < KS1; ++s) fy[s] = *(const long*)(base + voff + 16 * s);
  }
  // O = x / sc: lane = channel 32 t + (l & 31), register r = position (r&3) + 8 (r>>2) + 4 (l>>5)
  static __device__ __forceinline__ void init_o(const float* __restrict__ X, const float* __restrict__ aux, int m0, int lane,
                                                f32x16 (&O)[NT2]) {
    const int voff = 4 * (lane >> 5) * C + (lane & 31);
    float isc[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) isc[t] = aux[G::AUX_ISC + 32 * t + (lane & 31)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* row = X + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * C;  // scalar
#pragma unroll
      for (int t = 0; t < NT2; ++t) O[t][r] = row[voff + 32 * t] * isc[t];
    }
  }
  // x' = sc O + bb
  static __device__ __forceinline__ void store_o(float* __restrict__ X, const float* __restrict__ aux, int m0, int M, int lane,
                                                 const f32x16 (&O)[NT2]) {
    const int voff = 4 * (lane >> 5) * C + (lane & 31);
    const int plim = M - m0 - 4 * (lane >> 5);
    float sc[NT2], bb[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      sc[t] = aux[G::AUX_SC + 32 * t + (lane & 31)];
      bb[t] = aux[G::AUX_BB + 32 * t + (lane & 31)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* row = X + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * C;  // scalar
      if ((r & 3) + 8 * (r >> 2) < plim) {
#pragma unroll
        for (int t = 0; t < NT2; ++t) row[voff + 32 * t] = fmaf(O[t][r], sc[t], bb[t]);
      }
    }
  }
};

// ---- resident variant (C = 96): the whole stream (72 KB) lives in LDS -------------------------------------------------
template <int C, int NW>
__global__ __launch_bounds__(NW * 64) void cn_mlp_f8_resident_kernel(const unsigned char* __restrict__ Y, const char* __restrict__ WS,
                                                                     float* __restrict__ X, int M) {
  typedef Rc2F8Geom<C> G;
  typedef Rc2F8Wave<C> W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PIECES = (int)(G::STREAM_BYTES / 1024);
  for (int i = wave; i < PIECES; i += NW)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(WS + (size_t)i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
  const float* aux = (const float*)(WS + G::STREAM_BYTES);
  float* b1s = (float*)(smem + G::STREAM_BYTES);
  for (int i = tid; i < 4 * C; i += NW * 64) b1s[i] = aux[i];
  const float is1 = aux[G::AUX_IS1];
  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  long fy[G::KS1];
  int tile = t_lo + wave;
  if (tile < t_hi) W::load_y(Y, tile * 32, lane, fy);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* wl = smem + lane * 8;
  const float* bl = b1s + (lane >> 5) * 16;
  for (; tile < t_hi; tile += NW) {
    f32x16 O[G::NT2];
    W::init_o(X, aux, tile * 32, lane, O);
#pragma unroll 1
    for (int j = 0; j < G::NSTEP; ++j) W::step(wl + j * G::STEP_BYTES, bl + j * 32, fy, O, is1);
    if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, lane, fy);
    W::store_o(X, aux, tile * 32, M, lane, O);
  }
}

template <int C, int NW>
static int cn_launch_mlp_f8_resident(const unsigned char* Y, const void* WS, float* X, int M, int n_blocks, hipStream_t s) {
  typedef Rc2F8Geom<C> G;
  constexpr int SMEM = (int)G::STREAM_BYTES + 4 * C * 4;
  static_assert(SMEM <= 160 * 1024, "resident variant: the weight stream must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_f8_resident_kernel<C, NW>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_f8_resident_kernel<C, NW>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, (const char*)WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---- ring variant: steps stream L2 -> LDS through an NST-deep ring (see mlp_rc2.h for the vmcnt discipline) -------------
template <int C, int NW, int NST>
__global__ __launch_bounds__(NW * 64) void cn_mlp_f8_ring_kernel(const unsigned char* __restrict__ Y, const char* __restrict__ WS,
                                                                 float* __restrict__ X, int M) {
  typedef Rc2F8Geom<C> G;
  typedef Rc2F8Wave<C> W;
  constexpr int SB = G::STEP_BYTES, PIECES = SB / 1024;
  constexpr int DPW_LO = PIECES / NW, N_HI = PIECES % NW;  // waves < N_HI issue DPW_LO + 1 pieces per entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* aux = (const float*)(WS + G::STREAM_BYTES);
  float* b1s = (float*)(smem + NST * SB);
  for (int i = tid; i < 4 * C; i += NW * 64) b1s[i] = aux[i];
  const float is1 = aux[G::AUX_IS1];
  __syncthreads();

  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NW - 1) / NW;  // block-uniform: every wave runs the same number of steps

  const char* wsrc = WS + lane * 16;
  auto stage = [&](int g) {  // stream step g % NSTEP -> slot g % NST (this wave's pieces)
    const char* src = wsrc + (size_t)(g % G::NSTEP) * SB;
    char* dst = smem + (g % NST) * SB;
#pragma unroll
    for (int i = 0; i < DPW_LO + 1; ++i) {
      const int piece = wave + i * NW;
      if (i < DPW_LO || wave < N_HI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int g = 0; g < NST - 1; ++g) stage(g);
  long fy[G::KS1];
  if (t_lo + wave < t_hi) W::load_y(Y, (t_lo + wave) * 32, lane, fy);
  const char* wl = smem + lane * 8;
  const float* bl = b1s + (lane >> 5) * 16;
  int g = 0;
  for (int it = 0; it < max_it; ++it) {
    const int tile = t_lo + wave + it * NW;
    const bool valid = tile < t_hi;
    f32x16 O[G::NT2];
    for (int j = 0; j < G::NSTEP; ++j, ++g) {
      if (N_HI > 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (DPW_LO + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW_LO) : "memory");
      __builtin_amdgcn_s_barrier();
      stage(g + NST - 1);
      if (j == 0 && valid) W::init_o(X, aux, tile * 32, lane, O);
      if (valid) W::step(wl + (g % NST) * SB, bl + j * 32, fy, O, is1);
    }
    if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, lane, fy);
    if (valid) W::store_o(X, aux, tile * 32, M, lane, O);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 entries past the end
}

template <int C, int NW, int NST>
static int cn_launch_mlp_f8_ring(const unsigned char* Y, const void* WS, float* X, int M, int n_blocks, hipStream_t s) {
  typedef Rc2F8Geom<C> G;
  constexpr int SMEM = NST * G::STEP_BYTES + 4 * C * 4;
  static_assert(SMEM <= 160 * 1024, "ring must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_f8_ring_kernel<C, NW, NST>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_f8_ring_kernel<C, NW, NST>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, (const char*)WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
