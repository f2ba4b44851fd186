// Register-chained fused ConvNeXt MLP (16-bit operands HT = bf16_t | half_t, common.h), stages 0-2 -- second generation (see mlp_rc.h for the idea):
//
//     x[m][:] += scale * ( W2 . gelu( W1 . y[m][:] + b1 ) + b2 )          (convnext.py:66-74)
//
// * a wave owns 32 positions; per 32 hidden units: GEMM1 (A = W1 fragments from LDS, B = y fragments in registers,
//   bias as one extra k-step) -> GELU on the 16 accumulator registers -> GEMM2 with the converted accumulator as the
//   A operand (X^T . B, cdna guide 3) accumulating O[32 pos][C] with channels on the lanes;
// * a STEP covers NCK = 1 or 2 chunks and is one basic block whose MFMA / VALU / LDS-read interleave is laid down with
//   sched_group_barrier: NCK = 2:  M1a | M1b + GELU a | M2a + GELU b | M2b;   NCK = 1:  M1 | GELU lo | M2(k 0) + GELU hi | M2(k 1)
//   with the fragment reads running a few MFMAs ahead of their use;
// * GEMM2 runs TRANSPOSED since round 5 -- O^T[ch][pos] += W2c' . G^T, the W2 fragment as the A operand and the converted
//   accumulator as B (the two operands have the same per-lane layout, so the fragments did not change): a lane owns ONE
//   position and, per 32-channel tile, two runs of 8 consecutive channels (the rows of the W2 fragments are permuted for
//   that, cn_rc2_chan), so the residual moves as 16-byte pieces with the access pattern of the y fragments -- 6 loads + 6 stores
//   per tile at C = 96 with an fp16 residual stream (XT = half_t, the 16-bit precisions since round 5: 10 C -> 6 C bytes per
//   position and block for this kernel) where the channels-on-lanes layout of rounds 2-4 moved 48 + 48 dwords;
// * LayerScale is folded into the packed W2 rows (W2'[c][:] = bf16(s[c] W2[c][:])) and the accumulators START from the
//   residual plus the folded output bias: O = x + s b2, the bias through one MFMA per channel tile (a (hi, lo) fragment
//   against the constant "ones" fragment, like b1: the VALU is what these kernels run out of, the matrix pipe has slack);
//   the epilogue is a conversion and the stores.
//
// Packed stream per step (1 KB fragments, lane l reads 16 B at 16 l):
//   [chunk i: W1 fragments k-step 0 .. C/16-1, bias fragment] i < NCK, then [chunk i: W2 fragments (k 0, tile t) t < C/32, (k 1, t)] i < NCK
// followed by C/32 fragments holding bb = s b2 as (hi, lo) pairs, one per channel tile.
#pragma once
#include "common.h"

template <int C, int NCK> struct Rc2Geom {
  static constexpr int KS1 = C / 16, NT2 = C / 32, NCH = C / 8;
  static constexpr int F1 = KS1 + 1, F2 = 2 * NT2;
  static constexpr int NSTEP = NCH / NCK;            // steps per tile
  static constexpr int FRAGS = NCK * (F1 + F2);      // fragments per step
  static constexpr int STEP_BYTES = FRAGS * 1024;
  static constexpr size_t STREAM_BYTES = (size_t)NSTEP * STEP_BYTES;
  static constexpr size_t TOTAL_BYTES = STREAM_BYTES + NT2 * 1024;
};
// Row m of a W2 (or output-bias) fragment = accumulator row m of the transposed product = channel cn_rc2_chan(m) of the tile:
// lane (pos, h = l >> 5) holds rows 8 (r >> 2) + 4 h + (r & 3) in register r, and with bits 2 and 3 of m swapped those are the
// channels 16 (r >> 3) + 8 h + (r & 7): registers 0-7 and 8-15 are two runs of 8 consecutive channels.
__host__ __device__ constexpr int cn_rc2_chan(int m) { return (m & 19) | ((m & 4) << 1) | ((m & 8) >> 1); }

// GELU as x * sigmoid(x (a + b x^2 + c x^4)): minimax fit of the logit of the normal CDF on [-8, 8], max |error|
// against the exact erf form 2.5e-5 (tanh form: 4.7e-4).  x^2 is clamped at 64, where the quartic would bend back.
__device__ __forceinline__ float cn_gelu_sig2(float x) {
  constexpr float L2E = 1.4426950408889634f;
  const float x2 = fminf(x * x, 64.0f);
  float p = fmaf(x2, 0.0007030350670982541f * L2E, -0.07401130190658815f * L2E);
  p = fmaf(p, x2, -1.5950157568571721f * L2E);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// (cn_gelu_e1<DEG>: common.h -- the one-transcendental form of round 5)
// The GELU of the fused MLP kernels.  Unless this is an A/B build of the sigmoid form, the packed W1 / b1 operands carry a factor
// 1/2 (CN_MLP_XSCALE, applied by the packers below) and the accumulator of GEMM1 is x / 2.
#if defined(CN_GELU_SIG2)
#define CN_MLP_XSCALE 1.0f
#else
#define CN_MLP_XSCALE 0.5f
#endif
template <typename HT> __device__ __forceinline__ float cn_gelu_mlp(float x) {
#if defined(CN_GELU_SIG2)   // A/B builds: the two-transcendental sigmoid form of rounds 2-4
  return cn_gelu_sig2(x);
#elif defined(CN_GELU_E1_DEG)
  return cn_gelu_e1_half<CN_GELU_E1_DEG>(x);
#else
  return cn_gelu_e1_half<__is_same(HT, half_t) ? 5 : 3>(x);
#endif
}

// two elements at once: v_pk_mul / v_pk_fma / v_pk_add carry both (the two min, exp2 and rcp stay scalar)
__device__ __forceinline__ f32x2 cn_gelu_sig2_pk(f32x2 x) {
  constexpr float L2E = 1.4426950408889634f;
  f32x2 x2 = x * x;
  x2 = f32x2{fminf(x2[0], 64.0f), fminf(x2[1], 64.0f)};
  f32x2 p = x2 * (0.0007030350670982541f * L2E) + (-0.07401130190658815f * L2E);
  p = p * x2 + (-1.5950157568571721f * L2E);
  const f32x2 u = x * p;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.0f;
  return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// ---- packing (fp32 nn.Linear layouts -> fragment stream + aux) ------------------------------------------------------
template <typename HT>
static __global__ void pk_mlp_rc2(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                           const float* __restrict__ b2, const float* __restrict__ scale, int C, int NCK,
                           HT* __restrict__ dst) {
  const int KS1 = C / 16, NT2 = C / 32, NCH = C / 8, F1 = KS1 + 1, F2 = 2 * NT2, FRAGS = NCK * (F1 + F2), NSTEP = NCH / NCK;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (NSTEP * FRAGS + NT2) * 64) return;
  const int l = u & 63, q = (u >> 6) % FRAGS, st = (u >> 6) / FRAGS;
  const int r = l & 31, h = l >> 5;
  float v[8];
  if (st >= NSTEP) {  // behind the stream: bb = s b2 of channel tile q as a (hi, lo) k-step
    const int c = 32 * q + cn_rc2_chan(r);
    const float b = scale[c] * b2[c];
    const float hi = (float)(HT)b;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (h == 0) {
      v[0] = hi;
      v[1] = b - hi;
    }
  } else if (q < NCK * F1) {
    const int j = st * NCK + q / F1, s = q % F1;
    if (s < KS1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = CN_MLP_XSCALE * W1[(size_t)(32 * j + r) * C + 16 * s + 8 * h + i];
    } else {
      const float b = CN_MLP_XSCALE * b1[32 * j + r];
      const float hi = (float)(HT)b;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
      if (h == 0) {
        v[0] = hi;
        v[1] = b - hi;
      }
    }
  } else {
    const int q2 = q - NCK * F1;
    const int j = st * NCK + q2 / F2, s = (q2 % F2) / NT2, t = (q2 % F2) % NT2;
    const int c = 32 * t + cn_rc2_chan(r);
    const float sc = scale[c];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sc * W2[(size_t)c * (4 * C) + 32 * j + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = (HT)v[i];
}

// hidden chunks (of 32) per step: two at C = 192 (3-deep ring of 50 KB entries), one elsewhere
#define CN_RC2_NCK(C) ((C) == 192 ? 2 : 1)

// sched_group_barrier masks (LLVM AMDGPU IGroupLP)
#define CN_SG_VALU 0x002
#define CN_SG_MFMA 0x008
#define CN_SG_DSR 0x100

// Ring pieces go through cn_dma16_s (common.h: inline-asm LDS-DMA; the builtin cost a full lgkmcnt drain at every fifth
// MFMA of the step loop).  (Lab: issuing piece q INSIDE the step, behind MFMA q by wave q % NW, only moved the ~70 cycles
// a piece costs its wave from the top of the step into the step, MFMAs or not: 201 us against 189.  profiles/r02_notes.md)

template <int C, int NCK, typename HT = bf16_t, typename XT = float> struct Rc2Wave {
  typedef Rc2Geom<C, NCK> G;
  typedef cn_h8<HT> hx8;
  static constexpr int KS1 = G::KS1, NT2 = G::NT2, F1 = G::F1, F2 = G::F2;
  static constexpr int NV = 88;  // VALU instructions of the GELU + bf16 packing of 8 accumulator registers (approx.)

  static __device__ __forceinline__ hx8 frag(const char* wc, int f) { return *(const hx8*)(wc + f * 1024); }
  static __device__ __forceinline__ f32x16 mma(hx8 a, hx8 b, f32x16 c) { return cn_mma32(a, b, c); }
  static __device__ __forceinline__ hx8 gelu8(const f32x16& X, int o) {
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = cn_gelu_mlp<HT>(X[o + i]);
    return cn_sat8<HT>(cn_pack8<HT>(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]));
  }
  static __device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
  }

  // One step; fragments at wc (LDS, lane offset applied).  MFMA q of a step consumes fragment q of the packed stream.
  // The instruction order is laid down BY HAND (sched_barrier(0) between slices; sched_group_barrier pipelines of this
  // size were not honoured by the scheduler): fragment q + PRE is requested before MFMA q, and the GELU of an accumulator
  // is cut into per-element slices placed behind the MFMAs of the neighbouring GEMM:
  //   NCK = 2:  M1a | M1b + GELU a | M2a + GELU b | M2b        NCK = 1:  M1 | GELU lo | M2(k 0) + GELU hi | M2(k 1)
  static constexpr int PRE = 4;
  static constexpr int NM = NCK * (F1 + F2), R = PRE + 1;
  struct State {
    hx8 F[R];
    f32x16 X[NCK];
    float g[NCK][16];
    hx8 H[NCK][2];
  };
  // index of the MFMA behind which GELU element e of chunk i is placed
  static constexpr int gelu_at(int i, int e) {
    if (NCK == 2) return i == 0 ? F1 + 1 + e * (F1 - 1) / 16 : 2 * F1 + e * F2 / 16;
    return e < 8 ? F1 - 1 : F1 + (e - 8) * NT2 / 8;
  }
  template <int Q, int I, int E>
  static __device__ __forceinline__ void gelu_slices(State& st) {  // (compile-time recursion: every index is a constant)
#if defined(CN_RC2_GELU_PK) && defined(CN_GELU_SIG2)
    if constexpr ((E & 1) == 0 && gelu_at(I, E) == Q) {  // pairs (E, E + 1) ride together behind MFMA gelu_at(I, E)
      const f32x2 r = cn_gelu_sig2_pk(f32x2{st.X[I][E], st.X[I][E + 1]});
      st.g[I][E] = r[0];
      st.g[I][E + 1] = r[1];
    }
    if constexpr (gelu_at(I, E & ~1) == Q) {
#else
    if constexpr (gelu_at(I, E) == Q) {
      st.g[I][E] = cn_gelu_mlp<HT>(st.X[I][E]);
#endif
      if constexpr ((E & 7) == 7) {
        constexpr int o = E - 7;
        st.H[I][E >> 3] = cn_sat8<HT>(cn_pack8<HT>(st.g[I][o], st.g[I][o + 1], st.g[I][o + 2], st.g[I][o + 3],
                                                   st.g[I][o + 4], st.g[I][o + 5], st.g[I][o + 6], st.g[I][o + 7]));
      }
    }
    if constexpr (E + 1 < 16) gelu_slices<Q, I, E + 1>(st);
    else if constexpr (I + 1 < NCK) gelu_slices<Q, I + 1, 0>(st);
  }
  template <int Q>
  static __device__ __forceinline__ void mstep(const char* wc, const hx8 (&fy)[KS1], const hx8 ones, f32x16 (&O)[NT2],
                                               State& st) {
    if constexpr (Q + PRE < NM) st.F[(Q + PRE) % R] = frag(wc, Q + PRE);
    if constexpr (Q < NCK * F1) {
      constexpr int i = Q / F1, s = Q % F1;
      if constexpr (s == 0) st.X[i] = zero16();
      if constexpr (s < KS1) st.X[i] = mma(st.F[Q % R], fy[s], st.X[i]);
      else st.X[i] = mma(st.F[Q % R], ones, st.X[i]);
    } else {
      constexpr int q2 = Q - NCK * F1, i = q2 / F2, k = (q2 % F2) / NT2, t = q2 % NT2;
      O[t] = mma(st.F[Q % R], st.H[i][k], O[t]);  // O^T += W2c' . G^T
    }
    gelu_slices<Q, 0, 0>(st);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < NM) mstep<Q + 1>(wc, fy, ones, O, st);
  }
  template <int Q>
  static __device__ __forceinline__ void prefetch(const char* wc, State& st) {
    st.F[Q % R] = frag(wc, Q);
    if constexpr (Q + 1 < PRE) prefetch<Q + 1>(wc, st);
  }
  static __device__ __forceinline__ void step(const char* wc, const hx8 (&fy)[KS1], const hx8 ones, f32x16 (&O)[NT2]) {
    State st;
    prefetch<0>(wc, st);
    __builtin_amdgcn_sched_barrier(0);
    mstep<0>(wc, fy, ones, O, st);
  }

  // Addressing: tile bases are wave-uniform (scalar), the per-lane part is ONE loop-invariant 32-bit offset and the rest
  // are immediates.  Rows of a ragged last tile beyond M are READ (never written): Y and X must be readable for 31 rows
  // past M (the encoder workspace is padded accordingly).

  // y rows of a 32-position tile as B fragments: fy[s] = y[m0 + (l & 31)][16 s + 8 (l >> 5) .. + 8]
  static __device__ __forceinline__ void load_y(const HT* __restrict__ Y, int m0, int lane, hx8 (&fy)[KS1]) {
    const HT* base = Y + (size_t)m0 * C;                 // scalar
    const int voff = (lane & 31) * C + 8 * (lane >> 5);      // elements
#pragma unroll
    for (int s = 0; s < KS1; ++s) fy[s] = *(const hx8*)(base + voff + 16 * s);
  }

  // O^T = x^T + bb: lane = position m0 + (l & 31); register r of tile t = channel 32 t + 16 (r >> 3) + 8 (l >> 5) + (r & 7).
  // bbf: the NT2 output-bias fragments behind the stream (global memory, lane offset applied; L1 / L2 resident).
  static __device__ __forceinline__ void init_o(const XT* __restrict__ X, const char* __restrict__ bbf, const hx8 ones, int m0,
                                                int lane, f32x16 (&O)[NT2]) {
    const XT* base = X + (size_t)m0 * C;                    // scalar
    const int voff = (lane & 31) * C + 8 * (lane >> 5);     // elements
    hx8 fb[NT2];
    if constexpr (sizeof(XT) == 2) {
      cn_h8<XT> v[NT2][2];
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        v[t][0] = *(const cn_h8<XT>*)(base + voff + 32 * t);
        v[t][1] = *(const cn_h8<XT>*)(base + voff + 32 * t + 16);
      }
#pragma unroll
      for (int t = 0; t < NT2; ++t) fb[t] = *(const hx8*)(bbf + t * 1024);
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = (float)v[t][r >> 3][r & 7];
    } else {
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *(const f32x4*)(base + voff + 32 * t + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
          for (int e = 0; e < 4; ++e) O[t][4 * q + e] = v[e];
        }
#pragma unroll
      for (int t = 0; t < NT2; ++t) fb[t] = *(const hx8*)(bbf + t * 1024);
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t) O[t] = mma(fb[t], ones, O[t]);
  }

  // x' = O, converted to the stream's type: 16-byte pieces (the rows of a ragged last tile beyond M are not written)
  static __device__ __forceinline__ void store_o(XT* __restrict__ X, int m0, int M, int lane, const f32x16 (&O)[NT2]) {
    XT* base = X + (size_t)m0 * C;
    const int voff = (lane & 31) * C + 8 * (lane >> 5);
    if (m0 + (lane & 31) >= M) return;
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      if constexpr (sizeof(XT) == 2) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
          *(cn_h8<XT>*)(base + voff + 32 * t + 16 * o) = cn_pack8<XT>(O[t][8 * o], O[t][8 * o + 1], O[t][8 * o + 2], O[t][8 * o + 3],
                                                                      O[t][8 * o + 4], O[t][8 * o + 5], O[t][8 * o + 6], O[t][8 * o + 7]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(base + voff + 32 * t + 16 * (q >> 1) + 4 * (q & 1)) = f32x4{O[t][4 * q], O[t][4 * q + 1], O[t][4 * q + 2], O[t][4 * q + 3]};
      }
    }
  }
};

// The channels-on-lanes residual I/O of rounds 2-4 (O[pos][ch], dword pieces), still the layout of the exact precision's
// fused block (mlp_sp.h: its GEMM2 is not transposed): lane = channel 32 t + (l & 31), register r = position (r & 3) + 8 (r >> 2) + 4 (l >> 5).
template <int C> struct Rc2IoCl {
  static constexpr int NT2 = C / 32;
  static __device__ __forceinline__ void init_o(const float* __restrict__ X, int m0, int lane, f32x16 (&O)[NT2]) {
    const int voff = 4 * (lane >> 5) * C + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* row = X + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * C;  // scalar
#pragma unroll
      for (int t = 0; t < NT2; ++t) O[t][r] = row[voff + 32 * t];
    }
  }
  // x' = O + bb, straight from the accumulator layout (two whole 128-byte row pieces per store instruction)
  static __device__ __forceinline__ void store_o(float* __restrict__ X, const float* __restrict__ bbv, int m0, int M, int lane,
                                                 const f32x16 (&O)[NT2]) {
    const int voff = 4 * (lane >> 5) * C + (lane & 31);
    const int plim = M - m0 - 4 * (lane >> 5);  // this lane's rows (r&3) + 8 (r>>2) < plim are inside the tensor
    float bb[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) bb[t] = bbv[32 * t + (lane & 31)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* row = X + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * C;  // scalar
      if ((r & 3) + 8 * (r >> 2) < plim) {
#pragma unroll
        for (int t = 0; t < NT2; ++t) row[voff + 32 * t] = O[t][r] + bb[t];
      }
    }
  }
};

// Grid of a persistent launch: every wave walks ceil(tiles / waves) 32-position tiles, so the launch takes that many
// ROUNDS whatever the block count; use the fewest rounds max_blocks allows and then the SMALLEST grid that still finishes in
// that many rounds (441 four-wave block-tiles: 208 blocks need 3 rounds, 221 need 2 and leave 35 CUs to other streams).
static inline int cn_rc2_grid(int n_tiles, int waves_per_block, int max_blocks) {
  const int block_tiles = cn_cdiv(n_tiles, waves_per_block);
  if (max_blocks < 1) max_blocks = 1;
  const int rounds = cn_cdiv(block_tiles, max_blocks);
  return cn_cdiv(block_tiles, rounds);
}

// ---- resident variant (C = 96): the whole stream (156 KB) lives in LDS; persistent blocks; no barrier, no DMA after the fill
// ABL (kernel lab only, wrong results): 1 = accumulators start from zero (no residual loads), 2 = nothing stored, 8 = y loaded once
template <int C, int NW, int NCK, typename HT = bf16_t, int ABL = 0, typename XT = float>
__global__ __launch_bounds__(NW * 64) void cn_mlp_rc2_resident_kernel(const HT* __restrict__ Y, const HT* __restrict__ WS,
                                                                      XT* __restrict__ X, int M) {
  typedef Rc2Geom<C, NCK> G;
  typedef Rc2Wave<C, NCK, HT, XT> W;
  typedef cn_h8<HT> hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int TOTAL = G::NSTEP * G::FRAGS;
  for (int i = wave; i < TOTAL; i += NW)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)WS + (size_t)i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
  const char* bbf = (const char*)WS + G::STREAM_BYTES + lane * 16;  // output-bias fragments (global memory)
  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  hx8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (HT)((lane < 32 && i < 2) ? 1.0f : 0.0f);
  hx8 fy[G::KS1];
  int tile = t_lo + wave;
  if (tile < t_hi) W::load_y(Y, tile * 32, lane, fy);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* wl = smem + lane * 16;
  for (; tile < t_hi; tile += NW) {
    f32x16 O[G::NT2];
    if constexpr (ABL & 1) {
#pragma unroll
      for (int t = 0; t < G::NT2; ++t) O[t] = W::zero16();
    } else {
      W::init_o(X, bbf, ones, tile * 32, lane, O);
    }
#pragma unroll 1
    for (int j = 0; j < G::NSTEP; ++j) W::step(wl + j * G::STEP_BYTES, fy, ones, O);
    if constexpr (!(ABL & 8))
      if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, lane, fy);
    if constexpr (ABL & 2) {
      if (O[0][0] == 12345.678f) W::store_o(X, tile * 32, M, lane, O);
    } else {
      W::store_o(X, tile * 32, M, lane, O);
    }
  }
}

template <int C, int NW, int NCK, int ABL = 0, typename HT, typename XT>
static int cn_launch_mlp_rc2_resident(const HT* Y, const HT* WS, XT* X, int M, int n_blocks, hipStream_t s) {
  constexpr int SMEM = (int)Rc2Geom<C, NCK>::STREAM_BYTES;
  static_assert(SMEM <= 160 * 1024, "resident variant: the weight stream must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_rc2_resident_kernel<C, NW, NCK, HT, ABL, XT>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_rc2_resident_kernel<C, NW, NCK, HT, ABL, XT>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---- ring variant (C = 192: 8 waves, C = 384: 4 waves): steps stream L2 -> LDS through an NST-deep ring ----------------
// Step g of a block consumes ring entry g (stream step g % NSTEP); entry g + NST - 1 is requested at the start of step g,
// right after the barrier that publishes entry g and retires entry g - 1.  vmcnt: at most (NST - 2) younger entries of
// this wave's pieces may be outstanding when entry g is needed; any other vector-memory operation in between (residual
// loads, stores) only makes that wait stricter, never unsafe (the counter retires in order).
// PROF (kernel lab only): per-step s_memtime stamps at points where no LDS read is in flight -> prof[0..4] =
// wait for the DMA, barrier, DMA issue, step compute, tile boundary (sums over waves, in cycles), prof[5] = wave-steps
template <int C, int NW, int NCK, int NST, int PROF = 0, typename HT = bf16_t, typename XT = float>
__global__ __launch_bounds__(NW * 64) void cn_mlp_rc2_ring_kernel(const HT* __restrict__ Y, const HT* __restrict__ WS,
                                                                  XT* __restrict__ X, int M, unsigned long long* prof = nullptr) {
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0, nstep = 0;
  auto stamp = [&](int i) {
    if constexpr (PROF) {
      const unsigned long long t = clock64();
      tacc[i] += t - tprev;
      tprev = t;
    }
  };
  if constexpr (PROF) tprev = clock64();
  typedef Rc2Geom<C, NCK> G;
  typedef Rc2Wave<C, NCK, HT, XT> W;
  typedef cn_h8<HT> hx8;
  constexpr int FR = G::FRAGS, SB = G::STEP_BYTES;
  constexpr int DPW_LO = FR / NW, N_HI = FR % NW;  // waves < N_HI issue DPW_LO + 1 pieces per entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* bbf = (const char*)WS + G::STREAM_BYTES + lane * 16;  // output-bias fragments (global memory)

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
  hx8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (HT)((lane < 32 && i < 2) ? 1.0f : 0.0f);

#pragma unroll
  for (int g = 0; g < NST - 1; ++g) stage(g);
  hx8 fy[G::KS1];
  if (t_lo + wave < t_hi) W::load_y(Y, (t_lo + wave) * 32, lane, fy);
  const char* wl = smem + lane * 16;
  int g = 0;
  for (int it = 0; it < max_it; ++it) {
    const int tile = t_lo + wave + it * NW;
    const bool valid = tile < t_hi;
    f32x16 O[G::NT2];
    if (valid) W::init_o(X, bbf, ones, tile * 32, lane, O);
    {
      // The residual (and the y fragments requested at the end of the last tile) land HERE, once per tile, in a wait the
      // compiler sees and that dominates the step loop: otherwise the step code, shared by every j, carries the waits for
      // their first uses, and those -- counted or not -- drain the ring's pieces in flight at every step.
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), on every path into the loop (the compiler's bookkeeping is not path-sensitive)
    }
    for (int j = 0; j < G::NSTEP; ++j, ++g) {
      stamp(4);
      if (N_HI > 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (DPW_LO + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW_LO) : "memory");
      stamp(0);
      __builtin_amdgcn_s_barrier();
      stamp(1);
      stage(g + NST - 1);
      stamp(2);
      if (valid) W::step(wl + (g % NST) * SB, fy, ones, O);
      stamp(3);
      if constexpr (PROF) nstep += valid;
    }
    if (tile + NW < t_hi) W::load_y(Y, (tile + NW) * 32, lane, fy);
    if (valid) W::store_o(X, tile * 32, M, lane, O);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 entries past the end
  if constexpr (PROF) {
    if (lane == 0 && prof) {
#pragma unroll
      for (int i = 0; i < 5; ++i) atomicAdd(prof + i, tacc[i]);
      atomicAdd(prof + 5, nstep);
    }
  }
}

template <int C, int NW, int NCK, int NST, int PROF = 0, typename HT, typename XT>
static int cn_launch_mlp_rc2_ring(const HT* Y, const HT* WS, XT* X, int M, int n_blocks, hipStream_t s,
                                  unsigned long long* prof = nullptr) {
  constexpr int SMEM = NST * Rc2Geom<C, NCK>::STEP_BYTES;
  static_assert(SMEM <= 160 * 1024, "ring must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_rc2_ring_kernel<C, NW, NCK, NST, PROF, HT, XT>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NW, n_blocks);
  hipLaunchKernelGGL((cn_mlp_rc2_ring_kernel<C, NW, NCK, NST, PROF, HT, XT>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, Y, WS, X, M, prof);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
