// Role-split fused ConvNeXt MLP (bf16) -- third generation of the stage 0-2 pointwise kernel (mlp_rc2.h is the second):
//
//     x[m][:] += scale * ( W2 . gelu( W1 . y[m][:] + b1 ) + b2 )          (convnext.py:66-74)
//
// The register-chained kernel runs GEMM1 -> GELU -> GEMM2 of a hidden chunk inside ONE wave: at C = 384 that wave needs
// y (96 registers) + O (192) and is alone on its SIMD, so the chain's dependencies are paid in full (MFMA pipe 0.36 busy).
// Here the chain is cut across TWO waves that share a SIMD and a 32-position tile:
//   * an A wave holds y and computes X = W1c . y^T (+ b1 as a k-step), runs the GELU and hands the converted chunk
//     (32 pos x 32 hidden, bf16, already in MFMA A-operand order: the accumulator layout IS the operand layout of a
//     product that sums over the hidden units) to its partner through 2 KB of LDS;
//   * a B wave holds O (channels on the lanes, starts from the residual) and only runs O += G . W2c'^T.
// Software pipeline over the hidden chunks of a tile (one step = one chunk, one ring entry):
//     step g:   A: GEMM1(g) interleaved with the GELU of X(g-1), G(g-1) -> LDS      B: GEMM2(g-2) from G(g-2)
// so no wave ever waits for its own MFMA results: A's VALU work belongs to the previous chunk, B is a plain MFMA loop,
// and the two waves of a SIMD fill each other's issue gaps.  Registers: A = y + 2 X + fragments, B = O + fragments;
// both <= 256 at C = 384 -> 2 waves per SIMD where the chained kernel had one.
//
// Ring entry e (1 KB fragments, packed by pk_mlp_rs): [W1 fragments of chunk e, k-step 0 .. C/16-1, bias fragment]
// [W2' fragments of chunk (e - 2) mod NCH: (k 0, tile t) t < C/32, (k 1, t)], NCH = C/8 entries, then bb = s b2 (fp32, C).
// The W2 part of entry e belongs to the chunk B works on while A works on chunk e, so an entry is consumed whole by one
// step and the ring protocol is the chained kernel's: entry g + NST - 1 is requested at the start of step g.
//
// LDS: NST x (C/8 + 1) KB ring + NP x 2 KB hand-over buffers (single-buffered: two barriers per step -- the first
// publishes entry g and G(g-1), B takes G(g-1) into registers, the second lets A overwrite it).
#pragma once
#include "mlp_rc2.h"

static __global__ void pk_mlp_rs(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                 const float* __restrict__ b2, const float* __restrict__ scale, int C, bf16_t* __restrict__ dst) {
  const int KS1 = C / 16, NT2 = C / 32, NCH = C / 8, F1 = KS1 + 1, F2 = 2 * NT2, FRAGS = F1 + F2;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < C) ((float*)((char*)dst + (size_t)NCH * FRAGS * 1024))[u] = scale[u] * b2[u];  // bb behind the stream
  if (u >= NCH * FRAGS * 64) return;
  const int l = u & 63, q = (u >> 6) % FRAGS, e = (u >> 6) / FRAGS;
  const int r = l & 31, h = l >> 5;
  float v[8];
  if (q < F1) {
    const int j = e, s = q;
    if (s < KS1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = W1[(size_t)(32 * j + r) * C + 16 * s + 8 * h + i];
    } else {  // bias k-step: fp32 b1 as hi + lo bf16 against a "ones" fragment
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
    const int q2 = q - F1;
    const int j = (e + NCH - 2) % NCH, s = q2 / NT2, t = q2 % NT2;
    const int c = 32 * t + r;
    const float sc = scale[c];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sc * W2[(size_t)c * (4 * C) + 32 * j + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = (bf16_t)v[i];
}

// ABL (kernel lab only, wrong results): 1 = no ring refill after the prologue, 2 = GELU replaced by a copy, 4 = no residual / y
// traffic at tile boundaries
template <int C, int ABL = 0> struct RsWave {
  typedef Rc2Geom<C, 1> G;
  typedef Rc2Wave<C, 1> W;
  static constexpr int KS1 = G::KS1, NT2 = G::NT2, F1 = G::F1, F2 = G::F2;
  static constexpr int PRE = 4, R = PRE + 1;

  // ---- A: GEMM1 of this step's chunk into Xn, interleaved with the GELU of the previous chunk's Xp -> H ----------------
  // GELU element pair (e, e + 1) rides behind MFMA (e * F1) / 16
  static constexpr int gelu_at(int e) { return (e * F1) / 16; }
  struct AState {
    bf16x8 F[R];
    float g[16];
  };
  template <int Q, int E>
  static __device__ __forceinline__ void a_gelu(const f32x16& Xp, AState& st) {
    if constexpr (gelu_at(E) == Q) {
      const f32x2 r = (ABL & 2) ? f32x2{Xp[E], Xp[E + 1]} : cn_gelu_sig2_pk(f32x2{Xp[E], Xp[E + 1]});
      st.g[E] = r[0];
      st.g[E + 1] = r[1];
    }
    if constexpr (E + 2 < 16) a_gelu<Q, E + 2>(Xp, st);
  }
  template <int Q>
  static __device__ __forceinline__ void a_mstep(const char* wc, const bf16x8 (&fy)[KS1], const bf16x8 ones, const f32x16& Xp,
                                                 f32x16& Xn, AState& st) {
    if constexpr (Q + PRE < F1) st.F[(Q + PRE) % R] = W::frag(wc, Q + PRE);
    if constexpr (Q == 0) Xn = W::mma(st.F[0], fy[0], W::zero16());
    else if constexpr (Q < KS1) Xn = W::mma(st.F[Q % R], fy[Q], Xn);
    else Xn = W::mma(st.F[Q % R], ones, Xn);
    a_gelu<Q, 0>(Xp, st);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < F1) a_mstep<Q + 1>(wc, fy, ones, Xp, Xn, st);
  }
  template <int Q>
  static __device__ __forceinline__ void a_prefetch(const char* wc, AState& st) {
    st.F[Q % R] = W::frag(wc, Q);
    if constexpr (Q + 1 < PRE) a_prefetch<Q + 1>(wc, st);
  }
  // wc: this step's entry (lane offset applied); gdst: this pair's hand-over buffer (lane offset applied)
  static __device__ __forceinline__ void a_step(const char* wc, const bf16x8 (&fy)[KS1], const bf16x8 ones, const f32x16& Xp,
                                                f32x16& Xn, char* gdst) {
    AState st;
    a_prefetch<0>(wc, st);
    __builtin_amdgcn_sched_barrier(0);
    a_mstep<0>(wc, fy, ones, Xp, Xn, st);
    *(bf16x8*)gdst = bf16x8{(bf16_t)st.g[0], (bf16_t)st.g[1], (bf16_t)st.g[2],  (bf16_t)st.g[3],
                            (bf16_t)st.g[4], (bf16_t)st.g[5], (bf16_t)st.g[6],  (bf16_t)st.g[7]};
    *(bf16x8*)(gdst + 1024) = bf16x8{(bf16_t)st.g[8],  (bf16_t)st.g[9],  (bf16_t)st.g[10], (bf16_t)st.g[11],
                                     (bf16_t)st.g[12], (bf16_t)st.g[13], (bf16_t)st.g[14], (bf16_t)st.g[15]};
  }

  // ---- B: O += G . W2c'^T, fragments PRE ahead -----------------------------------------------------------------------
  struct BState {
    bf16x8 F[R];
  };
  template <int Q>
  static __device__ __forceinline__ void b_mstep(const char* w2, const bf16x8 (&H)[2], f32x16 (&O)[NT2], BState& st) {
    if constexpr (Q + PRE < F2) st.F[(Q + PRE) % R] = W::frag(w2, Q + PRE);
    constexpr int k = Q / NT2, t = Q % NT2;
    O[t] = W::mma(H[k], st.F[Q % R], O[t]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < F2) b_mstep<Q + 1>(w2, H, O, st);
  }
  template <int Q>
  static __device__ __forceinline__ void b_prefetch(const char* w2, BState& st) {
    st.F[Q % R] = W::frag(w2, Q);
    if constexpr (Q + 1 < PRE) b_prefetch<Q + 1>(w2, st);
  }
  static __device__ __forceinline__ void b_step(const char* w2, const bf16x8 (&H)[2], f32x16 (&O)[NT2]) {
    BState st;
    b_prefetch<0>(w2, st);
    __builtin_amdgcn_sched_barrier(0);
    b_mstep<0>(w2, H, O, st);
  }
};

// NP pairs per block (2 NP waves: waves [0, NP) are the A roles, [NP, 2 NP) the B roles: with waves dealt round-robin over the
// four SIMDs a pair shares its SIMD when NP is a multiple of 4); pair p owns tiles t_lo + p + it * NP of the block's range.
template <int C, int NP, int NST, int ABL = 0>
__global__ __launch_bounds__(2 * NP * 64) void cn_mlp_rs_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ WS,
                                                                float* __restrict__ X, int M) {
  typedef Rc2Geom<C, 1> G;
  typedef Rc2Wave<C, 1> W;
  typedef RsWave<C, ABL> RW;
  constexpr int NW = 2 * NP, NCH = G::NSTEP, FR = G::FRAGS, SB = G::STEP_BYTES;
  constexpr int DPW_LO = FR / NW, N_HI = FR % NW;  // waves < N_HI issue DPW_LO + 1 pieces per entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* gbuf = smem + NST * SB;  // NP x 2 KB hand-over buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_b = wave >= NP;
  const int pair = role_b ? wave - NP : wave;
  const float* aux = (const float*)((const char*)WS + G::STREAM_BYTES);

  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NP - 1) / NP;  // block-uniform
  const int n_steps = max_it * NCH + 2;            // + 2: B runs two chunks behind A

  const unsigned voff = lane * 16;
  const unsigned lds0 = cn_lds_addr(smem);
  auto stage = [&](int g) {  // stream entry g % NCH -> slot g % NST (this wave's pieces)
    if ((ABL & 1) && g >= NST - 1) return;
    const char* src = (const char*)WS + (size_t)(g % NCH) * SB;  // wave-uniform
    const unsigned dst = lds0 + (unsigned)((g % NST) * SB);
#pragma unroll
    for (int i = 0; i < DPW_LO + 1; ++i) {
      const int piece = wave + i * NW;
      if (i < DPW_LO || wave < N_HI) cn_dma16_s(src + piece * 1024, voff, dst + piece * 1024);
    }
  };
  auto ring_wait = [&]() {  // this wave's pieces of the entry about to be consumed have landed (NST - 2 younger entries may fly)
    if (N_HI > 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (DPW_LO + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW_LO) : "memory");
  };
#pragma unroll
  for (int g = 0; g < NST - 1; ++g) stage(g);
  const char* wl = smem + lane * 16;
  char* gl = gbuf + pair * 2048 + lane * 16;

  if (!role_b) {
    // ================================================= A ==============================================================
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)((lane < 32 && i < 2) ? 1.0f : 0.0f);
    bf16x8 fy[G::KS1];
    f32x16 Xa = W::zero16(), Xb = W::zero16();
    if (t_lo + pair < t_hi) W::load_y(Y, (t_lo + pair) * 32, lane, fy);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): y (and the first entries; once per block)
    int g = 0;
    for (int it = 0; it <= max_it; ++it) {  // (+ one trailing pass: steps max_it * NCH and + 1 only finish the last chunk's GELU)
      const int tile = t_lo + pair + it * NP;
      const bool last_pass = it == max_it;
      for (int j = 0; j < (last_pass ? 2 : NCH); j += 2) {
        // two steps per trip: the roles of the two X accumulators swap
        ring_wait();
        __builtin_amdgcn_s_barrier();
        stage(g + NST - 1);
        __builtin_amdgcn_s_barrier();
        RW::a_step(wl + (g % NST) * SB, fy, ones, Xb, Xa, gl);  // GEMM1(g) -> Xa, GELU of Xb (chunk g - 1) -> G
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++g;
        ring_wait();
        __builtin_amdgcn_s_barrier();
        stage(g + NST - 1);
        __builtin_amdgcn_s_barrier();
        RW::a_step(wl + (g % NST) * SB, fy, ones, Xa, Xb, gl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++g;
      }
      if (!last_pass && tile + NP < t_hi && !(ABL & 4)) {
        W::load_y(Y, (tile + NP) * 32, lane, fy);
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
    }
  } else {
    // ================================================= B ==============================================================
    f32x16 O[G::NT2];
    int g = 0;
    for (int it = 0; it <= max_it; ++it) {
      const bool last_pass = it == max_it;
      for (int j = 0; j < (last_pass ? 2 : NCH); ++j, ++g) {
        ring_wait();
        __builtin_amdgcn_s_barrier();
        stage(g + NST - 1);
        const int gb = g - 2;                       // the chunk step B works on
        const int itb = gb < 0 ? 0 : gb / NCH, cb = gb < 0 ? 0 : gb % NCH;
        const int tile = t_lo + pair + itb * NP;
        const bool valid = gb >= 0 && tile < t_hi;
        bf16x8 H[2];
        H[0] = *(const bf16x8*)gl;
        H[1] = *(const bf16x8*)(gl + 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (valid) {
          if (cb == 0 && !(ABL & 4)) {
            W::init_o(X, tile * 32, lane, O);
            __builtin_amdgcn_s_waitcnt(0x0F70);
          }
          RW::b_step(wl + (g % NST) * SB + RW::F1 * 1024, H, O);
          if (cb == NCH - 1 && (!(ABL & 4) || it == max_it)) {
            const float* bbv = aux;
            asm volatile("" : "+s"(bbv));
            W::store_o(X, bbv, tile * 32, M, lane, O);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 entries past the end
}

template <int C, int NP, int NST, int ABL = 0>
static int cn_launch_mlp_rs(const bf16_t* Y, const bf16_t* WS, float* X, int M, int n_blocks, hipStream_t s) {
  constexpr int SMEM = NST * Rc2Geom<C, 1>::STEP_BYTES + NP * 2048;
  static_assert(SMEM <= 160 * 1024, "ring + hand-over buffers must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_rs_kernel<C, NP, NST, ABL>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NP, n_blocks);
  hipLaunchKernelGGL((cn_mlp_rs_kernel<C, NP, NST, ABL>), dim3((unsigned)grid), dim3(2 * NP * 64), SMEM, s, Y, WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
