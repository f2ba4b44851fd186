// Role-split fused ConvNeXt MLP (16-bit operands HT = bf16_t | half_t) -- third generation of the stage 0-2 pointwise kernel (mlp_rc2.h is the second):
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
// publishes entry g and G(g-1), B takes G(g-1) into registers, the second lets A overwrite it) + bb (C fp32).
//
// The ring refill (1 KB LDS-DMA pieces) is issued by the B waves INSIDE their MFMA loop: B has no VALU work, so a piece
// fits the gap behind one of its MFMAs while the partner's MFMA runs (issued between the barriers instead, with every wave
// taking a share, the refill cost 60 us of a 209 us launch at C = 384; inside B's loop 15).
// Tile boundaries are software-pipelined into the steps around them:
//   * A re-loads y fragment s of the NEXT tile in place, right behind the last MFMA of the tile that reads fragment s
//     (the first step of the next tile meets them one step later);
//   * B keeps O transposed -- GEMM2 as W2c' . G^T: lane = position, two runs of eight consecutive channels per 32-channel
//     tile (the rows of the W2 fragments are permuted for that: cn_rc2_chan, mlp_rc2.h), so the residual moves in 16-byte
//     pieces -- and runs its MFMAs tile by tile (k 0, k 1 of one 32-channel tile back to back), so that in a tile's last
//     step the store of channel tile t and the load of the next position tile's residual follow the tile's last MFMA, one
//     tile behind.  With an fp16 residual stream (XT = half_t, round 5) a tile's row is 24 loads + 24 stores of 16 bytes;
//     the loaded halves wait in registers that the stored tile has just freed and are converted in the FIRST step of the
//     next position tile, right before the first MFMA of their channel tile (a conversion at the load would wait for HBM in
//     the middle of the step).
#pragma once
#include <type_traits>

#include "mlp_rc2.h"

template <typename HT>
static __global__ void pk_mlp_rs(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                 const float* __restrict__ b2, const float* __restrict__ scale, int C, HT* __restrict__ dst) {
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
      for (int i = 0; i < 8; ++i) v[i] = CN_MLP_XSCALE * W1[(size_t)(32 * j + r) * C + 16 * s + 8 * h + i];
    } else {  // bias k-step: fp32 b1 as hi + lo bf16 against a "ones" fragment
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
    const int q2 = q - F1;
    const int j = (e + NCH - 2) % NCH, s = q2 / NT2, t = q2 % NT2;
    const int c = 32 * t + cn_rc2_chan(r);
    const float sc = scale[c];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sc * W2[(size_t)c * (4 * C) + 32 * j + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[(size_t)u * 8 + i] = (HT)v[i];
}

// ABL (kernel lab only, wrong results): 1 = no ring refill after the prologue, 2 = GELU replaced by a copy, 4 = no residual / y
// traffic at tile boundaries
template <int C, int NP, int ABL = 0, typename HT = bf16_t, typename XT = float> struct RsWave {
  typedef Rc2Geom<C, 1> G;
  typedef Rc2Wave<C, 1, HT, XT> W;
  typedef cn_h8<HT> hx8;
  static constexpr int KS1 = G::KS1, NT2 = G::NT2, F1 = G::F1, F2 = G::F2, FR = G::FRAGS;
#ifndef CN_RS_PRE
#define CN_RS_PRE 4
#endif
  static constexpr int PRE = CN_RS_PRE, R = PRE + 1;
  static constexpr int NPW = (FR + NP - 1) / NP;  // ring pieces a B wave issues per step
  static_assert(NPW <= F2, "one piece per MFMA gap at most");

  // ---- A: GEMM1 of this step's chunk into Xn, interleaved with the GELU of the previous chunk's Xp -> H ----------------
  // GELU element pair (e, e + 1) rides behind MFMA (e * F1) / 16
  static constexpr int gelu_at(int e) { return (e * F1) / 16; }
  struct AState {
    hx8 F[R];
    float g[16];
  };
  template <int Q, int E>
  static __device__ __forceinline__ void a_gelu(const f32x16& Xp, AState& st) {
    if constexpr (gelu_at(E) == Q) {
#if defined(CN_GELU_SIG2)
      const f32x2 r = (ABL & 2) ? f32x2{Xp[E], Xp[E + 1]} : cn_gelu_sig2_pk(f32x2{Xp[E], Xp[E + 1]});
      st.g[E] = r[0];
      st.g[E + 1] = r[1];
#else
      st.g[E] = (ABL & 2) ? Xp[E] : cn_gelu_mlp<HT>(Xp[E]);
      st.g[E + 1] = (ABL & 2) ? Xp[E + 1] : cn_gelu_mlp<HT>(Xp[E + 1]);
#endif
    }
    if constexpr (E + 2 < 16) a_gelu<Q, E + 2>(Xp, st);
  }
  // RELOAD: the tile's last step -- fragment s of the next tile's y replaces fy[s] behind the MFMA that read it
  template <int Q, bool RELOAD>
  static __device__ __forceinline__ void a_mstep(const char* wc, hx8 (&fy)[KS1], const hx8 ones, const f32x16& Xp,
                                                 f32x16& Xn, AState& st, const HT* ynext) {
    if constexpr (Q + PRE < F1) st.F[(Q + PRE) % R] = W::frag(wc, Q + PRE);
    if constexpr (Q == 0) Xn = W::mma(st.F[0], fy[0], W::zero16());
    else if constexpr (Q < KS1) Xn = W::mma(st.F[Q % R], fy[Q], Xn);
    else Xn = W::mma(st.F[Q % R], ones, Xn);
    a_gelu<Q, 0>(Xp, st);
    if constexpr (RELOAD && Q >= 1 && Q <= KS1) fy[Q - 1] = *(const hx8*)(ynext + 16 * (Q - 1));  // (one MFMA behind its last reader)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < F1) a_mstep<Q + 1, RELOAD>(wc, fy, ones, Xp, Xn, st, ynext);
  }
  template <int Q>
  static __device__ __forceinline__ void a_prefetch(const char* wc, AState& st) {
    st.F[Q % R] = W::frag(wc, Q);
    if constexpr (Q + 1 < PRE) a_prefetch<Q + 1>(wc, st);
  }
  // wc: this step's entry (lane offset applied); gdst: this pair's hand-over buffer (lane offset applied);
  // ynext: the next tile's y rows with this lane's offset applied (RELOAD only)
  template <bool RELOAD>
  static __device__ __forceinline__ void a_step(const char* wc, hx8 (&fy)[KS1], const hx8 ones, const f32x16& Xp,
                                                f32x16& Xn, char* gdst, const HT* ynext) {
    AState st;
    a_prefetch<0>(wc, st);
    __builtin_amdgcn_sched_barrier(0);
    a_mstep<0, RELOAD>(wc, fy, ones, Xp, Xn, st, ynext);
    *(hx8*)gdst = cn_sat8<HT>(cn_pack8<HT>(st.g[0], st.g[1], st.g[2], st.g[3], st.g[4], st.g[5], st.g[6], st.g[7]));
    *(hx8*)(gdst + 1024) = cn_sat8<HT>(cn_pack8<HT>(st.g[8], st.g[9], st.g[10], st.g[11], st.g[12], st.g[13], st.g[14], st.g[15]));
  }

  // ---- B: O^T += W2c' . G^T; lane = position m0 + (l & 31), register r of tile t = channel 32 t + 16 (r >> 3) + 8 (l >> 5) + (r & 7)
  struct Dma {
    const char* src;   // entry to fetch (wave-uniform)
    unsigned dst;      // LDS byte address of its slot
    unsigned voff;     // lane * 16
    int first;         // this wave's first piece; the others follow at stride NP
    int n_pieces;      // pieces of the entry (0: nothing to issue)
  };
  template <int I>
  static __device__ __forceinline__ void dma_piece(const Dma& d) {
    const int piece = d.first + I * NP;
    if (piece < d.n_pieces) cn_dma16_s(d.src + piece * 1024, d.voff, d.dst + piece * 1024);
  }
  // The residual rows of a tile as they come from memory: 32 channels of this lane's position = 16 values (two 16-byte pieces
  // at fp16, four at fp32); row pointers carry the lane's position and channel-run offset, t is an immediate.
  static constexpr int RAWN = sizeof(XT) == 2 ? 2 : 4;
  struct Raw {
    u32x4 v[RAWN];
  };
  static __device__ __forceinline__ void load_raw(const XT* xrow, int t, Raw& raw) {
    if constexpr (sizeof(XT) == 2) {
      raw.v[0] = *(const u32x4*)(xrow + 32 * t);
      raw.v[1] = *(const u32x4*)(xrow + 32 * t + 16);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) raw.v[q] = *(const u32x4*)(xrow + 32 * t + 16 * (q >> 1) + 4 * (q & 1));
    }
  }
  static __device__ __forceinline__ void raw_to_acc(const Raw& raw, f32x16& Ot) {
    if constexpr (sizeof(XT) == 2) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const cn_h8<XT> h = __builtin_bit_cast(cn_h8<XT>, raw.v[o]);
#pragma unroll
        for (int i = 0; i < 8; ++i) Ot[8 * o + i] = (float)h[i];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 f = __builtin_bit_cast(f32x4, raw.v[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) Ot[4 * q + e] = f[e];
      }
    }
  }
  // x' = O + bb, converted to the stream's type (bbl: the fp32 bias vector in LDS with this lane's channel-run offset applied)
  static __device__ __forceinline__ void store_tile(XT* xrow, const char* bbl, int t, bool in_range, const f32x16& Ot) {
    float o[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const f32x4*)(bbl + (32 * t + 16 * (q >> 1) + 4 * (q & 1)) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * q + e] = Ot[4 * q + e] + b[e];
    }
    if (!in_range) return;
    if constexpr (sizeof(XT) == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *(cn_h8<XT>*)(xrow + 32 * t + 16 * h) = cn_pack8<XT>(o[8 * h], o[8 * h + 1], o[8 * h + 2], o[8 * h + 3], o[8 * h + 4],
                                                             o[8 * h + 5], o[8 * h + 6], o[8 * h + 7]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(xrow + 32 * t + 16 * (q >> 1) + 4 * (q & 1)) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
    }
  }
  struct BState {
    hx8 F[R];
  };
  // MFMA order: Q = 2 t + k (the k-steps of one channel tile back to back); fragment (k, t) sits at k * NT2 + t of the W2 part.
  // LAST: the tile's last chunk -- one tile behind the MFMAs, O[t] (+ bb) is stored and, if there is a next position
  // tile for this pair, its residual is loaded into the same registers.
  static constexpr int fidx(int q) { return (q & 1) * NT2 + (q >> 1); }
  static constexpr int piece_at(int i) { return (i * F2) / NPW; }  // piece i rides behind MFMA piece_at(i)
  template <int Q, int I>
  static __device__ __forceinline__ void b_dma(const Dma& d) {
    if constexpr (I < NPW) {
      if constexpr (piece_at(I) == Q) dma_piece<I>(d);
      else b_dma<Q, I + 1>(d);
    }
  }
  // LAST: the tile's last chunk -- one channel tile behind the MFMAs, O[t] (+ bb) is stored and the residual of the pair's
  // next position tile is loaded (raw[t]).  FIRST: the tile's first chunk -- raw[t] becomes O[t] right before the first MFMA
  // of channel tile t.  Both are UNCONDITIONAL instruction streams (rows outside the
  // tensor are masked per lane; a pair without a next tile re-loads rows it never uses): a branch around them, or a
  // run-time choice between two instantiations of the step, made the register allocator give O different registers on
  // the two paths and shuffle all 192 of them through scratch at the join.
  template <int Q, bool FIRST, bool LAST>
  static __device__ __forceinline__ void b_mstep(const char* w2, const hx8 (&H)[2], f32x16 (&O)[NT2], Raw (&raw)[NT2], BState& st,
                                                 const Dma& d, XT* xrow, const XT* xnext, const char* bbl, bool in_range) {
    if constexpr (Q + PRE < F2) st.F[(Q + PRE) % R] = W::frag(w2, fidx(Q + PRE));
    constexpr int k = Q & 1, t = Q >> 1;
    if constexpr (FIRST && k == 0) raw_to_acc(raw[t], O[t]);
    O[t] = W::mma(st.F[Q % R], H[k], O[t]);
    b_dma<Q, 0>(d);
    if constexpr (LAST && k == 1 && t >= 1) {
      store_tile(xrow, bbl, t - 1, in_range, O[t - 1]);
      load_raw(xnext, t - 1, raw[t - 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q + 1 < F2) b_mstep<Q + 1, FIRST, LAST>(w2, H, O, raw, st, d, xrow, xnext, bbl, in_range);
  }
  template <int Q>
  static __device__ __forceinline__ void b_prefetch(const char* w2, BState& st) {
    st.F[Q % R] = W::frag(w2, fidx(Q));
    if constexpr (Q + 1 < PRE) b_prefetch<Q + 1>(w2, st);
  }
  template <bool FIRST, bool LAST>
  static __device__ __forceinline__ void b_step(const char* w2, const hx8 (&H)[2], f32x16 (&O)[NT2], Raw (&raw)[NT2], const Dma& d,
                                                XT* xrow, const XT* xnext, const char* bbl, bool in_range) {
    BState st;
    b_prefetch<0>(w2, st);
    __builtin_amdgcn_sched_barrier(0);
    b_mstep<0, FIRST, LAST>(w2, H, O, raw, st, d, xrow, xnext, bbl, in_range);
    if constexpr (LAST) {
      store_tile(xrow, bbl, NT2 - 1, in_range, O[NT2 - 1]);
      load_raw(xnext, NT2 - 1, raw[NT2 - 1]);
    }
  }
  static __device__ __forceinline__ void issue_all(const Dma& d) {  // a step without a tile still owes the refill
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int piece = d.first + i * NP;
      if (piece < d.n_pieces) cn_dma16_s(d.src + piece * 1024, d.voff, d.dst + piece * 1024);
    }
  }
};

// s_waitcnt vmcnt(n) for a wave-uniform n (the immediate must be a constant); n >= 63: nothing to wait for (a wave never
// has more than 63 vector-memory operations outstanding)
__device__ __forceinline__ void cn_vm_wait(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
    case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
    case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
    case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
    case 51: asm volatile("s_waitcnt vmcnt(51)" ::: "memory"); break;
    case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 53: asm volatile("s_waitcnt vmcnt(53)" ::: "memory"); break;
    case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
    case 55: asm volatile("s_waitcnt vmcnt(55)" ::: "memory"); break;
    case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    case 57: asm volatile("s_waitcnt vmcnt(57)" ::: "memory"); break;
    case 58: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
    case 59: asm volatile("s_waitcnt vmcnt(59)" ::: "memory"); break;
    case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    case 61: asm volatile("s_waitcnt vmcnt(61)" ::: "memory"); break;
    case 62: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
    default: break;
  }
}

// NP pairs per block (2 NP waves: waves [0, NP) are the A roles, [NP, 2 NP) the B roles: with waves dealt round-robin over the
// four SIMDs a pair shares its SIMD when NP is a multiple of 4); pair p owns tiles t_lo + p + it * NP of the block's range.
template <int C, int NP, int NST, int ABL = 0, typename HT = bf16_t, typename XT = float>
__global__ __launch_bounds__(2 * NP * 64) void cn_mlp_rs_kernel(const HT* __restrict__ Y, const HT* __restrict__ WS,
                                                                XT* __restrict__ X, int M) {
  typedef Rc2Geom<C, 1> G;
  typedef Rc2Wave<C, 1, HT, XT> W;
  typedef RsWave<C, NP, ABL, HT, XT> RW;
  typedef cn_h8<HT> hx8;
  constexpr int NCH = G::NSTEP, FR = G::FRAGS, SB = G::STEP_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* gbuf = smem + NST * SB;          // NP x 2 KB hand-over buffers
  char* bbuf = gbuf + NP * 2048;         // bb = s b2 (fp32, C)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_b = wave >= NP;
  const int pair = role_b ? wave - NP : wave;
  const float* aux = (const float*)((const char*)WS + G::STREAM_BYTES);  // bb = s b2 (fp32, C: pk_mlp_rs)

  const int n_tiles = (M + 31) >> 5;
  const int t_lo = (int)((long)blockIdx.x * n_tiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
  const int max_it = (t_hi - t_lo + NP - 1) / NP;  // block-uniform

  const unsigned voff = lane * 16;
  const unsigned lds0 = cn_lds_addr(smem);
  const int n_mine = role_b ? (FR - pair + NP - 1) / NP : 0;  // ring pieces of this wave per entry
  auto entry = [&](int g) {
    typename RW::Dma d;
    d.src = (const char*)WS + (size_t)(g % NCH) * SB, d.dst = lds0 + (unsigned)((g % NST) * SB);
    d.voff = voff, d.first = pair, d.n_pieces = (!role_b || ((ABL & 1) && g >= NST - 1)) ? 0 : FR;
    return d;
  };
  for (int i = tid; i < C; i += 2 * NP * 64) ((float*)bbuf)[i] = aux[i];
#pragma unroll
  for (int g = 0; g < NST - 1; ++g) RW::issue_all(entry(g));
  const char* wl = smem + lane * 16;
  char* gl = gbuf + pair * 2048 + lane * 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // bb is in LDS before the first barrier

  // Both roles run max_it * NCH + 2 steps of two barriers each; their loops have NO run-time choice between step variants
  // (see b_mstep): the last step of a tile is peeled behind the loop over the others.
  if constexpr ((ABL & 24) != 0) {  // lab: static wave priority for one of the roles (8: B, 16: A)
    if (role_b == ((ABL & 8) != 0)) __builtin_amdgcn_s_setprio(1);
  }
  if (!role_b) {
    // ================================================= A ==============================================================
    hx8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (HT)((lane < 32 && i < 2) ? 1.0f : 0.0f);
    hx8 fy[G::KS1];
    f32x16 Xa = W::zero16(), Xb = W::zero16();
    const int yoff = (lane & 31) * C + 8 * (lane >> 5);  // elements: fy[s] = y[m0 + (l & 31)][16 s + 8 (l >> 5) .. + 8]
    const int tile0 = t_lo + pair < t_hi ? t_lo + pair : t_lo;  // (a pair without a tile multiplies rows nobody stores)
    W::load_y(Y, tile0 * 32, lane, fy);
    int g = 0;
    auto step2 = [&](auto reload_tag, const HT* ynext) {  // two steps: the roles of the two X accumulators swap
      constexpr bool RELOAD = decltype(reload_tag)::value;
      __builtin_amdgcn_s_barrier();
      if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
      RW::template a_step<false>(wl + (g % NST) * SB, fy, ones, Xb, Xa, gl, ynext);  // GEMM1(g) -> Xa, GELU of Xb (chunk g - 1) -> G
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++g;
      __builtin_amdgcn_s_barrier();
      if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
      RW::template a_step<RELOAD>(wl + (g % NST) * SB, fy, ones, Xa, Xb, gl, ynext);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++g;
    };
    for (int it = 0; it < max_it; ++it) {
      const int tile = t_lo + pair + it * NP;
      // No next tile: the re-load stays an unconditional instruction stream (see b_mstep) but every lane reads ROW 0 of the
      // pair's first tile -- one 768-byte row instead of a whole 24 KB tile.  (Round 3 re-read the first tile here: with two
      // tiles per pair and launch that was + 50 % of the y and residual reads, the "1.39x FETCH_SIZE" of VERDICT r03.)
      const bool has_next = tile + NP < t_hi && !(ABL & 4);
      const HT* ynext = has_next ? Y + (size_t)(tile + NP) * 32 * C + yoff : Y + (size_t)tile0 * 32 * C + 8 * (lane >> 5);
      for (int j = 0; j < NCH - 2; j += 2) step2(std::false_type{}, ynext);
      step2(std::true_type{}, ynext);  // the tile's last two steps: the second re-loads y in place
    }
    step2(std::false_type{}, Y);  // trailing: the GELU of the last chunk (B is two chunks behind)
  } else {
    // ================================================= B ==============================================================
    f32x16 O[G::NT2];
    typename RW::Raw raw[G::NT2];
    const int xoff = (lane & 31) * C + 8 * (lane >> 5);  // elements: this lane's position row and channel-run offset
    const char* bbl = bbuf + 32 * (lane >> 5);
    const int tile0 = t_lo + pair < t_hi ? t_lo + pair : t_lo;
    {
      const XT* x0 = X + (size_t)tile0 * 32 * C + xoff;
#pragma unroll
      for (int t = 0; t < G::NT2; ++t) RW::load_raw(x0, t, raw[t]);
    }
    int g = 0;
    // the last step in which this wave ran a tile boundary, and the vector-memory operations it really issued there: the
    // loads always, the stores only if some lane's row is inside the tensor (an all-masked store is branched over)
    constexpr int IO_N = RW::RAWN * G::NT2;
    int io_step = -1000, io_n = 0;
    hx8 H[2];
    auto head = [&]() {  // ring wait, first barrier, G(g - 2) -> registers, second barrier
      // Own pieces of the entry about to be consumed have landed.  They were issued in step g - (NST - 1); younger than them
      // (the counter retires in order) are the pieces of NST - 2 later entries and, if a tile boundary fell strictly
      // between, its stores and loads.
      cn_vm_wait((NST - 2) * n_mine + ((io_step > g - (NST - 1) && io_step < g) ? io_n : 0));
      __builtin_amdgcn_s_barrier();
      H[0] = *(const hx8*)gl;
      H[1] = *(const hx8*)(gl + 1024);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();  // (32: lab ablation, racy: what the second barrier costs)
    };
    for (int i = 0; i < 2; ++i, ++g) {  // B runs two chunks behind A: nothing to multiply yet, the refill is still owed
      head();
      RW::issue_all(entry(g + NST - 1));
    }
    for (int it = 0; it < max_it; ++it) {
      const int tile = t_lo + pair + it * NP;
      const bool valid = tile < t_hi;
      const int tcur = valid ? tile : tile0;
      const bool has_next = valid && tile + NP < t_hi && !(ABL & 4);
      XT* xrow = X + (size_t)tcur * 32 * C + xoff;
      // (no next tile: every lane re-loads row 0 of the pair's first tile, not a whole tile it never uses)
      const XT* xnext = has_next ? X + (size_t)(tile + NP) * 32 * C + xoff : X + (size_t)tile0 * 32 * C + 8 * (lane >> 5);
      const bool in_range = valid && tile * 32 + (lane & 31) < M && (!(ABL & 4) || it == max_it - 1);
      head();
      RW::template b_step<true, false>(wl + (g % NST) * SB + RW::F1 * 1024, H, O, raw, entry(g + NST - 1), xrow, xnext, bbl, false);
      ++g;
      for (int cb = 1; cb < NCH - 1; ++cb, ++g) {
        head();
        RW::template b_step<false, false>(wl + (g % NST) * SB + RW::F1 * 1024, H, O, raw, entry(g + NST - 1), xrow, xnext, bbl, false);
      }
      head();
      RW::template b_step<false, true>(wl + (g % NST) * SB + RW::F1 * 1024, H, O, raw, entry(g + NST - 1), xrow, xnext, bbl, in_range);
      io_step = g;
      io_n = IO_N + ((valid && (!(ABL & 4) || it == max_it - 1)) ? IO_N : 0);  // (lane 0's row of a valid tile is in range)
      ++g;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring was filled NST - 1 entries past the end
}

template <int C, int NP, int NST, int ABL = 0, typename HT, typename XT>
static int cn_launch_mlp_rs(const HT* Y, const HT* WS, XT* X, int M, int n_blocks, hipStream_t s) {
  constexpr int SMEM = NST * Rc2Geom<C, 1>::STEP_BYTES + NP * 2048 + C * 4;
  static_assert(SMEM <= 160 * 1024, "ring + hand-over buffers must fit in LDS");
  CN_TRY(cn_configure_lds((const void*)cn_mlp_rs_kernel<C, NP, NST, ABL, HT, XT>, SMEM));
  const int grid = cn_rc2_grid((M + 31) / 32, NP, n_blocks);
  hipLaunchKernelGGL((cn_mlp_rs_kernel<C, NP, NST, ABL, HT, XT>), dim3((unsigned)grid), dim3(2 * NP * 64), SMEM, s, Y, WS, X, M);
  CN_LAUNCH_CHECK();
  return CN_OK;
}
