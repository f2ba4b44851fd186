// bf16 NT GEMM v2 for gfx950: LDS-DMA staged, XOR-swizzled, double buffered.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]      A: (M, K) activations, W: (N, K) nn.Linear weight
//
// Differences from gemm.h (kept for the fp32 parity mode):
//  * tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass);
//    a wave-instruction writes 1 KiB linearly, so the bank-conflict swizzle is applied to the
//    per-lane SOURCE address and mirrored on the ds_read_b128 address (cdna guide rule 21):
//        16-byte chunk c of row r lives at slot  c ^ ((r / rows_per_256B) & (chunks_per_row-1))
//  * BK = 64 (32 only when K % 64 != 0, i.e. the K = 96 pointwise conv of stage 0); the skinny
//    decoder GEMMs (M = B*beam rows, K = 256 per slice) use BK = 256: the whole K extent is one
//    stage, so a block pays ONE memory latency instead of one per k-tile;
//  * the DMA of k-tile t+1 is issued before the fragments of tile t are read, one
//    vmcnt(0) + barrier per k-tile;
//  * optional split-K over blockIdx.y: the epilogue receives the slice index (partial slabs,
//    summed in a fixed order by the consumer -> deterministic, no atomics);
//  * tile shapes (cn_gemm2 below): 64 x 64 (decoder), 128 x 128 / 128 x 96 with 4 waves and two blocks per CU, and for the
//    large N % 256 == 0 products of stage 3 one block of 8 waves (2 x 4) per CU: 224 / 256 x 256 two-deep, or 224 x 192
//    with a three-deep ring when the whole product fits in one round over the chip.  A 128 x 128 tile asks the CU's load
//    path for 512 bytes per 32x32x16-sized MFMA -- all it delivers at full MFMA rate -- the big tiles for 256-300;
//  * the LDS-DMA pieces are issued through inline asm (common.h: cn_dma16_v) and are invisible to the compiler: every
//    ring wait below is counted by hand, per wave (the last wave may own fewer pieces per stage than the others).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

// out[ks][m][n] = acc  (partial sums of one K slice; bias / residual are added by the consumer)
struct EpiSlab {
  float* out;
  int ld;
  size_t slab_stride;
  typedef float stage_t;
  static constexpr bool kBatched = false;
  __device__ __forceinline__ float pre(int, float x, int) const { return x; }
  __device__ __forceinline__ void commit(int m, int n, const float* chunk, int N, int ks) const {
    float* p = out + (size_t)ks * slab_stride + (size_t)m * ld + n;
    if (n + 3 < N && (ld & 3) == 0) {
      *(f32x4*)p = *(const f32x4*)chunk;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n + i < N) p[i] = chunk[i];
    }
  }
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int N, int ks) const {
    float* p = out + (size_t)ks * slab_stride + (size_t)m * ld + n;
    if (n + 3 < N && (ld & 3) == 0) {
      *(f32x4*)p = v;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n + i < N) p[i] = v[i];
    }
  }
};

// ---- shared pieces ------------------------------------------------------------------------------
template <int BK> struct G2Geom {
  static constexpr int RBY = BK * 2;                       // bytes per tile row
  static constexpr int CPR = RBY / 16;                     // 16-byte chunks per row (4, 8 or 32)
  static constexpr int RPB = RBY >= 256 ? 1 : 256 / RBY;   // tile rows per 256-byte LDS bank row
  static constexpr int SWM = CPR > 16 ? 15 : CPR - 1;      // chunk ^= (row / RPB) & SWM
  static constexpr int RPI = 1024 / RBY;                   // tile rows per 1 KiB DMA piece
};

// all MFMAs of one staged k-tile: acc[a][b] += W-tile(a) . A-tile(b)^T
// (WM x WN waves per block, each owning a (BM / WM) x (BN / WN) sub-tile: 2 x 2 everywhere but the 256 x 256 tiles, 2 x 4)
template <int BM, int BN, int BK, int WM = 2, int WN = 2, typename HT = bf16_t>
__device__ __forceinline__ void cn_g2_compute(const char* sA, const char* sW, int lane, int wm, int wn,
                                              f32x4 (&acc)[BN / (16 * WN)][BM / (16 * WM)]) {
  typedef G2Geom<BK> G;
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), KS = BK / 32;
  const int lr = lane & 15;
  const int sw = (lr / G::RPB) & G::SWM;
  const int row_off = lr * G::RBY;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int coff = (((lane >> 4) + 4 * ks) ^ sw) * 16;
    cn_h8<HT> fw[TN], fa[TM];
#pragma unroll
    for (int a = 0; a < TN; ++a) fw[a] = *(const cn_h8<HT>*)(sW + (wn * (BN / WN) + a * 16) * G::RBY + row_off + coff);
#pragma unroll
    for (int b = 0; b < TM; ++b) fa[b] = *(const cn_h8<HT>*)(sA + (wm * (BM / WM) + b * 16) * G::RBY + row_off + coff);
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = cn_mma16(fw[a], fa[b], acc[a][b]);
  }
}

// The same k-tile for sp16 operands (common.h): the staged rows hold 4-byte {hi, lo} elements, so a tile of BK bf16-sized
// columns is BK / 2 elements and one MFMA k-step (32 elements) spans 128 bytes of a row: lane group g takes the two
// neighbouring 16-byte chunks 2 (g + 4 ks) and 2 (g + 4 ks) + 1 (both through the same XOR swizzle as the loader), splits the
// eight {hi, lo} dwords into a hi and a lo fp16 vector (v_perm_b32) and the product of a pair of fragments is three
// v_mfma_f32_16x16x32_f16: lo.hi + hi.lo + hi.hi, smallest terms first.
__device__ __forceinline__ void cn_sp_split(u32x4 c0, u32x4 c1, f16x8& hi, f16x8& lo) {
  u32x4 h, l;
  h[0] = __builtin_amdgcn_perm(c0[1], c0[0], 0x05040100u);
  h[1] = __builtin_amdgcn_perm(c0[3], c0[2], 0x05040100u);
  h[2] = __builtin_amdgcn_perm(c1[1], c1[0], 0x05040100u);
  h[3] = __builtin_amdgcn_perm(c1[3], c1[2], 0x05040100u);
  l[0] = __builtin_amdgcn_perm(c0[1], c0[0], 0x07060302u);
  l[1] = __builtin_amdgcn_perm(c0[3], c0[2], 0x07060302u);
  l[2] = __builtin_amdgcn_perm(c1[1], c1[0], 0x07060302u);
  l[3] = __builtin_amdgcn_perm(c1[3], c1[2], 0x07060302u);
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}
template <int BM, int BN, int BK, int WM = 2, int WN = 2>
__device__ __forceinline__ void cn_g2_compute_sp(const char* sA, const char* sW, int lane, int wm, int wn,
                                                 f32x4 (&acc)[BN / (16 * WN)][BM / (16 * WM)]) {
  typedef G2Geom<BK> G;
  static_assert(BK % 64 == 0, "sp16 tiles: a k-step is 32 elements = 64 bf16-sized columns");
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), KS = BK / 64;
  const int lr = lane & 15;
  const int sw = (lr / G::RPB) & G::SWM;
  const int row_off = lr * G::RBY;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c0 = ((2 * ((lane >> 4) + 4 * ks)) ^ sw) * 16, c1 = ((2 * ((lane >> 4) + 4 * ks) + 1) ^ sw) * 16;
    f16x8 wh[TN], wl[TN], ah[TM], al[TM];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const char* r = sW + (wn * (BN / WN) + a * 16) * G::RBY + row_off;
      cn_sp_split(*(const u32x4*)(r + c0), *(const u32x4*)(r + c1), wh[a], wl[a]);
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const char* r = sA + (wm * (BM / WM) + b * 16) * G::RBY + row_off;
      cn_sp_split(*(const u32x4*)(r + c0), *(const u32x4*)(r + c1), ah[b], al[b]);
    }
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        f32x4 c = acc[a][b];
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[a], ah[b], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a], al[b], c, 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a], ah[b], c, 0, 0, 0);
      }
  }
}

__device__ unsigned long long g_g2_prof[16];
// Phase stamps exist only in a profiling build (CN_G2_PROF=1 python build.py --force; tools/g2prof.py): the
// accumulators cost 32 registers, which pushed the 128 x 128 tiles from 148 to 180 and broke the co-residency
// budget of the decode kernels (dec_block.h).
#ifndef CN_G2_PROF
#define G2_STAMP(i)
#else
#define G2_STAMP(i)                                                        \
  if (dbg & 1) { /* accumulated in registers: an atomic per stamp would sit in the vmcnt queue the loop waits on */ \
    const unsigned long long t_ = wall_clock64();                          \
    t_acc[i] += t_ - t_prev;                                               \
    t_prev = t_;                                                           \
  }
#endif

// Epilogue.  fp32 outputs store 16 bytes per lane straight from the MFMA layout (64 contiguous
// bytes per row and instruction; staging a 64 KB fp32 tile only costs occupancy: pw2 +35 %).
// bf16 outputs go registers -> LDS tile [BM][BN] -> whole-row 16-byte chunks: direct stores would be
// 8 bytes per lane scattered over 16 rows = up to 2.2x write amplification at HBM (rocprof
// WRITE_SIZE).  Must be entered after a barrier (the LDS pipeline buffers are reused).
template <int BM, int BN, class Epi, int WM = 2, int WN = 2>
__device__ __forceinline__ void cn_g2_epilogue(char* smem, f32x4 (&acc)[BN / (16 * WN)][BM / (16 * WM)], int m0, int n0, int M, int N,
                                               const Epi& epi, int ks, int tid, int wm, int wn
#ifdef CN_G2_PROF
                                               , int dbg, unsigned long long& t_prev, unsigned long long (&t_acc)[8]
#endif
) {
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), NT = WM * WN * 64;
  const int lane = tid & 63;
  typedef typename Epi::stage_t ST;
  if constexpr (sizeof(ST) == 4) {
    if constexpr (Epi::kBatched) {
      // interior tiles (the whole 128 x 128 block inside the matrix): column parameters once, then per column
      // tile `a` all TM residual loads in flight before the first store, the next tile's loads issued before this
      // tile's stores
      if (epi.fast(N) && m0 + BM <= M && n0 + BN <= N) {
        const int mb = m0 + wm * (BM / WM) + (lane & 15);
        const int nb = n0 + wn * (BN / WN) + 4 * (lane >> 4);
        // (register budget: the 128 x 128 tiles must stay at 148 registers so that a decode block can start next to
        // one resident GEMM workgroup, dec_block.h; sched_barrier keeps hipcc from hoisting every accumulator read
        // and column load to the top of the epilogue)
        const auto* rp = epi.resid + (size_t)mb * epi.ld + nb;  // one 64-bit base, immediate / scalar offsets from here
        auto* op = epi.out + (size_t)mb * epi.ld + nb;          // (the residual stream's type: fp32 or fp16, common.h)
        const size_t rstride = (size_t)16 * epi.ld;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          const typename Epi::Cols c = epi.cols(nb + a * 16);
          f32x4 r[TM];
#pragma unroll
          for (int b = 0; b < TM; ++b) r[b] = cn_ld4(rp + b * rstride + a * 16);
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            const f32x4 v = acc[a][b];
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = r[b][i] + c.s[i] * (v[i] + c.b[i]);
            cn_store4(op + b * rstride + a * 16, o[0], o[1], o[2], o[3]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        G2_STAMP(7)
        return;
      }
    }
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (BM / WM) + b * 16 + (lane & 15);
        const int n = n0 + wn * (BN / WN) + a * 16 + 4 * (lane >> 4);
        if (m < M && n < N) epi(m, n, acc[a][b], N, ks);
      }
  } else {
    constexpr int EPC = 16 / (int)sizeof(ST);  // elements per 16-byte chunk
    constexpr int PITCH = BN + EPC;            // padded row pitch (elements)
    constexpr int CH = BN / EPC;               // chunks per row
    ST* tile = (ST*)smem;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int ml = wm * (BM / WM) + b * 16 + (lane & 15);
        const int nl = wn * (BN / WN) + a * 16 + 4 * (lane >> 4);
        const f32x4 v = epi.pre4(n0 + nl, acc[a][b], N);
        cn_store4(tile + ml * PITCH + nl, v[0], v[1], v[2], v[3]);
      }
    G2_STAMP(5)
    __syncthreads();
    G2_STAMP(6)
    for (int idx = tid; idx < BM * CH; idx += NT) {
      const int r = idx / CH, c = idx % CH;
      const int m = m0 + r, n = n0 + c * EPC;
      if (m < M && n < N) epi.commit(m, n, tile + r * PITCH + c * EPC, N, ks);
    }
  }
}

// OPK = the operand kind of the staged bytes: 0 bf16, 1 sp16, 2 fp16 (the kernel takes them as 2-byte columns either way).
// sp16: the operands are sp16 matrices handed over as bf16-sized columns (lda, ldw, K, k_slice all doubled by the launcher):
// staging is byte-for-byte the bf16 path, only the fragment reads / MFMAs of a k-tile differ (cn_g2_compute_sp)
template <int BM, int BN, int BK, int NST, class Epi, int WM = 2, int WN = 2, int OPK = 0>
__global__ __launch_bounds__(WM * WN * 64) void cn_gemm2_kernel(const bf16_t* __restrict__ A, int lda,
                                                       const bf16_t* __restrict__ W, int ldw, int M, int N, int K,
                                                       int k_slice, Epi epi, int dbg) {
#ifdef CN_G2_PROF
  unsigned long long t_prev = (dbg & 1) ? wall_clock64() : 0ull;
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  typedef G2Geom<BK> G;
  constexpr int RBY = G::RBY, CPR = G::CPR, RPB = G::RPB, SWM = G::SWM, RPI = G::RPI;
  constexpr int A_BYTES = BM * RBY, W_BYTES = BN * RBY, BUF = A_BYTES + W_BYTES;
  constexpr int N_DMA = BUF / 1024;
  static_assert(BUF % 1024 == 0 && A_BYTES % 1024 == 0, "tile must be a whole number of 1 KiB DMA pieces");
  constexpr int NWV = WM * WN;
  constexpr int DPW = (N_DMA + NWV - 1) / NWV;  // DMA instructions per wave and stage
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
  const int bid = cn_xcd_remap(blockIdx.x, gridDim.x);  // every XCD gets a contiguous run of tiles
  // ... ordered so that the LARGER operand is the one an XCD sees only a slice of: tall products (M >= N, the encoder's) walk the
  // n-tiles of one A panel first -- an XCD's L2 holds a few A panels and all of W; wide ones (N > M: the decoder's classifier,
  // 5 632 x 256 weights against a few hundred rows) walk the m-tiles of one W panel first, so each XCD pulls 1 / 8 of the
  // weights instead of all of them (speed only: which workgroup computes a tile does not change the tile)
  const bool wide = N > M;
  const int m0 = (wide ? bid % m_tiles : bid / n_tiles) * BM;
  const int n0 = (wide ? bid / m_tiles : bid % n_tiles) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int k_begin = blockIdx.y * k_slice;
  const int k_end = min(K, k_begin + k_slice);
  const int KT = (k_end - k_begin) / BK;

  // per-lane DMA sources (k offset added per stage)
  const char* src[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int inst = wave * DPW + i;
    const int row = inst * RPI + lane / CPR;  // row inside [A rows | W rows]
    const int slot = lane % CPR;
    if (inst < N_DMA) {
      if (row < BM) {
        const int chunk = slot ^ ((row / RPB) & SWM);
        src[i] = (const char*)(A + (size_t)min(m0 + row, M - 1) * lda + k_begin) + chunk * 16;
      } else {
        const int r = row - BM;
        const int chunk = slot ^ ((r / RPB) & SWM);
        src[i] = (const char*)(W + (size_t)min(n0 + r, N - 1) * ldw + k_begin) + chunk * 16;
      }
    } else {
      src[i] = (const char*)A;
    }
  }
  const unsigned lds0 = cn_lds_addr(smem);
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int inst = wave * DPW + i;
      if (N_DMA % NWV == 0 || inst < N_DMA)
        cn_dma16_v(src[i] + (size_t)kt * RBY, lds0 + (unsigned)(buf * BUF + inst * 1024));
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (NST == 1) {
    // single stage (BK = 256 skinny GEMMs): refill only after every wave has read the tile
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      if constexpr (OPK == 1) cn_g2_compute_sp<BM, BN, BK, WM, WN>(smem, smem + A_BYTES, lane, wm, wn, acc);
      else cn_g2_compute<BM, BN, BK, WM, WN, typename std::conditional<OPK == 2, half_t, bf16_t>::type>(smem, smem + A_BYTES, lane, wm, wn, acc);
      if (kt + 1 < KT) {
        __syncthreads();
        stage(0, kt + 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // NST-deep LDS ring: NST-1 k-tiles are in flight while one is consumed.  The DMA stays in flight
    // ACROSS the barrier (raw s_barrier + counted vmcnt; __syncthreads() would drain it -- cdna guide
    // "Pipelining across barriers"): each wave waits until all but its newest NST-2 stages landed,
    // the barrier makes every wave's pieces of tile kt visible and proves tile kt-1 is no longer read,
    // so its buffer is refilled right away.
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
      if (t < KT) stage(t, t);
    G2_STAMP(0)
    for (int kt = 0; kt < KT; ++kt) {
      const int newer = KT - 1 - kt;  // stages issued after tile kt that may stay in flight
      // (pieces are dealt DPW per wave in order, so only the LAST wave can own fewer: its count decides its wait -- with
      // vmcnt(DPW) it would let one of tile kt's own pieces stay in flight and read the tile before it has landed)
      constexpr int LASTW = N_DMA - (NWV - 1) * DPW;  // pieces per stage of the last wave
      static_assert(LASTW > 0 && LASTW <= DPW, "pieces are dealt to every wave");
      if (LASTW != DPW && wave == NWV - 1) {
        if (NST >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LASTW) : "memory");
        else if (NST >= 3 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LASTW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (NST >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory");
        else if (NST >= 3 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (kt == 0) { G2_STAMP(1) } else { G2_STAMP(2) }
      if (kt + NST - 1 < KT) stage((kt + NST - 1) % NST, kt + NST - 1);
      const char* sA = smem + (kt % NST) * BUF;
#ifdef CN_G2_PROF
      if (!(dbg & 4))
#endif
      {
        if constexpr (OPK == 1) cn_g2_compute_sp<BM, BN, BK, WM, WN>(sA, sA + A_BYTES, lane, wm, wn, acc);
        else cn_g2_compute<BM, BN, BK, WM, WN, typename std::conditional<OPK == 2, half_t, bf16_t>::type>(sA, sA + A_BYTES, lane, wm, wn, acc);
      }
      G2_STAMP(3)
    }
    __syncthreads();  // all fragment reads done before the epilogue reuses the LDS
  }
  G2_STAMP(4)
#ifdef CN_G2_PROF
  if (!(dbg & 2) || acc[0][0][0] == 12345.678f)  // experiment: skip the epilogue (wrong results)
    cn_g2_epilogue<BM, BN, Epi, WM, WN>(smem, acc, m0, n0, M, N, epi, (int)blockIdx.y, tid, wm, wn, dbg & 1, t_prev, t_acc);
#else
  cn_g2_epilogue<BM, BN, Epi, WM, WN>(smem, acc, m0, n0, M, N, epi, (int)blockIdx.y, tid, wm, wn);
#endif
  G2_STAMP(7)
#ifdef CN_G2_PROF
  if ((dbg & 1) && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_g2_prof[i], t_acc[i]);
    atomicAdd(&g_g2_prof[8], 1ull);
  }
#endif
}

// Phase-stamp / skip controls exist in the profiling build only (CN_G2_PROF=1 python build.py --force, tools/g2prof.py);
// the product library reads no environment variable on this path.
#ifdef CN_G2_PROF
static inline int g2_debug_level() {  // CN_G2_DEBUG = BM + BN of the tile shape to instrument (256: the 128 x 128 tiles)
  static const int v = getenv("CN_G2_DEBUG") ? atoi(getenv("CN_G2_DEBUG")) : 0;
  return v;
}
static inline int g2_debug_skip() {  // CN_G2_SKIP bit 1 (2) = no epilogue, bit 2 (4) = no MFMA phase
  static const int v = getenv("CN_G2_SKIP") ? atoi(getenv("CN_G2_SKIP")) : 0;
  return v;
}
static inline int g2_debug_epi() {  // CN_G2_DEBUG_EPI = 2: bf16-output (staged) epilogues only, 4: fp32-output only
  static const int v = getenv("CN_G2_DEBUG_EPI") ? atoi(getenv("CN_G2_DEBUG_EPI")) : 0;
  return v;
}
#else
static inline int g2_debug_level() { return 0; }
static inline int g2_debug_skip() { return 0; }
static inline int g2_debug_epi() { return 0; }
#endif

template <int BM, int BN, int BK, int NST, class Epi, int WM = 2, int WN = 2, int OPK = 0>
static int cn_launch_gemm2_t(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, int splits,
                             const Epi& epi, hipStream_t stream) {
  constexpr int EPI_BYTES = sizeof(typename Epi::stage_t) == 4 ? 0 : BM * (BN * 2 + 16);
  constexpr int PIPE_BYTES = NST * (BM + BN) * BK * 2;
  constexpr int SMEM = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  CN_TRY(cn_configure_lds((const void*)cn_gemm2_kernel<BM, BN, BK, NST, Epi, WM, WN, OPK>, SMEM));
  const long blocks = (long)cn_cdiv(M, BM) * cn_cdiv(N, BN);
  const int k_slice = K / splits;
  hipLaunchKernelGGL((cn_gemm2_kernel<BM, BN, BK, NST, Epi, WM, WN, OPK>), dim3((unsigned)blocks, (unsigned)splits), dim3(WM * WN * 64), SMEM,
                     stream, A, lda, W, ldw, M, N, K, k_slice, epi, (g2_debug_level() == BM + BN && (g2_debug_epi() == 0 || g2_debug_epi() == (int)sizeof(typename Epi::stage_t)) ? 1 : 0) | g2_debug_skip());
  CN_LAUNCH_CHECK();
  return CN_OK;
}

static inline int cn_g2_cus() {  // compute units of the current device (cached)
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// bf16 / fp16 dispatch (OPK 0 / 2).  splits > 1 only with a slab epilogue; K / splits must be a multiple of 64.
template <class Epi, int OPK = 0>
static int cn_gemm2(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const Epi& epi,
                    hipStream_t stream, int splits = 1) {
  if (K % 32 != 0 || M <= 0 || N <= 0 || splits < 1 || (K / splits) % 64 != 0 && splits > 1) {
    cn_set_error("cn_gemm2: bad shape M=%d N=%d K=%d splits=%d", M, N, K, splits);
    return CN_ERR_ARG;
  }
  const bool k64 = (K % 64 == 0);
  if (M >= 4096) {
    const bool n96 = (N % 96 == 0) && (N % 128 != 0);
    if (!k64) return cn_launch_gemm2_t<128, 128, 32, 2, Epi, 2, 2, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
    if (n96) return cn_launch_gemm2_t<128, 96, 64, 2, Epi, 2, 2, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
#ifndef CN_G2_NO256
    // 256 x 256 tiles, 8 waves (2 x 4), one block per CU: a 128 x 128 tile asks the CU's load path for 512 bytes per MFMA
    // -- all of the ~64 B/clk it delivers when four SIMDs run MFMAs back to back -- this one for 256
    // A launch of several rounds over the chip takes 224-row tiles when that saves row-rounds (M = 13 888, N = 3072: 3 rounds
    // of 744 tiles of 224 rows instead of 3 of 660 of 256: 103 -> 96 us).  A launch that fits in ONE round (N = 768: the
    // stage-3 pw2 and downsample products, K = 3072 / 1536) is bound by waiting for the next k-tile -- two 64 KB stages do not
    // cover the load latency -- and takes 224 x 192 tiles with a THREE-deep ring (156 KB, 248 of 256 compute units busy):
    // 96 -> 80 us and 61 -> 48 us (same-box traces; with two stages 186 tiles of 224 x 256 took 108 us, 219 of 192 x 256
    // 101).  The N = 3072 product has 12 k-tiles per tile and a heavy epilogue: there the wider two-stage tile wins (96 us
    // against 102).
    if (N % 256 == 0 && M >= 8192 && splits == 1) {
      const long ncu = cn_g2_cus(), nt = N / 256;
      const long r256 = cn_cdiv(cn_cdiv(M, 256) * (int)nt, (int)ncu), r224 = cn_cdiv(cn_cdiv(M, 224) * (int)nt, (int)ncu);
      if (r256 == 1 && N % 192 == 0 && cn_cdiv(M, 224) * (N / 192) <= ncu)
        return cn_launch_gemm2_t<224, 192, 64, 3, Epi, 2, 4, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
      if (r256 > 1 && r224 * 224 < r256 * 256)
        return cn_launch_gemm2_t<224, 256, 64, 2, Epi, 2, 4, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
      return cn_launch_gemm2_t<256, 256, 64, 2, Epi, 2, 4, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
    }
#endif
    return cn_launch_gemm2_t<128, 128, 64, 2, Epi, 2, 2, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
  }
  if (!k64) return cn_launch_gemm2_t<64, 64, 32, 2, Epi, 2, 2, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
  return cn_launch_gemm2_t<64, 64, 64, 2, Epi, 2, 2, OPK>(A, lda, W, ldw, M, N, K, splits, epi, stream);
}

template <class Epi>
static int cn_gemm2(const half_t* A, int lda, const half_t* W, int ldw, int M, int N, int K, const Epi& epi,
                    hipStream_t stream, int splits = 1) {
  return cn_gemm2<Epi, 2>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, epi, stream, splits);
}

// sp16 dispatch ("exact" precision): K elements of 4 bytes = 2 K bf16-sized columns; K % 32 == 0.  128 x 128 (or 128 x 96)
// tiles with four waves and two blocks per compute unit for the encoder's products, 64 x 64 for the decoder's.
// splits > 1 (slab epilogue only): split K over blockIdx.y; K / splits must be a multiple of 32 elements.
template <class Epi>
static int cn_gemm2_sp(const sp16_t* A, int lda, const sp16_t* W, int ldw, int M, int N, int K, const Epi& epi,
                       hipStream_t stream, int splits = 1) {
  if (K % 32 != 0 || M <= 0 || N <= 0 || splits < 1 || (K / splits) % 32 != 0) {
    cn_set_error("cn_gemm2_sp: bad shape M=%d N=%d K=%d splits=%d", M, N, K, splits);
    return CN_ERR_ARG;
  }
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* w = (const bf16_t*)W;
  if (M >= 4096) {
    const bool n96 = (N % 96 == 0) && (N % 128 != 0);
    // (splits is honoured by every tile shape: the exact decoder's FFN2 asks for split-K slabs at any row count, and a launch
    // that ignored it would leave slabs 1 .. splits - 1 unwritten for the LayerNorm that sums them -- ADVICE r03)
    if (n96) return cn_launch_gemm2_t<128, 96, 64, 2, Epi, 2, 2, 1>(a, 2 * lda, w, 2 * ldw, M, N, 2 * K, splits, epi, stream);
    return cn_launch_gemm2_t<128, 128, 64, 2, Epi, 2, 2, 1>(a, 2 * lda, w, 2 * ldw, M, N, 2 * K, splits, epi, stream);
  }
  return cn_launch_gemm2_t<64, 64, 64, 2, Epi, 2, 2, 1>(a, 2 * lda, w, 2 * ldw, M, N, 2 * K, splits, epi, stream);
}

// type-generic front end: bf16 -> v2, sp16 -> v2 with split fragments, fp32 -> gemm.h
template <class Epi>
static int cn_mm(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const Epi& epi,
                 hipStream_t stream) {
  return cn_gemm2(A, lda, W, ldw, M, N, K, epi, stream);
}
template <class Epi>
static int cn_mm(const half_t* A, int lda, const half_t* W, int ldw, int M, int N, int K, const Epi& epi,
                 hipStream_t stream) {
  return cn_gemm2(A, lda, W, ldw, M, N, K, epi, stream);
}
template <class Epi>
static int cn_mm(const sp16_t* A, int lda, const sp16_t* W, int ldw, int M, int N, int K, const Epi& epi,
                 hipStream_t stream) {
  return cn_gemm2_sp(A, lda, W, ldw, M, N, K, epi, stream);
}
template <class Epi>
static int cn_mm(const float* A, int lda, const float* W, int ldw, int M, int N, int K, const Epi& epi,
                 hipStream_t stream) {
  return cn_gemm<float>(A, lda, W, ldw, M, N, K, epi, stream);
}
