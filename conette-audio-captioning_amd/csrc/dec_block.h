// Fused decoder "attention block" for one new token per row (bf16 operands, d_model 256, 8 heads):
//
//   x  = E[tok] * sqrt(d) + PE[step]                     (layer 0)
//      | LN3_prev(x + FFN2 split-K slabs + b2_prev)      (layers > 0: the previous layer's tail)
//   q | k | v = x Win^T + bin
//   a  = SelfAttn(q, cached K/V of the row's ancestors + this step's k, v)   (writes this step's K/V)
//   x1 = LN1(x + a Wo^T + bo)
//   c  = CrossAttn(x1 Wq^T + bq, audio K/V of the row's clip, frame mask)
//   x2 = LN2(x1 + c Wo2^T + bo2)                                              -> x (fp32) and xt (bf16)
//
// i.e. torch's post-norm TransformerDecoderLayer up to the feed-forward (aac_tfmer.py:46-58,
// 100-115).  Everything here is row-local, so a block owns DB_ROWS rows (padded to one MFMA M tile of
// 16) and keeps them on chip across all sub-steps; the unfused path needs 10 dependent launches for
// the same work and the decode phase is bound by per-launch latency (~5 us each, rocprof), not bytes.
//
// The block is a dependent chain of tiny phases, so what matters is that nothing it will need is
// still un-requested when a phase starts.  Two wave roles with separate VMEM queues (vmcnt is per wave
// and returns in order, so mixing the streams would serialise them):
//
//   * 4 GEMM waves: wave w owns output columns 64w .. 64w+63 of all six 256 x 256 weight matrices and
//     holds the current matrix as 32 MFMA A-fragments in registers (128 VGPRs).  The matrices are
//     stored in fragment order (api.hip pk_block_stream: every load instruction moves 1 KB of
//     consecutive bytes -- strided 64-byte row pieces streamed at half the rate), and a fragment's
//     registers are re-loaded with the same fragment of the NEXT matrix right behind the MFMA that
//     consumed it, so one whole matrix (128 KB per block) is always in flight; hipcc counts the
//     vmcnt per fragment because these waves issue no other global access.  Activations are the
//     MFMA B operand from a swizzled LDS tile.  (An LDS ring filled by global_load_lds was as fast
//     in isolation but its 128 KB left no room for the encoder's blocks on the CU: rocprof, bench.)
//   * 4 row waves (wave = row, lane = 4 dims of one head): prologue, both attentions, both
//     LayerNorms, all global stores.  Self-attention K/V (<= 24 steps) are requested at kernel start,
//     cross-attention K/V (<= 32 frames) right after the self-attention, both into registers, so they
//     arrive while the GEMM waves work.  fp32 online softmax, wave-shuffle dot products and LN sums.
//
// The roles hand over through LDS with raw s_barrier (lgkmcnt(0) only: __syncthreads() would also
// drain the prefetch queues).  Per-CU ingest (~80 GB/s) bounds the kernel: 768 KB of weights per block.
//
// Exact precision (HT = sp16_t, round 4): the same kernel with every operand an fp16 hi / lo pair.  The weight stream holds
// each matrix twice -- its lo halves, then its hi halves, both in the fp16 fragment order -- and flows through the SAME 128
// fragment registers as twelve half-matrices: the lo pass starts the accumulators with W_lo . a_hi, the hi pass adds
// W_hi . a_lo and W_hi . a_hi (the three products of common.h's sp16 scheme, smallest terms first); the activation tile
// exists twice (hi and lo halves of the row waves' fp32 values), the K/V caches and the cross K/V are sp16 (16 bytes per
// lane and key), LayerNorm uses 1 / sqrtf like the per-sub-layer kernels it replaces.  3 launches per layer instead of 11.
#pragma once
#include "gemm2.h"

#ifndef DB_ROWS
#define DB_ROWS 4   // rows (= row waves) per block, 16-bit operands: many small blocks spread the K/V and weight streams over more CUs
#endif
#ifndef DB_WIDE_ROWS
#define DB_WIDE_ROWS 8   // rows per block of a wide search (R >= DB_WIDE_R rows), 16-bit operands
#endif
#ifndef DB_WIDE_ROWS_SP
#define DB_WIDE_ROWS_SP 8  // ... exact precision (mixed16 beside the encoder: +1.2-2.3 %, exact +0.3 %)
#endif
#ifndef DB_WIDE_R
#define DB_WIDE_R 512
#endif
#ifndef DB_XCDS
#define DB_XCDS 8   // XCDs whose workgroups work in the block kernel (8 = all: the product)
#endif
#ifndef DB_ROWS_SP
#define DB_ROWS_SP 4  // exact precision: the same (eight rows = two per row wave halve the weight re-streaming: solo search 5.3 -> 6.6 ms, mixed16 beside an encoder +0.3 %; profiles/r04_notes.md)
#endif
// operand traits of the block / FFN kernels: the element type of an MFMA fragment, a lane's four dims of one cached key,
// how many passes the weight stream makes per matrix (sp16: lo, hi) and how many activation tiles there are (sp16: hi, lo)
template <typename HT> struct DbOp {
  typedef HT frag_t;
  typedef cn_h4<HT> kv4_t;
  static constexpr int NPH = 1;
  static constexpr int ROWS = DB_ROWS;
  static constexpr bool kExactLn = false;
  static __device__ __forceinline__ float kvf(const kv4_t& v, int i) { return (float)v[i]; }
};
template <> struct DbOp<sp16_t> {
  typedef half_t frag_t;
  typedef u32x4 kv4_t;
  static constexpr int NPH = 2;
  static constexpr int ROWS = DB_ROWS_SP;
  static constexpr bool kExactLn = true;
  static __device__ __forceinline__ float kvf(const kv4_t& v, int i) { return cn_sp16_value(v[i]); }
};


// Wave-per-row attention: lane l owns dims 4l..4l+3 (head l >> 3); keys in batches of NB with all
// loads of a batch in flight; fp32 online softmax.  kp(s) / vp(s): this lane's 4 bf16 of key / value s.
template <int NB, typename HT> struct DbKV { typename DbOp<HT>::kv4_t k[NB], v[NB]; };

template <int NB, typename HT, class KeyPtr, class ValPtr>
__device__ __forceinline__ void db_kv_load(DbKV<NB, HT>& kv, int s0, int n_keys, KeyPtr kp, ValPtr vp) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int s = min(s0 + u, n_keys - 1);
    kv.k[u] = *(const typename DbOp<HT>::kv4_t*)kp(s);
    kv.v[u] = *(const typename DbOp<HT>::kv4_t*)vp(s);
  }
}

template <int NB, typename HT>
__device__ __forceinline__ void db_kv_consume(const DbKV<NB, HT>& kv, const f32x4& q, int s0, int n_keys, float& m,
                                              float& l, f32x4& o, unsigned long long valid = ~0ull) {
  float sc[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    float d = q[0] * DbOp<HT>::kvf(kv.k[u], 0) + q[1] * DbOp<HT>::kvf(kv.k[u], 1) + q[2] * DbOp<HT>::kvf(kv.k[u], 2) +
              q[3] * DbOp<HT>::kvf(kv.k[u], 3);
    d = cn_sum8_dpp(d);
    sc[u] = (s0 + u < n_keys && ((valid >> ((s0 + u) & 63)) & 1)) ? d : -INFINITY;
  }
  float mb = sc[0];
#pragma unroll
  for (int u = 1; u < NB; ++u) mb = fmaxf(mb, sc[u]);
  const float mn = fmaxf(m, mb);
  const float corr = __expf(m - mn);
  l *= corr;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] *= corr;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const float p = __expf(sc[u] - mn);
    l += p;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(p, DbOp<HT>::kvf(kv.v[u], i), o[i]);
  }
  m = mn;
}

// Rolling K/V pipeline over DEPTH register buffers: db_kv_prefetch issues batches 0 .. DEPTH-1 (early, while
// other work runs), db_kv_attend consumes batch b and re-issues its buffer with batch b + DEPTH.
template <int NB, int DEPTH, typename HT, class KeyPtr, class ValPtr>
__device__ __forceinline__ void db_kv_prefetch(DbKV<NB, HT> (&buf)[DEPTH], int n_keys, KeyPtr kp, ValPtr vp) {
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d * NB < n_keys) db_kv_load(buf[d], d * NB, n_keys, kp, vp);
}
template <int NB, int DEPTH, typename HT, class KeyPtr, class ValPtr>
__device__ __forceinline__ void db_kv_attend(DbKV<NB, HT> (&buf)[DEPTH], const f32x4& q, int n_keys, KeyPtr kp, ValPtr vp,
                                             float& m, float& l, f32x4& o, unsigned long long valid = ~0ull) {
  for (int s0 = 0; s0 < n_keys; s0 += NB * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int sb = s0 + d * NB;
      if (sb < n_keys) {
        db_kv_consume(buf[d], q, sb, n_keys, m, l, o, valid);
        if (sb + NB * DEPTH < n_keys) db_kv_load(buf[d], sb + NB * DEPTH, n_keys, kp, vp);
      }
    }
  }
}

// attention output of row `wave` (lane's 4 dims) -> bf16 -> swizzled A tile row `wave`
// (sp16: the hi halves go to the tile at sA, the lo halves to its twin DbL<HT>::TILE further)
// fp32 rows in LDS (residual, q | k | v, pre-LN, cross q): 256 values at a pitch of 260 -- the GEMM waves' epilogues write the
// same 16-byte column of the block's rows from neighbouring lanes, and at a 1 KB pitch those all hit one bank quad (rocprof
// r03_h / r04_a: 48 % of this kernel's LDS cycles were bank conflicts)
#define DB_RP 260
// NR = rows per block: DbOp<HT>::ROWS unless the launch picks another instantiation (decoder.hip: 8 for the 16-bit operand types at
// R >= 512 rows -- the grouped search of bench.py -- where halving the weight re-streaming pays: profiles/r05_notes.md section 8)
template <typename HT, int NR = DbOp<HT>::ROWS> struct DbL {   // LDS map (dynamic, bytes) of the block kernel for operand type HT
  static constexpr int ROWS = NR, NT = DbOp<HT>::NPH;
  static constexpr int RPW = ROWS / 4;                  // rows per row wave (four row waves)
  static_assert(ROWS % 4 == 0 && ROWS <= 12, "rows per block: a multiple of the four row waves, within one 16-row MFMA tile (+ the zero row)");
  static constexpr int THREADS = 512;                  // 4 GEMM waves + 4 row waves
  static constexpr int TILE = (ROWS + 1) * 512;        // one activation tile: ROWS rows + one zero row, 512 B each
  static constexpr int OFF_A = 0;                      // activation tile(s) (NT = 2 in sp16: hi, lo)
  static constexpr int OFF_X = OFF_A + NT * TILE;      // fp32 residual rows
  static constexpr int OFF_V = OFF_X + ROWS * DB_RP * 4;    // q | k | v rows (fp32); later: pre-LN rows (Y) and cross q (Q)
  static constexpr int OFF_P = OFF_V + 3 * ROWS * DB_RP * 4;  // parameters: bin 768 | bo | bq | bo2 | g1 | b1 | g2 | b2
  static constexpr int BYTES = OFF_P + 2560 * 4;
};
template <typename HT, int NR = DbOp<HT>::ROWS>
__device__ __forceinline__ void db_store_row(char* sA, int wave, int lane, const f32x4& o, float inv) {
  typedef G2Geom<256> G;
  char* p = sA + wave * G::RBY + (((lane >> 1) ^ (wave & G::SWM)) * 16) + (lane & 1) * 8;
  if constexpr (DbOp<HT>::NPH == 2) {
    const unsigned b0 = cn_sp16_bits(o[0] * inv), b1 = cn_sp16_bits(o[1] * inv), b2 = cn_sp16_bits(o[2] * inv), b3 = cn_sp16_bits(o[3] * inv);
    *(uint2*)p = uint2{__builtin_amdgcn_perm(b1, b0, 0x05040100u), __builtin_amdgcn_perm(b3, b2, 0x05040100u)};
    *(uint2*)(p + DbL<HT, NR>::TILE) = uint2{__builtin_amdgcn_perm(b1, b0, 0x07060302u), __builtin_amdgcn_perm(b3, b2, 0x07060302u)};
  } else {
    cn_store4((HT*)p, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  }
}

struct DbPrologue {
  // layer 0: x = emb[tok] * sqrt(d) + pe[step]   (aac_tfmer.py:100-106)
  const int* tok;
  const float* emb;
  const float* pe_row;
  float emb_scale;
  // layer > 0: x = LN3_prev( sum_s slabs[s] + b2_prev + x_prev )   (FFN2 split-K partials of the previous layer)
  const float* slabs;
  int nslab;  // <= 8
  size_t slab_stride;
  const float* b2_prev;
  const float* g3;
  const float* b3;
};

// the six 256 x 256 weight matrices in the order they are consumed, and the per-column parameters
struct DbWeights {
  const void* stream;    // CnLayerW::blk_w (16-bit operands, bf16_t or half_t): in_proj q | k | v rows, self out-proj, cross q-proj, cross out-proj in fragment order
  const float* params;   // CnLayerW::blk_p: bin 768 | bo | bq | bo2 | g1 | b1 | g2 | b2
};

// offsets inside the parameter block (floats)
#define DB_P_BO 768
#define DB_P_BQ 1024
#define DB_P_BO2 1280
#define DB_P_G1 1536
#define DB_P_B1 1792
#define DB_P_G2 2048
#define DB_P_B2 2304

#define DB_SYNC()                                        \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  } while (0)

// fragment (a, ks) of matrix m for this wave: fw[a][ks] = W_m[64w + 16a + (lane & 15)][32ks + 8(lane >> 4) .. +8],
// 1 KB of consecutive bytes per wave instruction in the packed stream.  Uniform base + 32-bit lane offset
// (voff = (wave * 16384 + lane * 8) * 2 bytes) keeps the address in an SGPR pair + one VGPR for all 192 loads.
struct DbStream {
  const char* base;  // packed 16-bit fragments
  unsigned voff;
};
// Written as inline asm: hipcc otherwise keeps a 64-bit VGPR address per 4 KB window (35 pairs) and renames the
// destination registers (+28), which pushed the kernel past the register budget that lets a block start on a CU
// that still runs one of the encoder's GEMM workgroups.  The loads are invisible to hipcc's waitcnt insertion,
// so db_gemm_regs counts them itself.
template <typename F8>
__device__ __forceinline__ void db_frag_load(F8& dst, const DbStream& st, int m, int a, int ks) {
  const char* p = st.base + ((size_t)(m * 16 + (ks >> 1)) * 4096 + (a * 2 + (ks & 1)) * 512) * 2;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(st.voff), "s"(p));
}

// acc[a] (columns n = 64w + 16a + 4(lane >> 4) + j, row m = lane & 15) = A-tile . W[M]^T for this wave's columns;
// each fragment register is re-loaded with the next pass of the stream right behind its MFMA.  Fragments are consumed in
// issue order and every consumed one is re-issued, so exactly 31 younger loads are in flight at each wait (fewer only
// while the last pass drains); the GEMM waves issue no other vector memory instruction.
// P = pass of the stream: 16-bit operands have one pass per matrix (P = matrix); sp16 has two -- P = 2 m: the lo halves of
// W_m against the hi activation tile (starts the sum), P = 2 m + 1: the hi halves against the lo and the hi tile.
template <int P, typename HT, int NR = DbOp<HT>::ROWS>
__device__ __forceinline__ void db_gemm_regs(const DbStream& wlane, cn_h8<typename DbOp<HT>::frag_t> (&fw)[4][8], const char* sA,
                                             int lane, f32x4 (&acc)[4]) {
  typedef G2Geom<256> G;
  typedef typename DbOp<HT>::frag_t FT;
  constexpr int NPH = DbOp<HT>::NPH, LASTP = 6 * NPH - 1;
  constexpr bool kHiPass = NPH == 2 && (P & 1) == 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int arow = lr < NR ? lr : NR;  // padding rows of the M tile all read the zero row
  const int asw = lr < NR ? (lr & G::SWM) : 0;
  if (!kHiPass) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  cn_h8<FT> fa = *(const cn_h8<FT>*)(sA + arow * G::RBY + ((lq ^ asw) * 16));
  cn_h8<FT> fl = fa;
  if (kHiPass) fl = *(const cn_h8<FT>*)(sA + DbL<HT, NR>::TILE + arow * G::RBY + ((lq ^ asw) * 16));
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const cn_h8<FT> fc = fa, fcl = fl;
    if (ks < 7) {
      fa = *(const cn_h8<FT>*)(sA + arow * G::RBY + (((lq + 4 * (ks + 1)) ^ asw) * 16));
      if (kHiPass) fl = *(const cn_h8<FT>*)(sA + DbL<HT, NR>::TILE + arow * G::RBY + (((lq + 4 * (ks + 1)) ^ asw) * 16));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (P < LASTP) {
        asm volatile("s_waitcnt vmcnt(31)" : "+v"(fw[a][ks]));
      } else {
        switch (31 - (ks * 4 + a)) {  // compile-time after unrolling
#define DB_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(fw[a][ks])); break;
          DB_WAIT_CASE(31) DB_WAIT_CASE(30) DB_WAIT_CASE(29) DB_WAIT_CASE(28) DB_WAIT_CASE(27) DB_WAIT_CASE(26)
          DB_WAIT_CASE(25) DB_WAIT_CASE(24) DB_WAIT_CASE(23) DB_WAIT_CASE(22) DB_WAIT_CASE(21) DB_WAIT_CASE(20)
          DB_WAIT_CASE(19) DB_WAIT_CASE(18) DB_WAIT_CASE(17) DB_WAIT_CASE(16) DB_WAIT_CASE(15) DB_WAIT_CASE(14)
          DB_WAIT_CASE(13) DB_WAIT_CASE(12) DB_WAIT_CASE(11) DB_WAIT_CASE(10) DB_WAIT_CASE(9) DB_WAIT_CASE(8)
          DB_WAIT_CASE(7) DB_WAIT_CASE(6) DB_WAIT_CASE(5) DB_WAIT_CASE(4) DB_WAIT_CASE(3) DB_WAIT_CASE(2)
          DB_WAIT_CASE(1) DB_WAIT_CASE(0)
#undef DB_WAIT_CASE
        }
      }
      if (kHiPass) acc[a] = cn_mma16(fw[a][ks], fcl, acc[a]);
      acc[a] = cn_mma16(fw[a][ks], fc, acc[a]);
      if (P < LASTP) db_frag_load(fw[a][ks], wlane, P + 1, a, ks);
    }
  }
}
// matrix M of the layer (0 q, 1 k, 2 v, 3 self out-proj, 4 cross q-proj, 5 cross out-proj) in the operand type's passes
template <int M, typename HT, int NR = DbOp<HT>::ROWS>
__device__ __forceinline__ void db_gemm_mat(const DbStream& wlane, cn_h8<typename DbOp<HT>::frag_t> (&fw)[4][8], const char* sA,
                                            int lane, f32x4 (&acc)[4]) {
  if constexpr (DbOp<HT>::NPH == 2) {
    db_gemm_regs<2 * M, HT, NR>(wlane, fw, sA, lane, acc);
    db_gemm_regs<2 * M + 1, HT, NR>(wlane, fw, sA, lane, acc);
  } else {
    db_gemm_regs<M, HT, NR>(wlane, fw, sA, lane, acc);
  }
}

// row LayerNorm (eps 1e-5) of a lane's 4 columns of one 256-wide row
template <bool EXACT = false>
__device__ __forceinline__ f32x4 db_row_ln(const f32x4& v, const float* g, const float* b, int lane) {
  const float mean = cn_wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s2 = fmaf(v[i] - mean, v[i] - mean, s2);
  const float var = cn_wave_sum_dpp(s2) * (1.0f / 256.0f) + 1e-5f;
  const float rstd = EXACT ? 1.0f / sqrtf(var) : __builtin_amdgcn_rsqf(var);
  const f32x4 gg = *(const f32x4*)(g + 4 * lane), bb = *(const f32x4*)(b + 4 * lane);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
  return r;
}

__device__ unsigned long long g_db_prof[16];
#define DB_STAMP(i)                                            \
  if (dbg) {                                                   \
    const unsigned long long t_ = wall_clock64();              \
    if (lane == 0 && rw == 0) atomicAdd(&g_db_prof[i], t_ - t_prev); \
    t_prev = t_;                                               \
  }

#define DB_NB_SELF 8       // self-attention keys per batch
#define DB_DEPTH_SELF 2    // batches in flight (register buffers): the first two are requested at kernel start
#define DB_NB_CROSS 8      // cross-attention frames per batch
#define DB_DEPTH_CROSS 2   // requested right after the self-attention; the rest roll while the first are consumed

template <typename HT, int NR = DbOp<HT>::ROWS>
__global__ __launch_bounds__((DbL<HT, NR>::THREADS), 1) void cn_dec_block_kernel(
    DbPrologue pro, DbWeights wt,
    HT* __restrict__ kc, HT* __restrict__ vc,    // self K/V cache of this layer [step][R][256]
    const int* __restrict__ anc, int step, int R, int beam, int maxp,
    const HT* __restrict__ kvx, int kv_ld, int kv_off, const int* __restrict__ lens, int Ta,  // cross K/V
    float* __restrict__ x /* in: previous layer's x2 (residual of its FFN), out: x2 */, HT* __restrict__ xt,
    float scale, const unsigned long long* __restrict__ kvalid /* teacher forcing: non-pad caption positions */,
    int dbg, const int* __restrict__ gate /* rows still searching at this step (device counter) or null */) {
  if (gate != nullptr && *gate == 0) return;  // every hypothesis has finished (beam.py:192-194 stops here)
  typedef G2Geom<256> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DbOp<HT>::frag_t FT;
  constexpr int NT = DbOp<HT>::NPH;  // activation tiles (sp16: hi and lo halves)
  constexpr bool kXL = DbOp<HT>::kExactLn;
  char* sA = smem + DbL<HT, NR>::OFF_A;
  float* sX = (float*)(smem + DbL<HT, NR>::OFF_X);
  float* sV = (float*)(smem + DbL<HT, NR>::OFF_V);
  float* sY = sV;                      // pre-LayerNorm rows (after q | k | v are dead)
  float* sQ = sV + NR * DB_RP;      // scaled cross-attention queries
  float* sP = (float*)(smem + DbL<HT, NR>::OFF_P);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // DB_XCDS < 8 (A/B build knob): only the workgroups dealt to the first DB_XCDS XCDs work (the grid is 8 / DB_XCDS times
  // larger, the others return at once), so that fewer L2s pull the layer's 768 KB weight stream
  int bx = blockIdx.x;
  if (DB_XCDS < 8) {
    if ((bx & 7) >= DB_XCDS) return;
    bx = (bx >> 3) * DB_XCDS + (bx & 7);
    if (bx * NR >= R) return;
  }
  const int r0 = bx * NR;

  if (wave < 4) {
    // ======================= GEMM waves ========================================================
    const int lr = lane & 15, lq = lane >> 4;
    const DbStream wlane{(const char*)wt.stream, (unsigned)(wave * 4 * 4096 + lane * 8) * 2u};
    cn_h8<FT> fw[4][8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a) db_frag_load(fw[a][ks], wlane, 0, a, ks);
    f32x4 acc[4];
    DB_SYNC();  // b1: x rows (sA) and parameters (sP) are in LDS
    {           // q | k | v
      db_gemm_mat<0, HT, NR>(wlane, fw, sA, lane, acc);
      if (lr < NR)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = 64 * wave + 16 * a + 4 * lq;
          const f32x4 bb = *(const f32x4*)(sP + n);
          *(f32x4*)(sV + (0 * NR + lr) * DB_RP + n) = f32x4{acc[a][0] + bb[0], acc[a][1] + bb[1], acc[a][2] + bb[2], acc[a][3] + bb[3]};
        }
      db_gemm_mat<1, HT, NR>(wlane, fw, sA, lane, acc);
      if (lr < NR)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = 64 * wave + 16 * a + 4 * lq;
          const f32x4 bb = *(const f32x4*)(sP + 256 + n);
          *(f32x4*)(sV + (1 * NR + lr) * DB_RP + n) = f32x4{acc[a][0] + bb[0], acc[a][1] + bb[1], acc[a][2] + bb[2], acc[a][3] + bb[3]};
        }
      db_gemm_mat<2, HT, NR>(wlane, fw, sA, lane, acc);
      if (lr < NR)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = 64 * wave + 16 * a + 4 * lq;
          const f32x4 bb = *(const f32x4*)(sP + 512 + n);
          *(f32x4*)(sV + (2 * NR + lr) * DB_RP + n) = f32x4{acc[a][0] + bb[0], acc[a][1] + bb[1], acc[a][2] + bb[2], acc[a][3] + bb[3]};
        }
    }
    DB_SYNC();  // b2: q | k | v ready
    DB_SYNC();  // b3: self-attention output in sA
    {
      db_gemm_mat<3, HT, NR>(wlane, fw, sA, lane, acc);
      if (lr < NR)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = 64 * wave + 16 * a + 4 * lq;
          const f32x4 bb = *(const f32x4*)(sP + DB_P_BO + n), rs = *(const f32x4*)(sX + lr * DB_RP + n);
          *(f32x4*)(sY + lr * DB_RP + n) = f32x4{acc[a][0] + bb[0] + rs[0], acc[a][1] + bb[1] + rs[1], acc[a][2] + bb[2] + rs[2], acc[a][3] + bb[3] + rs[3]};
        }
    }
    DB_SYNC();  // b4: pre-LN1 rows ready
    DB_SYNC();  // b5: x1 in sX / sA
    {
      db_gemm_mat<4, HT, NR>(wlane, fw, sA, lane, acc);
      if (lr < NR)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = 64 * wave + 16 * a + 4 * lq;
          const f32x4 bb = *(const f32x4*)(sP + DB_P_BQ + n);
          *(f32x4*)(sQ + lr * DB_RP + n) = f32x4{(acc[a][0] + bb[0]) * scale, (acc[a][1] + bb[1]) * scale, (acc[a][2] + bb[2]) * scale, (acc[a][3] + bb[3]) * scale};
        }
    }
    DB_SYNC();  // b6: cross queries ready
    DB_SYNC();  // b7: cross-attention output in sA
    {
      db_gemm_mat<5, HT, NR>(wlane, fw, sA, lane, acc);
      if (lr < NR)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = 64 * wave + 16 * a + 4 * lq;
          const f32x4 bb = *(const f32x4*)(sP + DB_P_BO2 + n), rs = *(const f32x4*)(sX + lr * DB_RP + n);
          *(f32x4*)(sY + lr * DB_RP + n) = f32x4{acc[a][0] + bb[0] + rs[0], acc[a][1] + bb[1] + rs[1], acc[a][2] + bb[2] + rs[2], acc[a][3] + bb[3] + rs[3]};
        }
    }
    DB_SYNC();  // b8: pre-LN2 rows ready
    return;
  }

  // ========================= row waves ===========================================================
  // Four row waves; wave rw works for the rows rw + 4 h of the block, h < RPW (16-bit operands: RPW = 1, one row per wave;
  // exact precision: RPW = 2, eight rows per block -- twelve waves would cap the kernel at 168 registers, the hi / lo GEMM
  // waves need 212).  Row h = 0 is the one whose K/V batches are requested ahead of the phases that consume them.
  constexpr int ROWS = NR, RPW = DbL<HT, NR>::RPW;
  const int rw = wave - 4;
  const int rt = tid - 256;
  unsigned long long t_prev = dbg ? wall_clock64() : 0ull;
  if (rt < 32 * NT) ((uint4*)(sA + (rt >> 5) * DbL<HT, NR>::TILE + ROWS * 512))[rt & 31] = uint4{0, 0, 0, 0};  // the zero row(s)
  // parameters -> LDS, 10 pieces of 1 KB by LDS-DMA (landed before this wave's younger P0 loads, i.e. before b1)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int piece = rw + 4 * c;
    if (piece < 10)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wt.params + piece * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)((char*)sP + piece * 1024), 16, 0, 0);
  }
  int rr[RPW], tr[RPW], rb[RPW], my_anc[RPW];
  bool live[RPW];
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    rr[h] = rw + 4 * h;                    // row of the block
    tr[h] = min(r0 + rr[h], R - 1);        // the row worked for (clamped: padding rows recompute the last row)
    live[h] = r0 + rr[h] < R;
    rb[h] = (tr[h] / beam) * beam;
    my_anc[h] = lane < step ? anc[(size_t)tr[h] * maxp + lane] : 0;  // step <= 63 (CN_MAX_PRED 64)
  }
  // ---- P0: the rows of the residual stream --------------------------------------------------------------
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    f32x4 xr;
    if (pro.slabs == nullptr) {
      const f32x4 e = *(const f32x4*)(pro.emb + (size_t)pro.tok[tr[h]] * 256 + 4 * lane);
      const f32x4 pe = *(const f32x4*)(pro.pe_row + 4 * lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[i] = e[i] * pro.emb_scale + pe[i];
    } else {
      f32x4 u[8];
      const int ns = pro.nslab;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl)
        u[sl] = sl < ns ? *(const f32x4*)(pro.slabs + sl * pro.slab_stride + (size_t)tr[h] * 256 + 4 * lane)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 rs = *(const f32x4*)(x + (size_t)tr[h] * 256 + 4 * lane);
      const f32x4 bb = *(const f32x4*)(pro.b2_prev + 4 * lane);
      f32x4 v = u[0];
#pragma unroll
      for (int sl = 1; sl < 8; ++sl)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += u[sl][i];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += bb[i] + rs[i];
      xr = db_row_ln<kXL>(v, pro.g3, pro.b3, lane);
    }
    *(f32x4*)(sX + rr[h] * DB_RP + 4 * lane) = xr;
    db_store_row<HT, NR>(sA, rr[h], lane, xr, 1.0f);
  }
  // self-attention K/V of the ancestors (row h = 0): requested now, consumed after the q | k | v GEMMs
  auto skp = [&](int h) { return [&, h](int s) { return kc + ((size_t)s * R + rb[h] + __builtin_amdgcn_readlane(my_anc[h], s)) * 256 + 4 * lane; }; };
  auto svp = [&](int h) { return [&, h](int s) { return vc + ((size_t)s * R + rb[h] + __builtin_amdgcn_readlane(my_anc[h], s)) * 256 + 4 * lane; }; };
  DbKV<DB_NB_SELF, HT> skv[DB_DEPTH_SELF];
  db_kv_prefetch(skv, step, skp(0), svp(0));
  DB_STAMP(0)
  DB_SYNC();  // b1
  DB_SYNC();  // b2: q | k | v ready
  DB_STAMP(1)

  // ---- P1: self-attention ----------------------------------------------------------------------------
  DbKV<DB_NB_CROSS, HT> xkv[DB_DEPTH_CROSS];
  int n_fr[RPW];
  const HT* xbase[RPW];
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    const int clip = tr[h] / beam;
    const int n = lens[clip];
    n_fr[h] = n < 1 ? 1 : (n > Ta ? Ta : n);
    xbase[h] = kvx + (size_t)clip * Ta * kv_ld + kv_off + 4 * lane;
  }
  auto xkp = [&](int h) { return [&, h](int t) { return xbase[h] + (size_t)t * kv_ld; }; };
  auto xvp = [&](int h) { return [&, h](int t) { return xbase[h] + (size_t)t * kv_ld + 256; }; };
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    f32x4 q = *(const f32x4*)(sV + (0 * ROWS + rr[h]) * DB_RP + 4 * lane);
    f32x4 kn = *(const f32x4*)(sV + (1 * ROWS + rr[h]) * DB_RP + 4 * lane);
    f32x4 vn = *(const f32x4*)(sV + (2 * ROWS + rr[h]) * DB_RP + 4 * lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      q[i] *= scale;
      kn[i] = cn_to_f32(cn_from_f32<HT>(kn[i]));  // cache precision
      vn[i] = cn_to_f32(cn_from_f32<HT>(vn[i]));
    }
    if (live[h]) {
      cn_store4(kc + ((size_t)step * R + tr[h]) * 256 + 4 * lane, kn[0], kn[1], kn[2], kn[3]);
      cn_store4(vc + ((size_t)step * R + tr[h]) * 256 + 4 * lane, vn[0], vn[1], vn[2], vn[3]);
    }
    float m = -INFINITY, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long valid = kvalid ? kvalid[tr[h]] : ~0ull;
    if (h > 0) db_kv_prefetch(skv, step, skp(h), svp(h));
    db_kv_attend(skv, q, step, skp(h), svp(h), m, l, o, valid);
    // the audio K/V of the clip (row h = 0): first batches requested here, consumed after out-proj, LN1 and the query GEMM
    if (h == RPW - 1) db_kv_prefetch(xkv, n_fr[0], xkp(0), xvp(0));
    if ((valid >> step) & 1) {  // own key / value
      float d = q[0] * kn[0] + q[1] * kn[1] + q[2] * kn[2] + q[3] * kn[3];
      d = cn_sum8_dpp(d);
      const float mn = fmaxf(m, d);
      const float corr = __expf(m - mn), p = __expf(d - mn);
      l = l * corr + p;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = o[i] * corr + p * vn[i];
    }
    db_store_row<HT, NR>(sA, rr[h], lane, o, 1.0f / l);
  }
  DB_STAMP(2)
  DB_SYNC();  // b3
  DB_SYNC();  // b4: pre-LN1 rows ready
  DB_STAMP(3)

  // ---- LN1 --------------------------------------------------------------------------------------------
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    const f32x4 y = *(const f32x4*)(sY + rr[h] * DB_RP + 4 * lane);
    const f32x4 x1 = db_row_ln<kXL>(y, sP + DB_P_G1, sP + DB_P_B1, lane);
    *(f32x4*)(sX + rr[h] * DB_RP + 4 * lane) = x1;
    db_store_row<HT, NR>(sA, rr[h], lane, x1, 1.0f);
  }
  DB_STAMP(4)
  DB_SYNC();  // b5
  DB_SYNC();  // b6: cross queries ready
  DB_STAMP(5)

  // ---- cross-attention over the clip's audio memory -----------------------------------------------------
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    const f32x4 q = *(const f32x4*)(sQ + rr[h] * DB_RP + 4 * lane);
    float m = -INFINITY, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (h > 0) db_kv_prefetch(xkv, n_fr[h], xkp(h), xvp(h));
    db_kv_attend(xkv, q, n_fr[h], xkp(h), xvp(h), m, l, o);
    db_store_row<HT, NR>(sA, rr[h], lane, o, 1.0f / l);
  }
  DB_STAMP(6)
  DB_SYNC();  // b7
  DB_SYNC();  // b8: pre-LN2 rows ready
  DB_STAMP(7)

  // ---- LN2 -> x, xt ---------------------------------------------------------------------------------------
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    const f32x4 y = *(const f32x4*)(sY + rr[h] * DB_RP + 4 * lane);
    const f32x4 x2 = db_row_ln<kXL>(y, sP + DB_P_G2, sP + DB_P_B2, lane);
    if (live[h]) {
      *(f32x4*)(x + (size_t)tr[h] * 256 + 4 * lane) = x2;
      cn_store4(xt + (size_t)tr[h] * 256 + 4 * lane, x2[0], x2[1], x2[2], x2[3]);
    }
  }
  DB_STAMP(8)
  if (dbg && lane == 0 && rw == 0) atomicAdd(&g_db_prof[9], 1ull);
}

template <typename HT, int NR = DbOp<HT>::ROWS> static inline int cn_dec_block_setup() {
  return cn_configure_lds((const void*)cn_dec_block_kernel<HT, NR>, DbL<HT, NR>::BYTES);
}
