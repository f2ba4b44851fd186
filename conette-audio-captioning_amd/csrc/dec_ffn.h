// Fused decoder feed-forward for one new token per row (operands HT = bf16_t | half_t | sp16_t, d_model 256, d_ff = 256 * NCH):
//
//   slab[c] = GELU(x W1[c]^T + b1[c]) W2[:, c]^T        c = hidden chunk of 256 columns
//
// i.e. linear2(gelu(linear1(x))) of torch's TransformerDecoderLayer (aac_tfmer.py:46-58) as
// split-K partial sums; bias b2, the residual and LayerNorm 3 are applied by the consumer (the next
// layer's block-kernel prologue, or cn_ln256_kernel in front of the classifier).  Replaces two
// launches (FFN1 GEMM + GELU -> HBM, FFN2 split-K) of ~6.5 us each by one of about that length:
// the hidden activations never leave the CU.
//
// Block = 32 rows x one hidden chunk; 4 waves, wave w owns 64 of the 256 columns of both GEMMs
// (hidden columns in GEMM 1, output columns in GEMM 2).  The two 256 x 256 weight tiles stream
// through 128 VGPRs exactly like dec_block.h's GEMM waves: fragment-ordered copy in HBM
// (pk_ffn_stream, api.hip; 1 KB of consecutive bytes per load), each fragment re-loaded with the W2
// tile right behind the MFMA that consumed the W1 tile, hand-counted vmcnt.  Activations are the MFMA
// B operand from swizzled LDS tiles (x tile by LDS-DMA with the swizzle on the source address).
//
// Exact precision (HT = sp16_t, round 4): the stream of a chunk is four half-tiles -- W1 lo, W1 hi, W2 lo, W2 hi, each in the
// fp16 fragment order -- through the same 128 registers; the x and hidden tiles exist twice (hi and lo halves); a pass over
// a lo half-tile starts the sum with W_lo . a_hi, the hi pass adds W_hi . a_lo and W_hi . a_hi (dec_block.h); the GELU is the
// exact precision's A&S erfc form (common.h cn_gelu_as2), as in the GEMM epilogue it replaces.
#pragma once
#include "dec_block.h"

#define DF_ROWS 32
#define DF_TILE_BYTES (DF_ROWS * 512)
#define DF_LDS_BYTES_T(NT) (2 * (NT) * DF_TILE_BYTES + 1024)
#define DF_LDS_BYTES DF_LDS_BYTES_T(1)

template <typename HT>
__global__ __launch_bounds__(256, 1) void cn_dec_ffn_kernel(const HT* __restrict__ xt, int R,
                                                            const void* __restrict__ stream /* [chunk][2 * NPH][...] */,
                                                            const float* __restrict__ b1, float* __restrict__ slabs,
                                                            size_t slab_stride, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;  // nothing left to decode at this step
  typedef G2Geom<256> G;
  typedef typename DbOp<HT>::frag_t FT;
  constexpr int NPH = DbOp<HT>::NPH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;                                  // x tile(s), 32 rows x 512 B, chunk-swizzled (sp16: hi tile, lo tile)
  char* sH = smem + NPH * DF_TILE_BYTES;            // hidden tile(s) (this chunk's 256 columns), same layout
  float* sB = (float*)(smem + 2 * NPH * DF_TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  // blockIdx.x = hidden chunk (the fast index): workgroups are dealt round-robin over the 8 XCDs, so with 8 chunks every block
  // of a chunk lands on the same XCD and the chunk's 256 KB of weights are pulled into ONE L2 instead of one per row tile
  // (speed only: rocprof r04_m counted 12.8 MB fetched per launch for 2 MB of weights -- six row tiles of a chunk on six
  // XCDs -- and the decode's fetch traffic is what it costs an encoder running beside it: profiles/r04_notes.md)
  const int r0 = blockIdx.y * DF_ROWS, chunk = blockIdx.x;

  u32x4 xs[8];  // sp16: the x tile goes through registers (16 bytes = 4 elements per load), split into the hi / lo tiles below
  if constexpr (NPH == 2) {
    // unit u = tid + 256 i: row u >> 6, elements 4 (u & 63) .. + 3.  Through inline asm: the fragment loads below are
    // invisible to hipcc's wait counting, so these are counted by hand too (vmcnt(32) once the 32 fragments are issued)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int u = tid + 256 * i, row = u >> 6;
      const sp16_t* src = xt + (size_t)min(r0 + row, R - 1) * 256 + 4 * (u & 63);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xs[i]) : "v"(src));
    }
  } else {
    // x tile: 16 pieces of 1 KB (2 rows each), 4 per wave; destination chunk cp of row r holds source chunk cp ^ (r & 15)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;
      const int row = 2 * piece + (lane >> 5), cp = lane & 31;
      const HT* src = xt + (size_t)min(r0 + row, R - 1) * 256 + ((cp ^ (row & G::SWM)) * 8);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sX + piece * 1024), 16, 0, 0);
    }
  }
  if (wave == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1 + chunk * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)sB, 16, 0, 0);
  const DbStream wl{(const char*)stream + (size_t)chunk * 2 * NPH * 16 * 4096 * 2, (unsigned)(wave * 4 * 4096 + lane * 8) * 2u};
  cn_h8<FT> fw[4][8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int a = 0; a < 4; ++a) db_frag_load(fw[a][ks], wl, 0, a, ks);
  if constexpr (NPH == 2) {
    asm volatile("s_waitcnt vmcnt(32)"
                 : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xs[4]), "+v"(xs[5]), "+v"(xs[6]), "+v"(xs[7])::"memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int u = tid + 256 * i, row = u >> 6, q = u & 63;
      char* p = sX + row * G::RBY + (((q >> 1) ^ (row & G::SWM)) * 16) + (q & 1) * 8;
      *(uint2*)p = uint2{__builtin_amdgcn_perm(xs[i][1], xs[i][0], 0x05040100u), __builtin_amdgcn_perm(xs[i][3], xs[i][2], 0x05040100u)};
      *(uint2*)(p + DF_TILE_BYTES) = uint2{__builtin_amdgcn_perm(xs[i][1], xs[i][0], 0x07060302u), __builtin_amdgcn_perm(xs[i][3], xs[i][2], 0x07060302u)};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // the tile pieces (older than the 32 fragment loads) have landed
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x4 acc[4][2];
  // one pass of the stream over an activation tile; PASS = index of the half-tile in the chunk's stream, LAST = it drains
  auto gemm = [&](const char* sT, auto pass_c, auto last_c) {
    constexpr int PASS = decltype(pass_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr bool kHiPass = NPH == 2 && (PASS & 1) == 1;
    if (!kHiPass) {
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cpos = ((lq + 4 * ks) ^ (lr & G::SWM)) * 16;
      const cn_h8<FT> f0 = *(const cn_h8<FT>*)(sT + lr * G::RBY + cpos);
      const cn_h8<FT> f1 = *(const cn_h8<FT>*)(sT + (16 + lr) * G::RBY + cpos);
      cn_h8<FT> l0 = f0, l1 = f1;
      if (kHiPass) {
        l0 = *(const cn_h8<FT>*)(sT + DF_TILE_BYTES + lr * G::RBY + cpos);
        l1 = *(const cn_h8<FT>*)(sT + DF_TILE_BYTES + (16 + lr) * G::RBY + cpos);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (!LAST) {
          asm volatile("s_waitcnt vmcnt(31)" : "+v"(fw[a][ks]));
        } else {
          switch (31 - (ks * 4 + a)) {  // compile-time after unrolling: the last tile drains
#define DF_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(fw[a][ks])); break;
            DF_WAIT_CASE(31) DF_WAIT_CASE(30) DF_WAIT_CASE(29) DF_WAIT_CASE(28) DF_WAIT_CASE(27) DF_WAIT_CASE(26)
            DF_WAIT_CASE(25) DF_WAIT_CASE(24) DF_WAIT_CASE(23) DF_WAIT_CASE(22) DF_WAIT_CASE(21) DF_WAIT_CASE(20)
            DF_WAIT_CASE(19) DF_WAIT_CASE(18) DF_WAIT_CASE(17) DF_WAIT_CASE(16) DF_WAIT_CASE(15) DF_WAIT_CASE(14)
            DF_WAIT_CASE(13) DF_WAIT_CASE(12) DF_WAIT_CASE(11) DF_WAIT_CASE(10) DF_WAIT_CASE(9) DF_WAIT_CASE(8)
            DF_WAIT_CASE(7) DF_WAIT_CASE(6) DF_WAIT_CASE(5) DF_WAIT_CASE(4) DF_WAIT_CASE(3) DF_WAIT_CASE(2)
            DF_WAIT_CASE(1) DF_WAIT_CASE(0)
#undef DF_WAIT_CASE
          }
        }
        if (kHiPass) {
          acc[a][0] = cn_mma16(fw[a][ks], l0, acc[a][0]);
          acc[a][1] = cn_mma16(fw[a][ks], l1, acc[a][1]);
        }
        acc[a][0] = cn_mma16(fw[a][ks], f0, acc[a][0]);
        acc[a][1] = cn_mma16(fw[a][ks], f1, acc[a][1]);
        if (!LAST) db_frag_load(fw[a][ks], wl, PASS + 1, a, ks);
      }
    }
  };
  typedef std::integral_constant<bool, false> no_t;
  typedef std::integral_constant<bool, true> yes_t;

  // ---- GEMM 1: hidden chunk = GELU(x W1[chunk]^T + b1[chunk]) -> sH ------------------------------------
  gemm(sX, std::integral_constant<int, 0>{}, no_t{});
  if constexpr (NPH == 2) gemm(sX, std::integral_constant<int, 1>{}, no_t{});
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = 64 * wave + 16 * a + 4 * lq;  // hidden column within the chunk
    const f32x4 bb = *(const f32x4*)(sB + n);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int row = 16 * ms + lr;
      f32x4 v = acc[a][ms];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += bb[j];
      char* p = sH + row * G::RBY + (((n >> 3) ^ (row & G::SWM)) * 16) + ((n >> 2) & 1) * 8;
      if constexpr (NPH == 2) {
        v = cn_gelu_as4(v);
        const unsigned b0 = cn_sp16_bits(v[0]), b1_ = cn_sp16_bits(v[1]), b2 = cn_sp16_bits(v[2]), b3 = cn_sp16_bits(v[3]);
        *(uint2*)p = uint2{__builtin_amdgcn_perm(b1_, b0, 0x05040100u), __builtin_amdgcn_perm(b3, b2, 0x05040100u)};
        *(uint2*)(p + DF_TILE_BYTES) = uint2{__builtin_amdgcn_perm(b1_, b0, 0x07060302u), __builtin_amdgcn_perm(b3, b2, 0x07060302u)};
      } else {
        v = cn_gelu_fast4(v);
        cn_store4((HT*)p, v[0], v[1], v[2], v[3]);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // raw: keeps the W2 tile's loads in flight
  asm volatile("" ::: "memory");

  // ---- GEMM 2: slab[chunk] = hidden chunk . W2[:, chunk]^T --------------------------------------------
  if constexpr (NPH == 2) {
    gemm(sH, std::integral_constant<int, 2>{}, no_t{});
    gemm(sH, std::integral_constant<int, 3>{}, yes_t{});
  } else {
    gemm(sH, std::integral_constant<int, 1>{}, yes_t{});
  }
  float* out = slabs + (size_t)chunk * slab_stride;
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    const int row = r0 + 16 * ms + lr;
    if (row < R) {
#pragma unroll
      for (int a = 0; a < 4; ++a) *(f32x4*)(out + (size_t)row * 256 + 64 * wave + 16 * a + 4 * lq) = acc[a][ms];
    }
  }
}

template <typename HT> static inline int cn_dec_ffn_setup() {
  return cn_configure_lds((const void*)cn_dec_ffn_kernel<HT>, DF_LDS_BYTES_T(DbOp<HT>::NPH));
}
