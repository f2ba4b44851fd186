// Fused decoder feed-forward for one new token per row (16-bit operands HT = bf16_t | half_t, d_model 256, d_ff = 256 * NCH):
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
#pragma once
#include "dec_block.h"

#define DF_ROWS 32
#define DF_LDS_BYTES (2 * DF_ROWS * 512 + 1024)

template <typename HT>
__global__ __launch_bounds__(256, 1) void cn_dec_ffn_kernel(const HT* __restrict__ xt, int R,
                                                            const HT* __restrict__ stream /* [chunk][2][...] */,
                                                            const float* __restrict__ b1, float* __restrict__ slabs,
                                                            size_t slab_stride, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;  // nothing left to decode at this step
  typedef G2Geom<256> G;
  __shared__ __attribute__((aligned(16))) char smem[DF_LDS_BYTES];
  char* sX = smem;                       // x tile, 32 rows x 512 B, chunk-swizzled
  char* sH = smem + DF_ROWS * 512;       // hidden tile (this chunk's 256 columns), same layout
  float* sB = (float*)(smem + 2 * DF_ROWS * 512);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int r0 = blockIdx.x * DF_ROWS, chunk = blockIdx.y;

  // x tile: 16 pieces of 1 KB (2 rows each), 4 per wave; destination chunk cp of row r holds source chunk cp ^ (r & 15)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int row = 2 * piece + (lane >> 5), cp = lane & 31;
    const HT* src = xt + (size_t)min(r0 + row, R - 1) * 256 + ((cp ^ (row & G::SWM)) * 8);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sX + piece * 1024), 16, 0, 0);
  }
  if (wave == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1 + chunk * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)sB, 16, 0, 0);
  const DbStream wl{(const char*)(stream + (size_t)chunk * 2 * 16 * 4096), (unsigned)(wave * 4 * 4096 + lane * 8) * 2u};
  cn_h8<HT> fw[4][8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int a = 0; a < 4; ++a) db_frag_load(fw[a][ks], wl, 0, a, ks);
  asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // the tile pieces (older than the 32 fragment loads) have landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x4 acc[4][2];
  auto gemm = [&](const char* sT, bool refill) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cpos = ((lq + 4 * ks) ^ (lr & G::SWM)) * 16;
      const cn_h8<HT> f0 = *(const cn_h8<HT>*)(sT + lr * G::RBY + cpos);
      const cn_h8<HT> f1 = *(const cn_h8<HT>*)(sT + (16 + lr) * G::RBY + cpos);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (refill) {
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
        acc[a][0] = cn_mma16(fw[a][ks], f0, acc[a][0]);
        acc[a][1] = cn_mma16(fw[a][ks], f1, acc[a][1]);
        if (refill) db_frag_load(fw[a][ks], wl, 1, a, ks);
      }
    }
  };

  // ---- GEMM 1: hidden chunk = GELU(x W1[chunk]^T + b1[chunk]) -> sH (bf16) ---------------------------
  gemm(sX, true);
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
      v = cn_gelu_fast4(v);
      cn_store4((HT*)(sH + row * G::RBY + (((n >> 3) ^ (row & G::SWM)) * 16) + ((n >> 2) & 1) * 8), v[0], v[1], v[2],
                v[3]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // raw: keeps the W2 tile's loads in flight
  asm volatile("" ::: "memory");

  // ---- GEMM 2: slab[chunk] = hidden chunk . W2[:, chunk]^T --------------------------------------------
  gemm(sH, false);
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
