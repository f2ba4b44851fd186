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
//    summed in a fixed order by the consumer -> deterministic, no atomics).
#pragma once
#include <stdlib.h>

#include "gemm.h"

// out[ks][m][n] = acc  (partial sums of one K slice; bias / residual are added by the consumer)
struct EpiSlab {
  float* out;
  int ld;
  size_t slab_stride;
  typedef float stage_t;
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

template <int BM, int BN, int BK, class Epi>
__global__ __launch_bounds__(256) void cn_gemm2_kernel(const bf16_t* __restrict__ A, int lda,
                                                       const bf16_t* __restrict__ W, int ldw, int M, int N, int K,
                                                       int k_slice, Epi epi) {
  constexpr int RBY = BK * 2;          // bytes per tile row
  constexpr int CPR = RBY / 16;        // 16-byte chunks per row (4 or 8)
  constexpr int RPB = RBY >= 256 ? 1 : 256 / RBY;  // tile rows per 256-byte LDS bank row (4, 2 or 1)
  constexpr int SWM = CPR > 16 ? 15 : CPR - 1;       // swizzle mask: XOR the chunk index with (row / RPB) & SWM
  constexpr int RPI = 1024 / RBY;      // tile rows written by one DMA wave-instruction (16, 8 or 2)
  constexpr int A_BYTES = BM * RBY, W_BYTES = BN * RBY, BUF = A_BYTES + W_BYTES;
  constexpr int N_DMA = BUF / 1024;
  static_assert(BUF % 1024 == 0 && A_BYTES % 1024 == 0, "tile must be a whole number of 1 KiB DMA pieces");
  constexpr int DPW = (N_DMA + 3) / 4;  // DMA instructions per wave and stage
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int KS = BK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_tiles = (N + BN - 1) / BN;
  const int bid = cn_xcd_remap(blockIdx.x, gridDim.x);  // the n-tiles of one A panel share an XCD / L2
  const int m0 = (bid / n_tiles) * BM;
  const int n0 = (bid % n_tiles) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int inst = wave * DPW + i;
      if (N_DMA % 4 == 0 || inst < N_DMA)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * RBY),
                                         (__attribute__((address_space(3))) void*)(smem + buf * BUF + inst * 1024), 16, 0,
                                         0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15;
  const int sw = (lr / RPB) & SWM;
  const int row_off = lr * RBY;

  constexpr int NBUF = BK >= 256 ? 1 : 2;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    if (NBUF == 2 && kt + 1 < KT) stage(buf ^ 1, kt + 1);
    const char* sA = smem + buf * BUF;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int coff = (((lane >> 4) + 4 * ks) ^ sw) * 16;
      bf16x8 fw[TN], fa[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) fw[a] = *(const bf16x8*)(sW + (wn * (BN / 2) + a * 16) * RBY + row_off + coff);
#pragma unroll
      for (int b = 0; b < TM; ++b) fa[b] = *(const bf16x8*)(sA + (wm * (BM / 2) + b * 16) * RBY + row_off + coff);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[a], fa[b], acc[a][b], 0, 0, 0);
    }
    if (NBUF == 1 && kt + 1 < KT) {  // single stage: refill only after every wave has read it
      __syncthreads();
      stage(0, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NBUF == 2) buf ^= 1;
  }

  // ---- staged epilogue: registers -> LDS tile [BM][BN] (output type) -> whole-row 16-byte stores.
  // Direct stores from the MFMA layout are 8/16 bytes per lane scattered over 16 rows, which cost
  // up to 2.2x write amplification at HBM (rocprof WRITE_SIZE); through LDS every wave-instruction
  // stores contiguous 16-byte chunks of a few rows.  (The loop above ended on a barrier.)
  typedef typename Epi::stage_t ST;
  if constexpr (sizeof(ST) == 4) {
    // fp32 outputs already store 16 bytes per lane (64 contiguous bytes per row and instruction):
    // staging a 64 KB fp32 tile only costs occupancy (measured: pw2 +35 %), so store directly
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (BM / 2) + b * 16 + (lane & 15);
        const int n = n0 + wn * (BN / 2) + a * 16 + 4 * (lane >> 4);
        if (m < M && n < N) epi(m, n, acc[a][b], N, (int)blockIdx.y);
      }
    return;
  }
  constexpr int EPC = 16 / (int)sizeof(ST);          // elements per 16-byte chunk
  constexpr int PITCH = BN + EPC;                     // padded row pitch (elements)
  constexpr int CH = BN / EPC;                        // chunks per row
  ST* tile = (ST*)smem;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int ml = wm * (BM / 2) + b * 16 + (lane & 15);
      const int nl = wn * (BN / 2) + a * 16 + 4 * (lane >> 4);
      const f32x4 v = acc[a][b];
      cn_store4(tile + ml * PITCH + nl, epi.pre(n0 + nl, v[0], N), epi.pre(n0 + nl + 1, v[1], N),
                epi.pre(n0 + nl + 2, v[2], N), epi.pre(n0 + nl + 3, v[3], N));
    }
  __syncthreads();
  for (int idx = tid; idx < BM * CH; idx += 256) {
    const int r = idx / CH, c = idx % CH;
    const int m = m0 + r, n = n0 + c * EPC;
    if (m < M && n < N) epi.commit(m, n, tile + r * PITCH + c * EPC, N, (int)blockIdx.y);
  }
}

template <int BM, int BN, int BK, class Epi>
static int cn_launch_gemm2_t(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, int splits,
                             const Epi& epi, hipStream_t stream) {
  constexpr int EPI_BYTES = sizeof(typename Epi::stage_t) == 4 ? 0 : BM * (BN * 2 + 16);
  constexpr int PIPE_BYTES = (BK >= 256 ? 1 : 2) * (BM + BN) * BK * 2;  // BK = 256: single stage per slice
  constexpr int SMEM = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  static bool configured = false;
  if (!configured) {
    CN_HIP(hipFuncSetAttribute((const void*)cn_gemm2_kernel<BM, BN, BK, Epi>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    configured = true;
  }
  const long blocks = (long)cn_cdiv(M, BM) * cn_cdiv(N, BN);
  const int k_slice = K / splits;
  hipLaunchKernelGGL((cn_gemm2_kernel<BM, BN, BK, Epi>), dim3((unsigned)blocks, (unsigned)splits), dim3(256), SMEM,
                     stream, A, lda, W, ldw, M, N, K, k_slice, epi);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// bf16 dispatch.  splits > 1 only with a slab epilogue; K / splits must be a multiple of 64.
template <class Epi>
static int cn_gemm2(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const Epi& epi,
                    hipStream_t stream, int splits = 1) {
  if (K % 32 != 0 || M <= 0 || N <= 0 || splits < 1 || (K / splits) % 64 != 0 && splits > 1) {
    cn_set_error("cn_gemm2: bad shape M=%d N=%d K=%d splits=%d", M, N, K, splits);
    return CN_ERR_ARG;
  }
  const bool k64 = (K % 64 == 0);
  if (M >= 4096) {
    const bool n96 = (N % 96 == 0) && (N % 128 != 0);
    if (!k64 || getenv("CN_GEMM_BK32")) return cn_launch_gemm2_t<128, 128, 32, Epi>(A, lda, W, ldw, M, N, K, splits, epi, stream);
    if (n96) return cn_launch_gemm2_t<128, 96, 64, Epi>(A, lda, W, ldw, M, N, K, splits, epi, stream);
    return cn_launch_gemm2_t<128, 128, 64, Epi>(A, lda, W, ldw, M, N, K, splits, epi, stream);
  }
  if (!k64) return cn_launch_gemm2_t<64, 64, 32, Epi>(A, lda, W, ldw, M, N, K, splits, epi, stream);
  if ((K / splits) % 256 == 0) return cn_launch_gemm2_t<64, 64, 256, Epi>(A, lda, W, ldw, M, N, K, splits, epi, stream);
  return cn_launch_gemm2_t<64, 64, 64, Epi>(A, lda, W, ldw, M, N, K, splits, epi, stream);
}

// type-generic front end: bf16 -> v2, fp32 -> gemm.h
template <class Epi>
static int cn_mm(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const Epi& epi,
                 hipStream_t stream) {
  return cn_gemm2(A, lda, W, ldw, M, N, K, epi, stream);
}
template <class Epi>
static int cn_mm(const float* A, int lda, const float* W, int ldw, int M, int N, int K, const Epi& epi,
                 hipStream_t stream) {
  return cn_gemm<float>(A, lda, W, ldw, M, N, K, epi, stream);
}
