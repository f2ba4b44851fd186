// NT GEMM for gfx950 MFMA:  C[m][n] = sum_k A[m][k] * W[n][k]   (A: activations, W: nn.Linear weight)
//
// Used for the ConvNeXt pointwise 1x1 convs, the 2x2/2 downsample convs (as patch GEMMs), the
// projection, and the decoder QKV / out / FFN / classifier GEMMs (north_star: MFMA only there).
//
// * operand type T = bf16 (v_mfma_f32_16x16x32_bf16) or float (v_mfma_f32_16x16x4_f32, exact
//   fp32 FMA chain -> the fp32 parity mode); accumulation always fp32.
// * "transposed" tile orientation: W rows feed the MFMA A operand and activation rows feed the
//   B operand, so each lane ends up with 4 CONSECUTIVE n for one m -> one 8/16-byte store per
//   lane in the epilogue instead of four scalar stores.
// * 256 threads = 4 waves as 2(M) x 2(N); block tile BM x BN x 32, register-staged double
//   buffered LDS (row pitch padded by 16 B against bank conflicts), one barrier per k-tile.
// * requires K % 32 == 0; M and N arbitrary (clamped loads, guarded stores).
#pragma once
#include "common.h"

template <typename T> struct GemmTraits;
template <> struct GemmTraits<bf16_t> {
  static constexpr int ROW_BYTES = 64 + 16;
  static constexpr int CPR = 4;  // 16-byte chunks per 32-element row
};
template <> struct GemmTraits<float> {
  static constexpr int ROW_BYTES = 128 + 16;
  static constexpr int CPR = 8;
};

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_GELU_FAST = 4, ACT_GELU_AS = 5 };
// GELU of a precision's GEMM epilogues: libm erff in the fp32 parity mode, A&S erfc without cancellation in the "exact"
// (sp16) mode, the A&S form of cn_gelu_fast in bf16
template <typename T> struct CnGeluAct { static constexpr int value = ACT_GELU; };
template <> struct CnGeluAct<bf16_t> { static constexpr int value = ACT_GELU_FAST; };
template <> struct CnGeluAct<half_t> { static constexpr int value = ACT_GELU_FAST; };
template <> struct CnGeluAct<sp16_t> { static constexpr int value = ACT_GELU_AS; };

// out[m][n] = act(acc + bias[n])
// ACT >= 0 fixes the activation at compile time: with the runtime switch every one of the 16 accumulator tiles of a
// 128 x 128 block carried the code of all activations (libm erff included) -- a 12,000-instruction kernel whose
// epilogue streamed ~90 KB through the instruction cache once per block (g2prof: 44 % of the block time).
template <typename TOut, int ACT = -1> struct EpiBiasAct {
  const float* bias;  // may be null
  TOut* out;
  int ldo;
  int act;
  static constexpr bool kBatched = false;
  __device__ __forceinline__ int the_act() const { return ACT >= 0 ? ACT : act; }
  // --- staged interface (gemm2.h): registers -> pre() -> LDS tile -> commit() of 16-byte row chunks
  typedef TOut stage_t;
  __device__ __forceinline__ float pre(int n, float x, int N) const {
    if (bias != nullptr && n < N) x += bias[n];
    if (the_act() == ACT_GELU) x = cn_gelu(x);
    else if (the_act() == ACT_GELU_FAST) x = cn_gelu_fast(x);
    else if (the_act() == ACT_GELU_AS) x = cn_gelu_as(x);
    else if (the_act() == ACT_RELU) x = fmaxf(x, 0.0f);
    else if (the_act() == ACT_SIGMOID) x = 1.0f / (1.0f + __expf(-x));
    return x;
  }
  // 4 consecutive columns at once (packed-math GELU)
  __device__ __forceinline__ f32x4 pre4(int n, f32x4 v, int N) const {
    if (the_act() == ACT_GELU_FAST && n + 3 < N) {
      if (bias != nullptr) {
        const f32x4 b = *(const f32x4*)(bias + n);
        v = f32x4{v[0] + b[0], v[1] + b[1], v[2] + b[2], v[3] + b[3]};
      }
      return cn_gelu_fast4(v);
    }
    return f32x4{pre(n, v[0], N), pre(n + 1, v[1], N), pre(n + 2, v[2], N), pre(n + 3, v[3], N)};
  }
  __device__ __forceinline__ void commit(int m, int n, const TOut* chunk, int N, int /*ks*/) const {
    constexpr int E = 16 / (int)sizeof(TOut);
    TOut* p = out + (size_t)m * ldo + n;
    if (n + E - 1 < N && (ldo % E) == 0) {
      *(uint4*)p = *(const uint4*)chunk;
    } else {
#pragma unroll
      for (int i = 0; i < E; ++i)
        if (n + i < N) p[i] = chunk[i];
    }
  }
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int N, int /*ks*/ = 0) const {
    float r[4];
    if (the_act() == ACT_GELU_AS && n + 3 < N && (ldo & 3) == 0) {  // four columns at once, packed math
      if (bias != nullptr) {
        const f32x4 b = *(const f32x4*)(bias + n);
        v = f32x4{v[0] + b[0], v[1] + b[1], v[2] + b[2], v[3] + b[3]};
      }
      const f32x4 g = cn_gelu_as4(v);
      cn_store4(out + (size_t)m * ldo + n, g[0], g[1], g[2], g[3]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = v[i];
      if (bias != nullptr && n + i < N) x += bias[n + i];
      if (the_act() == ACT_GELU) x = cn_gelu(x);
      else if (the_act() == ACT_GELU_AS) x = cn_gelu_as(x);
      else if (the_act() == ACT_GELU_FAST) x = cn_gelu_fast(x);
      else if (the_act() == ACT_RELU) x = fmaxf(x, 0.0f);
      else if (the_act() == ACT_SIGMOID) x = 1.0f / (1.0f + __expf(-x));
      r[i] = x;
    }
    TOut* p = out + (size_t)m * ldo + n;
    if (n + 3 < N && (ldo & 3) == 0) {
      cn_store4(p, r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n + i < N) p[i] = cn_from_f32<TOut>(r[i]);
    }
  }
};

// out[m][n] = resid[m][n] + scale[n] * (acc + bias[n])      (residual stream of type XT, common.h; in-place ok)
template <typename XT> struct EpiResidT {
  const float* bias;
  const float* scale;  // may be null (== 1)
  const XT* resid;
  XT* out;
  int ld;
  typedef float stage_t;
  __device__ __forceinline__ float pre(int n, float x, int N) const {
    if (n >= N) return 0.f;
    x += bias[n];
    return scale ? scale[n] * x : x;
  }
  __device__ __forceinline__ void commit(int m, int n, const float* chunk, int N, int /*ks*/) const {
    const size_t o = (size_t)m * ld + n;
    if (n + 3 < N && (ld & 3) == 0) {
      const f32x4 rs = cn_ld4(resid + o);
      const f32x4 c = *(const f32x4*)chunk;
      cn_store4(out + o, rs[0] + c[0], rs[1] + c[1], rs[2] + c[2], rs[3] + c[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n + i < N) out[o + i] = cn_from_f32<XT>(cn_ld1(resid + o + i) + chunk[i]);
    }
  }
  // --- batched interface (gemm2.h direct epilogue): every load of a batch of tiles is issued before its first store.
  // `resid` and `out` are the same buffer, so hipcc cannot move a tile's loads above the previous tile's store by
  // itself: 16 tiles = 16 dependent HBM round trips (g2prof: 14 us of a 46 us block).
  static constexpr bool kBatched = true;
  struct Cols {
    f32x4 b, s;
  };
  __device__ __forceinline__ bool fast(int N) const { return (ld & 3) == 0 && (N & 3) == 0; }
  __device__ __forceinline__ Cols cols(int n) const {
    Cols c;
    c.b = *(const f32x4*)(bias + n);
    c.s = scale ? *(const f32x4*)(scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    return c;
  }
  __device__ __forceinline__ f32x4 prefetch(int m, int n) const { return cn_ld4(resid + (size_t)m * ld + n); }
  __device__ __forceinline__ void finish(int m, int n, f32x4 v, const Cols& c, f32x4 rs) const {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = rs[i] + c.s[i] * (v[i] + c.b[i]);
    cn_store4(out + (size_t)m * ld + n, r[0], r[1], r[2], r[3]);
  }
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int N, int /*ks*/ = 0) const {
    const size_t o = (size_t)m * ld + n;
    if (n + 3 < N && (ld & 3) == 0) {
      f32x4 rs = cn_ld4(resid + o);
      f32x4 b = *(const f32x4*)(bias + n);
      f32x4 s = scale ? *(const f32x4*)(scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      f32x4 r;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = rs[i] + s[i] * (v[i] + b[i]);
      cn_store4(out + o, r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n + i < N) out[o + i] = cn_from_f32<XT>(cn_ld1(resid + o + i) + (scale ? scale[n + i] : 1.0f) * (v[i] + bias[n + i]));
    }
  }
};
typedef EpiResidT<float> EpiResid;

template <typename T> struct Frag8;
template <> struct Frag8<bf16_t> { typedef bf16x8 type; };
template <> struct Frag8<float> { struct type { f32x4 lo, hi; }; };

template <typename T> __device__ __forceinline__ typename Frag8<T>::type cn_lds_frag(const char* p);
template <> __device__ __forceinline__ bf16x8 cn_lds_frag<bf16_t>(const char* p) { return *(const bf16x8*)p; }
template <> __device__ __forceinline__ Frag8<float>::type cn_lds_frag<float>(const char* p) {
  Frag8<float>::type f;
  f.lo = *(const f32x4*)p;
  f.hi = *(const f32x4*)(p + 16);
  return f;
}

__device__ __forceinline__ f32x4 cn_mma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 cn_mma(const Frag8<float>::type& a, const Frag8<float>::type& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
  return c;
}

template <typename T, int BM, int BN, class Epi>
__global__ __launch_bounds__(256) void cn_gemm_nt_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W,
                                                        int ldw, int M, int N, int K, Epi epi) {
  constexpr int RB = GemmTraits<T>::ROW_BYTES;
  constexpr int CPR = GemmTraits<T>::CPR;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int FRAG_BYTES = 8 * (int)sizeof(T);
  constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 tiles per wave along M / N
  constexpr int A_CH = BM * CPR, W_CH = BN * CPR;
  constexpr int A_IT = (A_CH + 255) / 256, W_IT = (W_CH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF = (BM + BN) * RB;

  const int n_tiles = (N + BN - 1) / BN;
  const int m0 = (blockIdx.x / n_tiles) * BM;
  const int n0 = (blockIdx.x % n_tiles) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  uint4 ra[A_IT], rw[W_IT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + i * 256;
      if (A_CH % 256 == 0 || idx < A_CH) {
        const int row = idx / CPR, ch = idx % CPR;
        const int gm = min(m0 + row, M - 1);
        ra[i] = *(const uint4*)(A + (size_t)gm * lda + k0 + ch * EPC);
      }
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int idx = tid + i * 256;
      if (W_CH % 256 == 0 || idx < W_CH) {
        const int row = idx / CPR, ch = idx % CPR;
        const int gn = min(n0 + row, N - 1);
        rw[i] = *(const uint4*)(W + (size_t)gn * ldw + k0 + ch * EPC);
      }
    }
  };
  auto swrite = [&](int buf) {
    char* sA = smem + buf * BUF;
    char* sW = sA + BM * RB;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + i * 256;
      if (A_CH % 256 == 0 || idx < A_CH) *(uint4*)(sA + (idx / CPR) * RB + (idx % CPR) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int idx = tid + i * 256;
      if (W_CH % 256 == 0 || idx < W_CH) *(uint4*)(sW + (idx / CPR) * RB + (idx % CPR) * 16) = rw[i];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = K / 32;
  gload(0);
  swrite(0);
  __syncthreads();
  int buf = 0;
  const int frag_off = (lane & 15) * RB + (lane >> 4) * FRAG_BYTES;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) gload((kt + 1) * 32);
    const char* sA = smem + buf * BUF;
    const char* sW = sA + BM * RB;
    typename Frag8<T>::type fw[TN], fa[TM];
#pragma unroll
    for (int a = 0; a < TN; ++a) fw[a] = cn_lds_frag<T>(sW + (wn * (BN / 2) + a * 16) * RB + frag_off);
#pragma unroll
    for (int b = 0; b < TM; ++b) fa[b] = cn_lds_frag<T>(sA + (wm * (BM / 2) + b * 16) * RB + frag_off);
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = cn_mma(fw[a], fa[b], acc[a][b]);
    if (kt + 1 < KT) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = m0 + wm * (BM / 2) + b * 16 + (lane & 15);
      const int n = n0 + wn * (BN / 2) + a * 16 + 4 * (lane >> 4);
      if (m < M && n < N) epi(m, n, acc[a][b], N);
    }
}

template <typename T, int BM, int BN, class Epi>
static int cn_launch_gemm_t(const T* A, int lda, const T* W, int ldw, int M, int N, int K, const Epi& epi,
                            hipStream_t stream) {
  constexpr int SMEM = 2 * (BM + BN) * GemmTraits<T>::ROW_BYTES;
  CN_TRY(cn_configure_lds((const void*)cn_gemm_nt_kernel<T, BM, BN, Epi>, SMEM));
  const long blocks = (long)cn_cdiv(M, BM) * cn_cdiv(N, BN);
  hipLaunchKernelGGL((cn_gemm_nt_kernel<T, BM, BN, Epi>), dim3((unsigned)blocks), dim3(256), SMEM, stream, A, lda,
                     W, ldw, M, N, K, epi);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// Tile choice: BN = 96 when N is a multiple of 96 but not of 128 (ConvNeXt dims 96 / 192), else 128;
// BM = 128 for the big encoder GEMMs, 64 for the skinny decoder ones (more blocks in flight).
template <typename T, class Epi>
static int cn_gemm(const T* A, int lda, const T* W, int ldw, int M, int N, int K, const Epi& epi,
                   hipStream_t stream) {
  if (K % 32 != 0 || M <= 0 || N <= 0) {
    cn_set_error("cn_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    return CN_ERR_ARG;
  }
  const bool n96 = (N % 96 == 0) && (N % 128 != 0);
  const bool big = M >= 4096;
  if (big) {
    if (n96) return cn_launch_gemm_t<T, 128, 96, Epi>(A, lda, W, ldw, M, N, K, epi, stream);
    return cn_launch_gemm_t<T, 128, 128, Epi>(A, lda, W, ldw, M, N, K, epi, stream);
  }
  if (n96) return cn_launch_gemm_t<T, 64, 96, Epi>(A, lda, W, ldw, M, N, K, epi, stream);
  return cn_launch_gemm_t<T, 64, 128, Epi>(A, lda, W, ldw, M, N, K, epi, stream);
}
