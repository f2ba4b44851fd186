// Shared device/host helpers for the conette gfx950 library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <set>
#include <utility>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CN_F32X16

// ---- the two 16-bit operand types of the MFMA paths ------------------------------------------------------------------
// bf16_t (CONETTE_PREC_BF16: 8 significant bits, fp32 range) and half_t (CONETTE_PREC_F16: IEEE fp16, 11 significant bits,
// |x| <= 65504).  v_mfma_f32_*_bf16 and v_mfma_f32_*_f16 take the same cycles (MI355X guide), fragments and packed streams
// have the same bytes, so every kernel is written once over H and instantiated for both; an fp16 operand carries an eighth
// of the rounding error of a bf16 one.  Conversions to half_t saturate (cn_from_f32 / cn_sat8) instead of producing inf:
// post-LayerNorm activations, GELU outputs, attention outputs and weights of this model stay orders of magnitude below 65504.
typedef _Float16 half_t;
template <typename H> using cn_h8 = H __attribute__((ext_vector_type(8)));
template <typename H> using cn_h4 = H __attribute__((ext_vector_type(4)));
typedef _Float16 cn_h2 __attribute__((ext_vector_type(2)));   // one register of two fp16 values (v_dot2_f32_f16 operands)
template <typename T> struct CnIsH16 { static constexpr bool value = false; };
template <> struct CnIsH16<bf16_t> { static constexpr bool value = true; };
template <> struct CnIsH16<half_t> { static constexpr bool value = true; };
__device__ __forceinline__ f32x16 cn_mma32(cn_h8<bf16_t> a, cn_h8<bf16_t> b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 cn_mma32(cn_h8<half_t> a, cn_h8<half_t> b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 cn_mma16(cn_h8<bf16_t> a, cn_h8<bf16_t> b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 cn_mma16(cn_h8<half_t> a, cn_h8<half_t> b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// eight converted values that are known to be >= -65504 (GELU / ReLU outputs): +inf -> 65504 with four v_pk_min_f16
template <typename H> __device__ __forceinline__ cn_h8<H> cn_sat8(cn_h8<H> v) {
#ifndef CN_NO_SAT8  // (A/B builds only: what the saturation costs the fused MLP kernels)
  if constexpr (__is_same(H, half_t)) {
    cn_h8<H> m;
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = (H)65504.0f;
    return __builtin_elementwise_min(v, m);
  }
#endif
  return v;
}
// eight fp32 values -> one 16-byte fragment (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
template <typename H> __device__ __forceinline__ cn_h8<H> cn_pack8(float a, float b, float c, float d, float e, float f, float g, float h) {
  return cn_h8<H>{(H)a, (H)b, (H)c, (H)d, (H)e, (H)f, (H)g, (H)h};
}

// ---- LDS-DMA (global -> LDS, 16 bytes per lane, 1 KB per wave instruction) through inline asm ------------------------
// Not __builtin_amdgcn_global_load_lds: while a builtin piece is pending, the compiler's wait-count pass treats every later
// LDS read as possibly out of order and emits `s_waitcnt lgkmcnt(0)` -- a full drain, the read issued one instruction
// earlier included -- wherever a counted wait was meant (seen in every MFMA loop that reads fragments while a ring is
// being filled).  The compiler does not count these pieces in vmcnt: every kernel that uses them waits for its pieces
// with hand-written `s_waitcnt vmcnt(n)`, and a compiler-generated wait that does not know about pieces in flight only
// waits for more than it needs (the counter retires in order), never for less.
__device__ __forceinline__ unsigned cn_lds_addr(const void* p) {  // LDS byte address of a pointer into __shared__ memory
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// lane-private 64-bit source address; lds = wave-uniform LDS byte address of the 1 KB piece
__device__ __forceinline__ void cn_dma16_v(const void* src_lane, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds) : "memory", "m0");
}
// wave-uniform source base + 32-bit lane offset
__device__ __forceinline__ void cn_dma16_s(const void* src_base, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src_base), "s"(lds) : "memory", "m0");
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// ---- sp16: the operand type of the "exact" precision (CONETTE_PREC_F16X2) -------------------------------------------
// An fp32 value stored as the unevaluated sum of two fp16 numbers, hi = rn16(x), lo = rn16(x - hi): 22 significant bits
// (fp32 has 24), 4 bytes per element with hi in the low half of the dword.  A product of two such operands is taken with
// three fp16 MFMAs -- hi.hi + hi.lo + lo.hi, fp32 accumulation; the dropped lo.lo term is below 2^-22 of the product --
// so a GEMM in this type costs three bf16-rate MFMAs per k-step where the fp32 MFMA costs sixteen times one, and its error
// against an fp64 product is that of the fp32 MFMA path (tools/lab/split_probe.hip: mean |err| 1.1e-6 for both at K = 1536;
// the bf16 pair split is 6.5x worse; v_mfma_f32_16x16x32_f16 keeps subnormal inputs, so a small weight's lo part counts).
// Values beyond the fp16 range (65504) are clamped: post-LayerNorm activations, GELU outputs and weights stay far below.
struct sp16_t {
  _Float16 hi, lo;
};
__device__ __forceinline__ unsigned cn_sp16_bits(float x) {
  x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
  const _Float16 hi = (_Float16)x;
  const _Float16 lo = (_Float16)(x - (float)hi);
  return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}
__device__ __forceinline__ float cn_sp16_value(unsigned w) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
}

#define CN_OK 0
#define CN_ERR_ARG 1
#define CN_ERR_HIP 2
#define CN_ERR_WEIGHT 3
#define CN_ERR_WORKSPACE 4

void cn_set_error(const char* fmt, ...);

#define CN_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      cn_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
      return CN_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

#define CN_LAUNCH_CHECK()                                                                 \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) {                                                               \
      cn_set_error("%s:%d launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e));      \
      return CN_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

#define CN_TRY(expr)                 \
  do {                               \
    int _s = (expr);                 \
    if (_s != CN_OK) return _s;      \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remembered per (device, kernel), so a second
// context on another GPU of the same process configures its own device (was a process-wide flag: ADVICE r01)
static inline int cn_configure_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  CN_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(std::make_pair(dev, fn))) return CN_OK;
  CN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert(std::make_pair(dev, fn));
  return CN_OK;
}

static inline size_t cn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int cn_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- element conversion -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T cn_from_f32(float x);
template <> __device__ __forceinline__ float cn_from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t cn_from_f32<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ half_t cn_from_f32<half_t>(float x) { return (half_t)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
template <> __device__ __forceinline__ sp16_t cn_from_f32<sp16_t>(float x) { return __builtin_bit_cast(sp16_t, cn_sp16_bits(x)); }
__device__ __forceinline__ float cn_to_f32(float x) { return x; }
__device__ __forceinline__ float cn_to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float cn_to_f32(half_t x) { return (float)x; }
__device__ __forceinline__ float cn_to_f32(sp16_t x) { return (float)x.hi + (float)x.lo; }

// ---- the residual stream's element type XT ------------------------------------------------------------------------------
// float in the fp32 / exact precisions; half_t (IEEE fp16) in the 16-bit precisions since round 5: every kernel that
// touches the stream moves half the bytes (a ConvNeXt block: 16 C -> 10 C bytes per position), and the stream's 11 significant
// bits cost the bf16 precision nothing measurable (frame embeddings 4.43e-3 -> 4.48e-3 rel. rms off the fp32 oracle) and the
// f16 precision 5.4e-4 -> 8.5e-4 (oracle/rounding_study.py).  |x| <= 65504 is required of the stream (cn_from_f32<half_t>
// saturates in the element-wise kernels; the fused MLP's packed conversion does not).
__device__ __forceinline__ float cn_ld1(const float* p) { return *p; }
__device__ __forceinline__ float cn_ld1(const half_t* p) { return (float)*p; }
__device__ __forceinline__ f32x4 cn_ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 cn_ld4(const half_t* p) {
  const cn_h4<half_t> h = *(const cn_h4<half_t>*)p;
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// exact-erf GELU (torch F.gelu default; reference convnext.py:47, aac_tfmer.py:36)
__device__ __forceinline__ float cn_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): 1 rcp + 1 exp + 6 fma instead of the
// ~40-instruction libm erff; used by the bf16 path where the result is rounded to 8 bits anyway
__device__ __forceinline__ float cn_gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-z * z);  // erf(|x|/sqrt2)
  return 0.5f * x + 0.5f * fabsf(x) * e;            // 0.5 x (1 + sign(x) erf(|x|/sqrt2))
}

// Same approximation on 4 values with packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 do two lanes'
// worth per issue slot; the GELU epilogues are VALU-issue bound): ~8.5 instructions per element
// instead of ~15.
__device__ __forceinline__ f32x2 cn_gelu_fast2(f32x2 x) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  const f32x2 d = ax * (0.70710678118654752440f * 0.3275911f) + 1.0f;
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = t * 1.061405429f + (-1.453152027f);
  p = p * t + 1.421413741f;
  p = p * t + (-0.284496736f);
  p = p * t + 0.254829592f;
  const f32x2 u = x * 0.84932180028801904272f;  // sqrt(0.5 * log2(e)): exp(-x^2/2) = exp2(-u^2)
  const f32x2 zz = -(u * u);
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
  const f32x2 er = 1.0f - (p * t) * e;  // erf(|x| / sqrt2)
  const f32x2 h = x * 0.5f;
  return __builtin_elementwise_abs(h) * er + h;
}
// The "exact" precision's GELU: the same A&S 7.1.26 erfc (|abs error| <= 1.5e-7 -- about one fp32 ulp of the O(1)
// values it multiplies) written without the cancellation of 0.5 x + 0.5 |x| erf for negative arguments:
//     gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2)
// ~9 VALU per element in packed form against ~45 for libm's erff (the pw1 epilogue of stage 0 is 347 M elements per
// launch: erff alone was 380 us of a 620 us launch).
__device__ __forceinline__ f32x2 cn_gelu_as2(f32x2 x) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  const f32x2 d = ax * (0.70710678118654752440f * 0.3275911f) + 1.0f;
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = t * 1.061405429f + (-1.453152027f);
  p = p * t + 1.421413741f;
  p = p * t + (-0.284496736f);
  p = p * t + 0.254829592f;
  const f32x2 u = x * 0.84932180028801904272f;  // sqrt(0.5 * log2(e)): exp(-x^2/2) = exp2(-u^2)
  const f32x2 zz = -(u * u);
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
  const f32x2 q = (p * t) * e;                  // erfc(|x| / sqrt 2)
  const f32x2 pos = f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return pos - (ax * 0.5f) * q;
}
__device__ __forceinline__ float cn_gelu_as(float x) { return cn_gelu_as2(f32x2{x, x})[0]; }
__device__ __forceinline__ f32x4 cn_gelu_as4(f32x4 x) {
  const f32x2 a = cn_gelu_as2(f32x2{x[0], x[1]}), b = cn_gelu_as2(f32x2{x[2], x[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ f32x4 cn_gelu_fast4(f32x4 x) {
  const f32x2 a = cn_gelu_fast2(f32x2{x[0], x[1]}), b = cn_gelu_fast2(f32x2{x[2], x[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}

// GELU with ONE transcendental (round 5).  With a = |x|:   gelu(x) = x Phi(x) = max(x, 0) - a Phi(-a)   (both signs), and
// log2 Phi(-a) is smooth, concave and polynomial-like on a >= 0 (-1 at 0, ~ -a^2 / 2 log2 e beyond): Phi(-a) = 2^P(a) with P
// a minimax fit of the ABSOLUTE error of a 2^P(a) on [0, 10] (tools/lab/fit_gelu.py).  P decreases monotonically for every
// a >= 0: no clamp, 2^P underflows to 0 and gelu(x) -> max(x, 0).
//   degree 3: max |error| 5.5e-5 against the exact erf form   -- 5 VALU + 1 transcendental
//   degree 5: max |error| 8.6e-7 (fp32 evaluation)           -- 7 VALU + 1 transcendental
// against 7 VALU + 2 transcendentals (v_exp, v_rcp: 8 issue cycles each, MI355X guide) for the sigmoid form of rounds 2-4.
// The function takes h = x / 2 (the fused MLP kernels fold the factor into the packed W1 / b1 operands: a power of two, so
// every product and sum is the same bits, halved):   max(x, 0) = h + |h|  is ONE v_add with a source modifier -- fmaxf(x, 0)
// costs two instructions, a canonicalising `v_max x, x` first, because the accumulator of an MFMA is not known to be quiet --
// and   gelu(x) = (h + |h|) - |h| 2^(P(2 |h|) + 1):   3 (5) fma + v_exp + v_add + fma, nothing but compiler-visible
// instructions (inline-asm forms of v_max / v_fma were tried: the hazard recogniser does not see into them, and they read
// MFMA results before the matrix pipe had written them).  The bf16 kernels use degree 3 (their result is rounded to 8 bits:
// half an ulp is 1e-3 at 0.5), the fp16 kernels degree 5 (half an ulp 1.2e-4).
template <int DEG> __device__ __forceinline__ float cn_gelu_e1_half(float h) {
  const float a = __builtin_fabsf(h);
  float p;
  if constexpr (DEG == 3) {
    p = fmaf(a, 8.0f * -0.02487156353890896f, 4.0f * -0.49887558817863464f);
    p = fmaf(p, a, 2.0f * -1.1291841268539429f);
    p = fmaf(p, a, 1.0f - 1.0035500526428223f);
  } else {
    static_assert(DEG == 5, "degree 3 or 5");
    p = fmaf(a, 32.0f * -0.0004728270578198135f, 16.0f * 0.007081141695380211f);
    p = fmaf(p, a, 8.0f * -0.05181875079870224f);
    p = fmaf(p, a, 4.0f * -0.4600019156932831f);
    p = fmaf(p, a, 2.0f * -1.1507835388183594f);
    p = fmaf(p, a, 1.0f - 1.0000382661819458f);
  }
  float g = fmaf(-a, __builtin_amdgcn_exp2f(p), h + a);
  // (an empty asm: the result is an opaque scalar, so the SLP vectoriser -- seeded by the v_cvt_pk that packs two neighbouring
  // results -- does not pair the elements' closing instructions into v_pk_add_f32 / v_pk_fma_f32: those cannot take |h| as a
  // modifier, need a v_and per element to materialise it, and are an anti-lever beside MFMAs, MI355X guide)
  asm("" : "+v"(g));
  return g;
}
template <int DEG> __device__ __forceinline__ float cn_gelu_e1(float x) { return cn_gelu_e1_half<DEG>(0.5f * x); }

// XCD-aware block remap (cdna guide T1, bijective form): workgroups are dealt round-robin over the
// 8 XCDs, each with a private L2, so blocks b and b+8 share an L2.  Give every XCD one CONTIGUOUS
// chunk of the logical grid so that neighbouring tiles (shared halos / shared A panels) hit in L2
// instead of being re-fetched over the fabric once per XCD.  Affects speed only.
__device__ __forceinline__ int cn_xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// 64-lane butterfly reductions
__device__ __forceinline__ float cn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// DPP lane permutes (VALU, no LDS crossbar round trip like ds_bpermute): quad_perm [1,0,3,2] = 0xB1,
// quad_perm [2,3,0,1] = 0x4E, row_half_mirror (i <-> 7-i in each 8) = 0x141, row_mirror (i <-> 15-i in each 16) = 0x140.
// For commutative reductions the mirrors pair a lane with one from the other half exactly like xor 4 / xor 8.
template <int CTRL> __device__ __forceinline__ float cn_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int cn_dpp(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// sum over each aligned group of 8 / 16 lanes, result in every lane of the group
__device__ __forceinline__ float cn_sum8_dpp(float v) {
  v += cn_dpp<0xB1>(v);
  v += cn_dpp<0x4E>(v);
  v += cn_dpp<0x141>(v);
  return v;
}
__device__ __forceinline__ float cn_wave_sum_dpp(float v) {
  v = cn_sum8_dpp(v);
  v += cn_dpp<0x140>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float cn_wave_max_dpp(float v) {
  v = fmaxf(v, cn_dpp<0xB1>(v));
  v = fmaxf(v, cn_dpp<0x4E>(v));
  v = fmaxf(v, cn_dpp<0x141>(v));
  v = fmaxf(v, cn_dpp<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float cn_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// store 4 consecutive values (n-contiguous) as T; p must be 4-element aligned
__device__ __forceinline__ void cn_store4(float* p, float a, float b, float c, float d) {
  *(f32x4*)p = f32x4{a, b, c, d};
}
__device__ __forceinline__ void cn_store4(bf16_t* p, float a, float b, float c, float d) {
  *(bf16x4*)p = bf16x4{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
}
__device__ __forceinline__ void cn_store4(half_t* p, float a, float b, float c, float d) {
  *(cn_h4<half_t>*)p = cn_h4<half_t>{cn_from_f32<half_t>(a), cn_from_f32<half_t>(b), cn_from_f32<half_t>(c), cn_from_f32<half_t>(d)};
}
__device__ __forceinline__ void cn_store4(sp16_t* p, float a, float b, float c, float d) {
  *(u32x4*)p = u32x4{cn_sp16_bits(a), cn_sp16_bits(b), cn_sp16_bits(c), cn_sp16_bits(d)};
}
// store 8 consecutive values as T; p must be 8-element aligned (bf16: ONE 16-byte store)
__device__ __forceinline__ void cn_store8(float* p, const float (&o)[8]) {
  *(f32x4*)p = f32x4{o[0], o[1], o[2], o[3]};
  *(f32x4*)(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
}
__device__ __forceinline__ void cn_store8(sp16_t* p, const float (&o)[8]) {
  *(u32x4*)p = u32x4{cn_sp16_bits(o[0]), cn_sp16_bits(o[1]), cn_sp16_bits(o[2]), cn_sp16_bits(o[3])};
  *(u32x4*)(p + 4) = u32x4{cn_sp16_bits(o[4]), cn_sp16_bits(o[5]), cn_sp16_bits(o[6]), cn_sp16_bits(o[7])};
}
__device__ __forceinline__ void cn_store8(half_t* p, const float (&o)[8]) {
  *(f16x8*)p = f16x8{cn_from_f32<half_t>(o[0]), cn_from_f32<half_t>(o[1]), cn_from_f32<half_t>(o[2]), cn_from_f32<half_t>(o[3]),
                     cn_from_f32<half_t>(o[4]), cn_from_f32<half_t>(o[5]), cn_from_f32<half_t>(o[6]), cn_from_f32<half_t>(o[7])};
}
__device__ __forceinline__ void cn_store8(bf16_t* p, const float (&o)[8]) {
  *(bf16x8*)p = bf16x8{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3], (bf16_t)o[4], (bf16_t)o[5], (bf16_t)o[6], (bf16_t)o[7]};
}
