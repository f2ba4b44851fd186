// Log-mel frontend for gfx950 (row a2 of SURVEY.md section 8a).
//
// Reference: nn/encoders/convnext.py:270-292 -- torchlibrosa Spectrogram (reflect pad 512, hann,
// n_fft 1024, hop 320, power 2) -> LogmelFilterBank (melW matmul, 10*log10(clamp 1e-10)) -> bn0
// (eval BatchNorm2d over the 224 mel bins).  The reference evaluates the DFT as two dense
// conv1d (2.1 GFLOP/clip); here one wavefront does one frame as a 512-point complex FFT
// (radix-8 x 8 x 8, 8 complex values per lane, two LDS transposes) + the real-FFT untangle,
// then the banded mel projection, log10 and the bn0 affine -- one pass, fp32 throughout.
//
// The window and the mel matrix are taken from the checkpoint's own tensors
// (conv_real.weight[0,0,:] is the window because cos(0) = 1; melW dense, its per-bin non-zero
// band located at create time), so a checkpoint with other frozen tensors is honoured.
#include "ctx.h"

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float2 cmul_mi(float2 a) { return float2{a.y, -a.x}; }  // a * (-i)

// in-place 8-point DFT, natural order in and out
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
  const float h = 0.70710678118654752440f;
  float2 a[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = cadd(v[i], v[i + 4]);
    a[i + 4] = csub(v[i], v[i + 4]);
  }
  a[5] = float2{h * (a[5].x + a[5].y), h * (a[5].y - a[5].x)};   // * (1 - i)/sqrt2
  a[6] = cmul_mi(a[6]);
  a[7] = float2{h * (a[7].y - a[7].x), -h * (a[7].x + a[7].y)};  // * (-1 - i)/sqrt2
  float2 b0 = cadd(a[0], a[2]), b2 = csub(a[0], a[2]), b1 = cadd(a[1], a[3]), b3 = cmul_mi(csub(a[1], a[3]));
  float2 b4 = cadd(a[4], a[6]), b6 = csub(a[4], a[6]), b5 = cadd(a[5], a[7]), b7 = cmul_mi(csub(a[5], a[7]));
  v[0] = cadd(b0, b1);
  v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3);
  v[6] = csub(b2, b3);
  v[1] = cadd(b4, b5);
  v[5] = csub(b4, b5);
  v[3] = cadd(b6, b7);
  v[7] = csub(b6, b7);
}

// The four waves of a block work on their own frames and their own LDS buffers: the stages of a frame only need the LDS
// operations of ONE wave to stay in program order (they do: a wave's ds instructions execute in order), not a block barrier.
__device__ __forceinline__ void cn_wave_sync() { __builtin_amdgcn_wave_barrier(); }

#define FE_PITCH 72  // complex elements per transpose row (64 + 8 pad)

#ifndef FE_NW
#define FE_NW 16  // waves (= frames in flight) per block
#endif
// Static LDS of cn_logmel_kernel (the five arrays below) and the dynamic LDS its launch adds on top: together the WHOLE 160 KB of a
// compute unit (minus < 768 bytes of rounding and alignment slack), one block of 16 waves per CU.
// History: in round 2 frames came out wrong -- lanes 48-63 of one VALU result at a time -- whenever the decoder's GEMM workgroups
// shared a CU with this kernel, and taking the CU's LDS was the cure that kept them out.  Round 3 found the mechanism (profiles/r03_notes.md
// section 8, tools/lab/pk_mfma_probe.hip): a packed-fp32 instruction whose op_sel feeds the high half of src1 to the low result
// (hipcc's SLP vectoriser made 34 of them out of the complex arithmetic below) is executed wrongly by MI355X in lanes 48-63 while
// another wave on the SIMD runs v_mfma_f32_16x16x32_bf16.  The file is now compiled without SLP vectorisation (build.py FILE_FLAGS: no
// packed instruction at all, same speed), isa_lint.py rejects a library that contains the form, and the exclusive CU stays as it
// costs nothing.
#define FE_STATIC_BYTES ((512 + 513) * 8 + 1024 * 4 + FE_NW * 8 * FE_PITCH * 8 + FE_NW * 520 * 4)
#define FE_LDS_TOTAL (160 * 1024)
#define FE_FILL_BYTES ((FE_LDS_TOTAL - FE_STATIC_BYTES - 256) / 256 * 256)  // (256 bytes of slack for the arrays' alignment padding)
#define FE_MEL_LDS_MAX FE_FILL_BYTES  // the band-compact mel matrix lives in that dynamic LDS when it fits (it does: ~16 rows x 224)
static_assert(FE_STATIC_BYTES + FE_FILL_BYTES > FE_LDS_TOTAL - 768 && FE_STATIC_BYTES + FE_FILL_BYTES <= FE_LDS_TOTAL,
              "the log-mel block must own its compute unit's LDS");
template <bool MEL_LDS>
__global__ __launch_bounds__(FE_NW * 64) void cn_logmel_kernel(const float* __restrict__ wave, int L, int F, int total, int mel_rows,
                                                        const float* __restrict__ window,
                                                        const float2* __restrict__ tw512,
                                                        const float2* __restrict__ tw1024,
                                                        const float* __restrict__ melC, const int* __restrict__ band,
                                                        const float* __restrict__ bn_scale,
                                                        const float* __restrict__ bn_shift, float* __restrict__ out) {
  __shared__ float2 s_tw512[512];
  __shared__ float2 s_tw1024[513];
  __shared__ __attribute__((aligned(16))) float s_win[1024];
  __shared__ float2 s_x[FE_NW][8 * FE_PITCH];
  __shared__ float s_p[FE_NW][520];
  // A block takes a compute unit's LDS for itself (all 160 KB): nothing that allocates LDS can start beside it (see above).
  // (the launch adds >= FE_FILL_BYTES of dynamic LDS; it holds the mel matrix when that fits)
  extern __shared__ __attribute__((aligned(16))) float s_mel[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (MEL_LDS)
    for (int i = tid; i < mel_rows * CN_N_MELS; i += FE_NW * 64) s_mel[i] = melC[i];
  for (int i = tid; i < 512; i += FE_NW * 64) s_tw512[i] = tw512[i];
  for (int i = tid; i < 513; i += FE_NW * 64) s_tw1024[i] = tw1024[i];
  for (int i = tid; i < 1024; i += FE_NW * 64) s_win[i] = window[i];
  __syncthreads();
  float2* sx = s_x[wv];
  float* sp = s_p[wv];
  // mel bands of this lane's four bins; the trip count of a group of 64 bins is its widest band (rows past a bin's own band
  // hold zeros in melC, and acc + p * 0 = acc, so the sum is still the dense row's sum in the dense row's order)
  int m_lo[4], m_trip[4];
  float m_sc[4], m_sh[4];  // bn0 affine of this lane's bins (loaded once, not per frame)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int m = lane + 64 * g;
    const bool on = m < CN_N_MELS;
    m_sc[g] = on ? bn_scale[m] : 0.f;
    m_sh[g] = on ? bn_shift[m] : 0.f;
    m_lo[g] = on ? band[2 * m] : 0;
    int nb = on ? band[2 * m + 1] - m_lo[g] : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) nb = max(nb, __shfl_xor(nb, d));
    m_trip[g] = min(nb, mel_rows);
  }
  const bool pair_loads = (L & 1) == 0 && (((size_t)wave) & 7) == 0;  // every even sample index is 8-byte aligned

  for (int base = blockIdx.x * FE_NW; base < total; base += gridDim.x * FE_NW) {
    const int fr = base + wv;
    const bool act = fr < total;
    float2 v[8];
    // ---- stage 1: lane = b, v[a] = z[64a + b], z[n] = (s[2n] w[2n], s[2n+1] w[2n+1]) ---------
    if (act) {
      const int b = fr / F, f = fr - b * F;
      const float* x = wave + (size_t)b * L;
      const int p0 = f * CN_HOP - CN_N_FFT / 2;  // even
      if (pair_loads && p0 >= 0 && p0 + CN_N_FFT <= L) {  // the whole window inside the clip (wave-uniform): 8-byte loads
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int n2 = 2 * (64 * a + lane);
          const float2 xv = *(const float2*)(x + p0 + n2);
          const float2 wv2 = *(const float2*)(s_win + n2);
          v[a] = float2{xv.x * wv2.x, xv.y * wv2.y};
        }
      } else {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int n2 = 2 * (64 * a + lane);
          int q0 = p0 + n2, q1 = q0 + 1;
          q0 = q0 < 0 ? -q0 : (q0 >= L ? 2 * L - 2 - q0 : q0);  // reflect (no edge repeat)
          q1 = q1 < 0 ? -q1 : (q1 >= L ? 2 * L - 2 - q1 : q1);
          v[a] = float2{x[q0] * s_win[n2], x[q1] * s_win[n2 + 1]};
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < 8; ++a) v[a] = float2{0.f, 0.f};
    }
    dft8(v);
#pragma unroll
    for (int c = 1; c < 8; ++c) v[c] = cmul(v[c], s_tw512[lane * c]);
#pragma unroll
    for (int c = 0; c < 8; ++c) sx[c * FE_PITCH + lane] = v[c];
    cn_wave_sync();
    // ---- stage 2: lane = (c, b'), v[a'] = Y[c][8a' + b'] --------------------------------------
    const int c = lane >> 3, lo3 = lane & 7;
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = sx[c * FE_PITCH + 8 * a + lo3];
    dft8(v);
#pragma unroll
    for (int cp = 1; cp < 8; ++cp) v[cp] = cmul(v[cp], s_tw512[8 * lo3 * cp]);
    cn_wave_sync();
#pragma unroll
    for (int cp = 0; cp < 8; ++cp) sx[c * FE_PITCH + cp * 8 + lo3] = v[cp];
    cn_wave_sync();
    // ---- stage 3: lane = (c, c'), v[b'] = Y'[c][c'][b'] ---------------------------------------
#pragma unroll
    for (int bp = 0; bp < 8; ++bp) v[bp] = sx[c * FE_PITCH + lo3 * 8 + bp];
    dft8(v);
    cn_wave_sync();
#pragma unroll
    for (int dp = 0; dp < 8; ++dp) sx[c + 8 * lo3 + 64 * dp] = v[dp];  // Z[k], k = c + 8c' + 64d'
    cn_wave_sync();
    // ---- real-FFT untangle -> power spectrum ---------------------------------------------------
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = lane + 64 * j;
      const float2 A = sx[k];
      const float2 Bz = sx[(512 - k) & 511];
      const float2 E = float2{0.5f * (A.x + Bz.x), 0.5f * (A.y - Bz.y)};   // (A + conj B)/2
      const float2 O = float2{0.5f * (A.y + Bz.y), -0.5f * (A.x - Bz.x)};  // (A - conj B) * (-i/2)
      const float2 X = cadd(E, cmul(s_tw1024[k], O));
      sp[k] = X.x * X.x + X.y * X.y;
    }
    if (lane == 0) {
      const float2 A = sx[0];
      const float xr = A.x - A.y;  // X[512] = Re Z0 - Im Z0
      sp[512] = xr * xr;
    }
    cn_wave_sync();
    // ---- mel band sums, dB, bn0 affine ---------------------------------------------------------
    if (act) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = lane + 64 * g;
        if (m < CN_N_MELS) {
          const int lo = m_lo[g], trip = m_trip[g];
          const float* mc = MEL_LDS ? s_mel : melC;
          float acc = 0.f;
          int i = 0;
          for (; i + 4 <= trip; i += 4) {  // bins lo, lo+1, ..: same order as the dense row
            float p[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              p[u] = sp[min(lo + i + u, CN_N_BINS - 1)];
              w[u] = mc[(i + u) * CN_N_MELS + m];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(p[u], w[u], acc);
          }
          for (; i < trip; ++i) acc = fmaf(sp[min(lo + i, CN_N_BINS - 1)], mc[i * CN_N_MELS + m], acc);
          const float db = 10.0f * log10f(fmaxf(acc, 1e-10f));
          out[(size_t)fr * CN_N_MELS + m] = db * m_sc[g] + m_sh[g];
        }
      }
    }
    cn_wave_sync();
  }
}

int cn_frontend(conette_ctx* ctx, const float* wave, int B, int L, float* out, hipStream_t s) {
  if (L <= CN_N_FFT / 2) {
    cn_set_error("frontend: n_samples=%d must exceed %d (reflect padding)", L, CN_N_FFT / 2);
    return CN_ERR_ARG;
  }
  const int F = L / CN_HOP + 1;
  const long total = (long)B * F;
  int grid = (int)((total + FE_NW - 1) / FE_NW);
  if (grid > ctx->n_cu) grid = ctx->n_cu;  // one block per compute unit (its LDS), each walking its share of the frames
  const int mel_bytes = ctx->mel_rows * CN_N_MELS * 4;
  if (mel_bytes <= FE_MEL_LDS_MAX) {
    const int dyn = FE_FILL_BYTES;
    CN_TRY(cn_configure_lds((const void*)cn_logmel_kernel<true>, dyn));
    hipLaunchKernelGGL(cn_logmel_kernel<true>, dim3(grid), dim3(FE_NW * 64), dyn, s, wave, L, F, (int)total, ctx->mel_rows,
                       ctx->window, ctx->tw512, ctx->tw1024, ctx->melC, ctx->band, ctx->bn_scale, ctx->bn_shift, out);
  } else {  // a checkpoint with very wide mel bands: the matrix stays in global memory
    CN_TRY(cn_configure_lds((const void*)cn_logmel_kernel<false>, FE_FILL_BYTES));
    hipLaunchKernelGGL(cn_logmel_kernel<false>, dim3(grid), dim3(FE_NW * 64), FE_FILL_BYTES, s, wave, L, F, (int)total,
                       ctx->mel_rows, ctx->window, ctx->tw512, ctx->tw1024, ctx->melC, ctx->band, ctx->bn_scale, ctx->bn_shift, out);
  }
  CN_LAUNCH_CHECK();
  return CN_OK;
}
