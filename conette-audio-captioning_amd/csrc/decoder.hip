// Task-embedded Transformer decoder under beam search, KV-cached, whole loop on device
// (rows a9-a14 of SURVEY.md section 8a).
//
// Reference: pl_modules/conette.py:452-467 (projection + pad mask), nn/decoders/aac_tfmer.py:71-118
// (embedding * sqrt(d) + PE, 6 post-norm torch TransformerDecoderLayer, classifier),
// nn/decoding/beam.py:22-269 (search).  The reference re-decodes the whole prefix and
// re-projects the audio memory at every step, and replicates the memory per beam; here
//   * the cross-attention K/V of all layers are projected once per clip and shared by its beams,
//   * self-attention K/V are cached per (layer, step, row); beams are re-ordered through a
//     small ancestor table instead of moving the cache,
//   * EOS floor, forbid-repeat mask, log-softmax, running sums, per-clip top-k and the
//     finished / shrinking-k bookkeeping run in one kernel per step with no host sync.
#include "ctx.h"
#include <stdlib.h>

#include <type_traits>

#include "gemm2.h"
#include "dec_ffn.h"

#define CN_MAX_BEAM 16  // beams 1..8: the register-resident step kernel; 9..16 (BaselinePLM's default is 10, baseline.py:47): the generic one
#define CN_MAX_PRED 64
#define FF2_SPLITS 8
static int ff2_splits_default() { return 4; }  // split-K slabs of the unfused FFN2 GEMM at small R

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void cn_cvt_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = cn_from_f32<T>(in[i]);
}

__global__ void cn_init_state_kernel(int B, int beam, int maxp, const int* __restrict__ bos, int* n_active, int* slot,
                                     float* sum_lp, int* prefix, int* anc, int* cur_tok, int* out_preds,
                                     float* out_avg, int* out_len, int* sizes, int pad_id, int* trace_sel,
                                     float* trace_val, int* live, float* margins) {
  const int R = B * beam;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  for (int i = gid; i < CN_MAX_PRED + 2; i += gsz) live[i] = 0;
  for (int i = gid; i < B; i += gsz) n_active[i] = beam;
  for (int i = gid; i < R; i += gsz) {
    slot[i] = i % beam;
    sum_lp[i] = 0.f;
    cur_tok[i] = bos[i / beam];
    out_avg[i] = 0.f;
    out_len[i] = 0;
  }
  for (int i = gid; i < R * (maxp + 1); i += gsz) prefix[i] = (i % (maxp + 1)) == 0 ? bos[i / ((maxp + 1) * beam)] : pad_id;
  for (int i = gid; i < R * maxp; i += gsz) {
    anc[i] = 0;
    out_preds[i] = pad_id;
  }
  if (gid < 2) sizes[gid] = 0;
  if (trace_sel)
    for (int i = gid; i < maxp * R * 2; i += gsz) trace_sel[i] = -1;
  if (trace_val)
    for (int i = gid; i < maxp * R; i += gsz) trace_val[i] = 0.f;
  if (margins)  // a step a clip does not take (it has finished) decides nothing: +inf
    for (int i = gid; i < B * 2 * (maxp + 1); i += gsz) margins[i] = INFINITY;
}

// x = E[tok] * sqrt(d) + PE[step]   (aac_tfmer.py:100-106); d == 256, one wave per row
template <typename T>
__global__ __launch_bounds__(256) void cn_embed_kernel(const int* __restrict__ tok, const float* __restrict__ emb,
                                                       const float* __restrict__ pe, int step, int R, float scale,
                                                       float* __restrict__ x, T* __restrict__ xt) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const f32x4 e = *(const f32x4*)(emb + (size_t)tok[r] * 256 + 4 * lane);
  const f32x4 p = *(const f32x4*)(pe + (size_t)step * 256 + 4 * lane);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = e[i] * scale + p[i];
  *(f32x4*)(x + (size_t)r * 256 + 4 * lane) = o;
  cn_store4(xt + (size_t)r * 256 + 4 * lane, o[0], o[1], o[2], o[3]);
}

// x = LayerNorm(sum of nslab partial slabs [+ bias + residual]) (eps 1e-5, d == 256); one wave per row.
// With nslab == 1 and no bias / residual this is the plain LayerNorm of a finished GEMM output.
template <typename T>
__global__ __launch_bounds__(256) void cn_ln256_kernel(const float* __restrict__ in, int nslab, size_t slab_stride,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ resid, const float* __restrict__ w,
                                                       const float* __restrict__ b, int R, float* __restrict__ x,
                                                       T* __restrict__ xt) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  f32x4 v = *(const f32x4*)(in + (size_t)r * 256 + 4 * lane);
  for (int sl = 1; sl < nslab; ++sl) {
    const f32x4 u = *(const f32x4*)(in + sl * slab_stride + (size_t)r * 256 + 4 * lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += u[i];
  }
  if (bias) {
    const f32x4 u = *(const f32x4*)(bias + 4 * lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += u[i];
  }
  if (resid) {
    const f32x4 u = *(const f32x4*)(resid + (size_t)r * 256 + 4 * lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += u[i];
  }
  const float mean = cn_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s2 = fmaf(v[i] - mean, v[i] - mean, s2);
  const float rstd = 1.0f / sqrtf(cn_wave_sum(s2) * (1.0f / 256.0f) + 1e-5f);
  const f32x4 ww = *(const f32x4*)(w + 4 * lane), bb = *(const f32x4*)(b + 4 * lane);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (v[i] - mean) * rstd * ww[i] + bb[i];
  *(f32x4*)(x + (size_t)r * 256 + 4 * lane) = o;
  cn_store4(xt + (size_t)r * 256 + 4 * lane, o[0], o[1], o[2], o[3]);
}

template <typename T> __device__ __forceinline__ f32x4 cn_load4(const T* p);
template <> __device__ __forceinline__ f32x4 cn_load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 cn_load4<bf16_t>(const bf16_t* p) {
  const bf16x4 v = *(const bf16x4*)p;
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

template <> __device__ __forceinline__ f32x4 cn_load4<half_t>(const half_t* p) {
  const cn_h4<half_t> v = *(const cn_h4<half_t>*)p;
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 cn_load4<sp16_t>(const sp16_t* p) {
  const u32x4 v = *(const u32x4*)p;
  return f32x4{cn_sp16_value(v[0]), cn_sp16_value(v[1]), cn_sp16_value(v[2]), cn_sp16_value(v[3])};
}

__device__ __forceinline__ float cn_dot8(f32x4 a, f32x4 b) {  // 32-dim head dot: 4 local + 8-lane reduce
  float d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  d += __shfl_xor(d, 1);
  d += __shfl_xor(d, 2);
  d += __shfl_xor(d, 4);
  return d;
}

// Online-softmax update with a batch of NB scores / values (all loads were issued before the math,
// so a row pays one memory latency per batch instead of one per key).
template <int NB>
__device__ __forceinline__ void cn_attn_update(const float (&sc)[NB], const f32x4 (&vv)[NB], float& m, float& l,
                                               f32x4& acc) {
  float mb = sc[0];
#pragma unroll
  for (int u = 1; u < NB; ++u) mb = fmaxf(mb, sc[u]);
  const float mn = fmaxf(m, mb);
  const float corr = __expf(m - mn);
  l *= corr;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] *= corr;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const float p = __expf(sc[u] - mn);  // masked entries carry -inf -> p = 0
    l += p;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = fmaf(p, vv[u][i], acc[i]);
  }
  m = mn;
}

// Causal self-attention of ONE new position per row over the cached prefix (8 heads x 32).
// lane l owns dims 4l..4l+3 (head l>>3).  Writes this step's K/V into the cache.
template <typename T>
__global__ __launch_bounds__(256) void cn_self_attn_kernel(const float* __restrict__ qkv, T* __restrict__ kc,
                                                           T* __restrict__ vc, const int* __restrict__ anc, int step,
                                                           int R, int beam, int maxp, float scale,
                                                           T* __restrict__ out,
                                                           const unsigned long long* __restrict__ kvalid) {
  constexpr int NB = 8;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const unsigned long long valid = kvalid ? kvalid[r] : ~0ull;  // teacher forcing: padded caption positions are no keys
  const float* row = qkv + (size_t)r * 768 + 4 * lane;
  f32x4 q = *(const f32x4*)row;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] *= scale;
  const f32x4 kn = *(const f32x4*)(row + 256), vn = *(const f32x4*)(row + 512);
  T* kdst = kc + ((size_t)step * R + r) * 256 + 4 * lane;
  T* vdst = vc + ((size_t)step * R + r) * 256 + 4 * lane;
  cn_store4(kdst, kn[0], kn[1], kn[2], kn[3]);
  cn_store4(vdst, vn[0], vn[1], vn[2], vn[3]);
  const int rb = (r / beam) * beam;
  float m = -INFINITY, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < step; s0 += NB) {
    f32x4 kk[NB], vv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int s = min(s0 + u, step - 1);
      const int src = rb + anc[(size_t)r * maxp + s];
      kk[u] = cn_load4<T>(kc + ((size_t)s * R + src) * 256 + 4 * lane);
      vv[u] = cn_load4<T>(vc + ((size_t)s * R + src) * 256 + 4 * lane);
    }
    float sc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u)
      sc[u] = (s0 + u < step && ((valid >> (s0 + u)) & 1)) ? cn_dot8(q, kk[u]) : -INFINITY;
    cn_attn_update<NB>(sc, vv, m, l, acc);
  }
  if ((valid >> step) & 1) {  // own entry, at cache precision
    const f32x4 kk = f32x4{cn_to_f32(cn_from_f32<T>(kn[0])), cn_to_f32(cn_from_f32<T>(kn[1])),
                           cn_to_f32(cn_from_f32<T>(kn[2])), cn_to_f32(cn_from_f32<T>(kn[3]))};
    const f32x4 v1[1] = {f32x4{cn_to_f32(cn_from_f32<T>(vn[0])), cn_to_f32(cn_from_f32<T>(vn[1])),
                               cn_to_f32(cn_from_f32<T>(vn[2])), cn_to_f32(cn_from_f32<T>(vn[3]))}};
    const float s1[1] = {cn_dot8(q, kk)};
    cn_attn_update<1>(s1, v1, m, l, acc);
  }
  const float inv = 1.0f / l;
  cn_store4(out + (size_t)r * 256 + 4 * lane, acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
}

// Cross-attention of one query per row over its clip's projected audio memory (shared by beams);
// frames >= frame_lens[clip] are masked (key_padding_mask, conette.py:460-462).
template <typename T>
__global__ __launch_bounds__(256) void cn_cross_attn_kernel(const float* __restrict__ q, const T* __restrict__ kv,
                                                            int kv_ld, int kv_off, const int* __restrict__ lens,
                                                            int R, int beam, int Ta, float scale,
                                                            T* __restrict__ out) {
  constexpr int NB = 8;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const int b = r / beam;
  f32x4 qq = *(const f32x4*)(q + (size_t)r * 256 + 4 * lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) qq[i] *= scale;
  int n = lens[b];
  n = n < 1 ? 1 : (n > Ta ? Ta : n);
  const T* base = kv + (size_t)b * Ta * kv_ld + kv_off + 4 * lane;
  float m = -INFINITY, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < n; t0 += NB) {
    f32x4 kk[NB], vv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = min(t0 + u, n - 1);
      kk[u] = cn_load4<T>(base + (size_t)t * kv_ld);
      vv[u] = cn_load4<T>(base + (size_t)t * kv_ld + 256);
    }
    float sc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) sc[u] = (t0 + u < n) ? cn_dot8(qq, kk[u]) : -INFINITY;
    cn_attn_update<NB>(sc, vv, m, l, acc);
  }
  const float inv = 1.0f / l;
  cn_store4(out + (size_t)r * 256 + 4 * lane, acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
}

// ---------------------------------------------------------------------------------------------
// one search step for one clip (beam.py:125-203, _select_k_next_toks :230-269)
// ---------------------------------------------------------------------------------------------
struct ValIdx {
  float v;
  int i;
};
__device__ __forceinline__ ValIdx vi_better(ValIdx a, ValIdx b) {  // larger value, then lower flat index
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
template <int CTRL> __device__ __forceinline__ ValIdx vi_dpp_step(ValIdx x) {
  return vi_better(x, ValIdx{cn_dpp<CTRL>(x.v), cn_dpp<CTRL>(x.i)});
}
// same result as vi_wave (the order relation is total), 4 of the 6 exchange steps on DPP
__device__ __forceinline__ ValIdx vi_wave_dpp(ValIdx x) {
  x = vi_dpp_step<0xB1>(x);
  x = vi_dpp_step<0x4E>(x);
  x = vi_dpp_step<0x141>(x);
  x = vi_dpp_step<0x140>(x);
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    ValIdx y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    x = vi_better(x, y);
  }
  return x;
}
__device__ __forceinline__ ValIdx vi_wave(ValIdx x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ValIdx y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    x = vi_better(x, y);
  }
  return x;
}

// The effective margin of one top-k call (round 6, precision "certified"): how far the call is from ANY other outcome -- the gap
// between the last pick and the first rejected candidate, and the gaps between consecutive picks (their order decides which
// hypothesis lands in which slot: beam.py:165-169).  A 16-bit search whose every call (and whose final best-beam choice) has a
// margin above twice the precision's worst candidate error has taken the decisions an exact search takes.
// Two planes per clip, (B, 2, max_pred + 1): plane 0 = MEMBERSHIP, the gap that decides WHICH candidates continue (and, in its last
// column, which hypothesis is returned as the best); plane 1 = ORDER, the smallest gap between consecutive picks, which only decides
// which slot a hypothesis lands in -- the order of mult_preds, never best_preds.
__device__ __forceinline__ void cn_step_margin(float* margins, int b, int maxp, int step, const float* selv, int k, float runner_up) {
  float* row = margins + (size_t)b * 2 * (maxp + 1);
  row[step] = selv[k - 1] - runner_up;   // NaN (-inf - -inf: fewer finite candidates than picks) reads as "not certified" on the host
  float m = INFINITY;
  for (int c = 0; c + 1 < k; ++c) m = fminf(m, selv[c] - selv[c + 1]);
  row[(maxp + 1) + step] = m;
}

__global__ __launch_bounds__(256) void cn_search_step_kernel(float* __restrict__ logits, int ldv, int V, int beam,
                                                             int maxp, int step, int min_pred, int eos_id,
                                                             const uint8_t* __restrict__ forbid, int* n_active,
                                                             int* slot, float* sum_lp, int* prefix, int* anc,
                                                             int* cur_tok, int* out_preds, float* out_avg,
                                                             int* out_len, int* trace_sel, float* trace_val, int* live,
                                                             float* margins) {
  __shared__ float s_red[8];
  __shared__ ValIdx s_vi[4];
  __shared__ float s_mx[CN_MAX_BEAM], s_lg[CN_MAX_BEAM], s_base[CN_MAX_BEAM];
  __shared__ float s_selv[CN_MAX_BEAM + 1];
  __shared__ int s_self[CN_MAX_BEAM + 1];
  __shared__ int s_prefix[CN_MAX_BEAM][CN_MAX_PRED + 1];
  __shared__ int s_anc[CN_MAX_BEAM][CN_MAX_PRED];
  __shared__ int s_slot[CN_MAX_BEAM];
  __shared__ int s_newpos[CN_MAX_BEAM];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = n_active[b];
  if (k == 0) return;
  const int rb = b * beam;
  const int nrows = step == 0 ? 1 : k;

  // old state -> LDS
  for (int i = tid; i < k * (maxp + 1); i += 256) s_prefix[i / (maxp + 1)][i % (maxp + 1)] = prefix[(size_t)rb * (maxp + 1) + i];
  for (int i = tid; i < k * maxp; i += 256) s_anc[i / maxp][i % maxp] = anc[(size_t)rb * maxp + i];
  if (tid < k) {
    s_slot[tid] = slot[rb + tid];
    s_base[tid] = step == 0 ? 0.f : sum_lp[rb + tid];
  }
  __syncthreads();
  // EOS floor (beam.py:129-130) + forbid-repeat (beam.py:146-156), in place
  for (int i = tid; i < nrows * (step + 2); i += 256) {
    const int p = i / (step + 2), j = i % (step + 2);
    float* lg = logits + (size_t)(rb + p) * ldv;
    if (j == step + 1) {
      if (step < min_pred) lg[eos_id] = -INFINITY;
    } else if (forbid != nullptr) {
      const int tok = s_prefix[p][j];
      if (forbid[tok]) lg[tok] = -INFINITY;
    }
  }
  __syncthreads();
  // log-softmax statistics per row
  for (int p = 0; p < nrows; ++p) {
    const float* lg = logits + (size_t)(rb + p) * ldv;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) mx = fmaxf(mx, lg[v]);
    mx = cn_wave_max(mx);
    if (lane == 0) s_red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sm = 0.f;
    for (int v = tid; v < V; v += 256) sm += expf(lg[v] - mx);
    sm = cn_wave_sum(sm);
    if (lane == 0) s_red[4 + wv] = sm;
    __syncthreads();
    if (tid == 0) {
      s_mx[p] = mx;
      s_lg[p] = logf(s_red[4] + s_red[5] + s_red[6] + s_red[7]);
    }
    __syncthreads();
  }
  // top-k over the nrows * V candidates, k rounds of block arg-max (ties -> lowest flat index); with a margin output one
  // more round finds the first rejected candidate
  for (int c = 0; c < k + (margins != nullptr ? 1 : 0); ++c) {
    ValIdx best{-INFINITY, 0x7fffffff};
    for (int p = 0; p < nrows; ++p) {
      const float* lg = logits + (size_t)(rb + p) * ldv;
      const float mx = s_mx[p], lgs = s_lg[p], base = s_base[p];
      for (int v = tid; v < V; v += 256) {
        const int flat = p * V + v;
        bool taken = false;
        for (int d = 0; d < c; ++d) taken |= (s_self[d] == flat);
        if (taken) continue;
        float cand = (lg[v] - mx) - lgs;
        if (step != 0) cand = base + cand;
        best = vi_better(best, ValIdx{cand, flat});
      }
    }
    best = vi_wave(best);
    if (lane == 0) s_vi[wv] = best;
    __syncthreads();
    if (tid == 0) {
      ValIdx r0 = vi_better(vi_better(s_vi[0], s_vi[1]), vi_better(s_vi[2], s_vi[3]));
      s_selv[c] = r0.v;
      s_self[c] = r0.i;
    }
    __syncthreads();
  }
  // bookkeeping (beam.py:164-203)
  if (tid < k) {
    const size_t ti = ((size_t)step * gridDim.x + b) * beam + tid;
    if (trace_sel) {
      trace_sel[2 * ti] = s_self[tid] / V;
      trace_sel[2 * ti + 1] = s_self[tid] % V;
    }
    if (trace_val) trace_val[ti] = s_selv[tid];
  }
  if (tid == 0) {
    int cnt = 0;
    for (int c = 0; c < k; ++c) {
      const int token = s_self[c] % V;
      const bool fin = (token == eos_id) || (step == maxp - 1);
      s_newpos[c] = fin ? -1 : cnt++;
    }
    n_active[b] = cnt;
    if (cnt > 0) atomicAdd(&live[step + 1], cnt);  // rows that search on: gates the next step's kernels
    if (margins) cn_step_margin(margins, b, maxp, step, s_selv, k, s_selv[k]);
  }
  __syncthreads();
  for (int c = 0; c < k; ++c) {
    const int flat = s_self[c];
    const int parent = flat / V, token = flat % V;
    const int np = s_newpos[c];
    if (np < 0) {  // finished: write to the slot of this row position
      const int sl = rb + s_slot[c];
      for (int j = tid; j <= step; j += 256) out_preds[(size_t)sl * maxp + j] = (j == step) ? token : s_prefix[parent][j + 1];
      if (tid == 0) {
        out_avg[sl] = s_selv[c] / (float)(step + 1);
        out_len[sl] = step + 1;
      }
    } else {
      const int dst = rb + np;
      for (int j = tid; j <= step + 1; j += 256) prefix[(size_t)dst * (maxp + 1) + j] = (j == step + 1) ? token : s_prefix[parent][j];
      for (int j = tid; j <= step; j += 256) anc[(size_t)dst * maxp + j] = (j == step) ? parent : s_anc[parent][j];
      if (tid == 0) {
        sum_lp[dst] = s_selv[c];
        slot[dst] = s_slot[c];
        cur_tok[dst] = token;
      }
    }
  }
}

// Same step, register path (V <= 8192): 1024 threads per clip, every thread keeps its <= 8 logits of each live row
// in registers (one coalesced pass over the logits, nothing staged in LDS: the kernel starts on any CU with a free
// workgroup slot while the encoder's blocks hold the LDS), block reductions for the log-softmax, per-thread sorted
// top-k, then a two-level merge: k rounds of wave arg-max inside each wave, and the 16 x k wave winners merged
// redundantly by every wave.  Ties resolve to the lowest flat index exactly like the kernels above.
#define S3_T 1024
#define S3_VPT 8  // most logits per thread: V <= 8192
__device__ unsigned long long g_s3_prof[16];
#define S3_STAMP(i)                                            \
  if (dbg) {                                                   \
    const unsigned long long t_ = wall_clock64();              \
    if (tid == 0 && b == 0) atomicAdd(&g_s3_prof[i], t_ - t_prev); \
    t_prev = t_;                                               \
  }
template <int NR, int VPT>
__global__ __launch_bounds__(S3_T) void cn_search_step3_kernel(const float* __restrict__ logits, int ldv, int V, int beam,
                                                               int maxp, int step, int min_pred, int eos_id,
                                                               const uint8_t* __restrict__ forbid, int* n_active,
                                                               int* slot, float* sum_lp, int* prefix, int* anc,
                                                               int* cur_tok, int* out_preds, float* out_avg,
                                                               int* out_len, int* trace_sel, float* trace_val, int dbg,
                                                               int* live, float* margins) {
  __shared__ float s_redm[NR][16], s_reds[NR][16];
  __shared__ float s_ru[16];
  __shared__ ValIdx s_cand[16][CN_MAX_BEAM];
  __shared__ float s_base[CN_MAX_BEAM];
  __shared__ float s_selv[CN_MAX_BEAM];
  __shared__ int s_self[CN_MAX_BEAM];
  __shared__ int s_prefix[CN_MAX_BEAM][CN_MAX_PRED + 1];
  __shared__ int s_anc[CN_MAX_BEAM][CN_MAX_PRED];
  __shared__ int s_slot[CN_MAX_BEAM];
  __shared__ int s_newpos[CN_MAX_BEAM];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long t_prev = dbg ? wall_clock64() : 0ull;
  const int rb = b * beam;
  // every global read is issued up front and none depends on another (rows that turn out to be dead are read
  // and ignored): the step is a chain of dependent launches, a second round trip costs more than the bytes
  const int k_ld = n_active[b];
  const int rows_ld = step == 0 ? 1 : beam;
  float val[NR][VPT];
  bool fb[VPT];
#pragma unroll
  for (int sl = 0; sl < VPT; ++sl) {
    const int v = sl * S3_T + tid;
    fb[sl] = forbid != nullptr && v < V && forbid[v] != 0;
  }
#pragma unroll
  for (int p = 0; p < NR; ++p)
#pragma unroll
    for (int sl = 0; sl < VPT; ++sl) {
      const int v = sl * S3_T + tid;
      val[p][sl] = (p < rows_ld && v < V) ? logits[(size_t)(rb + p) * ldv + v] : -INFINITY;
    }
  for (int i = tid; i < beam * (maxp + 1); i += S3_T) s_prefix[i / (maxp + 1)][i % (maxp + 1)] = prefix[(size_t)rb * (maxp + 1) + i];
  for (int i = tid; i < beam * maxp; i += S3_T) s_anc[i / maxp][i % maxp] = anc[(size_t)rb * maxp + i];
  if (tid < beam) {
    s_slot[tid] = slot[rb + tid];
    s_base[tid] = step == 0 ? 0.f : sum_lp[rb + tid];
  }
  const int k = k_ld;
  if (k == 0) return;
  S3_STAMP(0)
  const int nrows = step == 0 ? 1 : k;
  __syncthreads();
  S3_STAMP(1)
  // forbid-repeat (beam.py:146-156) over the row's prefix and EOS floor (beam.py:129-130), on the owner's registers
#pragma unroll
  for (int p = 0; p < NR; ++p)
    if (p < nrows) {
      for (int j = 0; j <= step; ++j) {
        const int tok = s_prefix[p][j];
        if ((((unsigned)tok & (S3_T - 1)) >> 6) == (unsigned)wv) {  // wave-uniform: only the owner wave looks closer
          if ((tok & (S3_T - 1)) == tid) {
#pragma unroll
            for (int sl = 0; sl < VPT; ++sl)
              if (sl == (tok >> 10) && fb[sl]) val[p][sl] = -INFINITY;
          }
        }
      }
      if (step < min_pred && (eos_id & (S3_T - 1)) == tid) {
#pragma unroll
        for (int sl = 0; sl < VPT; ++sl)
          if (sl == (eos_id >> 10)) val[p][sl] = -INFINITY;
      }
    }
  S3_STAMP(2)
  // log-softmax statistics of every live row
#pragma unroll
  for (int p = 0; p < NR; ++p)
    if (p < nrows) {
      float mx = val[p][0];
#pragma unroll
      for (int sl = 1; sl < VPT; ++sl) mx = fmaxf(mx, val[p][sl]);
      mx = cn_wave_max_dpp(mx);
      if (lane == 0) s_redm[p][wv] = mx;
    }
  __syncthreads();
  float rmx[NR], rlg[NR];
#pragma unroll
  for (int p = 0; p < NR; ++p)
    if (p < nrows) {
      float mx = s_redm[p][0];
#pragma unroll
      for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_redm[p][w]);
      rmx[p] = mx;
      float sm = 0.f;
#pragma unroll
      for (int sl = 0; sl < VPT; ++sl) sm += __expf(val[p][sl] - mx);
      sm = cn_wave_sum_dpp(sm);
      if (lane == 0) s_reds[p][wv] = sm;
    }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NR; ++p)
    if (p < nrows) {
      float sm = s_reds[p][0];
#pragma unroll
      for (int w = 1; w < 16; ++w) sm += s_reds[p][w];
      rlg[p] = logf(sm);
    }
  S3_STAMP(3)
  // per-thread sorted top-k (strict > keeps the earlier = lower flat index first on ties)
  float bv[NR];
  int bi[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    bv[j] = -INFINITY;
    bi[j] = 0x7fffffff;
  }
#pragma unroll
  for (int p = 0; p < NR; ++p)
    if (p < nrows) {
      const float base = s_base[p];
#pragma unroll
      for (int sl = 0; sl < VPT; ++sl) {
        const int v = sl * S3_T + tid;
        float cand = (val[p][sl] - rmx[p]) - rlg[p];
        if (step != 0) cand = base + cand;
        if (v < V && cand > bv[NR - 1]) {
          bv[NR - 1] = cand;
          bi[NR - 1] = p * V + v;
#pragma unroll
          for (int j = NR - 1; j > 0; --j) {
            if (bv[j] > bv[j - 1]) {
              const float tv = bv[j];
              bv[j] = bv[j - 1];
              bv[j - 1] = tv;
              const int ti = bi[j];
              bi[j] = bi[j - 1];
              bi[j - 1] = ti;
            }
          }
        }
      }
    }
  S3_STAMP(4)
  // level 1: the wave's k best
  {
    int head = 0;
#pragma unroll
    for (int c = 0; c < NR; ++c)
      if (c < k) {
        ValIdx mine{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int j = 0; j < NR; ++j)
          if (head == j) mine = ValIdx{bv[j], bi[j]};
        const ValIdx best = vi_wave_dpp(mine);
        if (mine.i == best.i && mine.i != 0x7fffffff) ++head;
        if (lane == 0) s_cand[wv][c] = best;
      }
  }
  __syncthreads();
  S3_STAMP(5)
  // level 2: merge the 16 x k wave winners (every wave computes the same result)
  {
    const int nc = 16 * k;  // <= 128: two entries per lane
    ValIdx t0{-INFINITY, 0x7fffffff}, t1{-INFINITY, 0x7fffffff};
    if (lane < nc) t0 = s_cand[lane / k][lane % k];
    if (lane + 64 < nc) t1 = s_cand[(lane + 64) / k][(lane + 64) % k];
#pragma unroll
    for (int c = 0; c < NR; ++c)
      if (c < k) {
        const ValIdx best = vi_wave_dpp(vi_better(t0, t1));
        if (best.i != 0x7fffffff) {
          if (t0.i == best.i) t0 = ValIdx{-INFINITY, 0x7fffffff};
          else if (t1.i == best.i) t1 = ValIdx{-INFINITY, 0x7fffffff};
        }
        if (tid == 0) {
          s_selv[c] = best.v;
          s_self[c] = best.i;
        }
      }
  }
  __syncthreads();
  S3_STAMP(6)
  if (margins != nullptr) {  // the first rejected candidate: the best of everything the k picks left (same expression, same bits)
    float ru = -INFINITY;
#pragma unroll
    for (int p = 0; p < NR; ++p)
      if (p < nrows) {
        const float base = s_base[p];
#pragma unroll
        for (int sl = 0; sl < VPT; ++sl) {
          const int v = sl * S3_T + tid;
          float cand = (val[p][sl] - rmx[p]) - rlg[p];
          if (step != 0) cand = base + cand;
          const int flat = p * V + v;
          bool taken = v >= V;
          for (int c = 0; c < k; ++c) taken |= (s_self[c] == flat);
          if (!taken) ru = fmaxf(ru, cand);
        }
      }
    ru = cn_wave_max_dpp(ru);
    if (lane == 0) s_ru[wv] = ru;
    __syncthreads();
    if (tid == 0) {
      ru = s_ru[0];
#pragma unroll
      for (int w = 1; w < 16; ++w) ru = fmaxf(ru, s_ru[w]);
      cn_step_margin(margins, b, maxp, step, s_selv, k, ru);
    }
  }
  // bookkeeping (beam.py:164-203)
  if (tid < k) {
    const size_t ti = ((size_t)step * gridDim.x + b) * beam + tid;
    if (trace_sel) {
      trace_sel[2 * ti] = s_self[tid] / V;
      trace_sel[2 * ti + 1] = s_self[tid] % V;
    }
    if (trace_val) trace_val[ti] = s_selv[tid];
  }
  if (tid == 0) {
    int cnt = 0;
    for (int c = 0; c < k; ++c) {
      const int token = s_self[c] % V;
      const bool fin = (token == eos_id) || (step == maxp - 1);
      s_newpos[c] = fin ? -1 : cnt++;
    }
    n_active[b] = cnt;
    if (cnt > 0) atomicAdd(&live[step + 1], cnt);  // rows that search on: gates the next step's kernels
  }
  __syncthreads();
  for (int c = 0; c < k; ++c) {
    const int flat = s_self[c];
    const int parent = flat / V, token = flat % V;
    const int np = s_newpos[c];
    if (np < 0) {
      const int sl = rb + s_slot[c];
      for (int j = tid; j <= step; j += S3_T) out_preds[(size_t)sl * maxp + j] = (j == step) ? token : s_prefix[parent][j + 1];
      if (tid == 0) {
        out_avg[sl] = s_selv[c] / (float)(step + 1);
        out_len[sl] = step + 1;
      }
    } else {
      const int dst = rb + np;
      for (int j = tid; j <= step + 1; j += S3_T) prefix[(size_t)dst * (maxp + 1) + j] = (j == step + 1) ? token : s_prefix[parent][j];
      for (int j = tid; j <= step; j += S3_T) anc[(size_t)dst * maxp + j] = (j == step) ? parent : s_anc[parent][j];
      if (tid == 0) {
        sum_lp[dst] = s_selv[c];
        slot[dst] = s_slot[c];
        cur_tok[dst] = token;
      }
    }
  }
  S3_STAMP(7)
  if (dbg && tid == 0 && b == 0) atomicAdd(&g_s3_prof[9], 1ull);
}

// best beam per clip (beam.py:205-220): first max of the averaged log-prob; pred_size via atomicMax
__global__ void cn_finalize_kernel(int B, int beam, int maxp, int eos_id, const int* __restrict__ out_preds,
                                   const float* __restrict__ out_avg, const int* __restrict__ out_len,
                                   int* __restrict__ best_preds, float* __restrict__ best_lp,
                                   int* __restrict__ eos_idx, int* sizes, float* __restrict__ margins) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int best = 0, mlen = 0;
  float bv = out_avg[b * beam];
  for (int s = 0; s < beam; ++s) {
    if (out_avg[b * beam + s] > bv) {
      bv = out_avg[b * beam + s];
      best = s;
    }
    mlen = max(mlen, out_len[b * beam + s]);
  }
  atomicMax(&sizes[0], mlen);
  best_lp[b] = bv;
  if (margins) {  // how far the best hypothesis is from the second best (beam.py:214-217 takes the first maximum)
    float second = -INFINITY;
    for (int s = 0; s < beam; ++s)
      if (s != best) second = fmaxf(second, out_avg[b * beam + s]);
    margins[(size_t)b * 2 * (maxp + 1) + maxp] = bv - second;
  }
  int e = -1;
  for (int j = 0; j < maxp; ++j) {
    const int t = out_preds[((size_t)b * beam + best) * maxp + j];
    best_preds[(size_t)b * maxp + j] = t;
    if (e < 0 && t == eos_id) e = j;
  }
  eos_idx[b] = e;
}
// best_maxlen = max_b(first EOS index, or pred_size when absent) + 1 (beam.py:222-225)
__global__ void cn_finalize2_kernel(int B, const int* __restrict__ eos_idx, int* sizes) {
  __shared__ int s_m;
  if (threadIdx.x == 0) s_m = 0;
  __syncthreads();
  const int ps = sizes[0];
  int m = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int e = eos_idx[b];
    m = max(m, (e >= 0 && e < ps) ? e : ps);
  }
  atomicMax(&s_m, m);
  __syncthreads();
  if (threadIdx.x == 0) sizes[1] = min(s_m + 1, ps);
}

// ---------------------------------------------------------------------------------------------
// workspace + orchestration
// ---------------------------------------------------------------------------------------------
// ---- teacher forcing (nn/decoding/forcing.py:12-71): the step loop fed with the given caption instead of the search
__global__ void cn_force_init_kernel(const int32_t* __restrict__ caps, int B, int cap_len, int pad_id,
                                     unsigned long long* __restrict__ kvalid, int* __restrict__ anc) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  unsigned long long m = 0;
  for (int j = 0; j < cap_len; ++j) {
    if (caps[(size_t)r * cap_len + j] != pad_id) m |= 1ull << j;  // tensor_to_pad_mask(caps_in, pad_value=pad_id)
    anc[(size_t)r * cap_len + j] = 0;                             // beam 1: every ancestor is the row itself
  }
  kvalid[r] = m;
}
__global__ void cn_force_tok_kernel(const int32_t* __restrict__ caps, int B, int cap_len, int step, int* __restrict__ cur_tok) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < B) cur_tok[r] = caps[(size_t)r * cap_len + step];
}
__global__ void cn_force_store_kernel(const float* __restrict__ logits, int ldv, int V, int cap_len, int step,
                                      float* __restrict__ out) {
  const int r = blockIdx.y;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x)
    out[((size_t)r * cap_len + step) * V + v] = logits[(size_t)r * ldv + v];
}

// ---- greedy_search's full-logit output (nn/decoding/greedy.py:17-131): the step's logits of every unfinished clip
// with the EOS floor (:96-97) and the forbid-repeat mask (:99-105) applied, (-inf, pad_id -> 0) for finished clips (:64-69)
__global__ __launch_bounds__(256) void cn_greedy_logits_kernel(const float* __restrict__ logits, int ldv, int V, int maxp,
                                                               int step, int min_pred, int eos_id, int pad_id,
                                                               const uint8_t* __restrict__ forbid,
                                                               const int* __restrict__ n_active,
                                                               const int* __restrict__ prefix, float* __restrict__ out) {
  __shared__ int s_tok[CN_MAX_PRED + 1];
  const int b = blockIdx.y;
  const bool active = n_active[b] > 0;
  for (int j = threadIdx.x; j <= step; j += 256) s_tok[j] = prefix[(size_t)b * (maxp + 1) + j];
  __syncthreads();
  float* o = out + ((size_t)b * maxp + step) * V;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
    float x;
    if (!active) {
      x = v == pad_id ? 0.f : -INFINITY;
    } else {
      x = logits[(size_t)b * ldv + v];
      if (step < min_pred && v == eos_id) x = -INFINITY;
      if (forbid != nullptr && forbid[v]) {
        bool seen = false;
        for (int j = 0; j <= step; ++j) seen |= s_tok[j] == v;
        if (seen) x = -INFINITY;
      }
    }
    o[v] = x;
  }
}

struct DecWs {
  void *fe_t, *mem, *kvc, *xt, *attn_t, *ffh, *kc, *vc;
  float *x, *x2, *qkv, *q, *tmp, *logits, *slabs;
  int *n_active, *slot, *prefix, *anc, *cur_tok, *out_len, *eos_idx;
  int* live;  // [CN_MAX_PRED + 2] rows still searching when step s starts (summed by the search step of s - 1)
  unsigned long long* kvalid;  // teacher forcing: bit s of row r = caption position s is not padding
  float* sum_lp;
  int ldv;
  size_t total;
};

static DecWs dec_ws(const conette_ctx* ctx, int B, int Ta, int beam, int maxp, char* base) {
  const size_t es = ctx->esize;
  const int d = ctx->cfg.d_model, NL = ctx->cfg.n_layers, R = B * beam;
  DecWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += cn_align(bytes);
    return p;
  };
  w.ldv = (ctx->cfg.vocab_size + 7) / 8 * 8;
  w.fe_t = take((size_t)B * Ta * CN_FEAT * es);
  w.mem = take((size_t)B * Ta * d * es);
  w.kvc = take((size_t)B * Ta * NL * 2 * d * es);
  w.x = (float*)take((size_t)R * d * 4);
  w.x2 = (float*)take((size_t)R * d * 4);
  w.xt = take((size_t)R * d * es);
  w.qkv = (float*)take((size_t)R * 3 * d * 4);
  w.q = (float*)take((size_t)R * d * 4);
  w.attn_t = take((size_t)R * d * es);
  w.tmp = (float*)take((size_t)R * d * 4);
  w.slabs = (float*)take((size_t)FF2_SPLITS * R * d * 4);
  w.ffh = take((size_t)R * ctx->cfg.d_ff * es);
  w.logits = (float*)take((size_t)R * w.ldv * 4);
  w.kc = take((size_t)NL * maxp * R * d * es);
  w.vc = take((size_t)NL * maxp * R * d * es);
  w.n_active = (int*)take((size_t)B * 4);
  w.live = (int*)take((size_t)(CN_MAX_PRED + 2) * 4);
  w.slot = (int*)take((size_t)R * 4);
  w.sum_lp = (float*)take((size_t)R * 4);
  w.prefix = (int*)take((size_t)R * (maxp + 1) * 4);
  w.anc = (int*)take((size_t)R * maxp * 4);
  w.cur_tok = (int*)take((size_t)R * 4);
  w.out_len = (int*)take((size_t)R * 4);
  w.eos_idx = (int*)take((size_t)B * 4);
  w.kvalid = (unsigned long long*)take((size_t)R * 8);
  w.total = off;
  return w;
}

extern "C" size_t conette_decode_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t t_audio, int32_t beam,
                                                 int32_t max_pred) {
  return dec_ws(ctx, batch, t_audio, beam, max_pred, nullptr).total;
}

template <typename T>
static int decode_impl(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens, const int32_t* bos_ids,
                       const uint8_t* forbid, int B, int Ta, int beam, int min_pred, int maxp, int32_t* best_preds,
                       float* best_lprobs, int32_t* mult_preds, float* mult_lprobs, int32_t* out_sizes,
                       float* step0_logits, int32_t* trace_sel, float* trace_val, char* wsp, hipStream_t s,
                       const int32_t* force_caps = nullptr, float* force_logits = nullptr,
                       float* greedy_logits = nullptr, float* margins = nullptr) {
  const conette_config& cfg = ctx->cfg;
  const int d = cfg.d_model, NL = cfg.n_layers, R = B * beam, V = cfg.vocab_size, dff = cfg.d_ff;
  DecWs w = dec_ws(ctx, B, Ta, beam, maxp, wsp);
  const bool forcing = force_caps != nullptr;  // beam == 1, maxp == caption length, no search
  const unsigned long long* kvalid = forcing ? w.kvalid : nullptr;
  T* fe_t = (T*)w.fe_t;
  T* mem = (T*)w.mem;
  T* kvc = (T*)w.kvc;
  T* xt = (T*)w.xt;
  T* attn_t = (T*)w.attn_t;
  T* ffh = (T*)w.ffh;
  const int kv_ld = NL * 2 * d;
  const float scale = 1.0f / sqrtf((float)(d / cfg.nhead));
  const int rblocks = cn_cdiv(R, 4);

  // ---- once per batch: projection (conette.py:457) and cross K/V of every layer ---------------
  {
    CnProfScope ps(ctx, CONETTE_PROF_DEC_PREPARE, s);
    const size_t n = (size_t)B * Ta * CN_FEAT;
    hipLaunchKernelGGL((cn_cvt_kernel<T>), dim3((unsigned)((n + 1023) / 1024 < 4096 ? (n + 1023) / 1024 : 4096)), dim3(256), 0, s,
                       frame_embs, fe_t, n);
    CN_LAUNCH_CHECK();
    EpiBiasAct<T, ACT_RELU> ep{ctx->proj_b, mem, d, ACT_RELU};
    CN_TRY(cn_mm(fe_t, CN_FEAT, (const T*)ctx->proj_w, CN_FEAT, B * Ta, d, CN_FEAT, ep, s));
    EpiBiasAct<T, ACT_NONE> ekv{ctx->kv_b, kvc, kv_ld, ACT_NONE};
    CN_TRY(cn_mm(mem, d, (const T*)ctx->kv_w, d, B * Ta, kv_ld, d, ekv, s));
  }
  hipLaunchKernelGGL(cn_init_state_kernel, dim3(64), dim3(256), 0, s, B, beam, maxp, bos_ids, w.n_active, w.slot,
                     w.sum_lp, w.prefix, w.anc, w.cur_tok, mult_preds, mult_lprobs, w.out_len, out_sizes, cfg.pad_id,
                     trace_sel, trace_val, w.live, margins);
  CN_LAUNCH_CHECK();
  if (forcing) {
    hipLaunchKernelGGL(cn_force_init_kernel, dim3(cn_cdiv(B, 64)), dim3(64), 0, s, force_caps, B, maxp, cfg.pad_id,
                       w.kvalid, w.anc);
    CN_LAUNCH_CHECK();
  }

  for (int step = 0; step < maxp; ++step) {
    if (forcing) {
      hipLaunchKernelGGL(cn_force_tok_kernel, dim3(cn_cdiv(B, 64)), dim3(64), 0, s, force_caps, B, maxp, step, w.cur_tok);
      CN_LAUNCH_CHECK();
    }
    // fused path (dec_block.h + dec_ffn.h): 16-bit operands, and since round 4 the exact precision (hi / lo passes of the same
    // two kernels; needs both packed streams, i.e. d_ff % 256 == 0 and d_ff <= 2048)
    constexpr bool kBlockT = CnIsH16<T>::value || std::is_same<T, sp16_t>::value;
    const bool block_path = kBlockT && !ctx->dec_unfused && ctx->layers[0].blk_w != nullptr &&
                            (CnIsH16<T>::value || (ctx->layers[0].ffn_w != nullptr && dff / 256 <= FF2_SPLITS));
    // device-side early exit: once every hypothesis of every clip has finished the reference leaves its loop
    // (beam.py:192-194); the launch sequence is static (hipGraph), so the heavy kernels of the remaining steps read the
    // number of rows still searching and return at once
    const int* gate = (!forcing && step > 0) ? w.live + step : nullptr;
    if (!block_path) {  // (the block path embeds in the prologue of layer 0's block kernel)
      hipLaunchKernelGGL((cn_embed_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.cur_tok, ctx->emb, ctx->pe, step, R,
                         sqrtf((float)d), w.x, xt);
      CN_LAUNCH_CHECK();
    }
    bool fused_done = false;
#ifdef CN_G2_PROF  // profiling build: phase stamps of the block kernel (tools/dbprof.py)
    static const int db_debug = getenv("CN_DB_DEBUG") ? atoi(getenv("CN_DB_DEBUG")) : 0;
#else
    constexpr int db_debug = 0;
#endif
    if constexpr (kBlockT) if (block_path) {
      // default path: 3 launches per layer -- fused block (embedding | previous LN3, QKV, self-attention,
      // cross-attention: dec_block.h), FFN1 GEMM + GELU, FFN2 split-K slabs (summed by the next layer's block
      // prologue; the last layer's by the LN3 kernel in front of the classifier)
      fused_done = true;
      int splits = ff2_splits_default();
      if (splits < 1 || splits > FF2_SPLITS || splits > 8 || dff % (splits * 64) != 0) splits = 1;
      // fused FFN (dec_ffn.h): one launch per layer, one slab per 256-wide hidden chunk
      const bool ffn_fused = ctx->layers[0].ffn_w != nullptr && dff / 256 <= FF2_SPLITS;
      if (ffn_fused) splits = dff / 256;
      const size_t slab = (size_t)R * d;
      for (int l = 0; l < NL; ++l) {
        const CnLayerW& lw = ctx->layers[l];
        T* kc = (T*)w.kc + (size_t)l * maxp * R * d;
        T* vc = (T*)w.vc + (size_t)l * maxp * R * d;
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_ATTN, s);
          DbPrologue pro;
          pro.tok = w.cur_tok, pro.emb = ctx->emb, pro.pe_row = ctx->pe + (size_t)step * d, pro.emb_scale = sqrtf((float)d);
          pro.slabs = nullptr, pro.nslab = 0, pro.slab_stride = slab, pro.b2_prev = nullptr, pro.g3 = nullptr, pro.b3 = nullptr;
          if (l > 0) {
            const CnLayerW& pw = ctx->layers[l - 1];
            pro.slabs = w.slabs, pro.nslab = splits, pro.b2_prev = pw.ff2_b, pro.g3 = pw.n3w, pro.b3 = pw.n3b;
          }
          DbWeights wt;
          wt.stream = lw.blk_w;
          wt.params = lw.blk_p;
          // rows per block: DbOp<T>::ROWS (4: many small blocks spread a short search over the chip); a wide search -- R >= DB_WIDE_R
          // rows, the grouped search of four 64-clip batches -- runs 8 rows per block (two per row wave): half the blocks, half the
          // weight re-streaming from L2, +1.1 % end to end beside the encoder (profiles/r05_notes.md section 8).  Row-local arithmetic:
          // the same bits either way.
          auto launch_block = [&](auto nr_tag) -> int {
            constexpr int NR = decltype(nr_tag)::value;
            CN_TRY((cn_dec_block_setup<T, NR>()));
            hipLaunchKernelGGL((cn_dec_block_kernel<T, NR>), dim3(DB_XCDS < 8 ? 8 * cn_cdiv(cn_cdiv(R, NR), DB_XCDS) : cn_cdiv(R, NR)), dim3((DbL<T, NR>::THREADS)), (DbL<T, NR>::BYTES), s, pro, wt, kc, vc,
                               w.anc, step, R, beam, maxp, (const T*)kvc, kv_ld, l * 2 * d, frame_lens, Ta, w.x, xt,
                               scale, kvalid, db_debug, gate);
            CN_LAUNCH_CHECK();
            return CN_OK;
          };
          constexpr int kWide = CnIsH16<T>::value ? DB_WIDE_ROWS : DB_WIDE_ROWS_SP;
          if (kWide != DbOp<T>::ROWS && R >= DB_WIDE_R) CN_TRY(launch_block(std::integral_constant<int, kWide>{}));
          else CN_TRY(launch_block(std::integral_constant<int, DbOp<T>::ROWS>{}));
        }
        if (ffn_fused) {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
          CN_TRY(cn_dec_ffn_setup<T>());
          hipLaunchKernelGGL(cn_dec_ffn_kernel<T>, dim3(dff / 256, cn_cdiv(R, DF_ROWS)), dim3(256), DF_LDS_BYTES_T(DbOp<T>::NPH), s, (const T*)xt, R,
                             lw.ffn_w, lw.ff1_b, w.slabs, slab, gate);
          CN_LAUNCH_CHECK();
        } else if constexpr (CnIsH16<T>::value) {
          {
            CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
            EpiBiasAct<T, ACT_GELU_FAST> e1{lw.ff1_b, ffh, dff, ACT_GELU_FAST};
            CN_TRY(cn_gemm2(xt, d, (const T*)lw.ff1_w, d, R, dff, d, e1, s));
          }
          {
            CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
            EpiSlab e2{w.slabs, d, slab};
            CN_TRY(cn_gemm2(ffh, dff, (const T*)lw.ff2_w, dff, R, d, dff, e2, s, splits));
          }
        }
      }
      {
        CnProfScope ps(ctx, CONETTE_PROF_DEC_MISC, s);
        const CnLayerW& lw = ctx->layers[NL - 1];
        hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.slabs, splits, slab, lw.ff2_b, w.x,
                           lw.n3w, lw.n3b, R, w.x, xt);
        CN_LAUNCH_CHECK();
      }
      {
        CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
        EpiBiasAct<float, ACT_NONE> ec{ctx->cls_b, w.logits, w.ldv, ACT_NONE};
        CN_TRY(cn_mm(xt, d, (const T*)ctx->cls_w, d, R, V, d, ec, s));
      }
    }
    if (!fused_done) {
      for (int l = 0; l < NL; ++l) {
        const CnLayerW& lw = ctx->layers[l];
        T* kc = (T*)w.kc + (size_t)l * maxp * R * d;
        T* vc = (T*)w.vc + (size_t)l * maxp * R * d;
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
          EpiBiasAct<float, ACT_NONE> eq{lw.sa_in_b, w.qkv, 3 * d, ACT_NONE};
          CN_TRY(cn_mm(xt, d, (const T*)lw.sa_in_w, d, R, 3 * d, d, eq, s));
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_ATTN, s);
          hipLaunchKernelGGL((cn_self_attn_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.qkv, kc, vc, w.anc, step, R,
                             beam, maxp, scale, attn_t, kvalid);
          CN_LAUNCH_CHECK();
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
          EpiResid eo{lw.sa_out_b, nullptr, w.x, w.tmp, d};
          CN_TRY(cn_mm(attn_t, d, (const T*)lw.sa_out_w, d, R, d, d, eo, s));
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_MISC, s);
          hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.tmp, 1, (size_t)0, (const float*)nullptr,
                             (const float*)nullptr, lw.n1w, lw.n1b, R, w.x, xt);
          CN_LAUNCH_CHECK();
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
          EpiBiasAct<float, ACT_NONE> ecq{lw.ca_q_b, w.q, d, ACT_NONE};
          CN_TRY(cn_mm(xt, d, (const T*)lw.ca_q_w, d, R, d, d, ecq, s));
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_ATTN, s);
          hipLaunchKernelGGL((cn_cross_attn_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.q, kvc, kv_ld, l * 2 * d,
                             frame_lens, R, beam, Ta, scale, attn_t);
          CN_LAUNCH_CHECK();
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
          EpiResid eco{lw.ca_out_b, nullptr, w.x, w.tmp, d};
          CN_TRY(cn_mm(attn_t, d, (const T*)lw.ca_out_w, d, R, d, d, eco, s));
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_MISC, s);
          hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.tmp, 1, (size_t)0, (const float*)nullptr,
                             (const float*)nullptr, lw.n2w, lw.n2b, R, w.x, xt);
          CN_LAUNCH_CHECK();
        }
        {
          CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
          EpiBiasAct<T, CnGeluAct<T>::value> e1{lw.ff1_b, ffh, dff, CnGeluAct<T>::value};
          CN_TRY(cn_mm(xt, d, (const T*)lw.ff1_w, d, R, dff, d, e1, s));
        }
        if constexpr (CnIsH16<T>::value) {
          // K = d_ff is long and M = R is small: split K over blockIdx.y into partial slabs, summed
          // (fixed order, with bias + residual) by the LayerNorm kernel that follows
          const int splits = (dff % (FF2_SPLITS * 64) == 0) ? FF2_SPLITS : 1;
          {
            CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
            EpiSlab e2{w.slabs, d, (size_t)R * d};
            CN_TRY(cn_gemm2(ffh, dff, (const T*)lw.ff2_w, dff, R, d, dff, e2, s, splits));
          }
          CnProfScope ps(ctx, CONETTE_PROF_DEC_MISC, s);
          hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.slabs, splits, (size_t)R * d,
                             lw.ff2_b, w.x, lw.n3w, lw.n3b, R, w.x, xt);
          CN_LAUNCH_CHECK();
        } else if constexpr (std::is_same<T, sp16_t>::value) {
          // exact precision: the same split-K as bf16 (M = R rows, K = d_ff: 12 blocks looping over 64 k-tiles took 30 us,
          // a quarter of the whole decode) -- partial slabs, summed in a fixed order with bias + residual by the LayerNorm
          const int splits = (dff % (FF2_SPLITS * 32) == 0) ? FF2_SPLITS : 1;
          {
            CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
            EpiSlab e2{w.slabs, d, (size_t)R * d};
            CN_TRY(cn_gemm2_sp((const sp16_t*)ffh, dff, (const sp16_t*)lw.ff2_w, dff, R, d, dff, e2, s, splits));
          }
          CnProfScope ps(ctx, CONETTE_PROF_DEC_MISC, s);
          hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.slabs, splits, (size_t)R * d,
                             lw.ff2_b, w.x, lw.n3w, lw.n3b, R, w.x, xt);
          CN_LAUNCH_CHECK();
        } else {
          {
            CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
            EpiResid e2{lw.ff2_b, nullptr, w.x, w.tmp, d};
            CN_TRY(cn_mm(ffh, dff, (const T*)lw.ff2_w, dff, R, d, dff, e2, s));
          }
          CnProfScope ps(ctx, CONETTE_PROF_DEC_MISC, s);
          hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.tmp, 1, (size_t)0,
                             (const float*)nullptr, (const float*)nullptr, lw.n3w, lw.n3b, R, w.x, xt);
          CN_LAUNCH_CHECK();
        }
      }
      {
        CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
        EpiBiasAct<float, ACT_NONE> ec{ctx->cls_b, w.logits, w.ldv, ACT_NONE};
        CN_TRY(cn_mm(xt, d, (const T*)ctx->cls_w, d, R, V, d, ec, s));
      }
    }
    if (step == 0 && step0_logits)
      CN_HIP(hipMemcpyAsync(step0_logits, w.logits, (size_t)R * w.ldv * 4, hipMemcpyDeviceToDevice, s));
    if (forcing) {
      hipLaunchKernelGGL(cn_force_store_kernel, dim3(cn_cdiv(V, 1024), R), dim3(256), 0, s, w.logits, w.ldv, V, maxp, step,
                         force_logits);
      CN_LAUNCH_CHECK();
      continue;
    }
    if (greedy_logits) {  // beam == 1: rows == clips; before the search step updates prefix / n_active
      hipLaunchKernelGGL(cn_greedy_logits_kernel, dim3(cn_cdiv(V, 1024), B), dim3(256), 0, s, w.logits, w.ldv, V, maxp, step,
                         min_pred, cfg.eos_id, cfg.pad_id, forbid, w.n_active, w.prefix, greedy_logits);
      CN_LAUNCH_CHECK();
    }
    CnProfScope ps_search(ctx, CONETTE_PROF_SEARCH, s);
    if (V <= S3_T * S3_VPT && beam <= 8) {  // register-resident step (one block of 1024 threads per clip)
#define S3_LAUNCH(NR_, VPT_)                                                                                          \
  hipLaunchKernelGGL((cn_search_step3_kernel<NR_, VPT_>), dim3(B), dim3(S3_T), 0, s, w.logits, w.ldv, V, beam, maxp,  \
                     step, min_pred, cfg.eos_id, forbid, w.n_active, w.slot, w.sum_lp, w.prefix, w.anc, w.cur_tok,    \
                     mult_preds, mult_lprobs, w.out_len, trace_sel, trace_val, db_debug, w.live, margins)
      const int vpt = cn_cdiv(V, S3_T);
      if (beam <= 4) {
        if (vpt <= 2) S3_LAUNCH(4, 2);
        else if (vpt <= 4) S3_LAUNCH(4, 4);
        else if (vpt <= 6) S3_LAUNCH(4, 6);
        else S3_LAUNCH(4, 8);
      } else {
        if (vpt <= 4) S3_LAUNCH(8, 4);
        else S3_LAUNCH(8, 8);
      }
#undef S3_LAUNCH
    } else {  // vocabularies beyond 8192 entries, beams beyond 8: the generic step (masking in place, top-k over global memory)
      hipLaunchKernelGGL(cn_search_step_kernel, dim3(B), dim3(256), 0, s, w.logits, w.ldv, V, beam, maxp, step,
                         min_pred, cfg.eos_id, forbid, w.n_active, w.slot, w.sum_lp, w.prefix, w.anc, w.cur_tok,
                         mult_preds, mult_lprobs, w.out_len, trace_sel, trace_val, w.live, margins);
    }
    CN_LAUNCH_CHECK();
  }
  if (forcing) return CN_OK;
  hipLaunchKernelGGL(cn_finalize_kernel, dim3(cn_cdiv(B, 64)), dim3(64), 0, s, B, beam, maxp, cfg.eos_id, mult_preds,
                     mult_lprobs, w.out_len, best_preds, best_lprobs, w.eos_idx, out_sizes, margins);
  CN_LAUNCH_CHECK();
  hipLaunchKernelGGL(cn_finalize2_kernel, dim3(1), dim3(256), 0, s, B, w.eos_idx, out_sizes);
  CN_LAUNCH_CHECK();
  return CN_OK;
}

// ---- hipGraph replay of the (static) decode launch sequence ---------------------------------------
struct DecKey {
  const void *fe, *lens, *bos, *forbid, *bp, *bl, *mp, *ml, *sz, *s0, *ts, *tv, *mg, *ws;
  int B, Ta, beam, min_pred, maxp;
  bool operator==(const DecKey& o) const { return memcmp(this, &o, sizeof(DecKey)) == 0; }
};
struct DecGraph {
  DecKey key;
  hipGraphExec_t exec;
  hipGraph_t graph;
  hipEvent_t last;  // recorded behind every launch of `exec`: what an eviction waits for (never the whole device)
  int seen;         // 1: seen once (ran eagerly); 2: being captured by some thread (others run eagerly meanwhile)
  int nodes;        // kernel nodes of `graph`
};
static void dec_graph_release(DecGraph& g) {
  if (!g.exec) return;
  // the graph may still be executing on another stream: wait for ITS last launch.  (Not hipDeviceSynchronize: that is illegal
  // while any stream of the process is being captured -- e.g. another thread's first capture -- and would invalidate it.)
  if (g.last) {
    if (hipEventSynchronize(g.last) != hipSuccess) (void)hipGetLastError();
    (void)hipEventDestroy(g.last);
  }
  (void)hipGraphExecDestroy(g.exec);
  (void)hipGraphDestroy(g.graph);
  g.exec = nullptr, g.graph = nullptr, g.last = nullptr;
}
#define CN_MAX_DEC_GRAPHS 64  // (bucket, pipeline slot) keys: conette_amd.engine sizes its buffer cache from the same number
// The cache is shared by every host thread that decodes on this context (the pipelines run two or three decode streams,
// usually from one thread, but nothing in the ABI says so).  Its lock is PER CONTEXT and covers only the table itself --
// lookup, LRU order, insertion, the hand-over of an evicted entry: the ~300 eager launches of a first sighting, stream
// capture + instantiation and the blocking wait for an evicted graph's last launch all run outside it, so threads that
// drive other contexts (other GPUs) never meet, and an eviction stalls nobody but the caller that caused it.
struct DecGraphCache {
  std::mutex mu;
  DecGraph g[CN_MAX_DEC_GRAPHS] = {};
  int n = 0;
  int enabled = 1;
  int last_nodes = 0;  // nodes of the graph most recently launched or captured (conette_decode_graph_nodes)
};
static std::mutex g_dec_cache_create_mu;   // only the creation / destruction of a context's cache object
static DecGraphCache* graph_cache(conette_ctx* ctx, bool create) {
  std::lock_guard<std::mutex> lock(g_dec_cache_create_mu);
  if (ctx->dec_graphs == nullptr && create) ctx->dec_graphs = new DecGraphCache();
  return (DecGraphCache*)ctx->dec_graphs;
}
void cn_decode_graphs_free(conette_ctx* ctx) {
  DecGraphCache* c = graph_cache(ctx, false);
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(c->mu);
    for (int i = 0; i < c->n; ++i) dec_graph_release(c->g[i]);
    c->n = 0;
  }
  std::lock_guard<std::mutex> lock(g_dec_cache_create_mu);
  delete c;
  ctx->dec_graphs = nullptr;
}
extern "C" int conette_set_option(conette_ctx* ctx, int32_t option, int32_t value) {
  if (!ctx) return CN_ERR_ARG;
  if (option == CONETTE_OPT_DECODE_GRAPH) {
    DecGraphCache* c = graph_cache(ctx, true);
    if (c) {
      std::lock_guard<std::mutex> lock(c->mu);
      c->enabled = value ? 1 : 0;
    }
    return CN_OK;
  }
  if (option == CONETTE_OPT_DECODE_FUSION) {
    ctx->dec_unfused = value ? 0 : 1;
    DecGraphCache* c = graph_cache(ctx, false);
    if (c) {  // cached graphs hold the other launch sequence
      std::lock_guard<std::mutex> lock(c->mu);
      for (int i = 0; i < c->n; ++i) dec_graph_release(c->g[i]);
      c->n = 0;
    }
    return CN_OK;
  }
  if (option == CONETTE_OPT_FORCING_STEPWISE) {
    ctx->forcing_stepwise = value ? 1 : 0;
    return CN_OK;
  }
  if (option == CONETTE_OPT_ENCODE_RESERVED_CUS) {
    if (value < 0 || value > ctx->n_cu / 2) {
      cn_set_error("set_option: reserved CUs %d outside 0 .. %d", value, ctx->n_cu / 2);
      return CN_ERR_ARG;
    }
    ctx->enc_reserved_cus = value;
    return CN_OK;
  }
  cn_set_error("set_option: unknown option %d", option);
  return CN_ERR_ARG;
}

extern "C" int conette_decode(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens,
                              const int32_t* bos_ids, const uint8_t* forbid_mask, int32_t batch, int32_t t_audio,
                              int32_t beam, int32_t min_pred, int32_t max_pred, int32_t* best_preds,
                              float* best_lprobs, int32_t* mult_preds, float* mult_lprobs, int32_t* out_sizes,
                              float* step0_logits, int32_t* trace_sel, float* trace_val, float* margins,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!ctx || !frame_embs || !frame_lens || !bos_ids || !best_preds || !best_lprobs || !mult_preds || !mult_lprobs ||
      !out_sizes || !workspace || batch <= 0 || t_audio <= 0) {
    cn_set_error("decode: bad argument");
    return CN_ERR_ARG;
  }
  if (beam < 1 || beam > CN_MAX_BEAM || max_pred < 1 || max_pred > CN_MAX_PRED || min_pred < 0) {
    cn_set_error("decode: beam=%d (1..%d) max_pred=%d (1..%d) min_pred=%d unsupported", beam, CN_MAX_BEAM, max_pred,
                 CN_MAX_PRED, min_pred);
    return CN_ERR_ARG;
  }
  if (ctx->cfg.d_model != 256 || ctx->cfg.nhead != 8) {
    cn_set_error("decode: kernels are specialised for d_model=256, nhead=8");
    return CN_ERR_ARG;
  }
  if (max_pred > ctx->pe_len) {
    cn_set_error("decode: max_pred exceeds positional table");
    return CN_ERR_ARG;
  }
  const size_t need = conette_decode_workspace_bytes(ctx, batch, t_audio, beam, max_pred);
  if (workspace_bytes < need) {
    cn_set_error("decode: workspace %zu < %zu", workspace_bytes, need);
    return CN_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  auto run = [&]() -> int {
    CN_BY_PRECISION(ctx, decode_impl<OT>(ctx, frame_embs, frame_lens, bos_ids, forbid_mask, batch, t_audio, beam, min_pred,
                                 max_pred, best_preds, best_lprobs, mult_preds, mult_lprobs, out_sizes, step0_logits,
                                 trace_sel, trace_val, (char*)workspace, s, nullptr, nullptr, nullptr, margins));
  };
  DecGraphCache* cache = graph_cache(ctx, true);
  const uint32_t dec_classes = (1u << CONETTE_PROF_DEC_PREPARE) | (1u << CONETTE_PROF_DEC_GEMM) |
                               (1u << CONETTE_PROF_DEC_ATTN) | (1u << CONETTE_PROF_DEC_MISC) |
                               (1u << CONETTE_PROF_SEARCH);
  if (!cache || (ctx->prof_mask & dec_classes) != 0) return run();

  DecKey key;
  memset(&key, 0, sizeof(key));
  key.fe = frame_embs, key.lens = frame_lens, key.bos = bos_ids, key.forbid = forbid_mask, key.bp = best_preds;
  key.bl = best_lprobs, key.mp = mult_preds, key.ml = mult_lprobs, key.sz = out_sizes, key.s0 = step0_logits;
  key.ts = trace_sel, key.tv = trace_val, key.mg = margins, key.ws = workspace;
  key.B = batch, key.Ta = t_audio, key.beam = beam, key.min_pred = min_pred, key.maxp = max_pred;
  auto find = [&]() -> DecGraph* {  // (under the lock) least recently used first: a hit moves to the back
    for (int i = 0; i < cache->n; ++i)
      if (cache->g[i].key == key) {
        const DecGraph hit = cache->g[i];
        memmove(&cache->g[i], &cache->g[i + 1], sizeof(DecGraph) * (cache->n - 1 - i));
        cache->g[cache->n - 1] = hit;
        return &cache->g[cache->n - 1];
      }
    return nullptr;
  };
  enum { EAGER, CAPTURE } todo = EAGER;
  DecGraph evicted;
  memset(&evicted, 0, sizeof(evicted));
  {
    std::lock_guard<std::mutex> lock(cache->mu);
    if (!cache->enabled) {
      todo = EAGER;
    } else if (DecGraph* e = find()) {
      if (e->exec) {  // replay (the launch and the event record stay under the lock: an eviction must see this launch's event)
        CN_HIP(hipGraphLaunch(e->exec, s));
        CN_HIP(hipEventRecord(e->last, s));
        cache->last_nodes = e->nodes;
        return CN_OK;
      }
      if (e->seen == 1) {  // second sighting: this call captures; the same key from another thread meanwhile runs eagerly
        e->seen = 2;
        todo = CAPTURE;
      }
    } else {  // first sighting: run eagerly (also performs the one-time kernel attribute setup)
      if (cache->n == CN_MAX_DEC_GRAPHS) {
        // Evict the least recently used entry.  Its graph may still be executing on another stream (two or three decode
        // streams replay graphs side by side), and destroying an executable graph under a running launch is undefined: its
        // own last launch is waited for (an event per graph) -- below, after the lock has been dropped.  This is the one
        // place an entry point blocks, and it is off the steady state (more than CN_MAX_DEC_GRAPHS distinct (shape, buffer)
        // keys alive at once); conette_set_option(DECODE_GRAPH, 0) avoids it.
        evicted = cache->g[0];
        memmove(&cache->g[0], &cache->g[1], sizeof(DecGraph) * (CN_MAX_DEC_GRAPHS - 1));
        cache->n--;
      }
      DecGraph* fresh = &cache->g[cache->n++];
      memset(fresh, 0, sizeof(*fresh));
      fresh->key = key;
      fresh->seen = 1;
    }
  }
  dec_graph_release(evicted);
  if (todo == EAGER) return run();

  // capture, instantiate, launch -- outside the lock; the entry is found again by its key when the graph exists
  auto give_up = [&]() -> int {   // capture is not available here: every later call runs eagerly
    std::lock_guard<std::mutex> lock(cache->mu);
    cache->enabled = 0;
    return CN_OK;
  };
  hipError_t ce = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  if (ce != hipSuccess) {
    (void)hipGetLastError();
    give_up();
    return run();
  }
  const int rc = run();
  hipGraph_t graph = nullptr;
  ce = hipStreamEndCapture(s, &graph);
  if (rc != CN_OK || ce != hipSuccess || graph == nullptr) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    give_up();
    return run();
  }
  hipGraphExec_t exec = nullptr;
  ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (ce != hipSuccess || exec == nullptr) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(graph);
    give_up();
    return run();
  }
  hipEvent_t last = nullptr;
  if (hipEventCreateWithFlags(&last, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    give_up();
    return run();
  }
  size_t n_nodes = 0;
  if (hipGraphGetNodes(graph, nullptr, &n_nodes) != hipSuccess) (void)hipGetLastError();
  {
    std::lock_guard<std::mutex> lock(cache->mu);
    DecGraph* e = find();
    if (e && !e->exec) {
      e->exec = exec, e->graph = graph, e->last = last, e->nodes = (int)n_nodes;
      cache->last_nodes = e->nodes;
      CN_HIP(hipGraphLaunch(exec, s));
      CN_HIP(hipEventRecord(last, s));
      return CN_OK;
    }
  }
  // (the entry was evicted, or the cache reset, while this call captured: launch once and drop the graph)
  CN_HIP(hipGraphLaunch(exec, s));
  CN_HIP(hipEventRecord(last, s));
  DecGraph tmp;
  memset(&tmp, 0, sizeof(tmp));
  tmp.exec = exec, tmp.graph = graph, tmp.last = last;
  dec_graph_release(tmp);
  return CN_OK;
}

extern "C" int32_t conette_decode_graph_nodes(const conette_ctx* ctx) {
  DecGraphCache* c = ctx ? (DecGraphCache*)ctx->dec_graphs : nullptr;
  if (!c) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  return c->last_nodes;
}

// ---- teacher forcing as ONE causal pass (nn/decoding/forcing.py:12-71 is a single decoder forward over the caption) ----
// Rows r = clip * cap_len + position: every GEMM of the layer runs once on all B * cap_len rows (the unfused per-sub-layer
// kernels of the step path with M = B * cap_len), the cross-attention kernel is reused with "beam" = cap_len, and the
// self-attention reads the K / V of the clip's earlier positions from this pass's own QKV output -- at cache precision
// (rounded to the operand type exactly as the step path stores and re-reads them), masked to valid (non-pad) keys <= t.
template <typename T>
__global__ __launch_bounds__(256) void cn_embed_caps_kernel(const int32_t* __restrict__ caps, const float* __restrict__ emb,
                                                            const float* __restrict__ pe, int cap_len, int R, float scale,
                                                            float* __restrict__ x, T* __restrict__ xt) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const int t = r % cap_len;
  const f32x4 e = *(const f32x4*)(emb + (size_t)caps[r] * 256 + 4 * lane);
  const f32x4 p = *(const f32x4*)(pe + (size_t)t * 256 + 4 * lane);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = e[i] * scale + p[i];
  *(f32x4*)(x + (size_t)r * 256 + 4 * lane) = o;
  cn_store4(xt + (size_t)r * 256 + 4 * lane, o[0], o[1], o[2], o[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void cn_self_attn_causal_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ caps,
                                                                  int cap_len, int R, int pad_id, float scale,
                                                                  T* __restrict__ out) {
  constexpr int NB = 8;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const int b = r / cap_len, t = r % cap_len;
  f32x4 q = *(const f32x4*)(qkv + (size_t)r * 768 + 4 * lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] *= scale;
  const float* base = qkv + (size_t)b * cap_len * 768 + 256 + 4 * lane;
  const int32_t* cap = caps + (size_t)b * cap_len;
  float m = -INFINITY, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 <= t; s0 += NB) {
    f32x4 kk[NB], vv[NB];
    float sc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int sidx = min(s0 + u, t);
      const f32x4 k4 = *(const f32x4*)(base + (size_t)sidx * 768), v4 = *(const f32x4*)(base + (size_t)sidx * 768 + 256);
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // cache precision
        kk[u][i] = cn_to_f32(cn_from_f32<T>(k4[i]));
        vv[u][i] = cn_to_f32(cn_from_f32<T>(v4[i]));
      }
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) sc[u] = (s0 + u <= t && cap[min(s0 + u, t)] != pad_id) ? cn_dot8(q, kk[u]) : -INFINITY;
    cn_attn_update<NB>(sc, vv, m, l, acc);
  }
  const float inv = 1.0f / l;
  cn_store4(out + (size_t)r * 256 + 4 * lane, acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
}

template <typename T>
static int forcing_prefill_impl(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens, const int32_t* caps,
                                int B, int Ta, int cap_len, float* logits, char* wsp, hipStream_t s) {
  const conette_config& cfg = ctx->cfg;
  const int d = cfg.d_model, NL = cfg.n_layers, R = B * cap_len, V = cfg.vocab_size, dff = cfg.d_ff;
  DecWs w = dec_ws(ctx, B, Ta, cap_len, 1, wsp);  // rows = B * cap_len ("beam" = cap_len, one step)
  T* fe_t = (T*)w.fe_t;
  T* mem = (T*)w.mem;
  T* kvc = (T*)w.kvc;
  T* xt = (T*)w.xt;
  T* attn_t = (T*)w.attn_t;
  T* ffh = (T*)w.ffh;
  const int kv_ld = NL * 2 * d;
  const float scale = 1.0f / sqrtf((float)(d / cfg.nhead));
  const int rblocks = cn_cdiv(R, 4);
  {
    CnProfScope ps(ctx, CONETTE_PROF_DEC_PREPARE, s);
    const size_t n = (size_t)B * Ta * CN_FEAT;
    hipLaunchKernelGGL((cn_cvt_kernel<T>), dim3((unsigned)((n + 1023) / 1024 < 4096 ? (n + 1023) / 1024 : 4096)), dim3(256), 0, s,
                       frame_embs, fe_t, n);
    CN_LAUNCH_CHECK();
    EpiBiasAct<T, ACT_RELU> ep{ctx->proj_b, mem, d, ACT_RELU};
    CN_TRY(cn_mm(fe_t, CN_FEAT, (const T*)ctx->proj_w, CN_FEAT, B * Ta, d, CN_FEAT, ep, s));
    EpiBiasAct<T, ACT_NONE> ekv{ctx->kv_b, kvc, kv_ld, ACT_NONE};
    CN_TRY(cn_mm(mem, d, (const T*)ctx->kv_w, d, B * Ta, kv_ld, d, ekv, s));
  }
  hipLaunchKernelGGL((cn_embed_caps_kernel<T>), dim3(rblocks), dim3(256), 0, s, caps, ctx->emb, ctx->pe, cap_len, R,
                     sqrtf((float)d), w.x, xt);
  CN_LAUNCH_CHECK();
  for (int l = 0; l < NL; ++l) {
    const CnLayerW& lw = ctx->layers[l];
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
      EpiBiasAct<float, ACT_NONE> eq{lw.sa_in_b, w.qkv, 3 * d, ACT_NONE};
      CN_TRY(cn_mm(xt, d, (const T*)lw.sa_in_w, d, R, 3 * d, d, eq, s));
    }
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_ATTN, s);
      hipLaunchKernelGGL((cn_self_attn_causal_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.qkv, caps, cap_len, R, cfg.pad_id,
                         scale, attn_t);
      CN_LAUNCH_CHECK();
    }
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
      EpiResid eo{lw.sa_out_b, nullptr, w.x, w.tmp, d};
      CN_TRY(cn_mm(attn_t, d, (const T*)lw.sa_out_w, d, R, d, d, eo, s));
    }
    hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.tmp, 1, (size_t)0, (const float*)nullptr,
                       (const float*)nullptr, lw.n1w, lw.n1b, R, w.x, xt);
    CN_LAUNCH_CHECK();
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
      EpiBiasAct<float, ACT_NONE> ecq{lw.ca_q_b, w.q, d, ACT_NONE};
      CN_TRY(cn_mm(xt, d, (const T*)lw.ca_q_w, d, R, d, d, ecq, s));
    }
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_ATTN, s);
      hipLaunchKernelGGL((cn_cross_attn_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.q, kvc, kv_ld, l * 2 * d, frame_lens, R,
                         cap_len, Ta, scale, attn_t);
      CN_LAUNCH_CHECK();
    }
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
      EpiResid eco{lw.ca_out_b, nullptr, w.x, w.tmp, d};
      CN_TRY(cn_mm(attn_t, d, (const T*)lw.ca_out_w, d, R, d, d, eco, s));
    }
    hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.tmp, 1, (size_t)0, (const float*)nullptr,
                       (const float*)nullptr, lw.n2w, lw.n2b, R, w.x, xt);
    CN_LAUNCH_CHECK();
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
      EpiBiasAct<T, CnGeluAct<T>::value> e1{lw.ff1_b, ffh, dff, CnGeluAct<T>::value};
      CN_TRY(cn_mm(xt, d, (const T*)lw.ff1_w, d, R, dff, d, e1, s));
    }
    {
      CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
      EpiResid e2{lw.ff2_b, nullptr, w.x, w.tmp, d};
      CN_TRY(cn_mm(ffh, dff, (const T*)lw.ff2_w, dff, R, d, dff, e2, s));
    }
    hipLaunchKernelGGL((cn_ln256_kernel<T>), dim3(rblocks), dim3(256), 0, s, w.tmp, 1, (size_t)0, (const float*)nullptr,
                       (const float*)nullptr, lw.n3w, lw.n3b, R, w.x, xt);
    CN_LAUNCH_CHECK();
  }
  CnProfScope ps(ctx, CONETTE_PROF_DEC_GEMM, s);
  EpiBiasAct<float, ACT_NONE> ec{ctx->cls_b, logits, V, ACT_NONE};  // (B, cap_len, V) written in place
  CN_TRY(cn_mm(xt, d, (const T*)ctx->cls_w, d, R, V, d, ec, s));
  return CN_OK;
}

extern "C" size_t conette_forcing_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t t_audio,
                                                  int32_t cap_len) {
  if (!ctx || batch <= 0 || t_audio <= 0 || cap_len <= 0) return 0;
  // step path: decode workspace at beam 1 + the int32 scratch the search bookkeeping of a normal decode writes into;
  // one-pass path: the same buffers laid out for B * cap_len rows
  const size_t stepwise = conette_decode_workspace_bytes(ctx, batch, t_audio, 1, cap_len) + cn_align((size_t)batch * cap_len * 8) +
                          cn_align((size_t)batch * 16);
  const size_t onepass = dec_ws(ctx, batch, t_audio, cap_len, 1, nullptr).total;
  return stepwise > onepass ? stepwise : onepass;
}

extern "C" int conette_forcing(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens,
                               const int32_t* caps_in, int32_t batch, int32_t t_audio, int32_t cap_len, float* logits,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!ctx || !frame_embs || !frame_lens || !caps_in || !logits || !workspace || batch <= 0 || t_audio <= 0) {
    cn_set_error("forcing: bad argument");
    return CN_ERR_ARG;
  }
  if (cap_len < 1 || cap_len > CN_MAX_PRED || cap_len > ctx->pe_len) {
    cn_set_error("forcing: cap_len=%d unsupported (1..%d)", cap_len, CN_MAX_PRED);
    return CN_ERR_ARG;
  }
  if (ctx->cfg.d_model != 256 || ctx->cfg.nhead != 8) {
    cn_set_error("forcing: kernels are specialised for d_model=256, nhead=8");
    return CN_ERR_ARG;
  }
  const size_t need = conette_forcing_workspace_bytes(ctx, batch, t_audio, cap_len);
  if (workspace_bytes < need) {
    cn_set_error("forcing: workspace %zu < %zu", workspace_bytes, need);
    return CN_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!ctx->forcing_stepwise) {  // default: one causal pass over all caption positions
    CN_BY_PRECISION(ctx, forcing_prefill_impl<OT>(ctx, frame_embs, frame_lens, caps_in, batch, t_audio, cap_len, logits,
                                          (char*)workspace, s));
  }
  // CONETTE_OPT_FORCING_STEPWISE: the KV-cached step kernels fed with the caption (cross-check of the pass above)
  // outputs of the search bookkeeping that a forced pass does not produce: parked in the workspace tail
  char* tail = (char*)workspace + conette_decode_workspace_bytes(ctx, batch, t_audio, 1, cap_len);
  int32_t* mult_preds = (int32_t*)tail;
  float* mult_lprobs = (float*)(tail + cn_align((size_t)batch * cap_len * 4));
  tail += cn_align((size_t)batch * cap_len * 8);
  int32_t* sizes = (int32_t*)tail;
  int32_t* bos = sizes + 2;  // init kernel input; any valid ids: column 0 of the captions
  (void)bos;
  CN_BY_PRECISION(ctx, decode_impl<OT>(ctx, frame_embs, frame_lens, caps_in, nullptr, batch, t_audio, 1, 0, cap_len, mult_preds,
                               mult_lprobs, mult_preds, mult_lprobs, sizes, nullptr, nullptr, nullptr, (char*)workspace, s,
                               caps_in, logits));
}

extern "C" int conette_greedy(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens, const int32_t* bos_ids,
                              const uint8_t* forbid_mask, int32_t batch, int32_t t_audio, int32_t min_pred,
                              int32_t max_pred, float* logits, int32_t* preds, int32_t* out_sizes, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!ctx || !frame_embs || !frame_lens || !bos_ids || !logits || !preds || !out_sizes || !workspace || batch <= 0 ||
      t_audio <= 0) {
    cn_set_error("greedy: bad argument");
    return CN_ERR_ARG;
  }
  if (max_pred < 1 || max_pred > CN_MAX_PRED || max_pred > ctx->pe_len || min_pred < 0) {
    cn_set_error("greedy: max_pred=%d (1..%d) min_pred=%d unsupported", max_pred, CN_MAX_PRED, min_pred);
    return CN_ERR_ARG;
  }
  if (ctx->cfg.d_model != 256 || ctx->cfg.nhead != 8) {
    cn_set_error("greedy: kernels are specialised for d_model=256, nhead=8");
    return CN_ERR_ARG;
  }
  const size_t base = conette_decode_workspace_bytes(ctx, batch, t_audio, 1, max_pred);
  const size_t need = base + cn_align((size_t)batch * max_pred * 4) + 2 * cn_align((size_t)batch * 4);
  if (workspace_bytes < need) {
    cn_set_error("greedy: workspace %zu < %zu", workspace_bytes, need);
    return CN_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* tail = (char*)workspace + base;
  int32_t* mult_preds = (int32_t*)tail;  // beam 1: the single hypothesis of every clip
  float* mult_lprobs = (float*)(tail + cn_align((size_t)batch * max_pred * 4));
  float* best_lprobs = (float*)(tail + cn_align((size_t)batch * max_pred * 4) + cn_align((size_t)batch * 4));
  CN_BY_PRECISION(ctx, decode_impl<OT>(ctx, frame_embs, frame_lens, bos_ids, forbid_mask, batch, t_audio, 1, min_pred, max_pred, preds,
                               best_lprobs, mult_preds, mult_lprobs, out_sizes, nullptr, nullptr, nullptr, (char*)workspace,
                               s, nullptr, nullptr, logits));
}

extern "C" size_t conette_greedy_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t t_audio, int32_t max_pred) {
  if (!ctx || batch <= 0 || t_audio <= 0 || max_pred <= 0) return 0;
  return conette_decode_workspace_bytes(ctx, batch, t_audio, 1, max_pred) + cn_align((size_t)batch * max_pred * 4) +
         2 * cn_align((size_t)batch * 4);
}

extern "C" int conette_debug_dbprof(unsigned long long* out16, int reset) {
  if (out16) {
    CN_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_db_prof), 128));
    CN_HIP(hipMemcpyFromSymbol(out16 + 16, HIP_SYMBOL(g_s3_prof), 128));
  }
  if (reset) {
    unsigned long long z[16] = {0};
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_db_prof), z, 128));
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_s3_prof), z, 128));
  }
  return CN_OK;
}
