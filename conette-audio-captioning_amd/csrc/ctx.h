// Internal context: packed weights (GEMM operands in the context's operand type, everything
// else fp32) + constant tables, all inside one device arena allocated by conette_create.
#pragma once
#include "../../include/conette_hip.h"
#include "common.h"

#define CN_N_MELS 224
#define CN_N_FFT 1024
#define CN_HOP 320
#define CN_N_BINS 513
#define CN_N_TAGS 527
#define CN_FEAT 768
#define CN_MAX_LAYERS 12

static const int CN_DEPTHS[4] = {3, 3, 9, 3};
static const int CN_DIMS[4] = {96, 192, 384, 768};

struct CnBlockW {
  const float* dw_w;  // [49][C]
  const unsigned* dw_wp;  // fp16 stream only: [42][C] fp16 pairs of consecutive kernel rows, (k[2a][j], k[2a+1][j]) at a * 7 + j, (k[2a+1][j], k[2a+2][j]) at 21 + a * 7 + j
  const float* dw_b;
  const float* ln_w;
  const float* ln_b;
  const void* w1;  // [4C][C] operand type
  const float* b1;
  const void* w2;  // [C][4C]
  const float* b2;
  const float* scale;
  // bf16, C <= 384: pwconv1 / pwconv2 (+ b1, LayerScale, b2) as the MFMA-fragment stream of mlp_rc2.h:
  // [hidden chunk C/8][fragment C/8 + 1][lane 64][8 bf16] followed by s * b2 (fp32, C)
  const void* mlp_stream;
  // CONETTE_PREC_F16X2, C <= 192: the block as the fp16 hi / lo fragment stream of mlp_sp.h (else nullptr)
  const void* mlp_sp;
};

struct CnDownW {
  const float* ln_w;
  const float* ln_b;
  const void* w;  // [2C][(kh,kw,c)] operand type
  const float* bias;
  const void* fused;  // down_fused.h stream (bf16, stage 0 -> 1 only), else nullptr
};

struct CnLayerW {
  const void* sa_in_w;  // [768][256]
  const float* sa_in_b;
  const void* sa_out_w;  // [256][256]
  const float* sa_out_b;
  const void* ca_q_w;  // [256][256]
  const float* ca_q_b;
  const void* ca_out_w;
  const float* ca_out_b;
  const void* ff1_w;  // [d_ff][256]
  const float* ff1_b;
  const void* ff2_w;  // [256][d_ff]
  const float* ff2_b;
  const float *n1w, *n1b, *n2w, *n2b, *n3w, *n3b;
  // 16-bit operands (and, as twelve hi / lo passes, the exact precision): the six attention-side 256 x 256 matrices (in_proj q, k, v rows; self out-proj; cross q-proj;
  // cross out-proj) re-ordered into the MFMA-fragment stream of dec_block.h:
  // [matrix 6][wave 4][quarter 4][piece (a, kk) 8][lane 64][8 bf16]
  const void* blk_w;
  // 16-bit operands (exact precision: every tile as a lo pass and a hi pass), d_ff % 256 == 0, d_ff <= 2048: linear1 / linear2 as per-hidden-chunk fragment streams of dec_ffn.h:
  // [chunk d_ff/256][tile 2 (W1 rows of the chunk | W2 columns of the chunk)][wave 4][quarter 4][piece 8][lane 64][8 bf16]
  const void* ffn_w;
  const float* blk_p;  // 2560 floats: in_proj bias 768 | bo | bq | bo2 | g1 | b1 | g2 | b2 (one contiguous LDS fill)
};

struct CnRuntime;  // C++ side state: profiling events, decode graphs (api.hip)

struct conette_ctx {
  conette_config cfg;
  CnRuntime* rt;
  void* dec_graphs;  // DecGraphCache of decoder.hip (hipGraph replay of conette_decode), owned by the context
  uint32_t prof_mask;
  int dec_unfused;  // CONETTE_OPT_DECODE_FUSION = 0: one launch per sub-layer (the cross-check path of the tests)
  int esize;  // operand element size (2 or 4)
  int sp16;   // CONETTE_PREC_F16X2: operands are sp16_t (fp16 hi/lo pairs, 4 bytes)
  int f16;    // CONETTE_PREC_F16: operands are half_t (the bf16 kernels instantiated for fp16), esize 2
  int no_encoder;  // decoder-only context (a BaselinePLM-layout checkpoint: no "preprocessor.encoder." tensors at create)
  // frontend tables
  const float* window;     // [1024]
  const float2* tw512;     // [512]
  const float2* tw1024;    // [513]
  const int* band;         // [224][2] lo, hi
  int mel_rows;            // rows of melC (the widest band)
  const float* melC;       // [max band width][224]: melC[i][m] = melW[band lo(m) + i][m] (lanes = mel bins read consecutive words)
  const float* bn_scale;   // [224]
  const float* bn_shift;   // [224]
  // stem
  const float* stem_w;  // [16][96]
  const float* stem_b;
  const float* stem_ln_w;
  const float* stem_ln_b;
  CnBlockW blocks[18];
  CnDownW down[3];
  const float* norm_w;
  const float* norm_b;
  const void* head_w;  // [527][768]
  const float* head_b;
  // decoder
  const void* proj_w;  // [256][768]
  const float* proj_b;
  const void* kv_w;  // [n_layers*2*d][d]  (k rows then v rows per layer)
  const float* kv_b;
  CnLayerW layers[CN_MAX_LAYERS];
  const float* emb;  // [V][d] fp32
  const float* pe;   // [5000][d] fp32
  int pe_len;
  const void* cls_w;  // [V][d]
  const float* cls_b;
  int n_cu;  // compute units of the device the context lives on (persistent-kernel grids)
  int enc_reserved_cus;  // CONETTE_OPT_ENCODE_RESERVED_CUS
  int forcing_stepwise;  // CONETTE_OPT_FORCING_STEPWISE
  // arena
  char* arena;
  size_t arena_bytes;
  size_t arena_used;
};

// event pair around the launches of one kernel class when that class is being profiled
void cn_prof_begin(conette_ctx* ctx, int cls, hipStream_t s);
void cn_prof_end(conette_ctx* ctx, int cls, hipStream_t s);
struct CnProfScope {
  conette_ctx* ctx;
  int cls;
  hipStream_t s;
  CnProfScope(conette_ctx* c, int k, hipStream_t st) : ctx(c), cls(k), s(st) {
    if (ctx->prof_mask & (1u << cls)) cn_prof_begin(ctx, cls, s);
  }
  ~CnProfScope() {
    if (ctx->prof_mask & (1u << cls)) cn_prof_end(ctx, cls, s);
  }
};

// ---- stage entry points implemented in the .hip files ---------------------------------------
int cn_frontend(conette_ctx* ctx, const float* wave, int B, int L, float* out, hipStream_t s);

// `return CALL;` with OT = the operand type of the context's precision
#define CN_BY_PRECISION(ctx, ...)                                                          \
  do {                                                                                     \
    switch ((ctx)->cfg.precision) {                                                        \
      case CONETTE_PREC_BF16: { typedef bf16_t OT; return __VA_ARGS__; }                   \
      case CONETTE_PREC_F16: { typedef half_t OT; return __VA_ARGS__; }                    \
      case CONETTE_PREC_F16X2: { typedef sp16_t OT; return __VA_ARGS__; }                  \
      default: { typedef float OT; return __VA_ARGS__; }                                   \
    }                                                                                      \
  } while (0)
// a statement with HT = the 16-bit operand type of the context (esize == 2)
#define CN_H16_CALL(ctx, ...)                          \
  do {                                                 \
    if ((ctx)->f16) { typedef half_t HT; __VA_ARGS__; } \
    else { typedef bf16_t HT; __VA_ARGS__; }            \
  } while (0)
