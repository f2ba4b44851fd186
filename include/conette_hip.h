/* conette_hip.h -- C ABI of the MI355X (gfx950) CoNeTTE inference library (libconette_hip.so).
 *
 * The reference (Labbeti/conette-audio-captioning v0.3.1) is pure Python and has NO FFI layer
 * (SURVEY.md section 8b): these entry points sit UNDER the Python operator API that the build's
 * host package (conette_amd) keeps, one per dense stage of the hot path.  Each declaration
 * names the reference interface it replaces (paths relative to /root/reference/src/conette/).
 *
 * Conventions
 *   - every pointer marked "dev" is a device pointer owned by the caller (PyTorch caching
 *     allocator); the library allocates only inside conette_create (packed weights + tables);
 *   - all entry points are asynchronous on `stream` (a hipStream_t passed as void*), never
 *     call hipDeviceSynchronize, and may be captured into a hipGraph;
 *   - return value: 0 = ok, non-zero = error code, text via conette_last_error();
 *     no C++ exception crosses the boundary;
 *   - one host thread per device/process.
 */
#ifndef CONETTE_HIP_H
#define CONETTE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONETTE_ABI_VERSION 3 /* 2: conette_encode_taps carries its size; CONETTE_PREC_F16X2 / _FP8 / _F16; conette_decode_graph_nodes
                                 3: conette_decode's `margins` output (the id certificate of the 16-bit precisions), beams up to 16,
                                    conette_encode_nonfinite */

/* precision of GEMM operands / intermediate activations.  Accumulation is always fp32 and the decoder's residual stream is
 * always fp32; the ENCODER's residual stream is IEEE fp16 in the two 16-bit precisions (BF16, F16) since round 5 -- 11
 * significant bits, |x| <= 65504 required: frame embeddings 4.43e-3 -> 4.48e-3 (BF16) and 5.4e-4 -> 8.5e-4 (F16) rel. rms off the
 * fp32 reference for 10 C instead of 16 C bytes moved per position and block -- and fp32 in the other precisions */
#define CONETTE_PREC_F32 0  /* v_mfma_f32_16x16x4_f32: exact-fp32 parity mode          */
#define CONETTE_PREC_BF16 1 /* v_mfma_f32_16x16x32_bf16: throughput mode (BASELINE cfg) */
#define CONETTE_PREC_F16X2 2 /* "exact": every GEMM operand an fp16 hi + lo pair (22 bits), products as three
                                v_mfma_f32_16x16x32_f16 (hi.hi + hi.lo + lo.hi): fp32-MFMA accuracy at MFMA-f16 rate,
                                exact-erf GELU; token ids equal the fp32 mode's / the reference's */

/* 3: the experimental fp8 precision of ABI 2 (pointwise convolutions of stages 0-2 on the non-scaled e4m3 MFMA) -- WITHDRAWN in
 * ABI 3, conette_create refuses it.  BASELINE.json configs[4]'s "fp8 MFMA pointwise GEMMs" is not offered: the block-scaled
 * v_mfma_scale_f32_16x16x128_f8f6f4 does sustain 4.0 PFLOP/s in register-fed loops (3.5x the bf16 16x16x32 loop's 1.13), but e4m3
 * operands -- per-tensor scaled or MX block-scaled alike -- move the frame embeddings by 4 % (MX in stage 2 alone: 3.7 %; the
 * bf16 mode: 0.44 %), an order of magnitude beyond what keeps captions: profiles/r06_notes.md section 3, oracle/mx_study.py */

#define CONETTE_PREC_F16 4 /* the bf16 mode's kernels instantiated for IEEE fp16 operands (v_mfma_f32_*_f16: the same cycles,
                              the same bytes): 11 significant bits instead of 8, i.e. an eighth of the bf16 mode's operand
                              rounding error at the bf16 mode's speed.  Conversions saturate at +-65504; weights below
                              6.1e-5 in magnitude are fp16 subnormals (absolute error <= 3e-8, kept by the MFMA) */

typedef struct conette_ctx conette_ctx; /* opaque: packed weights + constant tables */

/* Hyper-parameters; mirrors CoNeTTEConfig (huggingface/config.py:13-88) + tokenizer ids
 * (tokenization/constants.py:15). */
typedef struct conette_config {
  int32_t precision;    /* CONETTE_PREC_* */
  int32_t vocab_size;   /* rows of model.decoder.classifier.weight */
  int32_t d_model;      /* 256 */
  int32_t nhead;        /* 8   */
  int32_t n_layers;     /* 6   */
  int32_t d_ff;         /* 2048 */
  int32_t pad_id;       /* 0 */
  int32_t bos_id;       /* 1 */
  int32_t eos_id;       /* 2 */
  int32_t reserved[7];
} conette_config;

/* Optional per-stage outputs of conette_encode for parity tests: fp32, channels-last
 * (B, H, W, C) unless noted; any pointer may be NULL. */
typedef struct conette_encode_taps {
  size_t struct_bytes;  /* sizeof(conette_encode_taps) of the CALLER's header: members beyond it are never read, so a caller
                           built against an older (shorter) layout stays valid when members are appended */
  float* logmel;        /* (B, F, 224)  bn0 output (convnext.py:276-292) */
  float* stem;          /* (B, 252, 56, 96)  downsample_layers[0] output */
  float* stage_block0[4]; /* output of the first block of stage i */
  float* stage[4];      /* output of stage i */
  float* down[4];       /* output of downsample_layers[i], i = 1..3 ([0] unused) */
  float* block[18];     /* output of every ConvNeXt block in network order (stage 0 block 0 .. stage 3 block 2) */
} conette_encode_taps;

const char* conette_last_error(void);
int conette_abi_version(void);

/* Build a context from the reference state dict: `names[i]` is the
 * reference key (e.g. "preprocessor.encoder.stages.0.0.pwconv1.weight"), `tensors[i]` a dev
 * pointer to the contiguous fp32 tensor (bool tensors as uint8, int64 as int64), `numel[i]`
 * its element count.  Replaces module construction + load_state_dict of
 * huggingface/model.py:41-107,126-163.  Synchronous (runs once).
 * A list WITHOUT any "preprocessor.encoder." tensor creates a decoder-only context -- the reference's BaselinePLM family
 * (pl_modules/baseline.py:84-140: FrameIdentEncoder + projection + decoder over precomputed frame embeddings; the caller maps its
 * keys "projection.*" / "decoder.*" / "forbid_rep_mask" under "model."): conette_decode / conette_greedy / conette_forcing work,
 * conette_encode and conette_frontend_logmel return an error. */
int conette_create(const conette_config* cfg, int32_t n_tensors, const char* const* names,
                   const void* const* tensors, const int64_t* numel, conette_ctx** out);
void conette_destroy(conette_ctx* ctx);

/* Geometry helpers: STFT frames F = L/320 + 1; encoder frames T (convnext.py stem + 3 /2). */
int32_t conette_num_frames(int32_t n_samples);
int32_t conette_num_audio_frames(int32_t n_samples);

/* Bytes of caller-provided scratch needed by conette_encode / conette_decode. */
size_t conette_encode_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t n_samples);
size_t conette_decode_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t t_audio, int32_t beam,
                                      int32_t max_pred);

/* a2: log-mel frontend (convnext.py:270-292 + torchlibrosa Spectrogram/LogmelFilterBank + bn0).
 * wave: dev (B, L) fp32 mono @ 32 kHz, zero-padded to L.  out: dev (B, F, 224) fp32. */
int conette_frontend_logmel(conette_ctx* ctx, const float* wave, int32_t batch, int32_t n_samples, float* out,
                            void* stream);

/* a2-a7: ConvNeXt.forward (convnext.py:264-336) + the transpose of preprocessor.py:66.
 *   frame_embs : dev (B, T, 768) fp32      ("audio" of preprocessor.py:71-75)
 *   clip_probs : dev (B, 527) fp32         (sigmoid(head_audioset), convnext.py:324-334) */
int conette_encode(conette_ctx* ctx, const float* wave, int32_t batch, int32_t n_samples, float* frame_embs,
                   float* clip_probs, const conette_encode_taps* taps, void* workspace, size_t workspace_bytes,
                   void* stream);

/* The 16-bit precisions keep the encoder's residual stream in IEEE fp16 (|x| <= 65504).  A value beyond that is inf where the
 * fused MLP stored it and NaN after the next LayerNorm (convnext.py:61-66) -- and the saturating conversions further down the
 * encoder turn such NaNs back into finite garbage, so the frame embeddings do not reliably show it.  Every LayerNorm of
 * conette_encode that reads the stream (the next block's depthwise conv + LN, the downsample layers, the frame-mean head) counts
 * the positions whose statistics are not finite in one per-device counter.  This call waits for `stream`, returns the count
 * accumulated by the encodes of this process on the context's device since the previous call in *count and resets it.
 * Non-zero = some clip's embeddings are unusable in this precision: run it through a CONETTE_PREC_F16X2 / _F32 context (fp32
 * stream).  The ONE blocking entry point besides conette_create; a host that never calls it loses the diagnosis. */
int conette_encode_nonfinite(conette_ctx* ctx, void* stream, int32_t* count);

/* a9-a14: CoNeTTEPLM.encode_audio + decode_audio("generate") = nn/decoding/beam.py:22-227 with
 * the decoder of nn/decoders/aac_tfmer.py:71-118, KV-cached, whole search loop on device.
 *   frame_embs  : dev (B, T, 768) fp32           frame_lens : dev (B) int32 valid frames
 *   bos_ids     : dev (B) int32 task tokens      forbid_mask: dev (V) uint8 or NULL
 * Outputs (all dev, full width; trim on host with out_sizes):
 *   best_preds  (B, max_pred) int32   best_lprobs (B) fp32
 *   mult_preds  (B, beam, max_pred) int32   mult_lprobs (B, beam) fp32
 *   out_sizes   (2) int32: [0] = pred_size (beam.py:192-194,207-211), [1] = best_maxlen (:222-225)
 *   step0_logits: optional dev (B*beam, ld_v) fp32 copy of the step-0 logits (parity), or NULL
 *   trace_sel   : optional dev (max_pred, B, beam, 2) int32 = (parent row, token) picked by the
 *                 per-clip top-k of each step in descending order, -1 where unused; or NULL
 *   trace_val   : optional dev (max_pred, B, beam) fp32 running log-prob sums of those picks
 *   margins     : optional dev (B, 2, max_pred + 1) fp32, or NULL -- how far each decision of the search is from any other
 *                 outcome.  Plane [b][0] = MEMBERSHIP: [i], i < max_pred = last pick minus first rejected candidate of clip b's
 *                 top-k call of step i (_select_k_next_toks, beam.py:230-269) -- which candidates continue; [max_pred] = best
 *                 averaged log-prob minus the second best (the final choice, beam.py:214-217), +inf for beam 1.  Plane [b][1] =
 *                 ORDER: [i] = the smallest gap between consecutive picks of that call (their order assigns the slots,
 *                 beam.py:165-169: the order of mult_preds, never best_preds), +inf with one pick; [max_pred] unused (+inf).
 *                 +inf for steps the clip no longer takes; NaN = fewer finite candidates than picks.  A search whose margins
 *                 all exceed twice the precision's candidate error took the decisions an exact search takes: the certificate
 *                 behind the host's precision "certified" (conette_amd/engine.py), which re-runs only the other clips through a
 *                 CONETTE_PREC_F16X2 context; with plane 0 alone ("certified-best") best_preds / best_lprobs are certified and
 *                 mult_preds as a set of hypotheses.
 * beam <= 16 (1..8 on the register-resident step kernel, 9..16 -- BaselinePLM's default is 10, pl_modules/baseline.py:47 -- on
 * the generic one), max_pred <= 64.  With identical arguments (pointers included) the launch sequence is replayed from a cached
 * hipGraph from the third call on (see conette_set_option). */
int conette_decode(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens, const int32_t* bos_ids,
                   const uint8_t* forbid_mask, int32_t batch, int32_t t_audio, int32_t beam,
                   int32_t min_pred, int32_t max_pred, int32_t* best_preds, float* best_lprobs,
                   int32_t* mult_preds, float* mult_lprobs, int32_t* out_sizes, float* step0_logits,
                   int32_t* trace_sel, float* trace_val, float* margins, void* workspace, size_t workspace_bytes,
                   void* stream);

/* teacher forcing (nn/decoding/forcing.py:12-71 as called by CoNeTTEPLM.decode_audio(..., "forcing",
 * caps_in=...), pl_modules/conette.py:392-417, after the projection of conette.py:457): the logits of every position of
 * given input captions under the causal mask, padded caption positions masked as keys
 * (tensor_to_pad_mask(caps_in, pad_value=pad_id)).
 *   caps_in : dev (B, cap_len) int32, column 0 = the task token that replaced <bos>, right-padded with pad_id
 *   logits  : dev (B, cap_len, vocab) fp32 -- the reference returns the same values permuted to (B, vocab, cap_len)
 * Runs the KV-cached step kernels of conette_decode with the next token taken from caps_in instead of the search. */
size_t conette_forcing_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t t_audio, int32_t cap_len);
int conette_forcing(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens, const int32_t* caps_in,
                    int32_t batch, int32_t t_audio, int32_t cap_len, float* logits, void* workspace,
                    size_t workspace_bytes, void* stream);

/* greedy_search (nn/decoding/greedy.py:17-131; BaselinePLM's decoder, not reachable from
 * CoNeTTEPLM): the arg-max chain with the full masked logits of every step as output.
 *   logits : dev (B, max_pred, vocab) fp32 -- per step the logits of every unfinished clip with the EOS floor
 *            (greedy.py:96-97) and the forbid-repeat mask (:99-105) applied; finished clips hold (-inf, pad_id -> 0)
 *            (:64-69).  The reference returns the same values permuted to (B, vocab, pred_size).
 *   preds  : dev (B, max_pred) int32 arg-max tokens, pad_id after <eos>
 *   out_sizes[0] = pred_size (steps until every clip had finished), out_sizes[1] = longest caption incl. <eos> */
size_t conette_greedy_workspace_bytes(const conette_ctx* ctx, int32_t batch, int32_t t_audio, int32_t max_pred);
int conette_greedy(conette_ctx* ctx, const float* frame_embs, const int32_t* frame_lens, const int32_t* bos_ids,
                   const uint8_t* forbid_mask, int32_t batch, int32_t t_audio, int32_t min_pred, int32_t max_pred,
                   float* logits, int32_t* preds, int32_t* out_sizes, void* workspace, size_t workspace_bytes,
                   void* stream);

/* a1: torchaudio.functional.resample (preprocessor.py:134-141), sinc_interpolation width 6,
 * rolloff 0.99.  in: dev (rows, n_in) fp32; out: dev (rows, n_out), n_out = ceil(n_in*new/orig). */
int conette_resample(const float* in, int32_t rows, int32_t n_in, int32_t orig_sr, int32_t new_sr, float* out,
                     void* stream);
int32_t conette_resample_len(int32_t n_in, int32_t orig_sr, int32_t new_sr);

/* CU-partitioned streams (hipExtStreamCreateWithCUMask).  The decode phase is a chain of ~10^3 tiny
 * dependent launches; sharing the whole chip with the encoder's big grids makes every one of them
 * queue behind resident encoder workgroups.  Giving the decode stream a private slice of CUs and
 * the encode stream the complement lets both phases of consecutive batches run side by side.
 * mask_words: 32-bit words, bit i = CU i enabled; returns a hipStream_t in *out_stream. */
int conette_stream_create_masked(const uint32_t* mask_words, int32_t n_words, void** out_stream);
int conette_stream_destroy(void* stream);

/* Runtime options. */
#define CONETTE_OPT_DECODE_GRAPH 1 /* 1 (default): replay conette_decode from a cached hipGraph */
#define CONETTE_OPT_DECODE_FUSION 2 /* 1 (default): fused decoder-layer kernels (bf16); 0: one launch per sub-layer */
#define CONETTE_OPT_ENCODE_RESERVED_CUS 3 /* default 0: compute units the encoder's persistent kernels leave free, so that
                                            the small dependent kernels of a conette_decode running on another stream are
                                            not queued behind them (0 .. n_cu / 2) */
#define CONETTE_OPT_FORCING_STEPWISE 4 /* default 0: conette_forcing is one causal pass over all caption positions
                                         (forcing.py:12-71); 1: the KV-cached step kernels fed with the caption */
int conette_set_option(conette_ctx* ctx, int32_t option, int32_t value);

/* Kernel / copy nodes of the decode hipGraph that conette_decode launched (or captured) most recently on this context
 * (0: none yet): the launch count of that whole search, for bench.py's decode roofline entry. */
int32_t conette_decode_graph_nodes(const conette_ctx* ctx);
/* Decode graphs a context keeps (least recently used evicted beyond it; the evicting call waits for the evicted graph's own
 * last launch, outside the context's cache lock).  A caller that cycles through more (shape, buffer) keys than this replays nothing. */
#define CONETTE_MAX_DECODE_GRAPHS 64

/* Per-kernel-class timing with HIP events recorded on the caller's stream around each launch
 * of the selected classes (bench.py's roofline leg).  While a decoder class is selected, decode
 * graphs are bypassed.  conette_profile_read synchronises the recorded events, ADDS the elapsed
 * milliseconds / launch counts per class into ms[CONETTE_PROF_NCLASS] / counts[...] and
 * clears the recording. */
#define CONETTE_PROF_FRONTEND 0
#define CONETTE_PROF_STEM 1
#define CONETTE_PROF_DWCONV_LN 2
#define CONETTE_PROF_PW1_GEMM 3
#define CONETTE_PROF_PW2_GEMM 4
#define CONETTE_PROF_DOWNSAMPLE 5
#define CONETTE_PROF_HEADS 6
#define CONETTE_PROF_DEC_PREPARE 7
#define CONETTE_PROF_DEC_GEMM 8
#define CONETTE_PROF_DEC_ATTN 9
#define CONETTE_PROF_DEC_MISC 10
#define CONETTE_PROF_SEARCH 11
#define CONETTE_PROF_NCLASS 12
int conette_profile_enable(conette_ctx* ctx, uint32_t class_mask);
int conette_profile_read(conette_ctx* ctx, float* ms, int32_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* CONETTE_HIP_H */
