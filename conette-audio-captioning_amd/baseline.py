"""BaselinePLM ("CNext-trans") checkpoints on the MI355X path -- SURVEY.md section 8 (f) 4.

The reference's second model family (``pl_modules/baseline.py``) is the same Transformer decoder WITHOUT task tokens: its
encoder is ``FrameIdentEncoder`` (``nn/encoders/ident.py:9-34``: the input IS the precomputed ConvNeXt frame embeddings that
``conette_amd.offline`` / the reference's ``get_resample_mean_convnext`` write), followed by the lin768 projection and
``AACTransformerDecoder`` prompted with the plain ``<bos>`` token.  A Lightning checkpoint of it holds

    tokenizers.0._extra_state        the fitted AACTokenizer (a dict, ``tokenization/aac_tokenizer.py:819-837``)
    projection.2.weight / .bias      ``build_proj_lin(768, d_model, False)`` (``pl_modules/common.py:59-78``)
    decoder.*                        ``AACTransformerDecoder`` (``nn/decoders/aac_tfmer.py``)
    forbid_rep_mask                  ``get_forbid_rep_mask("content_words", ...)`` (``baseline.py:127-135``)

-- no audio encoder at all.  ``BaselinePLM`` maps those keys onto the library's names (``model.projection.2.*``,
``model.decoder.*``) and creates a DECODER-ONLY context (``conette_create`` without ``preprocessor.encoder.*`` tensors);
its ``forward(batch, decode_method)`` / ``encode_audio`` / ``decode_audio`` mirror ``baseline.py:309-410`` for inference
(``training_step`` / ``validation_step`` / ``mix_audio`` belong to the training stack: out of scope).
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional, Union

import torch
from torch import Tensor

from .engine import Engine
from .tokenizer import AACTokenizer

DECODE_METHODS = ("forcing", "greedy", "generate")


class BaselinePLM:
    """Inference surface of the reference's ``BaselinePLM`` over a decoder-only HIP context."""

    def __init__(self, state_dict: Mapping[str, Any], *, tokenizer_state: Optional[Mapping[str, Any]] = None,
                 beam_size: int = 10, min_pred_size: int = 3, max_pred_size: Optional[int] = None, d_model: int = 256,
                 nhead: int = 8, num_decoder_layers: int = 6, dim_feedforward: int = 2048, proj_name: str = "lin768",
                 acti_name: str = "gelu", precision: str = "bf16", device: Union[str, torch.device, None] = "cuda") -> None:
        # (constructor defaults of the reference's BaselinePLM.__init__, pl_modules/baseline.py:36-52: proj_name "lin768",
        # min_pred_size 3, max_pred_size None -> the tokenizer's longest sentence (:88-99), beam_size 10, nhead 8, d_model 256,
        # num_decoder_layers 6, dim_feedforward 2048, acti_name "gelu"; tests/test_baseline_plm.py holds them against that list)
        if proj_name != "lin768" or acti_name != "gelu":
            raise ValueError(f"Unsupported hyper-parameters for the MI355X path: proj_name={proj_name!r}, acti_name={acti_name!r} "
                             "(expected 'lin768' and 'gelu'; lin2048 belongs to the PANN encoders, out of scope)")
        sd = dict(state_dict)
        tok_state = sd.pop("tokenizers.0._extra_state", None)
        if tok_state is None:
            tok_state = tokenizer_state
        if tok_state is None:
            raise RuntimeError("Cannot build the model from state_dict. (tokenizer is not fit)")   # base.py:95-97
        self.tokenizer = AACTokenizer.from_txt_state(tok_state)
        self.device = torch.device("cuda" if device in (None, "auto", "cuda_if_available") else device)
        if self.device.type != "cuda":
            raise RuntimeError("conette_amd runs on a ROCm GPU only; there is no CPU fallback.")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        mapped: Dict[str, Tensor] = {}
        for k, v in sd.items():
            if not isinstance(v, torch.Tensor):
                continue
            if k.startswith("projection.") or k.startswith("decoder."):
                mapped["model." + k] = v
            elif k == "forbid_rep_mask":
                mapped["model.forbid_rep_mask"] = v
            elif k.startswith("encoder."):
                raise ValueError(f"BaselinePLM checkpoint with encoder weights ({k}): only FrameIdentEncoder checkpoints "
                                 "(precomputed ConvNeXt frame embeddings) are supported")
        vocab = self.tokenizer.get_vocab_size()
        cls_rows = int(mapped["model.decoder.classifier.weight"].shape[0])
        if cls_rows != vocab:
            raise RuntimeError(f"vocab size mismatch: tokenizer {vocab} vs classifier {cls_rows}")
        frm = mapped.get("model.forbid_rep_mask")
        self.forbid_rep_mask: Optional[Tensor] = None if frm is None else frm.to(torch.bool).to(self.device)
        self.hp = dict(beam_size=beam_size, min_pred_size=min_pred_size,
                       max_pred_size=int(tok_state["tokenizer"]["max_sentence_size"]) if max_pred_size is None else int(max_pred_size),
                       d_model=d_model, nhead=nhead, num_decoder_layers=num_decoder_layers, dim_feedforward=dim_feedforward)
        with torch.cuda.device(self.device):
            self.engine = Engine(mapped, precision=precision, d_model=d_model, nhead=nhead, n_layers=num_decoder_layers,
                                 d_ff=dim_feedforward, pad_id=self.pad_id, bos_id=self.bos_id, eos_id=self.eos_id, device=self.device)

    @classmethod
    def from_checkpoint(cls, path: str, **kwargs) -> "BaselinePLM":
        """A Lightning ``.ckpt``: {"state_dict": ..., "hyper_parameters": ...} (the hyper-parameters fill the constructor)."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        hp = dict(ckpt.get("hyper_parameters", {})) if isinstance(ckpt, dict) else {}
        known = ("beam_size", "min_pred_size", "max_pred_size", "d_model", "nhead", "num_decoder_layers", "dim_feedforward",
                 "proj_name", "acti_name")
        args = {k: hp[k] for k in known if k in hp}
        args.update(kwargs)
        return cls(sd, **args)

    # ---- AACLightningModule properties (base.py) ----------------------------------------------------------------------
    @property
    def bos_id(self) -> int:
        return self.tokenizer.bos_token_id

    @property
    def eos_id(self) -> int:
        return self.tokenizer.eos_token_id

    @property
    def pad_id(self) -> int:
        return self.tokenizer.pad_token_id

    def decode_text(self, preds: Tensor):
        return self.tokenizer.decode_rec(preds)

    # ---- baseline.py:309-335 ------------------------------------------------------------------------------------------
    def forward(self, batch: Dict[str, Any], decode_method: str = "generate", **kwargs) -> Any:
        audio, audio_shape = batch["audio"], batch["audio_shape"]
        encoder_outs = self.encode_audio(audio, audio_shape)
        if decode_method == "forcing" and "captions" in batch:
            kwargs["caps_in"] = batch["captions"][:, :-1]
        outs = self.decode_audio(encoder_outs, decode_method, **kwargs)
        if decode_method == "generate":
            preds, lprobs, mult_preds, mult_lprobs = outs
            return {"cands": self.decode_text(preds), "preds": preds, "lprobs": lprobs,
                    "mult_cands": self.decode_text(mult_preds), "mult_preds": mult_preds, "mult_lprobs": mult_lprobs}
        return outs

    __call__ = forward

    def encode_audio(self, audio: Tensor, audio_shape: Tensor) -> Dict[str, Tensor]:
        """FrameIdentEncoder (ident.py:19-34): (B, 1, T, 768) or (B, T, 768) frame embeddings, lengths = audio_shape[:, 1].  The
        projection of baseline.py:409 runs inside the engine's decode entry points (``conette_decode`` projects once per clip)."""
        audio = torch.as_tensor(audio)
        if audio.ndim == 4:
            audio = audio.squeeze(dim=1)
        if audio.ndim != 3 or audio.shape[2] != 768:
            raise ValueError(f"expected frame embeddings of shape (bsize, [1,] time, 768), found {tuple(audio.shape)}")
        return {"frame_embs": audio, "frame_embs_lens": torch.as_tensor(audio_shape)[:, 1].to(torch.int32)}

    def decode_audio(self, encoder_outs: Dict[str, Tensor], decode_method: str, **kwargs) -> Any:
        """baseline.py:339-401: "forcing" -> logits (B, vocab, cap_len); "greedy" -> the masked logits of every step
        (B, vocab, pred_size) (greedy.py:17-131); "generate" -> (preds, lprobs, mult_preds, mult_lprobs) (beam.py:22-227)."""
        fe, lens = encoder_outs["frame_embs"], encoder_outs["frame_embs_lens"]
        b = fe.shape[0]
        hp = self.hp
        forbid = kwargs.get("forbid_rep_mask", self.forbid_rep_mask)
        bos = torch.full((b,), int(kwargs.get("bos_id", self.bos_id)), dtype=torch.int32)
        min_pred = int(kwargs.get("min_pred_size", hp["min_pred_size"]))
        max_pred = int(kwargs.get("max_pred_size", hp["max_pred_size"]))
        if decode_method == "forcing":
            if "caps_in" not in kwargs:
                raise ValueError(f"Please provide a 'caps_in' keyword argument with {decode_method=}. "
                                 f"(found {tuple(kwargs.keys())})")
            caps_in = torch.as_tensor(kwargs["caps_in"])
            return self.engine.forcing(fe, lens, caps_in).permute(0, 2, 1)
        if decode_method == "greedy":
            return self.engine.greedy(fe, lens, bos, forbid, min_pred, max_pred)["logits"].permute(0, 2, 1)
        if decode_method == "generate":
            beam = int(kwargs.get("beam_size", hp["beam_size"]))
            if self.engine.certified:   # given embeddings: uncertified clips re-run through the exact DECODER
                res = self.engine.generate_certified(None, fe, lens, bos, forbid, beam, min_pred, max_pred)
            else:
                res = self.engine.decode(fe, lens, bos, forbid, beam, min_pred, max_pred)
            pred_size, best_maxlen = (int(v) for v in res["sizes"].tolist())
            return (res["best_preds"][:, :best_maxlen].to(torch.long).contiguous(), res["best_lprobs"],
                    res["mult_preds"][:, :, :pred_size].to(torch.long).contiguous(), res["mult_lprobs"])
        raise ValueError(f"Unknown argument {decode_method=}. (expected one of {DECODE_METHODS})")
