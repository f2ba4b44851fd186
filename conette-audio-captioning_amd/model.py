"""CoNeTTEModel -- drop-in for the reference's inference API on MI355X.

Mirrors ``conette.huggingface.model.CoNeTTEModel`` (reference huggingface/model.py:38-289) and
the inference methods of ``CoNeTTEPLM`` (pl_modules/conette.py:352-525): same constructor
arguments, ``from_pretrained(name_or_dir, config=...)``, ``model(x, sr, x_shapes, preprocess,
threshold, task, beam_size, min_pred_size, max_pred_size, forbid_rep_mode)`` and the same
output dict.  All dense work runs in libconette_hip.so through ``engine.Engine``.
"""
from __future__ import annotations

import csv
import json
import logging
import os
import os.path as osp
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional, Union

import torch
from torch import Size, Tensor

from .config import CoNeTTEConfig
from .engine import Engine, PREC_BF16, PREC_F16
from .preprocessor import CoNeTTEPreprocessor
from .tokenizer import AACTokenizer, ENGLISH_STOPWORDS, unpickle_extra_state

pylog = logging.getLogger(__name__)

FORBID_MODES = ("none", "all", "content_words")
# What `CoNeTTEModel.from_pretrained(dir)(x)` runs when the caller names no precision (round 6): a 16-bit base pipeline (fp16
# encoder + exact decoder: engine.CERT_DEFAULT_BASE) with the device-side id certificate -- clips whose search margins do not
# certify their token ids are re-run through the exact context from the waveform, so the ids are the reference's (fp32) ids.
# "bf16" is the benchmark's throughput mode (BASELINE configs[1-3]).
DEFAULT_PRECISION = "certified"


def load_audioset_idx_to_name(offline: bool = False, cache_path: Union[str, Path, None] = None) -> Dict[int, str]:
    """transforms/audioset_mapping.py:102-107: index -> display_name from class_labels_indices.csv."""
    if cache_path is None:  # reference default: ~/.cache/audioset_mapping (audioset_mapping.py:16-24)
        cache_path = os.environ.get("CONETTE_AUDIOSET_CACHE")
    cache = Path(cache_path) if cache_path is not None else Path.home().joinpath(".cache", "audioset_mapping")
    fpath = cache.joinpath("class_labels_indices.csv")
    if not osp.isfile(fpath):
        if offline:
            raise FileNotFoundError(
                f"Cannot find or download audioset mapping file in '{fpath}' with mode offline={offline}.")
        from torch.hub import download_url_to_file
        os.makedirs(cache, exist_ok=True)
        download_url_to_file("http://storage.googleapis.com/us_audioset/youtube_corpus/v1/csv/class_labels_indices.csv",
                             str(fpath), progress=False)
    with open(fpath, "r") as file:
        data = list(csv.DictReader(file, skipinitialspace=True, strict=True))
    return {int(d["index"]): d["display_name"] for d in data}


def probs_to_names(probs: Tensor, threshold: Union[float, Tensor], idx_to_name: Dict[int, str]) -> List[List[str]]:
    mask = (probs >= threshold).cpu()
    return [[idx_to_name[int(j)] for j in torch.where(row)[0].tolist()] for row in mask]


def _read_state_dict(path: str) -> Dict[str, Tensor]:
    st = osp.join(path, "model.safetensors")
    if osp.isfile(st):
        from safetensors.torch import load_file
        return load_file(st)
    for name in ("pytorch_model.bin", "model.pt", "model.bin"):
        fp = osp.join(path, name)
        if osp.isfile(fp):
            return torch.load(fp, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"No weights file (model.safetensors / pytorch_model.bin) in '{path}'.")


class CoNeTTEModel:
    """CoNeTTE for inference; weights live packed on the GPU inside the HIP context."""

    config_class = CoNeTTEConfig

    def __init__(self, config: CoNeTTEConfig, device: Union[str, torch.device, None] = "cuda_if_available",
                 inference: bool = True, offline: bool = False, model_override: Any = None, *,
                 state_dict: Optional[Dict[str, Tensor]] = None, precision: str = DEFAULT_PRECISION,
                 audioset_idx_to_name: Optional[Dict[int, str]] = None,
                 stopwords: Optional[Iterable[str]] = None) -> None:
        if model_override is not None:
            raise NotImplementedError("model_override (Lightning checkpoints) is not supported by the MI355X path.")
        if not inference:
            raise NotImplementedError("The MI355X path is inference-only (reference training stack is out of scope).")
        if state_dict is None:
            raise ValueError("CoNeTTEModel needs weights: use CoNeTTEModel.from_pretrained(dir) or pass state_dict=.")
        if device in ("cuda_if_available", None, "auto"):
            device = "cuda"
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("conette_amd runs on a ROCm GPU only; there is no CPU fallback.")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.precision = precision
        self._stopwords = list(ENGLISH_STOPWORDS if stopwords is None else stopwords)
        self.audioset_idx_to_name = (load_audioset_idx_to_name(offline=offline) if audioset_idx_to_name is None
                                     else dict(audioset_idx_to_name))
        self.training = False
        self.last_recomputed: Optional[Tensor] = None
        self._build(dict(state_dict))

    def _build(self, state_dict: Dict[str, Tensor]) -> None:
        """module construction + load_state_dict of the reference (model.py:41-107,126-163): tokenizer from the pickled extra
        state, task tokens, forbid mask, then the packed weights inside the HIP contexts on ``self.device``."""
        config = self.config
        # non-tensor state: pickled dict in `_extra_state_` (model.py:126-139)
        tok_state = None
        if "_extra_state_" in state_dict:
            extra = unpickle_extra_state(state_dict.pop("_extra_state_"))
            tok_state = extra.get("model.tokenizers.0._extra_state")
        if tok_state is None:
            tok_state = state_dict.pop("model.tokenizers.0._extra_state", None)   # an already-unpacked state dict
        if tok_state is None:
            tok_state = config.tokenizer_state
        if tok_state is None:
            raise RuntimeError("Cannot build the model from state_dict. (tokenizer is not fit)")
        self.tokenizer = AACTokenizer.from_txt_state(tok_state)
        self._tok_state = tok_state
        # task tokens (conette.py:103-129); present in a trained tokenizer, appended otherwise
        self.task_name_to_token_id: Dict[str, int] = {}
        if config.task_mode in ("ds", "ds_src"):
            for name in config.task_names:
                token = f"<bos_{name}>"
                self.task_name_to_token_id[name] = (self.tokenizer.token_to_id(token) if self.tokenizer.has(token)
                                                    else self.tokenizer.add_special_token(token))
        elif config.task_mode != "none":
            raise ValueError(f"Invalid argument {config.task_mode=}.")
        vocab = self.tokenizer.get_vocab_size()
        cls_rows = int(state_dict["model.decoder.classifier.weight"].shape[0])
        if cls_rows != vocab:
            raise RuntimeError(f"vocab size mismatch: tokenizer {vocab} vs classifier {cls_rows}")
        if "model.task_id_to_token_id" in state_dict:
            self.task_id_to_token_id = state_dict["model.task_id_to_token_id"].to(torch.long).cpu()
        else:
            self.task_id_to_token_id = torch.as_tensor([self.task_name_to_token_id[n] for n in config.task_names])
        frm = state_dict.get("model.forbid_rep_mask")
        self.forbid_rep_mask: Optional[Tensor] = None if frm is None else frm.to(torch.bool).to(self.device)
        # the HIP path implements the published architecture only: fail loudly instead of silently running GELU / lin768
        if config.acti_name != "gelu" or config.proj_name != "lin768":
            raise ValueError(f"Unsupported config for the MI355X path: acti_name={config.acti_name!r}, "
                             f"proj_name={config.proj_name!r} (expected 'gelu' and 'lin768').")
        with torch.cuda.device(self.device):
            self.engine = Engine(state_dict, precision=self.precision, d_model=config.d_model, nhead=config.nhead,
                                 n_layers=config.num_decoder_layers, d_ff=config.dim_feedforward,
                                 pad_id=self.tokenizer.pad_token_id, bos_id=self.tokenizer.bos_token_id,
                                 eos_id=self.tokenizer.eos_token_id, device=self.device)
        self.preprocessor = CoNeTTEPreprocessor(self.engine, verbose=config.verbose)
        # the tensors the contexts were packed from, kept by reference (host memory the caller already holds) so that
        # state_dict() / save_pretrained() round-trip the reference layout (model.py:163-183)
        self._weights: Dict[str, Tensor] = {k: v for k, v in state_dict.items() if isinstance(v, Tensor)}

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *args, config: Optional[CoNeTTEConfig] = None,
                        **kwargs) -> "CoNeTTEModel":
        path = str(pretrained_model_name_or_path)
        if not osp.isdir(path):
            from huggingface_hub import snapshot_download  # needs network; raises offline
            path = snapshot_download(path)
        if config is None:
            config = CoNeTTEConfig.from_pretrained(path)
        return cls(config, *args, state_dict=_read_state_dict(path), **kwargs)

    # ---- PreTrainedModel surface (model.py:38,126-183): the checkpoint round trip -------------------------------------------
    def state_dict(self) -> Dict[str, Tensor]:
        """model.py:163-183: every tensor of the checkpoint the contexts were packed from (reference key names) + the non-tensor
        states -- the fitted tokenizer -- pickled into the uint8 tensor ``_extra_state_``; tensors contiguous."""
        import pickle
        out = {k: v.contiguous() for k, v in self._weights.items()}
        non_tensor = {"model.tokenizers.0._extra_state": self._tok_state}
        out["_extra_state_"] = torch.frombuffer(bytearray(pickle.dumps(non_tensor)), dtype=torch.uint8)
        return out

    def load_state_dict(self, state_dict: Dict[str, Tensor], strict: bool = True) -> "CoNeTTEModel":
        """Re-packs the HIP contexts from another checkpoint of the same layout (the pre-hook of model.py:126-161 unpickles
        ``_extra_state_`` and re-fits the tokenizer first: same here).  ``strict``: missing tensors are an error either way --
        ``conette_create`` names the first one it cannot find."""
        old = getattr(self, "engine", None)
        self._build(dict(state_dict))
        del old
        return self

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **kwargs) -> None:
        """``config.json`` + ``model.safetensors`` (or ``pytorch_model.bin``) readable by ``from_pretrained`` here AND by the
        reference's ``CoNeTTEModel.from_pretrained`` (transformers' PreTrainedModel layout; key names are the reference's)."""
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            # (safetensors refuses aliased storage: `.contiguous().clone()` for tensors that share memory)
            save_file({k: v.clone() for k, v in sd.items()}, osp.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, osp.join(save_directory, "pytorch_model.bin"))

    def to(self, device: Union[str, torch.device, None] = None, *args, **kwargs) -> "CoNeTTEModel":
        """nn.Module.to for devices: the device the contexts live on is a no-op, another ROCm GPU re-packs the weights there,
        a CPU is refused (the path has no CPU fallback); dtype arguments are ignored -- ``precision`` is chosen at construction."""
        if device is None or isinstance(device, torch.dtype):
            return self
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("conette_amd runs on a ROCm GPU only; there is no CPU fallback.")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev != self.device:
            self.device = dev
            self._build(self.state_dict())
        return self

    def cuda(self, device: Union[int, torch.device, None] = None) -> "CoNeTTEModel":
        return self.to("cuda" if device is None else (f"cuda:{device}" if isinstance(device, int) else device))

    def cpu(self) -> "CoNeTTEModel":
        return self.to("cpu")

    # ---- reference properties / methods (model.py:109-124) -------------------------------------
    @property
    def default_task(self) -> str:
        return next(iter(self.config.task_names))

    @property
    def tasks(self) -> List[str]:
        return list(self.config.task_names)

    def train_and_enable_grad(self, mode: bool = True) -> "CoNeTTEModel":
        if mode:
            raise NotImplementedError("The MI355X path is inference-only.")
        return self

    def eval_and_disable_grad(self, mode: bool = True) -> "CoNeTTEModel":
        return self.train_and_enable_grad(not mode)

    def eval(self) -> "CoNeTTEModel":
        return self

    # ---- helpers ---------------------------------------------------------------------------------
    def get_forbid_rep_mask(self, mode: Optional[str]) -> Optional[Tensor]:
        """pl_modules/common.py:222-299; None -> the checkpoint's persisted mask (conette.py:427-433)."""
        if mode is None:
            return self.forbid_rep_mask
        vocab = self.tokenizer.get_vocab_size()
        if mode == "none":
            return None
        if mode == "all":
            return torch.ones((vocab,), dtype=torch.bool, device=self.device)
        if mode == "content_words":
            mask = torch.ones((vocab,), dtype=torch.bool)
            for word in set(self._stopwords):
                if self.tokenizer.has(word):
                    mask[self.tokenizer.token_to_id(word)] = False
            return mask.to(self.device)
        raise ValueError(f"Invalid argument forbid_rep_mode={mode!r}. (expected one of {FORBID_MODES})")

    def batch_to_task_token_ids(self, datasets: List[str], sources: List[Optional[str]]) -> Tensor:
        """conette.py:486-525."""
        bsize = len(datasets)
        mode = self.config.task_mode
        if mode == "none":
            return torch.full((bsize,), self.tokenizer.bos_token_id, dtype=torch.long)
        task_to_idx = {name: i for i, name in enumerate(self.config.task_names)}
        if mode == "ds":
            idx = [task_to_idx[ds] for ds in datasets]
        elif mode == "ds_src":
            names = [ds if src is None else f"{ds}_{src}".lower() for ds, src in zip(datasets, sources)]
            idx = [task_to_idx[n] for n in names]
        else:
            raise ValueError(f"Invalid task mode {mode} for batch_to_task_token_ids.")
        return self.task_id_to_token_id[torch.as_tensor(idx, dtype=torch.long)]

    # ---- forward (model.py:185-261) -----------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: Union[Tensor, str, Iterable[str], Iterable[Tensor]],
                sr: Union[None, int, Iterable[int]] = None, x_shapes: Union[Tensor, None, List[Size]] = None,
                preprocess: bool = True, threshold: Union[float, Tensor] = 0.3,
                task: Union[str, List[str], None] = None, beam_size: Optional[int] = None,
                min_pred_size: Optional[int] = None, max_pred_size: Optional[int] = None,
                forbid_rep_mode: Optional[str] = None) -> Dict[str, Any]:
        with torch.cuda.device(self.device):
            if preprocess:
                batch = self.preprocessor(x, sr, x_shapes)
                clip_probs = batch.pop("clip_probs")
                tags = True   # (names from the probabilities below, once the certified precision has patched re-run clips)
            else:
                assert isinstance(x, Tensor) and isinstance(x_shapes, Tensor)
                batch = {"audio": x.to(self.device), "audio_shape": x_shapes.to(self.device)}
                clip_probs, tags = None, None
            wave = batch.pop("_wave", None)

            bsize = len(batch["audio"])
            if task is None:
                tasks = [self.default_task] * bsize
            elif isinstance(task, str):
                tasks = [task] * bsize
            elif len(task) != bsize:
                raise ValueError(f"Invalid number of tasks with input. (found {len(task)} tasks but {bsize} elements)")
            else:
                tasks = task
            del task
            for task in tasks:
                if task not in self.config.task_names:
                    raise ValueError(f"Invalid argument {tasks=}. (task {task} is not in {self.config.task_names})")
            dataset_lst = [self.default_task] * bsize
            source_lst: List[Optional[str]] = [None] * bsize
            for i, task in enumerate(tasks):
                parts = task.split("_")
                dataset_lst[i] = parts[0]
                if len(parts) >= 2:
                    source_lst[i] = "_".join(parts[1:])

            overflow = False
            if preprocess and self.engine.precision in (PREC_BF16, PREC_F16):
                # the fp16 residual stream of the 16-bit encoders holds |x| <= 65504; beyond it the embeddings are garbage
                # (include/conette_hip.h: conette_encode_nonfinite).  The certified precision re-runs the whole batch through
                # the exact context (fp32 stream) then; the others fail loudly.
                bad = self.engine.encode_nonfinite()
                if bad and not self.engine.certified:
                    raise RuntimeError(
                        f"precision={self.engine.precision_name!r}: the encoder's fp16 residual stream overflowed ({bad} "
                        "positions with non-finite LayerNorm statistics); use precision='certified', 'exact' or 'fp32' for "
                        "this checkpoint")
                overflow = bad > 0
            outs = self._generate(batch["audio"], batch["audio_shape"], dataset_lst, source_lst,
                                  beam_size=beam_size, min_pred_size=min_pred_size, max_pred_size=max_pred_size,
                                  forbid_rep_mode=forbid_rep_mode, wave=wave, recompute_all=overflow)
            outs["tasks"] = tasks
            patch = outs.pop("_clip_probs_patch", None)
            if clip_probs is not None and tags is not None:
                if patch is not None:    # certified: clips the exact context re-encoded carry its tag probabilities too
                    clip_probs.index_copy_(0, patch[0], patch[1])
                outs["tags_probs"] = clip_probs
                outs["tags"] = probs_to_names(clip_probs, threshold, self.audioset_idx_to_name)
            return outs

    def __call__(self, x, sr=None, x_shapes=None, preprocess: bool = True, threshold=0.3, task=None, beam_size=None,
                 min_pred_size=None, max_pred_size=None, forbid_rep_mode=None) -> Dict[str, Any]:
        return self.forward(x=x, sr=sr, x_shapes=x_shapes, preprocess=preprocess, threshold=threshold, task=task,
                            beam_size=beam_size, min_pred_size=min_pred_size, max_pred_size=max_pred_size,
                            forbid_rep_mode=forbid_rep_mode)

    def teacher_forcing(self, x, caps_in: Tensor, sr=None, x_shapes=None, preprocess: bool = True) -> Tensor:
        """CoNeTTEPLM.decode_audio(encode_audio(...), "forcing", caps_in=caps_in) (pl_modules/conette.py:392-417,
        nn/decoding/forcing.py:12-71): logits (B, vocab, cap_len) of given input captions in one causal pass.

        ``caps_in`` (B, cap_len): token ids, column 0 = the task token that replaced <bos>
        (``batch_to_task_token_ids``), right-padded with pad_id.  ``x`` as in ``forward`` (waveforms, or the
        preprocessor's output dict / (B, T, 768) embeddings with ``preprocess=False``)."""
        caps_in = torch.as_tensor(caps_in)
        if caps_in.ndim != 2 or caps_in.is_floating_point():
            raise ValueError("caps_in must be an integer tensor of shape (bsize, caps_size).")
        if bool(caps_in[:, 0].eq(self.tokenizer.bos_token_id).any()):
            raise ValueError("BOS was not replaced in input captions for decode_method='forcing'.")
        if preprocess:
            batch = self.preprocessor(x, sr, x_shapes)
            audio, audio_shape = batch["audio"], batch["audio_shape"]
        elif isinstance(x, dict):
            audio, audio_shape = x["audio"], x["audio_shape"]
        else:
            audio = x
            audio_shape = x_shapes if x_shapes is not None else torch.as_tensor([list(a.shape) for a in x])
        if audio.ndim == 4:
            audio = audio.squeeze(dim=1)
        if caps_in.shape[0] != audio.shape[0]:
            raise ValueError(f"Invalid number of captions {caps_in.shape[0]} for {audio.shape[0]} audio clips.")
        lens = torch.as_tensor(audio_shape)[:, 1].to(torch.int32)
        logits = self.engine.forcing(audio, lens, caps_in)  # (B, cap_len, V)
        return logits.permute(0, 2, 1)

    def greedy_search(self, x, sr=None, x_shapes=None, preprocess: bool = True, bos_id: Optional[int] = None,
                      min_pred_size: Optional[int] = None, max_pred_size: Optional[int] = None,
                      forbid_rep_mode: Optional[str] = None) -> Tensor:
        """nn/decoding/greedy.py:17-131 (the decoder of BaselinePLM, baseline.py:339-401): the arg-max chain, returned
        as the masked logits of every step, (B, vocab, pred_size).  ``bos_id`` defaults to the plain <bos> token
        (BaselinePLM has no task token); pass a task token id for CoNeTTE-style prompting."""
        if preprocess:
            batch = self.preprocessor(x, sr, x_shapes)
            audio, audio_shape = batch["audio"], batch["audio_shape"]
        elif isinstance(x, dict):
            audio, audio_shape = x["audio"], x["audio_shape"]
        else:
            audio = x
            audio_shape = x_shapes if x_shapes is not None else torch.as_tensor([list(a.shape) for a in x])
        if audio.ndim == 4:
            audio = audio.squeeze(dim=1)
        cfg = self.config
        min_pred = cfg.min_pred_size if min_pred_size is None else int(min_pred_size)
        max_pred = cfg.max_pred_size if max_pred_size is None else int(max_pred_size)
        bos = torch.full((audio.shape[0],), self.tokenizer.bos_token_id if bos_id is None else int(bos_id), dtype=torch.int32)
        lens = torch.as_tensor(audio_shape)[:, 1].to(torch.int32)
        res = self.engine.greedy(audio, lens, bos, self.get_forbid_rep_mask(forbid_rep_mode), min_pred, max_pred)
        return res["logits"].permute(0, 2, 1)

    def decode_audio(self, encoder_outs: Dict[str, Tensor], decode_method: str, **kwargs) -> Any:
        """The decode_audio surface shared by CoNeTTEPLM (pl_modules/conette.py:386-450) and BaselinePLM
        (pl_modules/baseline.py:339-401): ``encoder_outs`` = the preprocessor's output ({"audio": (B, T, 768),
        "audio_shape": (B, 2)}; the projection of conette.py:457 / baseline.py:409 runs inside the engine),
        ``decode_method`` in ("forcing", "greedy", "generate").

        * "forcing": ``caps_in`` required -> logits (B, vocab, cap_len)
        * "greedy": kwargs bos_id (default <bos>: BaselinePLM has no task token), min_pred_size, max_pred_size,
          forbid_rep_mode -> logits (B, vocab, pred_size)
        * "generate": kwargs bos_id (int or (B,) tensor; REQUIRED for CoNeTTE-style task prompting, default <bos>),
          beam_size, min_pred_size, max_pred_size, forbid_rep_mode -> (preds, lprobs, mult_preds, mult_lprobs)"""
        if decode_method == "forcing":
            if "caps_in" not in kwargs:
                raise ValueError(f"Please provide a 'caps_in' keyword argument with {decode_method=}. "
                                 f"(found {tuple(kwargs.keys())})")
            return self.teacher_forcing(encoder_outs, kwargs["caps_in"], preprocess=False)
        if decode_method == "greedy":
            return self.greedy_search(encoder_outs, preprocess=False, bos_id=kwargs.get("bos_id"),
                                      min_pred_size=kwargs.get("min_pred_size"), max_pred_size=kwargs.get("max_pred_size"),
                                      forbid_rep_mode=kwargs.get("forbid_rep_mode"))
        if decode_method == "generate":
            audio, audio_shape = encoder_outs["audio"], torch.as_tensor(encoder_outs["audio_shape"])
            if audio.ndim == 4:
                audio = audio.squeeze(dim=1)
            cfg = self.config
            bos = kwargs.get("bos_id")
            if bos is None:
                bos = self.tokenizer.bos_token_id
            bos = torch.as_tensor(bos, dtype=torch.int32).reshape(-1)
            if bos.numel() == 1:
                bos = bos.expand(audio.shape[0])
            beam = cfg.beam_size if kwargs.get("beam_size") is None else int(kwargs["beam_size"])
            min_pred = cfg.min_pred_size if kwargs.get("min_pred_size") is None else int(kwargs["min_pred_size"])
            max_pred = cfg.max_pred_size if kwargs.get("max_pred_size") is None else int(kwargs["max_pred_size"])
            if self.engine.certified:
                res = self.engine.generate_certified(None, audio, audio_shape[:, 1].to(torch.int32), bos.contiguous(),
                                                     self.get_forbid_rep_mask(kwargs.get("forbid_rep_mode")), beam, min_pred,
                                                     max_pred)
            else:
                res = self.engine.decode(audio, audio_shape[:, 1].to(torch.int32), bos.contiguous(),
                                         self.get_forbid_rep_mask(kwargs.get("forbid_rep_mode")), beam, min_pred, max_pred)
            pred_size, best_maxlen = (int(v) for v in res["sizes"].tolist())
            return (res["best_preds"][:, :best_maxlen].to(torch.long).contiguous(), res["best_lprobs"],
                    res["mult_preds"][:, :, :pred_size].to(torch.long).contiguous(), res["mult_lprobs"])
        raise ValueError(f"Unknown argument {decode_method=}. (expected one of ('forcing', 'greedy', 'generate'))")

    def _generate(self, audio: Tensor, audio_shape: Tensor, datasets: List[str], sources: List[Optional[str]], *,
                  beam_size=None, min_pred_size=None, max_pred_size=None, forbid_rep_mode=None,
                  wave: Optional[Tensor] = None, recompute_all: bool = False) -> Dict[str, Any]:
        """CoNeTTEPLM.forward("generate") = encode_audio + decode_audio + decode_text (conette.py:352-450)."""
        if audio.ndim == 4:  # FrameIdentEncoder (nn/encoders/ident.py:19-21)
            audio = audio.squeeze(dim=1)
        cfg = self.config
        beam = cfg.beam_size if beam_size is None else int(beam_size)
        min_pred = cfg.min_pred_size if min_pred_size is None else int(min_pred_size)
        max_pred = cfg.max_pred_size if max_pred_size is None else int(max_pred_size)
        assert beam > 0 and min_pred >= 0
        lens = audio_shape[:, 1].to(torch.int32)
        bos = self.batch_to_task_token_ids(datasets, sources)
        if self.engine.certified:   # base-precision search + margins; uncertified clips re-run through the exact context
            res = self.engine.generate_certified(wave, audio, lens, bos, self.get_forbid_rep_mask(forbid_rep_mode), beam,
                                                 min_pred, max_pred, tol=(float("inf"), 0.0, 0.0) if recompute_all else None)
        else:
            res = self.engine.decode(audio, lens, bos, self.get_forbid_rep_mask(forbid_rep_mode), beam, min_pred, max_pred)
        self.last_recomputed = res.get("recomputed")   # certified: (B,) bool, the clips the exact context re-ran
        pred_size, best_maxlen = (int(v) for v in res["sizes"].tolist())  # the one host sync of the path
        preds = res["best_preds"][:, :best_maxlen].to(torch.long).contiguous()
        mult_preds = res["mult_preds"][:, :, :pred_size].to(torch.long).contiguous()
        out = {
            "cands": self.tokenizer.decode_rec(preds), "preds": preds, "lprobs": res["best_lprobs"],
            "mult_cands": self.tokenizer.decode_rec(mult_preds), "mult_preds": mult_preds,
            "mult_lprobs": res["mult_lprobs"],
        }
        if "recomputed_clip_probs" in res:
            out["_clip_probs_patch"] = (res["recomputed_idx"], res["recomputed_clip_probs"])   # (popped by forward)
        return out
