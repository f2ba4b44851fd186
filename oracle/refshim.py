"""ORACLE tooling (runs only in the build container, never on the GPU box).

Imports the reference's OWN hot-path Python files from /root/reference with stand-in modules
for the third-party leaves that are not installed here (torchaudio, torchlibrosa, torchoutil,
pytorch_lightning, nltk, spacy) -- SURVEY.md section 8(c).  The arithmetic stand-ins are the
restatements in oracle/thirdparty.py.  Used by oracle/gen_golden.py to produce the committed
fixtures under tests/golden/ and by tests that validate oracle/cpu_ref.py against the import
(skipped when /root/reference is absent).
"""
from __future__ import annotations

import os
import sys
import types
from argparse import Namespace

REF_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "conette"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = None
    sys.modules[name] = m
    return m


def install() -> None:
    """Register the stub package + stand-ins.  Idempotent."""
    if "conette" in sys.modules and getattr(sys.modules["conette"], "_refshim", False):
        return
    import torch
    import transformers  # noqa: F401  must be imported BEFORE the torchaudio stub
    from transformers import PreTrainedModel, PretrainedConfig  # noqa: F401
    from torch import nn

    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.dirname(here) not in sys.path:
        sys.path.insert(0, os.path.dirname(here))
    from oracle import thirdparty as tp

    # --- the reference package object, skipping its __init__.py --------------------------
    pkg = types.ModuleType("conette")
    pkg.__path__ = [os.path.join(REF_SRC, "conette")]
    pkg._refshim = True
    sys.modules["conette"] = pkg

    # --- torchlibrosa --------------------------------------------------------------------
    _mod("torchlibrosa")
    _mod("torchlibrosa.stft", Spectrogram=tp.Spectrogram, LogmelFilterBank=tp.LogmelFilterBank, STFT=tp.STFT)
    _mod("torchlibrosa.augmentation", SpecAugmentation=tp.SpecAugmentation)

    # --- torchaudio ------------------------------------------------------------------------
    def _load(path, *a, **k):
        from oracle.cpu_ref import load_wav
        return load_wav(path)

    ta = _mod("torchaudio", load=_load)
    taf = _mod("torchaudio.functional", resample=tp.resample)
    ta.functional = taf

    # --- torchoutil ------------------------------------------------------------------------
    def count_parameters(m, only_trainable=False):
        return sum(p.numel() for p in m.parameters() if p.requires_grad or not only_trainable)

    def randperm_diff(n, device=None):
        return torch.randperm(n, device=device)

    def masked_mean(x, mask, dim=None):
        return (x * mask).sum(dim=dim) / mask.sum(dim=dim).clamp(min=1)

    class _Dim(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    fn = dict(
        generate_square_subsequent_mask=tp.generate_square_subsequent_mask,
        indices_to_multihot=tp.indices_to_multihot,
        repeat_interleave_nd=tp.repeat_interleave_nd,
        tensor_to_lengths=tp.tensor_to_lengths,
        lengths_to_pad_mask=tp.lengths_to_pad_mask,
        tensor_to_pad_mask=tp.tensor_to_pad_mask,
        randperm_diff=randperm_diff,
        count_parameters=count_parameters,
        get_device=tp.get_device,
        pad_dim=tp.pad_dim,
        masked_mean=masked_mean,
        probs_to_names=tp.probs_to_names,
    )
    _mod("torchoutil")
    _mod("torchoutil.nn")
    _mod("torchoutil.nn.functional", **fn)
    _mod("torchoutil.nn.functional.get", get_device=tp.get_device)
    _mod("torchoutil.nn.functional.pad", pad_dim=tp.pad_dim)
    _mod("torchoutil.nn.functional.mask", masked_mean=masked_mean)
    _mod("torchoutil.nn.functional.multilabel", probs_to_names=tp.probs_to_names)
    _mod("torchoutil.nn.modules", CropDim=_Dim, PadDim=_Dim, Transpose=tp.Transpose)
    _mod("torchoutil.nn.modules.tensor", Transpose=tp.Transpose)
    _mod("torchoutil.utils")
    _mod("torchoutil.utils.collections", all_eq=tp.all_eq)

    # --- pytorch_lightning -----------------------------------------------------------------
    import inspect

    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._hparams = {}
            self._trainer = None

        def save_hyperparameters(self, *args, ignore=(), **kwargs):
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            hp = {k: info.locals[k] for k in info.args if k not in ("self",) and k not in ignore}
            self._hparams = dict(hp)
            self._hparams_initial = dict(hp)

        @property
        def hparams(self):
            return self._hparams

        @property
        def hparams_initial(self):
            return self._hparams_initial

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                bufs = list(self.buffers())
                return bufs[0].device if bufs else torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    _mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=_Dummy, Trainer=_Dummy)
    _mod("pytorch_lightning.utilities")
    _mod("pytorch_lightning.utilities.types", _METRIC_COLLECTION=object)

    # --- nltk (stopwords corpus): synthetic vocabulary -> synthetic stop-words ---------------
    class _Stop:
        @staticmethod
        def words(lang="english"):
            from oracle.cpu_ref import SYNTH_STOPWORDS
            return list(SYNTH_STOPWORDS)

    nl = _mod("nltk", download=lambda *a, **k: True)
    nlc = _mod("nltk.corpus", stopwords=_Stop)
    nl.corpus = nlc

    # --- spacy: constructor only at inference (tokenizers/spacy.py:22) -----------------------
    _mod("spacy", load=lambda name: None)


def ref():
    """Return a namespace with the reference symbols on the hot path."""
    install()
    from conette.huggingface.config import CoNeTTEConfig
    from conette.huggingface.model import CoNeTTEModel
    from conette.huggingface.preprocessor import CoNeTTEPreprocessor
    from conette.nn.decoders.aac_tfmer import AACTransformerDecoder
    from conette.nn.decoding.beam import generate
    from conette.nn.encoders.convnext import convnext_tiny
    from conette.pl_modules.conette import CoNeTTEPLM
    from conette.tokenization.aac_tokenizer import AACTokenizer

    return Namespace(
        CoNeTTEConfig=CoNeTTEConfig, CoNeTTEModel=CoNeTTEModel, CoNeTTEPreprocessor=CoNeTTEPreprocessor,
        AACTransformerDecoder=AACTransformerDecoder, generate=generate, convnext_tiny=convnext_tiny,
        CoNeTTEPLM=CoNeTTEPLM, AACTokenizer=AACTokenizer,
    )
