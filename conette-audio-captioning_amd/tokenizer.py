"""Decode side of the reference's AACTokenizer (ids -> caption strings).

Reference: tokenization/aac_tokenizer.py:197-209 (detokenize_batch), :327-384 (decode_batch /
decode_rec), :691-707 (from_txt_state), :953-963 + tokenization/normalizers.py:68-188 (post-decoding
normalisers).  Encoding / fitting (spaCy, PTB) is training-only and out of scope.
"""
from __future__ import annotations

import io
import pickle
import re
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Union

import torch
from torch import Tensor

SPECIAL_TOKENS = ("<pad>", "<bos>", "<eos>", "<unk>")

# NLTK 3.8.1 english stop-word corpus (used by forbid_rep_mode="content_words",
# reference pl_modules/common.py:261-299)
ENGLISH_STOPWORDS = (
    "i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself yourselves he him his "
    "himself she she's her hers herself it it's its itself they them their theirs themselves what which who whom this "
    "that that'll these those am is are was were be been being have has had having do does did doing a an the and but "
    "if or because as until while of at by for with about against between into through during before after above below "
    "to from up down in out on off over under again further then once here there when where why how all any both each "
    "few more most other some such no nor not only own same so than too very s t can will just don don't should "
    "should've now d ll m o re ve y ain aren aren't couldn couldn't didn didn't doesn doesn't hadn hadn't hasn hasn't "
    "haven haven't isn isn't ma mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't wasn wasn't "
    "weren weren't won won't wouldn wouldn't"
).split()

_POST = (
    (re.compile("(" + "|".join(SPECIAL_TOKENS) + ")"), ""),   # CleanSpecialTokens
    (re.compile(r'\s+([,.!?;:"\'])'), r"\1"),                  # CleanSpacesBeforePunctuation
    None,                                                      # Strip
    (re.compile(" +"), " "),                                   # CleanDoubleSpaces
    (re.compile(r"(\s*)(\-)(\s*)"), r"\2"),                    # CleanHyphenSpaces
)


class _BuiltinsUnpickler(pickle.Unpickler):
    """The reference stores non-tensor state as a pickled dict of builtins
    (huggingface/model.py:165-183); refuse anything else."""

    _OK = {("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "str", "int", "float", "bool",
                                     "bytes", "bytearray", "complex", "NoneType")} | {("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name}")


def unpickle_extra_state(t: Tensor) -> Dict[str, Any]:
    return _BuiltinsUnpickler(io.BytesIO(bytes(t.cpu().tolist()))).load()


class AACTokenizer:
    """ids <-> tokens maps + post-decoding normalisation."""

    def __init__(self) -> None:
        self._hparams: Dict[str, Any] = {"level": "word", "lowercase": True, "punctuation_mode": "remove",
                                         "normalize": True}
        self._normalize = True
        self._added_special_tokens: List[str] = []
        self._max_sentence_size = -1
        self._min_sentence_size = 0
        self._n_sentences_fit = 0
        self._itos: Dict[int, str] = {}
        self._stoi: Dict[str, int] = {}
        self._vocab: Dict[str, int] = {}

    @classmethod
    def from_txt_state(cls, state: Mapping[str, Any]) -> "AACTokenizer":
        data = state["tokenizer"]
        tok = cls()
        tok._hparams = dict(data["hparams"])
        tok._normalize = data["normalize"]
        tok._added_special_tokens = list(data["added_special_tokens"])
        tok._max_sentence_size = data["max_sentence_size"]
        tok._min_sentence_size = data["min_sentence_size"]
        tok._n_sentences_fit = data["n_sentences_fit"]
        tok._itos = {int(k): v for k, v in data["itos"].items()}  # JSON round trips turn int keys into str
        tok._stoi = {k: int(v) for k, v in data["stoi"].items()}
        tok._vocab = dict(data["vocab"])
        return tok

    def get_txt_state(self) -> Dict[str, Any]:
        return {
            "_target_": "conette.tokenization.aac_tokenizer.AACTokenizer", "_version_": "2.2.0", "_type_": "txt",
            "tokenizer": {
                "hparams": self._hparams, "normalize": self._normalize,
                "added_special_tokens": self._added_special_tokens, "max_sentence_size": self._max_sentence_size,
                "min_sentence_size": self._min_sentence_size, "n_sentences_fit": self._n_sentences_fit,
                "itos": self._itos, "stoi": self._stoi, "vocab": self._vocab,
            },
        }

    # -- properties / queries (aac_tokenizer.py:150-170, 590-660) --
    bos_token, eos_token, pad_token, unk_token = "<bos>", "<eos>", "<pad>", "<unk>"

    @property
    def bos_token_id(self) -> int:
        return self._stoi[self.bos_token]

    @property
    def eos_token_id(self) -> int:
        return self._stoi[self.eos_token]

    @property
    def pad_token_id(self) -> int:
        return self._stoi[self.pad_token]

    @property
    def unk_token_id(self) -> int:
        return self._stoi[self.unk_token]

    def get_vocab_size(self) -> int:
        return len(self._vocab)

    def has(self, token: str) -> bool:
        return token in self._vocab

    def is_fit(self) -> bool:
        return self._n_sentences_fit > 0

    def get_level(self) -> str:
        return self._hparams.get("level", "word")

    def id_to_token(self, index: Union[int, Tensor]) -> str:
        if isinstance(index, Tensor):
            if index.ndim != 0 or index.is_floating_point():
                raise ValueError(f"Invalid argument {index=}. (expected an int or a scalar integer tensor)")
            index = int(index.item())
        return self._itos[index]

    def token_to_id(self, token: str) -> int:
        return self._stoi[token]

    def add_special_token(self, token: str, count: int = 0) -> int:
        """aac_tokenizer.py:302-316."""
        if token in self._vocab:
            raise ValueError(f"Invalid argument {token=}. (already in vocab)")
        new_id = max(max(self._itos.keys()), max(self._stoi.values())) + 1
        self._itos[new_id] = token
        self._stoi[token] = new_id
        self._vocab[token] = count
        self._added_special_tokens.append(token)
        return new_id

    # -- decoding --
    def detokenize_batch(self, sentences: Iterable[Iterable[str]]) -> List[str]:
        out = [" ".join(s) for s in sentences]
        if not self._normalize:
            return out
        res = []
        lowercase = self._hparams.get("lowercase", True)
        for s in out:
            for step in _POST:
                s = s.strip() if step is None else step[0].sub(step[1], s)
            res.append(s.lower() if lowercase else s)
        return res

    def decode_batch(self, sentences: Union[Tensor, Sequence[Sequence[int]]]) -> List[str]:
        if isinstance(sentences, Tensor):
            sentences = sentences.tolist()
        sentences = list(sentences)
        if len(sentences) == 0:
            return []
        return self.detokenize_batch([[self._itos[int(t)] for t in s] for s in sentences])

    def decode_single(self, sentence: Union[Tensor, Sequence[int]]) -> str:
        return self.decode_batch([sentence])[0]

    def decode_rec(self, nested: Union[Tensor, Sequence]) -> Union[str, list]:
        if isinstance(nested, Tensor):
            if nested.ndim == 0:
                raise TypeError("decode_rec expects a Tensor of ndim > 0 or a list")
            nested = nested.tolist()
        nested = list(nested)
        if len(nested) == 0 or isinstance(nested[0], int):
            return self.decode_single(nested)
        if all(len(s) == 0 or isinstance(s[0], int) for s in nested):
            return self.decode_batch(nested)
        return [self.decode_rec(s) for s in nested]
