"""conette_amd -- MI355X-native CoNeTTE inference path.

Drop-in for the reference's ``from conette import CoNeTTEConfig, CoNeTTEModel``
(reference src/conette/__init__.py:19-55): the same config class, ``from_pretrained`` and
``model(audio, sr=..., task=...)`` call, with every dense stage executed by hand-written HIP
kernels for gfx950 (libconette_hip.so, C ABI in include/conette_hip.h).  See DESIGN.md.
"""
from .config import CoNeTTEConfig  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need a GPU or the built library
    import importlib
    if name == "CoNeTTEModel":
        return importlib.import_module(__name__ + ".model").CoNeTTEModel
    if name == "Engine":
        return importlib.import_module(__name__ + ".engine").Engine
    if name == "BaselinePLM":   # the reference's second checkpoint family (pl_modules/baseline.py): no task tokens, no audio encoder
        return importlib.import_module(__name__ + ".baseline").BaselinePLM
    raise AttributeError(name)


DEFAULT_MODEL_NAME = "Labbeti/conette"


def conette(pretrained_model_name_or_path=DEFAULT_MODEL_NAME, config_kwds=None, model_kwds=None):
    """Create a pretrained CoNeTTEModel for inference (reference src/conette/__init__.py:25-49)."""
    import importlib
    CoNeTTEModel = importlib.import_module(__name__ + ".model").CoNeTTEModel
    config_kwds = {} if config_kwds is None else config_kwds
    model_kwds = {} if model_kwds is None else model_kwds
    if pretrained_model_name_or_path is None:
        return CoNeTTEModel(CoNeTTEConfig(**config_kwds), **model_kwds)
    config = CoNeTTEConfig.from_pretrained(pretrained_model_name_or_path, **config_kwds)
    return CoNeTTEModel.from_pretrained(pretrained_model_name_or_path, config=config, **model_kwds)


def get_offline_transform(model):
    """Offline feature producer with the per-file (batch = 1) semantics of the reference's
    ``get_resample_mean_convnext`` (transforms/get.py:240-310):
    ``f(waveform (C, T), sr) -> {"audio": (T', 768), "audio_shape": (2,), "clip_probs": (527,)}``,
    the columns the training HDF files store and ``model(..., preprocess=False)`` consumes."""
    import importlib
    offline = importlib.import_module(__name__ + ".offline")

    def transform(waveform, sr=32000):
        return offline.transform_one(model, waveform, sr)
    return transform
