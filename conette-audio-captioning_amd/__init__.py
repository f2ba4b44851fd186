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
    raise AttributeError(name)


def conette(pretrained_model_name_or_path: str = "Labbeti/conette", **kwargs):
    """Factory mirroring reference src/conette/__init__.py:25-49."""
    import importlib
    CoNeTTEModel = importlib.import_module(__name__ + ".model").CoNeTTEModel
    config = CoNeTTEConfig.from_pretrained(pretrained_model_name_or_path)
    return CoNeTTEModel.from_pretrained(pretrained_model_name_or_path, config=config, **kwargs)
