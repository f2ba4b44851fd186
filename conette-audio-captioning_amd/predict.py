"""``conette-predict`` for the MI355X path: same flags and CSV format as the reference CLI
(reference src/conette/predict.py:27-232).

    python -m conette_amd.predict --audio a.wav b.wav --task clotho --model_name DIR_OR_HUB_NAME \
        [--csv_export out.csv] [--precision certified|certified:BASE|bf16|bf16+f16dec|f16|mixed|mixed16|exact|fp32]

``--model_path`` (a Lightning training log directory with hydra/config.yaml + checkpoints/best.ckpt, predict.py:123-178) is
accepted when the audio encoder's weights come with it: the reference builds its HF wrapper around the Lightning module and leaves
the ConvNeXt of the preprocessor at its RANDOM initialisation there (huggingface/preprocessor.py:23-33: ``pretrained=False``,
nothing in such a directory is loaded into it), so its captions from that path are noise; this CLI wants ``preprocessor.encoder.*``
tensors -- inside best.ckpt's state dict, in ``checkpoints/encoder.ckpt`` or through ``--encoder_ckpt`` -- and refuses otherwise.
Audio files are PCM WAV (see preprocessor.load_audio).
"""
from __future__ import annotations

import csv
import logging
import os.path as osp
import sys
from argparse import ArgumentParser, Namespace
from typing import List, Optional

pylog = logging.getLogger("conette_amd.predict")


def _opt_str(x: str) -> Optional[str]:
    return None if x.lower() in ("none", "null", "") else x


def _opt_int(x: str) -> Optional[int]:
    return None if x.lower() in ("none", "null", "") else int(x)


def get_predict_args(argv: Optional[List[str]] = None) -> Namespace:
    parser = ArgumentParser(description="CoNeTTE audio captioning on MI355X.")
    parser.add_argument("--audio", type=str, help="Path to an audio file.", default=(), required=True, nargs="+")
    parser.add_argument("--task", type=_opt_str, help="CoNeTTE task embedding input.", default=None, nargs="+")
    parser.add_argument("--model_name", type=_opt_str, help="Model name on huggingface or local HF directory.",
                        default="Labbeti/conette")
    parser.add_argument("--model_path", type=_opt_str, default=None,
                        help="Path to a trained model directory (hydra/config.yaml + checkpoints/best.ckpt of a CoNeTTEPLM run). The "
                             "reference leaves the ConvNeXt audio encoder at its random initialisation on this path; here the "
                             "encoder weights (preprocessor.encoder.* tensors) must be in best.ckpt, in checkpoints/encoder.ckpt "
                             "or given by --encoder_ckpt.")
    parser.add_argument("--encoder_ckpt", type=_opt_str, default=None,
                        help="With --model_path: a state dict of the ConvNeXt audio encoder (keys preprocessor.encoder.*, encoder.* or bare).")
    parser.add_argument("--device", type=str, help="Torch device used to run the model.", default="cuda_if_available")
    parser.add_argument("--token", type=_opt_str, help="Optional access token.", default=None)
    parser.add_argument("--seed", type=_opt_int, help="Random seed value (inference is deterministic).", default=1234)
    parser.add_argument("--csv_export", type=_opt_str, help="Path to CSV output file.", default=None)
    parser.add_argument("--verbose", type=int, help="Verbose level.", default=1)
    parser.add_argument("--precision", type=str, default=None,
                        help="certified (default: fp16 pipeline + id certificate, uncertified clips re-run exactly), certified:<base>, "
                             "bf16, bf16+f16dec, f16, mixed, mixed16, exact, fp32")
    return parser.parse_args(argv)


def format_results(fpaths: List[str], tasks: List[str], cands: List[str]) -> List[dict]:
    """Row format of predict.py:214-218."""
    return [{"audio": osp.basename(f), "task": t, "candidate": c} for f, t, c in zip(fpaths, tasks, cands)]


def write_csv(path: str, results: List[dict]) -> None:
    with open(path, "w") as file:
        writer = csv.DictWriter(file, fieldnames=["audio", "task", "candidate"])
        writer.writeheader()
        writer.writerows(results)


def _check_model_path(model_path: str) -> None:
    """predict.py:123-141 (same three messages)."""
    cfg_fpath = osp.join(model_path, "hydra", "config.yaml")
    ckpt_fpath = osp.join(model_path, "checkpoints", "best.ckpt")
    if not osp.isdir(model_path):
        raise FileNotFoundError(f"Cannot find model_path directory. ({model_path} is not a directory)")
    if not osp.isfile(cfg_fpath):
        raise FileNotFoundError(f"Cannot find config file in model_path directory. ({cfg_fpath} is not a file)")
    if not osp.isfile(ckpt_fpath):
        raise FileNotFoundError(f"Cannot find checkpoint file in model_path directory. ({ckpt_fpath} is not a file)")


ENCODER_PREFIX = "preprocessor.encoder."


def model_path_state_dict(model_path: str, encoder_ckpt: Optional[str] = None):
    """(CoNeTTEConfig, HF-layout state dict) of a training log directory (predict.py:144-178): the Lightning module's tensors
    under ``model.`` (the attribute the HF wrapper holds it by, huggingface/model.py:86-98) + the audio encoder's under
    ``preprocessor.encoder.``."""
    import torch
    import yaml

    from . import CoNeTTEConfig

    _check_model_path(model_path)
    with open(osp.join(model_path, "hydra", "config.yaml"), "r") as file:
        raw_cfg = yaml.safe_load(file) or {}
    pl_cfg = dict(raw_cfg.get("pl", {}) or {})
    target = pl_cfg.pop("_target_", "unknown")
    if "CoNeTTEPLM" not in target:
        if "BaselinePLM" in target:
            raise NotImplementedError("model_path holds a BaselinePLM run: load it with conette_amd.BaselinePLM.from_checkpoint("
                                      "'<model_path>/checkpoints/best.ckpt') (no task tokens, precomputed frame embeddings).")
        raise NotImplementedError(f"Unsupported pretrained model type '{target}'.")
    ckpt = torch.load(osp.join(model_path, "checkpoints", "best.ckpt"), map_location="cpu", weights_only=False)
    plm_sd = ckpt["state_dict"]
    sd = {}
    for k, v in plm_sd.items():
        sd[k if k.startswith(ENCODER_PREFIX) else "model." + k] = v
    enc_file = encoder_ckpt or osp.join(model_path, "checkpoints", "encoder.ckpt")
    if not any(k.startswith(ENCODER_PREFIX) for k in sd) and osp.isfile(enc_file):
        enc = torch.load(enc_file, map_location="cpu", weights_only=False)
        enc = enc.get("state_dict", enc.get("model", enc)) if isinstance(enc, dict) else enc
        for k, v in enc.items():
            for pre in (ENCODER_PREFIX, "encoder."):
                if k.startswith(pre):
                    k = k[len(pre):]
                    break
            sd[ENCODER_PREFIX + k] = v
    if not any(k.startswith(ENCODER_PREFIX) for k in sd):
        raise ValueError(
            f"--model_path {model_path}: no audio-encoder weights.  The reference wraps the Lightning checkpoint in its HF model and "
            "leaves the ConvNeXt of the preprocessor at its random initialisation on this path (predict.py:144-178, "
            "huggingface/preprocessor.py:23-33); this CLI needs preprocessor.encoder.* tensors in checkpoints/best.ckpt, "
            "a checkpoints/encoder.ckpt, or --encoder_ckpt FILE.")
    import inspect
    known = set(inspect.signature(CoNeTTEConfig.__init__).parameters) - {"self", "kwargs"}
    config = CoNeTTEConfig(**{k: v for k, v in pl_cfg.items() if k in known})
    return config, sd


def main_predict(argv: Optional[List[str]] = None) -> List[dict]:
    args = get_predict_args(argv)
    logging.basicConfig(level=logging.INFO if args.verbose >= 1 else logging.WARNING, format="%(message)s")
    from . import CoNeTTEConfig
    from .model import DEFAULT_PRECISION, CoNeTTEModel

    precision = DEFAULT_PRECISION if args.precision is None else args.precision
    if args.model_path is not None:     # (the reference gives model_path the precedence too: predict.py:196-201)
        config, sd = model_path_state_dict(args.model_path, args.encoder_ckpt)
        model = CoNeTTEModel(config, device=args.device, state_dict=sd, precision=precision)
    elif args.model_name is not None:
        config = CoNeTTEConfig.from_pretrained(args.model_name)
        model = CoNeTTEModel.from_pretrained(args.model_name, config=config, device=args.device, precision=precision)
    else:
        raise ValueError(f"Invalid arguments {args.model_name=} and {args.model_path=}. (expected at one str value)")
    model.eval_and_disable_grad()
    fpaths = list(args.audio)
    tasks = args.task
    if tasks is not None and len(tasks) == 1:
        tasks = tasks[0]
    outs = model(fpaths, task=tasks)
    results = format_results(fpaths, outs["tasks"], outs["cands"])
    for r in results:
        pylog.info(f"File '{r['audio']}' with task '{r['task']}':\n - '{r['candidate']}'")
    if args.csv_export is not None:
        write_csv(args.csv_export, results)
    return results


if __name__ == "__main__":
    main_predict(sys.argv[1:])
