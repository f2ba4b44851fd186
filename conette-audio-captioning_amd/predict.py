"""``conette-predict`` for the MI355X path: same flags and CSV format as the reference CLI
(reference src/conette/predict.py:27-232).

    python -m conette_amd.predict --audio a.wav b.wav --task clotho --model_name DIR_OR_HUB_NAME \
        [--csv_export out.csv] [--precision bf16|bf16+f16dec|f16|fp8|mixed|mixed16|exact|fp32]

``--model_path`` (a Lightning training log directory with hydra/config.yaml + checkpoints/best.ckpt,
predict.py:144-178) belongs to the training stack and is out of scope: it is rejected with a
clear message.  Audio files are PCM WAV (see preprocessor.load_audio).
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
    parser.add_argument("--model_path", type=_opt_str, help="Path to trained model directory (unsupported).",
                        default=None)
    parser.add_argument("--device", type=str, help="Torch device used to run the model.", default="cuda_if_available")
    parser.add_argument("--token", type=_opt_str, help="Optional access token.", default=None)
    parser.add_argument("--seed", type=_opt_int, help="Random seed value (inference is deterministic).", default=1234)
    parser.add_argument("--csv_export", type=_opt_str, help="Path to CSV output file.", default=None)
    parser.add_argument("--verbose", type=int, help="Verbose level.", default=1)
    parser.add_argument("--precision", type=str, choices=("bf16", "bf16+f16dec", "f16", "fp8", "mixed", "mixed16", "exact", "fp32"), default="bf16")
    return parser.parse_args(argv)


def format_results(fpaths: List[str], tasks: List[str], cands: List[str]) -> List[dict]:
    """Row format of predict.py:214-218."""
    return [{"audio": osp.basename(f), "task": t, "candidate": c} for f, t, c in zip(fpaths, tasks, cands)]


def write_csv(path: str, results: List[dict]) -> None:
    with open(path, "w") as file:
        writer = csv.DictWriter(file, fieldnames=["audio", "task", "candidate"])
        writer.writeheader()
        writer.writerows(results)


def main_predict(argv: Optional[List[str]] = None) -> List[dict]:
    args = get_predict_args(argv)
    logging.basicConfig(level=logging.INFO if args.verbose >= 1 else logging.WARNING, format="%(message)s")
    if args.model_path is not None:
        raise ValueError("--model_path (Lightning checkpoints of the training stack) is not supported by the "
                         "MI355X inference path; export the model to an HF directory and use --model_name.")
    if args.model_name is None:
        raise ValueError(f"Invalid arguments {args.model_name=} and {args.model_path=}. (expected at one str value)")
    from . import CoNeTTEConfig
    from .model import CoNeTTEModel

    config = CoNeTTEConfig.from_pretrained(args.model_name)
    model = CoNeTTEModel.from_pretrained(args.model_name, config=config, device=args.device, precision=args.precision)
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
