"""GPU: length-bucketed batching (BASELINE config 5, part 1).  Contract: a clip's outputs equal the reference run on ITS
bucket (pad-to-bucket-max, nn/functional/pad.py:11-17; frame lengths from the bucket's padded length,
nn/encoders/convnext.py:312-315) -- checked against the CPU oracle bucket by bucket, fp32 ids exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("planner", ["budget", "cost"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_bucketed_captions_equal_reference_per_bucket(prec, planner, tmp_path, synth_weights, synth_cfg):
    from conette_amd import CoNeTTEModel, synth
    from conette_amd.bucketing import caption_bucketed, padding_waste
    from oracle import cpu_ref as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    d = synth.write_pretrained_dir(str(tmp_path / "ckpt"))
    m = CoNeTTEModel.from_pretrained(d, precision=prec, offline=True, audioset_idx_to_name={i: f"tag{i}" for i in range(527)})
    secs = [9.7, 1.0, 14.2, 3.3, 1.6, 20.0, 6.1, 2.9, 10.0, 4.4, 1.2]
    lengths = [int(s * 32000) for s in secs]
    wav = synth.synth_waveforms(len(secs), max(lengths), 9090, lengths=lengths)
    clips = [torch.from_numpy(wav[i, :n].copy())[None] for i, n in enumerate(lengths)]
    # (planner "cost": plan_buckets_by_cost with a fixed cost of 8 audio seconds per bucket -- another partition, same contract)
    kw = dict(max_padded_seconds=42.0) if planner == "budget" else dict(fixed_cost_seconds=8.0)
    out = caption_bucketed(m, clips, sr=32000, task="clotho", **kw)
    buckets = out["buckets"]
    assert sorted(i for b in buckets for i in b) == list(range(len(secs))) and len(buckets) >= 3
    assert padding_waste(lengths, buckets) < padding_waste(lengths, [list(range(len(secs)))])
    assert len(out["cands"]) == len(secs) and out["preds"].shape[0] == len(secs)
    for idx in buckets:
        with torch.no_grad():
            ref = O.model_forward(synth_weights, synth_cfg, [clips[i] for i in idx], sr=32000, task="clotho")
        w = ref["preds"].shape[1]
        for j, i in enumerate(idx):
            if prec == "fp32":
                assert out["preds"][i, :w].cpu().tolist() == ref["preds"][j].tolist(), (idx, i)
                assert not bool(out["preds"][i, w:].any())
                assert out["cands"][i] == ref["cands"][j]
                np.testing.assert_allclose(float(out["lprobs"][i]), float(ref["lprobs"][j]), atol=2e-4)
                np.testing.assert_allclose(out["tags_probs"][i].cpu().numpy(), ref["tags_probs"][j].numpy(), rtol=1e-3, atol=1e-4)
            else:
                assert abs(float(out["lprobs"][i]) - float(ref["lprobs"][j])) < 0.3
                np.testing.assert_allclose(out["tags_probs"][i].cpu().numpy(), ref["tags_probs"][j].numpy(), atol=0.03)
    # the same clips as ONE pad-to-max batch give (legitimately) different results for the short clips: bucketing is not
    # an approximation of that batch, it is a different choice of reference batches
    one = m(clips, sr=32000, task="clotho")
    assert one["preds"].shape[0] == len(secs)
