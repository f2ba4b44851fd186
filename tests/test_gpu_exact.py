"""GPU: the exact precision (CONETTE_PREC_F16X2: fp16 hi + lo operand pairs, three MFMAs per product) beyond the fixture
checks of test_gpu_parity.py (which hold it to the fp32 tolerances and to bit-exact ids):

  * the benchmark's shape and a ragged one: B = 64 / 40 as copies of the B = 8 fixture batch -- every copy bit-identical to
    copy 0 in every tap (the fused blocks of stages 0-1, mlp_sp.h, are persistent kernels whose tiles straddle clip
    boundaries; stages 2-3 run other GEMM tile shapes at other M), and copy 0 within the fp32 tolerance of the reference;
  * the mixed precision (bf16 encoder + exact decoder) is exactly "bf16 engine's encode fed to the exact engine's decode";
  * the split itself: hi + lo reproduces an fp32 value to 2^-22 relative (or 2^-25 absolute in the subnormal range of lo)."""
import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(synth_weights):
    from conette_amd.engine import Engine
    return {p: Engine(synth_weights, precision=p) for p in ("exact", "bf16", "mixed")}


def _wave(g):
    from conette_amd import synth
    n = [int(v) for v in g["lengths"]]
    return torch.from_numpy(synth.synth_waveforms(len(n), max(n), int(g["seed0"]), lengths=n))


@pytest.mark.parametrize("copies", [8, 5])
def test_exact_encoder_is_batch_invariant(copies, engines):
    eng = engines["exact"]
    g = G.load("b8_10s_beam3_all")
    w8 = _wave(g)
    fe, clip, taps = eng.encode(w8.repeat(copies, 1).cuda(), taps=True)
    fe8, clip8, taps8 = eng.encode(w8.cuda(), taps=True)
    torch.cuda.synchronize()
    for k in ["stem", "stage0_block0", "stage0", "down1", "stage1", "down2", "stage2", "down3", "stage3"]:
        v = taps[k].view(copies, 8, *taps[k].shape[1:])
        assert torch.equal(v, v[0:1].expand_as(v)), k
        assert torch.equal(v[0], taps8[k]), k
        got = G.sub(taps8[k].permute(0, 3, 1, 2).contiguous())
        np.testing.assert_allclose(got, g["sub_" + k], rtol=1e-3, atol=2e-4, err_msg=k)
    assert torch.equal(fe.view(copies, 8, *fe.shape[1:]), fe8[None].expand(copies, *fe8.shape))
    np.testing.assert_allclose(fe8.cpu().numpy(), g["frame_embs"], rtol=1e-3, atol=2e-4)


def test_mixed_is_bf16_encode_plus_exact_decode(engines, synth_weights):
    g = G.load("b4_10s_beam3_clotho")
    w = _wave(g).cuda()
    fe_m, clip_m = engines["mixed"].encode(w)
    fe_b, clip_b = engines["bf16"].encode(w)
    assert torch.equal(fe_m, fe_b) and torch.equal(clip_m, clip_b)
    t = fe_b.shape[1]
    lens = torch.full((w.shape[0],), t, dtype=torch.int32)
    bos = synth_weights["model.task_id_to_token_id"][torch.zeros(w.shape[0], dtype=torch.long)]
    fm = synth_weights["model.forbid_rep_mask"]
    a = engines["mixed"].decode(fe_b, lens, bos, fm, 3, 3, 20)
    b = engines["exact"].decode(fe_b, lens, bos, fm, 3, 3, 20)
    for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("name,enc,dec", [("mixed16", "f16", "exact"), ("bf16+f16dec", "bf16", "f16")])
def test_two_context_precisions_are_their_halves(name, enc, dec, synth_weights):
    """every two-context precision is bit for bit "engine A's encode fed to engine B's decode" (frame_embs fp32 is the interface)"""
    from conette_amd.engine import Engine
    g = G.load("b4_10s_beam3_clotho")
    w = _wave(g).cuda()
    both, e_enc, e_dec = Engine(synth_weights, precision=name), Engine(synth_weights, precision=enc), Engine(synth_weights, precision=dec)
    fe_m, clip_m = both.encode(w)
    fe_a, clip_a = e_enc.encode(w)
    assert torch.equal(fe_m, fe_a) and torch.equal(clip_m, clip_a)
    lens = torch.full((w.shape[0],), fe_a.shape[1], dtype=torch.int32)
    bos = synth_weights["model.task_id_to_token_id"][torch.zeros(w.shape[0], dtype=torch.long)]
    fm = synth_weights["model.forbid_rep_mask"]
    a = both.decode(fe_a, lens, bos, fm, 3, 3, 20)
    b = e_dec.decode(fe_a, lens, bos, fm, 3, 3, 20)
    for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs"):
        assert torch.equal(a[k], b[k]), k


def test_sp16_split_holds_22_bits():
    """host restatement of common.h cn_sp16_bits: hi = rn16(x), lo = rn16(x - hi)"""
    x = (torch.randn(100000) * torch.logspace(-4, 3, 100000)).float()
    x = x.clamp(-65504, 65504)
    hi = x.half()
    lo = (x - hi.float()).half()
    err = (hi.float() + lo.float() - x).abs()
    bound = torch.maximum(x.abs() * 2.0 ** -22, torch.full_like(x, 2.0 ** -25))
    assert bool((err <= bound).all()), float((err / bound).max())
