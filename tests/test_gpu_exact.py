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


def _decode_inputs(synth_weights, n, seed=99, t=31):
    g = torch.Generator().manual_seed(seed)
    fe = torch.randn((n, t, 768), generator=g) * 0.5
    lens = torch.randint(t // 2, t + 1, (n,), generator=g, dtype=torch.int32)
    bos = synth_weights["model.task_id_to_token_id"][torch.randint(0, 3, (n,), generator=g)]
    return fe.cuda(), lens, bos, synth_weights["model.forbid_rep_mask"]


@pytest.mark.parametrize("beam,max_pred,n", [(3, 20, 64), (8, 12, 9), (1, 20, 5), (5, 30, 7)])
def test_exact_fused_decoder_equals_the_per_sublayer_path(beam, max_pred, n, engines, synth_weights):
    """Round 4: the exact decode runs the fused block / FFN kernels on hi / lo passes (dec_block.h, dec_ffn.h: 3 launches per
    layer).  The per-sub-layer path it replaces (11 launches per layer, cn_gemm2<SP>) stays as the cross-check: same ids, same
    hypotheses, scores within the fp32 tolerance -- both are 22-bit-operand evaluations of the same sums in another order."""
    eng = engines["exact"]
    fe, lens, bos, fm = _decode_inputs(synth_weights, n)
    a = eng.decode(fe, lens, bos, fm, beam, 3, max_pred, want_trace=True)
    eng.set_decode_fusion(False)
    try:
        b = eng.decode(fe, lens, bos, fm, beam, 3, max_pred, want_trace=True)
    finally:
        eng.set_decode_fusion(True)
    torch.cuda.synchronize()
    # (a caption may differ only where the search meets a tie narrower than the two paths' rounding noise: none on these seeds)
    for k in ("best_preds", "mult_preds", "trace_sel"):
        assert torch.equal(a[k], b[k]), k
    torch.testing.assert_close(a["best_lprobs"], b["best_lprobs"], rtol=0, atol=2e-5)
    torch.testing.assert_close(a["mult_lprobs"], b["mult_lprobs"], rtol=0, atol=2e-5)
    torch.testing.assert_close(a["trace_val"], b["trace_val"], rtol=0, atol=2e-4)


def test_exact_decode_graph_is_the_fused_launch_sequence(engines, synth_weights):
    """<= 400 graph nodes per 20-step search (1 386 with one launch per sub-layer; the bf16 structure has 306)"""
    eng = engines["exact"]
    fe, lens, bos, fm = _decode_inputs(synth_weights, 16)
    side = torch.cuda.Stream()  # (the legacy default stream cannot be captured)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):  # eager, capture, replay
            eng.decode(fe, lens, bos, fm, 3, 3, 20, clone=False, slot=7)
    torch.cuda.synchronize()
    assert 0 < eng.decode_graph_nodes() <= 400, eng.decode_graph_nodes()


@pytest.mark.parametrize("fused", [True, False])
def test_exact_decode_at_4096_rows(fused, engines, synth_weights):
    """R = batch x beam >= 4096 takes the 128-row GEMM tiles in the per-sub-layer path; its split-K FFN2 must write every slab
    the LayerNorm sums (`splits` was dropped there).  Checked against the same clips decoded 64 at a time."""
    eng = engines["exact"]
    n, beam = 512, 8
    fe, lens, bos, fm = _decode_inputs(synth_weights, n, seed=5)
    fe = fe[:64].repeat(n // 64, 1, 1).contiguous()
    lens, bos = lens[:64].repeat(n // 64), bos[:64].repeat(n // 64)
    eng.set_decode_fusion(fused)
    try:
        big = eng.decode(fe, lens, bos, fm, beam, 3, 8)
        small = eng.decode(fe[:64], lens[:64], bos[:64], fm, beam, 3, 8)
    finally:
        eng.set_decode_fusion(True)
    torch.cuda.synchronize()
    for k in ("best_preds", "mult_preds"):
        v = big[k].view(n // 64, 64, *big[k].shape[1:])
        assert torch.equal(v, small[k][None].expand_as(v)), k
    torch.testing.assert_close(big["best_lprobs"].view(n // 64, 64), small["best_lprobs"][None].expand(n // 64, 64), rtol=0, atol=2e-5)


@pytest.mark.parametrize("prec", ["exact", "bf16"])
def test_wide_search_blocks_return_the_narrow_search_bits(prec, engines, synth_weights):
    """Round 5: a search of >= 512 rows runs the decoder block kernel with 8 rows per block (two per row wave), a shorter one
    with 4 (dec_block.h DB_WIDE_ROWS / DB_WIDE_R, chosen at launch).  The arithmetic is row-local: 32 clips x beam 3 = 96 rows
    (narrow) tiled eight times = 768 rows (wide) must return the same ids AND the same score bits eight times."""
    eng = engines[prec]
    fe, lens, bos, fm = _decode_inputs(synth_weights, 32, seed=11)
    narrow = eng.decode(fe, lens, bos, fm, 3, 3, 20)
    wide = eng.decode(fe.repeat(8, 1, 1).contiguous(), lens.repeat(8), bos.repeat(8), fm, 3, 3, 20)
    torch.cuda.synchronize()
    for k in ("best_preds", "mult_preds", "best_lprobs", "mult_lprobs"):
        v = wide[k].view(8, 32, *wide[k].shape[1:])
        assert torch.equal(v, narrow[k][None].expand_as(v)), k
