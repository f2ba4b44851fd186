"""GPU: results must not depend on what else runs on the GPU at the same time.

bench.py overlaps the encode of batch i with the decodes of batches i-1 and i-2 on three HIP streams; these tests pin that a
step gives the SAME BITS alone and inside that pipeline.  They exist because it once did not: with the decoder's small GEMM
workgroups sharing a compute unit, the log-mel front end produced wrong frames (a 16-lane quarter of one VALU result at a
time, 1-10 frames per 64 000), until the front end's workgroups were made to own their compute unit's LDS
(frontend.hip, profiles/r02_notes.md).  Also here: no entry point writes past the workspace size it asks for."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, BEAM, MAX_PRED, MIN_PRED, L = 64, 3, 20, 3, 320000


@pytest.fixture(scope="module")
def eng(synth_weights_np):
    from conette_amd.engine import Engine
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth_weights_np.items()}
    return Engine(sd, precision="bf16")


@pytest.fixture(scope="module")
def batch(eng):
    from conette_amd import synth
    dev = torch.device("cuda:0")
    wave = torch.from_numpy(synth.synth_waveforms(B, L, 1234)).to(dev)
    t = eng.lib.conette_num_audio_frames(L)
    lens = torch.full((B,), t, dtype=torch.int32, device=dev)
    bos = torch.full((B,), 1, dtype=torch.int32, device=dev)
    return dev, wave, t, lens, bos


def test_frontend_beside_a_decode_is_bit_stable(eng, batch):
    """log-mel of the same waveforms, alone and while decodes (fused and one-launch-per-sub-layer) run on another stream"""
    dev, wave, t, lens, bos = batch
    fe, _ = eng.encode(wave)
    ref = eng.frontend_logmel(wave)
    torch.cuda.synchronize()
    s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
    try:
        for fused in (True, False):
            eng.set_decode_fusion(fused)
            bad = 0
            for _ in range(12):
                with torch.cuda.stream(s_b):
                    eng.decode(fe, lens, bos, None, BEAM, MIN_PRED, MAX_PRED, slot=1)
                with torch.cuda.stream(s_a):
                    got = [eng.frontend_logmel(wave) for _ in range(4)]
                torch.cuda.synchronize()
                bad += sum(0 if torch.equal(g, ref) else 1 for g in got)
            assert bad == 0, (fused, bad)
    finally:
        eng.set_decode_fusion(True)


def test_every_encoder_tap_beside_decodes_is_bit_stable(eng, batch):
    """Every intermediate of the encoder (log-mel, stem, the output of each of the 18 blocks, the three downsample layers,
    frame_embs, clip_probs) alone and while decodes run on two other streams: the front end was the kernel that broke, but
    the stem, depthwise-conv and MLP kernels share compute units with the same decode kernels."""
    dev, wave, t, lens, bos = batch
    nb = 16
    w16 = wave[:nb].contiguous()
    fe_ref, clip_ref, taps_ref = eng.encode(w16, taps="blocks")
    fe64, _ = eng.encode(wave)
    torch.cuda.synchronize()
    s_enc = torch.cuda.Stream(dev)
    s_decs = [torch.cuda.Stream(dev, priority=-1) for _ in range(2)]
    bad = {}
    for rnd in range(8):
        for k, sd in enumerate(s_decs):
            with torch.cuda.stream(sd):
                eng.decode(fe64, lens, bos, None, BEAM, MIN_PRED, MAX_PRED, slot=30 + k)
        with torch.cuda.stream(s_enc):
            fe, clip, taps = eng.encode(w16, taps="blocks", slot=1)
        torch.cuda.synchronize()
        for name, ref in list(taps_ref.items()) + [("frame_embs", fe_ref), ("clip_probs", clip_ref)]:
            got = fe if name == "frame_embs" else clip if name == "clip_probs" else taps[name]
            if not torch.equal(got, ref):
                bad[name] = bad.get(name, 0) + 1
        del fe, clip, taps
    assert bad == {}, bad


@pytest.mark.parametrize("n_slot", [2, 3])
def test_pipelined_steps_equal_the_solo_pass(n_slot, eng, batch):
    """bench.py's pipeline (one encode stream, n_slot - 1 decode streams, per-slot buffers): every step's frame embeddings,
    captions and scores equal the un-pipelined pass of the same batch bit for bit"""
    dev, wave, t, lens, bos = batch
    fe0 = eng.decode_input_buffer(B, t, BEAM, MAX_PRED, slot=0)
    clip = torch.empty((B, 527), device=dev)
    eng.encode(wave, out=(fe0, clip))
    out0 = eng.decode(fe0, lens, bos, None, BEAM, MIN_PRED, MAX_PRED, clone=True, slot=0)
    torch.cuda.synchronize()
    fe_ref, p_ref, l_ref = fe0.clone(), out0["best_preds"], out0["best_lprobs"]
    s_enc = torch.cuda.Stream(dev)
    n_dec = n_slot - 1
    s_decs = [torch.cuda.Stream(dev, priority=-1) for _ in range(n_dec)]
    sl = [dict(fe=eng.decode_input_buffer(B, t, BEAM, MAX_PRED, slot=10 + k), clip=torch.empty((B, 527), device=dev),
               enc_done=torch.cuda.Event(), dec_done=torch.cuda.Event()) for k in range(n_slot)]
    keep = []
    for i in range(72):  # (72 steps: the race of an under-counted ring wait showed in ~1 of 30 steps)
        s = sl[i % n_slot]
        sd = s_decs[i % n_dec]
        with torch.cuda.stream(s_enc):
            if i >= n_slot:
                s_enc.wait_event(s["dec_done"])
            eng.encode(wave, out=(s["fe"], s["clip"]), slot=i & 1)
            s["enc_done"].record(s_enc)
        with torch.cuda.stream(sd):
            sd.wait_event(s["enc_done"])
            out = eng.decode(s["fe"], lens, bos, None, BEAM, MIN_PRED, MAX_PRED, clone=True, slot=10 + i % n_slot)
            fe_c = s["fe"].clone()
            s["dec_done"].record(sd)
        keep.append((out["best_preds"], out["best_lprobs"], fe_c))
    torch.cuda.synchronize()
    bad = [i for i, (p, lp, f) in enumerate(keep) if not (torch.equal(f, fe_ref) and torch.equal(p, p_ref) and torch.equal(lp, l_ref))]
    assert bad == [], bad


def test_entry_points_stay_inside_their_workspaces(eng):
    """1 MiB of guard bytes behind every workspace the library is handed (encode, decode, forcing, greedy; three shapes)"""
    from conette_amd import synth
    dev = torch.device("cuda:0")
    guard = 1 << 20
    state = {}
    orig = eng._workspace

    def guarded(key, nbytes):
        ws = eng._ws.get(key)
        if ws is None or ws.numel() < nbytes + guard:
            ws = torch.empty(int(nbytes) + guard, dtype=torch.uint8, device=dev)
            eng._ws[key] = ws
        ws[nbytes:].fill_(0xAB)
        state["last"] = (key, ws, int(nbytes))
        return ws[:nbytes]

    def clean():
        torch.cuda.synchronize()
        key, ws, n = state["last"]
        return bool((ws[n:] == 0xAB).all()), key

    eng._workspace = guarded
    try:
        for b, secs in ((16, 10), (3, 7), (8, 3)):
            n = secs * 32000
            wave = torch.from_numpy(synth.synth_waveforms(b, n, 77)).to(dev)
            t = eng.lib.conette_num_audio_frames(n)
            lens = torch.full((b,), t, dtype=torch.int32, device=dev)
            bos = torch.full((b,), 1, dtype=torch.int32, device=dev)
            fe, _ = eng.encode(wave)
            assert clean() == (True, "enc")
            eng.decode(fe, lens, bos, None, BEAM, MIN_PRED, MAX_PRED, slot=20)
            assert clean()[0]
            eng.forcing(fe, lens, torch.randint(3, 5000, (b, 12), device=dev))
            assert clean()[0]
            eng.greedy(fe, lens, bos, None, MIN_PRED, MAX_PRED)
            assert clean()[0]
    finally:
        eng._workspace = orig
        for k in [k for k in eng._ws]:
            del eng._ws[k]
