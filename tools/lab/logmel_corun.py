"""Which KIND of work beside the log-mel kernel (lab build sharing its compute units) disturbs it?  Synthetic co-runners
(csrc/frontend.hip, CN_LAB): 0 allocation only, 1 LDS write+barrier+read, 2 global loads, 3 fp32 VALU, 4 MFMA, 5 LDS-DMA, 6 LDS reads,
7 LDS writes, 8 packed fp32 with modifiers, 9 barriers; 'decode' = the real decoder."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
dev = torch.device("cuda:0")
B = 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).to(dev)
t = eng.lib.conette_num_audio_frames(320000)
lens = torch.full((B,), t, dtype=torch.int32, device=dev)
bos = torch.full((B,), 1, dtype=torch.int32, device=dev)
fe0, _ = eng.encode(wave)
lm0 = eng.frontend_logmel(wave)
torch.cuda.synchronize()
src = torch.randn(1 << 20, device=dev)
dst = torch.zeros(4096, device=dev)
s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
fn = eng.lib.conette_lab_corun
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 5 + [C.c_void_p] * 3
grid, iters, reps, lds = (int(os.environ.get(k, d)) for k, d in (("GRID", "96"), ("ITERS", "200"), ("REPS", "300"), ("LDS", "32768")))
eng.set_decode_fusion(False)
for kind in os.environ.get("KINDS", "none,decode,0,1,2,3,4,5,6,7,8,9").split(","):
    bad_runs = bad_frames = 0
    n = int(os.environ.get("RUNS", "6"))
    for it in range(n):
        with torch.cuda.stream(s_b):
            if kind == "decode":
                eng.decode(fe0, lens, bos, None, 3, 3, 20, slot=1)
            elif kind != "none":
                rc = fn(int(kind), grid, iters, reps, lds, src.data_ptr(), dst.data_ptr(), s_b.cuda_stream)
                assert rc == 0, rc
        with torch.cuda.stream(s_a):
            lms = [eng.frontend_logmel(wave) for _ in range(2)]
        torch.cuda.synchronize()
        for x in lms:
            nb = int((x != lm0).flatten(2).any(dim=2).sum()) if x.ndim == 3 else int((x != lm0).any(dim=-1).sum())
            bad_frames += nb
            bad_runs += nb > 0
    print(f"co-runner {kind:7s}: {bad_runs} of {2 * n} concurrent log-mel runs differ from the solo result ({bad_frames} wrong frames)", flush=True)
