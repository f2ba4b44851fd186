"""Same-box A/B of library builds for an encoder kernel class: per build (a child process each) the class time of one encode at
B = 64 (HIP events around every launch of the class), the encoder-alone time per pass, and a hash of the frame embeddings.
    python tools/lab/dw_ab.py libA.so libB.so ... [--cls dwconv_ln] [--prec bf16]
(libraries are looked up in tools/lab/; the in-tree library is restored afterwards)"""
import hashlib, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "conette-audio-captioning_amd", "libconette_hip.so")

CHILD = r'''
import sys, os, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
cls, prec = sys.argv[1], sys.argv[2]
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision=prec)
eng.set_encode_reserved_cus(24)
B = 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
fe = torch.empty((B, 31, 768), device="cuda"); clip = torch.empty((B, 527), device="cuda")
for _ in range(5): eng.encode(wave, out=(fe, clip))
torch.cuda.synchronize()
h = hashlib.sha256(fe.cpu().numpy().tobytes()).hexdigest()[:16]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 30
ev[0].record()
for _ in range(n): eng.encode(wave, out=(fe, clip))
ev[1].record(); torch.cuda.synchronize()
enc_ms = ev[0].elapsed_time(ev[1]) / n
eng.profile_enable((cls,))
for _ in range(10): eng.encode(wave, out=(fe, clip))
torch.cuda.synchronize()
ms, cnt = eng.profile_read()[cls]
print(f"class {cls}: {ms / 10:.4f} ms per encode ({cnt // 10} launches) | encoder alone {enc_ms:.3f} ms | frame_embs sha {h}")
''' % ROOT

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cls = next((a.split("=")[1] for a in sys.argv if a.startswith("--cls=")), "dwconv_ln")
    prec = next((a.split("=")[1] for a in sys.argv if a.startswith("--prec=")), "bf16")
    keep = LIB + ".keep"
    shutil.copy(LIB, keep)
    try:
        for rnd in range(2):
            for lib in args:
                shutil.copy(os.path.join(ROOT, "tools", "lab", lib), LIB)
                r = subprocess.run([sys.executable, "-c", CHILD, cls, prec], capture_output=True, text=True, timeout=600)
                print(f"{lib:28s}", (r.stdout.strip().splitlines() or ["(no output)"])[-1], flush=True)
                if r.returncode != 0:
                    print(r.stderr[-1500:], flush=True)
    finally:
        shutil.move(keep, LIB)

if __name__ == "__main__":
    main()
