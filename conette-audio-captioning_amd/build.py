"""Build the gfx950 shared library in-tree: hipcc -> conette-audio-captioning_amd/libconette_hip.so."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libconette_hip.so")
SOURCES = ["api.hip", "frontend.hip", "encoder.hip", "decoder.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DCN_RC2_GELU_PK"]
if os.environ.get("CN_G2_PROF"):  # profiling build: phase stamps inside the encoder GEMM (tools/g2prof.py)
    FLAGS.append("-DCN_G2_PROF")
if os.environ.get("CN_NO_RS"):     # A/B build: the chained stage-2 kernel of round 2 instead of the role-split one
    FLAGS.append("-DCN_NO_RS")
if os.environ.get("CN_NO_SAT8"):   # A/B build: fp16 GELU outputs of the fused MLP without the saturating v_pk_min_f16
    FLAGS.append("-DCN_NO_SAT8")
if os.environ.get("CN_DB_ROWS"):   # A/B build: rows per decoder block kernel (dec_block.h: 4)
    FLAGS.append("-DDB_ROWS=" + os.environ["CN_DB_ROWS"])
if os.environ.get("CN_G2_NOACT"):
    FLAGS.append("-DCN_G2_NOACT")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "conette_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cc = hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
