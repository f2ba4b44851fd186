"""Build the gfx950 shared library in-tree: hipcc -> conette-audio-captioning_amd/libconette_hip.so."""
from __future__ import annotations

import json
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
for knob in ("FE_NOFILL", "FE_POISON", "FE_NW", "FE_VARIANT"):   # lab builds of the log-mel kernel (tools/lab/logmel_repro.sh)
    if os.environ.get(knob):
        FLAGS.append(f"-D{knob}" + ("" if os.environ[knob] == "1" and knob != "FE_NW" else "=" + os.environ[knob]))
if os.environ.get("CN_EXTRA_FLAGS"):   # lab builds: extra compiler flags for every file (e.g. -fno-slp-vectorize)
    FLAGS += os.environ["CN_EXTRA_FLAGS"].split()
if os.environ.get("CN_NO_SAT8"):   # A/B build: fp16 GELU outputs of the fused MLP without the saturating v_pk_min_f16
    FLAGS.append("-DCN_NO_SAT8")
if os.environ.get("CN_DB_ROWS"):   # A/B build: rows per decoder block kernel (dec_block.h: 4)
    FLAGS.append("-DDB_ROWS=" + os.environ["CN_DB_ROWS"])
if os.environ.get("CN_DB_ROWS_SP"):   # A/B build: rows per decoder block kernel in the exact precision (dec_block.h: 4)
    FLAGS.append("-DDB_ROWS_SP=" + os.environ["CN_DB_ROWS_SP"])
if os.environ.get("CN_DB_WIDE_ROWS"):   # A/B build: rows per block of a wide search (dec_block.h: 8; 4 = off)
    FLAGS.append("-DDB_WIDE_ROWS=" + os.environ["CN_DB_WIDE_ROWS"])
if os.environ.get("CN_DB_WIDE_ROWS_SP"):   # A/B build: the same for the exact precision (dec_block.h: 8; 4 = off)
    FLAGS.append("-DDB_WIDE_ROWS_SP=" + os.environ["CN_DB_WIDE_ROWS_SP"])
if os.environ.get("CN_DB_WIDE_R"):      # A/B build: rows from which a search counts as wide (dec_block.h: 512)
    FLAGS.append("-DDB_WIDE_R=" + os.environ["CN_DB_WIDE_R"])
if os.environ.get("CN_DB_XCDS"):   # A/B build: XCDs whose workgroups work in the decoder block kernel (dec_block.h: 8 = all)
    FLAGS.append("-DDB_XCDS=" + os.environ["CN_DB_XCDS"])
if os.environ.get("CN_G2_NOACT"):
    FLAGS.append("-DCN_G2_NOACT")
if os.environ.get("CN_RS_PRIO"):    # A/B build: static wave priority in the role-split fused MLP (encoder.hip: 8 = B waves)
    FLAGS.append("-DCN_RS_PRIO=" + os.environ["CN_RS_PRIO"])
if os.environ.get("CN_RS16"):       # A/B build: 0 = stage 2 of the bf16 / f16 precisions on the 32x32x16 role-split kernel (mlp_rs.h)
    FLAGS.append("-DCN_RS16=" + os.environ["CN_RS16"])
if os.environ.get("CN_STEM_MFMA"):  # A/B build: 0 = the VALU stem kernel
    FLAGS.append("-DCN_STEM_MFMA=" + os.environ["CN_STEM_MFMA"])
for _k in ("CN_DW96_S", "CN_DW96_TH", "CN_DW192_S", "CN_DW192_TH"):   # A/B builds: tile of the depthwise kernel of stages 0 / 1 (encoder.hip)
    if os.environ.get(_k):
        FLAGS.append(f"-D{_k}=" + os.environ[_k])
if os.environ.get("CN_DW_DOT2"):    # A/B build: 0 = the depthwise conv of the fp16 stream with one v_fma_mix_f32 per tap (encoder.hip)
    FLAGS.append("-DCN_DW_DOT2=" + os.environ["CN_DW_DOT2"])
if os.environ.get("CN_FW_SPLIT"):   # A/B build: threads per channel of the full-width depthwise kernel at C = 384 (encoder.hip: 2)
    FLAGS.append("-DCN_FW_SPLIT=" + os.environ["CN_FW_SPLIT"])
if os.environ.get("CN_FW_TH"):      # A/B build: output rows per block of the full-width depthwise kernel at C = 384 (encoder.hip: 4)
    FLAGS.append("-DCN_FW_TH=" + os.environ["CN_FW_TH"])


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


# Files whose complex / horizontal arithmetic hipcc's SLP vectoriser would turn into packed-fp32 instructions with a swizzled src1
# (`v_pk_*_f32 ... op_sel:[0,1]`): the form that MI355X executes wrongly in lanes 48-63 beside another wave's
# v_mfma_f32_16x16x32_bf16 (isa_lint.py; profiles/r03_notes.md section 8).  Explicit two-element vector code (the GELU epilogues)
# is not affected by the flag and does not use that form; isa_lint.lint_library() checks the linked library either way.
FILE_FLAGS = {} if os.environ.get("CN_ALLOW_PK_HAZARD") else {"frontend.hip": ["-fno-slp-vectorize"], "decoder.hip": ["-fno-slp-vectorize"]}


SIDECAR = os.path.join(HERE, "libconette_hip.build.json")   # what LIB was built from: source hash + flags (travels with the .so)


_TOOLCHAIN = None


def toolchain_id() -> str:
    """What identifies the compiler: the first lines of `hipcc --version` (HIP version, clang version + commit), cached."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
            _TOOLCHAIN = "\n".join(l.strip() for l in out.splitlines()[:3])
        except Exception as e:   # no compiler here (a box that only runs the prebuilt library): the identity is "unknown"
            _TOOLCHAIN = f"unknown ({type(e).__name__})"
    return _TOOLCHAIN


def source_hash() -> str:
    """sha256 over every file of csrc/ + the public header + the compiler flags (per file) -- the identity of a build.  A
    rebuild of the same sources gives another BINARY hash (checked in round 3), so staleness and 'which build do these PMC
    tables describe' (bench.py pmc_is_current, tools/profile_round.sh) are both keyed on this instead."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith(".")) + [os.path.join(HERE, "..", "include", "conette_hip.h")]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
    h.update(repr((FLAGS, sorted(FILE_FLAGS.items()), os.environ.get("CN_FE_SRC", ""), os.environ.get("CN_FE_FLAGS", ""))).encode())
    h.update(toolchain_id().encode())   # a compiler upgrade is another build: objects are not reused, PMC tables turn stale
    return h.hexdigest()


def built_source_hash():
    """the source hash recorded beside the library when it was linked (None: no library or no record)"""
    try:
        return json.load(open(SIDECAR)).get("source_sha256") if os.path.exists(LIB) else None
    except (OSError, ValueError):
        return None


def _stale() -> bool:
    return built_source_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cc = hipcc()
    src_hash = source_hash()   # taken BEFORE compiling: an edit made while hipcc runs must leave the result stale
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def one(src):
        import hashlib
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        extra = list(FILE_FLAGS.get(src, []))
        if src == "frontend.hip":
            extra += os.environ.get("CN_FE_FLAGS", "").split()   # lab: flags for the front end only
        path = os.path.join(CSRC, src)
        if src == "frontend.hip" and os.environ.get("CN_FE_SRC"):   # lab: the instrumented copy (tools/lab/frontend_lab.hip)
            path = os.path.abspath(os.environ["CN_FE_SRC"])
        cmd = [cc, *FLAGS, *extra, "-c", path, "-o", obj]
        # per-object cache: the object is reused when the command line and every repo file it was compiled from (-MD list) are
        # unchanged -- an edit to decoder.hip or dec_block.h recompiles one file, not four
        stamp = obj + ".stamp"

        def digest(files):
            h = hashlib.sha256((repr(cmd) + toolchain_id()).encode())
            for f in sorted(files):
                h.update(f.encode() + b"\0" + open(f, "rb").read())
            return h.hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(stamp):
            try:
                rec = json.load(open(stamp))
                if all(os.path.exists(f) for f in rec["deps"]) and digest(rec["deps"]) == rec["digest"]:
                    return obj
            except (OSError, ValueError, KeyError):
                pass
        import time
        t_start = time.time()
        r = subprocess.run(cmd + ["-MD", "-MF", obj + ".d"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        try:   # repo files among the dependencies (system / ROCm headers belong to the image)
            toks = open(obj + ".d").read().replace("\\\n", " ").split()
            root = os.path.abspath(os.path.join(HERE, ".."))
            deps = sorted({os.path.abspath(t) for t in toks[1:] if os.path.abspath(t).startswith(root)})
            if all(os.path.getmtime(f) < t_start for f in deps):   # (a file edited while hipcc ran: no stamp, recompiled next time)
                json.dump({"deps": deps, "digest": digest(deps)}, open(stamp, "w"))
            elif os.path.exists(stamp):
                os.remove(stamp)
        except OSError:
            pass
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SOURCES))
    # link to a temporary path, lint THAT, and only then move it into place: a library that has not passed the lint (a hazard
    # hit, or the lint itself failing: llvm-objdump missing) never sits at the path engine.py loads
    tmp = LIB + ".tmp"
    try:
        r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if not os.environ.get("CN_ALLOW_PK_HAZARD"):   # (lab builds that reproduce the fault set it)
            import importlib.util   # (by path: this file is loaded as a package module, as a script, and by __graft_entry__ under a private name)
            spec = importlib.util.spec_from_file_location("_conette_isa_lint", os.path.join(HERE, "isa_lint.py"))
            isa_lint = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(isa_lint)
            bad = isa_lint.lint_library(tmp)
            if bad:
                raise RuntimeError("isa_lint: the library contains packed-fp32 instructions of the form MI355X executes wrongly beside bf16 MFMAs "
                                   f"(v_pk_*_f32 op_sel:[0,1..]): {bad[:4]} ... {len(bad)} in all")
        if os.path.exists(SIDECAR):
            os.remove(SIDECAR)       # (never a new library beside an old record)
        os.replace(tmp, LIB)
        with open(SIDECAR, "w") as f:
            json.dump({"source_sha256": src_hash, "flags": FLAGS, "file_flags": FILE_FLAGS,
                       "linted": not os.environ.get("CN_ALLOW_PK_HAZARD")}, f)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
