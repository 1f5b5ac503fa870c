"""Compile csrc/*.hip for gfx950 into csrc/libtdeed_hip.so (in-tree; travels with the snapshot).

hipcc cross-compiles without a GPU.  Objects are cached by source mtime so a rebuild after
touching one file takes seconds.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libtdeed_hip.so")
SOURCES = ["gemm.hip", "conv.hip", "front.hip", "gsf.hip", "sgp.hip", "sgp_fused.hip", "sgp_gemm.hip", "bneck.hip", "sgp_bwd.hip", "trunk_bwd.hip", "trunk_bwd2.hip", "trunk_bwd3.hip", "gsf_bwd.hip", "train.hip", "misc.hip", "augment.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wall",
         "-Wno-unused-function"]
# MFMA accumulators in VGPRs instead of AGPRs for the files whose epilogues are VALU-bound: every accumulator element
# otherwise costs a v_accvgpr_read before the BatchNorm / ReLU arithmetic (s1_front: 24 of ~170 vector instructions per
# 16-pixel tile).  Not for the tiled GEMM: its 64 accumulator registers would leave the VGPR budget of 3 waves per SIMD.
VGPR_MFMA = set(filter(None, os.environ.get("TDEED_VGPR_MFMA", "front.hip").split(",")))
EXTRA = {s: ["-mllvm", "-amdgpu-mfma-vgpr-form"] for s in VGPR_MFMA}


def _hipcc():
    for p in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(p) or p == "hipcc":
            return p


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


DEBUG_FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-DTDEED_DEBUG=1", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wall",
               "-Wno-unused-function"]
LIB_DEBUG = os.path.join(CSRC, "libtdeed_hip_dbg.so")


def build(force=False, verbose=True, flavour="release", only=None):
    """flavour "debug": -O1 -g -DTDEED_DEBUG=1 (device asserts on LDS offsets and table indices, common.h) into
    csrc/libtdeed_hip_dbg.so; loaded instead of the release library when TDEED_LIB_FLAVOUR=debug.  only: compile just these
    sources and do not link (a quick does-it-compile check)."""
    debug = flavour == "debug"
    flags, lib, osuf = (DEBUG_FLAGS, LIB_DEBUG, ".dbg.o") if debug else (FLAGS, LIB, ".o")
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "sgp_tile.h"), os.path.join(CSRC, "se_excite.h"),
            os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "tdeed_hip.h")]
    jobs = []
    objs = []
    for s in (only or SOURCES):
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", osuf))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc(), *flags, *([] if debug else EXTRA.get(s, [])), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if only:
        return objs
    if jobs or not os.path.exists(lib):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, flavour="debug" if "--debug" in sys.argv else "release"))
