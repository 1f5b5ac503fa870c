"""Build an A/B flavour of the library: the current tree with some csrc files taken from another git revision, linked as
csrc/libtdeed_hip_<name>.so and loaded with TDEED_LIB_FLAVOUR=<name>.  Boxes differ by 2-3 % in the latency-bound launches,
so two forms of a kernel are compared inside ONE gpurun call.
    python tools/build_ab.py NAME REV file.hip [file.hip ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "t-deed_amd"))
import build as B  # noqa: E402


def main():
    name, rev, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    objs = []
    with tempfile.TemporaryDirectory() as td:
        # the headers of REV beside the files taken from it (an #include "common.h" resolves to the including file's directory
        # first): a change to a shared header is then part of the A/B too
        for h in ("common.h", "sgp_tile.h", "se_excite.h"):
            r = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:t-deed_amd/csrc/{h}"], capture_output=True, text=True)
            if r.returncode == 0:
                with open(os.path.join(td, h), "w") as f:
                    f.write(r.stdout)
        for s in B.SOURCES:
            obj = os.path.join(B.CSRC, s.replace(".hip", ".o"))
            if s in files:
                src = os.path.join(td, s)
                with open(src, "w") as f:
                    f.write(subprocess.run(["git", "-C", ROOT, "show", f"{rev}:t-deed_amd/csrc/{s}"], check=True,
                                           capture_output=True, text=True).stdout)
                obj = os.path.join(td, s.replace(".hip", ".o"))
                subprocess.run([B._hipcc(), *B.FLAGS, *B.EXTRA.get(s, []), "-I", B.CSRC, "-c", src, "-o", obj], check=True)
            objs.append(obj)
        lib = B.LIB.replace("libtdeed_hip.so", f"libtdeed_hip_{name}.so")
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
