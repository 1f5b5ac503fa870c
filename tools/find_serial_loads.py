"""Find SERIALISED load loops in the device code: an inner loop whose body holds a global/buffer load and waits for it
(`s_waitcnt vmcnt(0)`) before the back edge -- every trip is one exposed memory round trip (0.5-1 us from L2 / fabric).
Round 6 found the SGP stage's GroupNorm prologue (4 such loops in a row = the 2.7 us the round-5 stamps saw) and the
LayerNorm statistics loads of the front kernels this way.

    python tools/find_serial_loads.py [file.hip ...]      (default: every csrc/*.hip)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "t-deed_amd", "csrc")


def scan(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                        "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", out, src], check=True,
                       capture_output=True)
        lines = open(out).read().splitlines()
    res = []
    kern = None
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m and "Inner Loop Header" in " ".join(lines[i:i + 4]):
            lab = m.group(1)
            body = []
            j = i + 1
            while j < len(lines) and lines[j].lstrip().startswith(";"):
                j += 1
            while j < len(lines) and not re.match(r"^\.LBB", lines[j]) and j - i < 400:
                body.append(lines[j])
                if re.search(r"s_cbranch_\w+\s+" + re.escape(lab) + r"\b", lines[j]):
                    break
                j += 1
            txt = "\n".join(body)
            closes = bool(body) and re.search(r"s_cbranch_\w+\s+" + re.escape(lab) + r"\b", body[-1])
            nload = len(re.findall(r"\b(global_load|buffer_load|flat_load)", txt))
            if closes and nload and re.search(r"s_waitcnt vmcnt\(0\)", txt) and "v_mfma" not in txt:
                res.append((kern, lab, len(body), nload))
        i += 1
    return res


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        seen = {}
        for kern, lab, n, nload in scan(f):
            d = subprocess.run(["c++filt", "-p", kern], capture_output=True, text=True).stdout.strip()
            base = re.sub(r"<.*", "", d.replace("(anonymous namespace)::", ""))
            seen.setdefault(base, []).append((d, lab, n, nload))
        for base, hits in seen.items():
            insts = sorted(set(h[0] for h in hits))
            per = len(hits) // max(len(insts), 1)
            print(f"{os.path.basename(f)}: {base}: {len(insts)} instance(s), ~{per} serial loop(s) each; e.g. {hits[0][0][:90]} "
                  + ", ".join(f"{h[1]}({h[2]} instr, {h[3]} ld)" for h in hits[:per]))


if __name__ == "__main__":
    main()
