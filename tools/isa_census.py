"""Per-kernel census of the instruction kinds that three round-6 fixes were about, from the compiled ISA: ds_bpermute
(cross-lane moves through the LDS crossbar: a `__shfl_xor` tree is a chain of them), integer divisions (v_rcp_iflag: one per
32-bit division, a 64-bit one is ~120 instructions around it), float divisions (v_div_scale pairs), dword-sized global loads
(per-element loads of constants) -- for kernels whose workgroups live a few microseconds each of these is on the critical path.
    python tools/isa_census.py [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "t-deed_amd", "csrc")
PAT = dict(bpermute=r"\bds_bpermute_b32", idiv=r"\bv_rcp_iflag_f32", fdiv=r"\bv_div_scale_f32", ld_dword=r"\bglobal_load_dword\b",
           ld_x4=r"\bglobal_load_dwordx4", mfma=r"\bv_mfma", waitcnt0=r"s_waitcnt vmcnt\(0\)", barrier=r"\bs_barrier")


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("conv.hip", "gsf.hip", "front.hip", "gemm.hip", "bneck.hip", "sgp_gemm.hip",
                                                              "sgp_fused.hip", "misc.hip")]
    print(f"{'kernel':70s} {'lines':>6} " + " ".join(f"{k:>8}" for k in PAT))
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                            "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", out, f], check=True, capture_output=True)
            txt = open(out).read()
        for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", txt, re.S | re.M):
            name = subprocess.run(["c++filt", "-p", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            body = m.group(2)
            cnt = {k: len(re.findall(p, body)) for k, p in PAT.items()}
            print(f"{name[:70]:70s} {body.count(chr(10)):6d} " + " ".join(f"{cnt[k]:8d}" for k in PAT))


if __name__ == "__main__":
    main()
