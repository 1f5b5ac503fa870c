"""Rounds of workgroups per launch, from a rocprofv3 kernel trace: for every distinct (kernel, grid, workgroup, LDS, VGPR)
of the trace the workgroups the launch has, how many fit the chip at once (256 CUs x the per-CU limit of registers, LDS and
32 waves) and the quotient -- a launch a little above a whole number of rounds pays a nearly empty extra round (round 6: the
GroupNorm prologue that took sgp_gemm from three resident workgroups per CU to two made 736 workgroups two rounds).
    python tools/launch_rounds.py <dir with *kernel_trace.csv> [min_us]"""
import csv
import glob
import math
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        key = (r["Kernel_Name"][:70], grid // max(wg, 1), wg, int(r.get("LDS_Block_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0)
               + int(r.get("Accum_VGPR_Count", 0) or 0))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = []
    for (name, nwg, wg, lds, vg), (n, us) in agg.items():
        waves = (wg + 63) // 64
        alloc = max(8, (vg + 7) // 8 * 8)
        by_reg = min(8, 512 // alloc) * 4 // waves if waves <= 4 * min(8, 512 // alloc) else 0
        by_lds = (160 * 1024) // lds if lds else 99
        by_waves = 32 // waves
        occ = max(1, min(by_reg if by_reg else 1, by_lds, by_waves, 16))
        slots = occ * 256
        rows.append((us / n, name, nwg, wg, lds, vg, occ, nwg / slots, n))
    rows.sort(reverse=True)
    print(f"{'avg us':>8} {'WGs':>7} {'thr':>5} {'LDS':>7} {'VGPR':>5} {'WG/CU':>5} {'rounds':>7}  kernel (launches)")
    for us, name, nwg, wg, lds, vg, occ, rounds, n in rows:
        if us < min_us:
            continue
        flag = " <-- just over a round" if 0.02 < (rounds - math.floor(rounds)) < 0.25 and rounds > 1 else ""
        print(f"{us:8.1f} {nwg:7d} {wg:5d} {lds:7d} {vg:5d} {occ:5d} {rounds:7.2f}  {name} ({n}){flag}")


if __name__ == "__main__":
    main()
