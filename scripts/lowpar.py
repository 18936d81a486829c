#!/usr/bin/env python3
"""List long kernels that launch few workgroups (candidates for more parallelism) from a rocprofv3 kernel trace of bench.py."""
import collections
import csv
import sys


def main(path, n=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
    seg = rows[ad[-1 - n] + 1:ad[-1] + 1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // (
            int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if wgs < 512 and d > 12:
            k = (r["Kernel_Name"][:50], wgs)
            agg[k][0] += 1
            agg[k][1] += d
    tot = 0
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t / n / 1e3:6.3f} ms/step {c // n:4d} x {t / c:6.1f} us  wgs={k[1]:5d}  {k[0]}")
        tot += t / n / 1e3
    print("total", round(tot, 2), "ms/step in kernels with < 512 workgroups and > 12 us")


if __name__ == "__main__":
    main(sys.argv[1])
