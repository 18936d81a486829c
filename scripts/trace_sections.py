#!/usr/bin/env python3
"""Critical-path estimate from a SERIALISED trace (POSE_MARKERS=1: the branches of every parallel region run one after the other on
one stream, each preceded by a k_marker launch of 100+i / 200+i workgroups, the region closed by 99 / 199).  Every kernel then runs
alone at full speed, so  sum over regions of max(branch time) + time outside regions  is the step time a perfect overlap of the
branches could reach, and the longest branch of each region is the one worth shortening.

    python scripts/trace_sections.py <kernel_trace.csv>"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
    seg = rows[ad[-2] + 1:ad[-1] + 1]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    regions, cur, branch, outside = [], None, None, 0.0
    out_names = collections.defaultdict(float)
    for r in seg:
        if "k_marker" in r["Kernel_Name"]:
            wid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
            if wid in (99, 199):
                if cur is not None:
                    regions.append(cur)
                cur = branch = None
            else:
                kind, b = ("fwd", wid - 100) if wid < 199 else ("bwd", wid - 200)
                if cur is None:
                    cur = {"kind": kind, "br": collections.OrderedDict()}
                branch = cur["br"].setdefault(b, {"us": 0.0, "n": 0, "names": collections.defaultdict(float)})
            continue
        d = dur(r)
        if branch is None:
            outside += d
            out_names[r["Kernel_Name"].replace("void ", "")[:40]] += d
        else:
            branch["us"] += d
            branch["n"] += 1
            branch["names"][r["Kernel_Name"].replace("void ", "")[:40]] += d
    crit = sum(max(b["us"] for b in g["br"].values()) for g in regions)
    tot = sum(b["us"] for g in regions for b in g["br"].values())
    print(f"{len(regions)} regions; kernel time inside regions {tot / 1e3:.2f} ms, longest-branch sum {crit / 1e3:.2f} ms, outside regions {outside / 1e3:.2f} ms")
    print(f"=> perfect-overlap step estimate {(crit + outside) / 1e3:.2f} ms; serial {(tot + outside) / 1e3:.2f} ms")
    for k, g in enumerate(regions):
        print(f"{g['kind']} region {k:3d}: " + " | ".join(f"b{b}: {v['us']:7.1f} us ({v['n']:3d})" for b, v in g["br"].items()))
    win = collections.defaultdict(float)
    names = collections.defaultdict(float)
    for g in regions:
        b, v = max(g["br"].items(), key=lambda kv: kv[1]["us"])
        win[(g["kind"], b)] += v["us"]
        for n, d in v["names"].items():
            names[n] += d
    print("longest branch by (pass, branch index): " + ", ".join(f"{k}: {v / 1e3:.2f} ms" for k, v in sorted(win.items())))
    print("kernels on the longest branches (ms):")
    for n, d in sorted(names.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {d / 1e3:6.3f}  {n}")
    print("kernels outside regions (ms):")
    for n, d in sorted(out_names.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {d / 1e3:6.3f}  {n}")


if __name__ == "__main__":
    main(sys.argv[1])
