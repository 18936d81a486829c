#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV of bench.py: per-stream busy time, union busy time, top kernels per step.

    python scripts/trace_summary.py <kernel_trace.csv> [n_last_steps]
Steps are delimited by the k_adamw launch that ends each step."""
import collections
import csv
import sys


def main(path, nsteps=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
    seg = rows[ad[-1 - nsteps] + 1:ad[-1] + 1]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / nsteps
    print(f"{(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6 / nsteps:.2f} ms/step in trace, "
          f"{len(seg) / nsteps:.0f} launches/step")
    bys, cnt = collections.defaultdict(float), collections.Counter()
    for r in seg:
        bys[r["Stream_Id"]] += dur(r)
        cnt[r["Stream_Id"]] += 1
    for k, v in sorted(bys.items(), key=lambda kv: -kv[1]):
        print(f"  stream {k}: {v:6.2f} ms of kernels, {cnt[k] // nsteps} launches")
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"  GPU busy (union over streams): {busy / 1e6 / nsteps:.2f} ms/step")
    tot, c2 = collections.defaultdict(float), collections.Counter()
    for r in seg:
        n = r["Kernel_Name"][:64]
        tot[n] += dur(r)
        c2[n] += 1
    for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:30]:
        print(f"{v:7.3f} ms {c2[n] // nsteps:5d} x {v / (c2[n] / nsteps) * 1e3:7.1f} us  {n}")
    print(f"sum of kernel time {sum(tot.values()):.2f} ms/step")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
