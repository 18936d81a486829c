#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV of bench.py: per-stream busy time, union busy time, top kernels per step.

    python scripts/trace_summary.py <kernel_trace.csv> [n_last_steps]
Steps are delimited by the k_adamw launch that ends each step."""
import collections
import csv
import sys


def main(path, nsteps=3, union_json=None):
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
    # Concurrency profile: sweep the timeline; for every instant the number of kernels in flight.  A kernel's "alone" time (it is the
    # only one running) is on the critical path with nothing hiding it; "shared" time is split evenly between the kernels in flight.
    ev = []
    for k, r in enumerate(seg):
        ev.append((int(r["Start_Timestamp"]), 1, k))
        ev.append((int(r["End_Timestamp"]), 0, k))
    ev.sort()
    live, last = set(), ev[0][0]
    alone, share, hist = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    alone_k = collections.defaultdict(float)
    for t, kind, k in ev:
        dt = (t - last) / 1e6 / nsteps
        if dt > 0:
            hist[min(len(live), 6)] += dt
            for j in live:
                n = seg[j]["Kernel_Name"][:64]
                share[n] += dt / len(live)
                if len(live) == 1:
                    alone[n] += dt
                    alone_k[j] += dt
        last = t
        (live.add if kind else live.discard)(k)
    print("time by number of kernels in flight (ms/step): " + ", ".join(f"{k}: {v:.2f}" for k, v in sorted(hist.items())))
    print("kernel time when nothing else runs (alone) / fair share of wall time, top 25:")
    for n, v in sorted(share.items(), key=lambda kv: -kv[1])[:25]:
        print(f"{alone[n]:7.3f} ms alone {v:7.3f} ms share  {n}")
    by_shape(seg, nsteps, alone_k)
    if union_json:
        write_union(seg, nsteps, union_json)
    per = len(seg) // nsteps
    timeline(seg[len(seg) - 2 * per:len(seg) - per] if nsteps >= 2 else seg)


def write_union(seg, nsteps, path):
    """Per kernel name: the wall time during which at least one instance runs (union of its intervals).  Instances of one kernel that
    share the chip on concurrent streams (the three head branches' convolutions) each look slow in the per-launch average; total work
    over the union is the throughput the kernel family actually delivers in the step (bench.py: `frac_in_step_union`)."""
    import json
    iv = collections.defaultdict(list)
    for r in seg:
        iv[r["Kernel_Name"].replace("void ", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = {}
    for n, lst in iv.items():
        lst.sort()
        tot, (cs, ce) = 0, lst[0]
        for a, b in lst[1:]:
            if a > ce:
                tot += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        tot += ce - cs
        out[n] = {"launches_per_step": len(lst) / nsteps, "union_ms_per_step": tot / 1e6 / nsteps,
                  "sum_ms_per_step": sum(b - a for a, b in lst) / 1e6 / nsteps}
    with open(path, "w") as f:
        json.dump({"steps": nsteps, "kernels": out}, f, indent=1, sort_keys=True)


def _wgs(r):
    n = 1
    for a in "XYZ":
        n *= max(1, int(r.get("Grid_Size_" + a, 1)) // max(1, int(r.get("Workgroup_Size_" + a, 1))))
    return n


def by_shape(seg, nsteps, alone_of):
    """Aggregate by (kernel, workgroups): the launch shapes inside one kernel family."""
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for k, r in enumerate(seg):
        key = r["Kernel_Name"].replace("void ", "")[:40] + f" [{_wgs(r)}]"
        tot[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / nsteps
        cnt[key] += 1
    print("by (kernel, workgroups), top 40 by alone time:  alone ms | total ms | launches x us")
    al = collections.defaultdict(float)
    for k, v in alone_of.items():
        r = seg[k]
        al[r["Kernel_Name"].replace("void ", "")[:40] + f" [{_wgs(r)}]"] += v
    for key, v in sorted(al.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{v:7.3f} | {tot[key]:7.3f} | {cnt[key] / nsteps:5.0f} x {tot[key] / (cnt[key] / nsteps) * 1e3:7.1f}  {key}")


def timeline(seg, bin_ms=0.5):
    """One step as a sequence of `bin_ms` bins: mean number of kernels in flight and the kernels (name, workgroups) with most time."""
    t0 = int(seg[0]["Start_Timestamp"])
    nb = int((int(seg[-1]["End_Timestamp"]) - t0) / 1e6 / bin_ms) + 1
    occ = [collections.defaultdict(float) for _ in range(nb)]
    for r in seg:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        name = r["Kernel_Name"].replace("void ", "")[:34] + f" [{_wgs(r)}]"
        b = int(s / bin_ms)
        while b < nb and b * bin_ms < e:
            occ[b][name] += min(e, (b + 1) * bin_ms) - max(s, b * bin_ms)
            b += 1
    print(f"timeline of the last step, {bin_ms} ms bins: mean kernels in flight | top kernels [workgroups] ms")
    for b, o in enumerate(occ):
        top = sorted(o.items(), key=lambda kv: -kv[1])[:3]
        print(f"{b * bin_ms:5.1f} ms  {sum(o.values()) / bin_ms:4.2f} | " + "; ".join(f"{n} {v:.2f}" for n, v in top))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, sys.argv[3] if len(sys.argv) > 3 else None)
