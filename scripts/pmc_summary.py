#!/usr/bin/env python3
"""Per-kernel mean of FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc passes -> JSON (bytes per launch).

gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KB; FETCH_SIZE reports half of the bytes of
wide coalesced reads -> doubled; WRITE_SIZE is exact.  The counters sit on the fabric side of L2: Infinity-Cache hits are included,
so the sum is L2-miss traffic, an upper bound on HBM bytes."""
import collections
import csv
import json
import sys


def means(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("void ", "").split("(")[0]
        tot[n] += float(r["Counter_Value"])
        cnt[n] += 1
    return {n: (tot[n] / cnt[n], cnt[n]) for n in tot}


def main(fetch_csv, write_csv):
    f, w = means(fetch_csv, "FETCH_SIZE"), means(write_csv, "WRITE_SIZE")
    out = {}
    for n in sorted(set(f) | set(w)):
        if not n.startswith("k_"):
            continue
        fb = 2.0 * 1024.0 * f.get(n, (0.0, 0))[0]
        wb = 1024.0 * w.get(n, (0.0, 0))[0]
        out[n] = {"launches": f.get(n, (0, 0))[1], "fetch_bytes": round(fb), "write_bytes": round(wb), "traffic_bytes": round(fb + wb)}
    print(json.dumps({"note": __doc__.strip().split("\n\n")[1].replace("\n", " "), "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
