#!/usr/bin/env python3
"""Per-kernel means of SQ counters from rocprofv3 --pmc passes -> JSON, with the ratios that say what bounds a kernel.

    python scripts/pmc_sq_summary.py <counter_collection.csv> [<counter_collection.csv> ...]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles (16 per 16x16x32 bf16 MFMA); SQ_INSTS_* count wave-instructions; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles."""
import collections
import csv
import json
import sys


def main(paths):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in paths:
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if not n.startswith("k_"):
                continue
            tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[n][r["Counter_Name"]].add((path, r["Dispatch_Id"]))
    out = {}
    for n in sorted(tot):
        c = {k: tot[n][k] / max(1, len(cnt[n][k])) for k in tot[n]}
        e = {"launches": max(len(v) for v in cnt[n].values()), "counters": {k: round(v) for k, v in sorted(c.items())}}
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for k, name in (("SQ_ACTIVE_INST_VALU", "valu_active_frac"), ("SQ_ACTIVE_INST_LDS", "lds_active_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"),
                            ("SQ_WAIT_ANY", "waitcnt_or_barrier_frac"), ("SQ_ACTIVE_INST_ANY", "any_active_frac")):
                if k in c:
                    e[name] = round(c[k] / wc, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_over_wave_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 3)
        if c.get("SQ_INSTS_VALU") and "SQ_INSTS_MFMA" in c:
            e["valu_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / max(1.0, c["SQ_INSTS_MFMA"]), 2)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        out[n] = e
    print(json.dumps({"note": __doc__.strip().split("\n\n")[1].replace("\n", " ") if "\n\n" in __doc__.strip() else "", "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
