#!/usr/bin/env python
"""Per-kernel average of the PMC counters stored in a rocprofv3 rocpd database
(`rocprofv3 --pmc ... --kernel-trace -d DIR -o NAME`).  Prints one markdown
table: kernel | dispatches | avg duration us | avg of every counter (summed over
the counter's instances/dimensions per dispatch).
Usage: rocpd_pmc.py results.db [name-filter] [--longest F]
--longest F: per kernel name, only the dispatches whose duration is >= F x the 90th-percentile duration of that
name (the launches of a kernel differ in size when the stages run on prefixes of the edge list)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    longest = float(sys.argv[sys.argv.index("--longest") + 1]) if "--longest" in sys.argv else 0.0
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "--schema" in sys.argv:
        for n in names:
            if "pmc" in n.lower() or n in ("kernels",):
                print(n, [r[1] for r in cur.execute(f"pragma table_info('{n}')")])
        return
    if "pmc_events" in names:
        cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
        # expected columns: ... dispatch_id / event_id, counter name (pmc name / symbol), value
        name_col = next(c for c in ("counter_name", "pmc_name", "symbol", "name") if c in cols)
        disp_col = next(c for c in ("dispatch_id", "event_id", "kernel_dispatch_id") if c in cols)
        val_col = next(c for c in ("value", "counter_value") if c in cols)
        rows = cur.execute(f"select {disp_col}, {name_col}, sum({val_col}) from pmc_events group by 1, 2").fetchall()
    else:
        raise SystemExit("no pmc_events view; run with --schema")
    per_disp = defaultdict(dict)
    for d, n, v in rows:
        per_disp[d][n] = v
    kcols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    key = "dispatch_id" if disp_col == "dispatch_id" else "id"
    if key not in kcols:
        key = "dispatch_id"
    kern = {r[0]: (r[1], r[2]) for r in cur.execute(f"select {key}, name, end-start from kernels")}
    agg = defaultdict(lambda: [0, 0.0, defaultdict(float)])
    clean = lambda nm: re.sub(r"\(.*", "", nm).replace("void ", "")
    durs = defaultdict(list)
    for d in per_disp:
        if d in kern:
            durs[clean(kern[d][0])].append(kern[d][1])
    dmax = {k: sorted(v)[min(len(v) - 1, int(0.9 * len(v)))] for k, v in durs.items()}
    for d, ctr in per_disp.items():
        if d not in kern:
            continue
        nm, dur = kern[d]
        nm = clean(nm)
        if flt and flt not in nm:
            continue
        if dur < longest * dmax[nm]:
            continue
        a = agg[nm]
        a[0] += 1
        a[1] += dur
        for k, v in ctr.items():
            a[2][k] += v
    ctrs = sorted({k for a in agg.values() for k in a[2]})
    print("| kernel | n | avg us | " + " | ".join(ctrs) + " |")
    print("|---|---|---|" + "---|" * len(ctrs))
    for nm, (n, dur, c) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print(f"| `{nm[:60]}` | {n} | {dur / n / 1e3:.1f} | " + " | ".join(f"{c[k] / n:.4g}" for k in ctrs) + " |")


if __name__ == "__main__":
    main()
