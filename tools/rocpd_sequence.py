#!/usr/bin/env python
"""Dispatch sequence of ONE forward call from a rocprofv3 rocpd database: the
launches between two consecutive `prep_assemble_kernel` dispatches (default: the last
complete call), one line per dispatch with its duration and the gap to the
previous one, plus a summary grouped by (kernel, grid).
Usage: rocpd_sequence.py results.db [call_index_from_end=2] [marker kernel, e.g. tn_pack_kernel for a training step]"""
import re
import sqlite3
import sys
from collections import OrderedDict


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("dsbdd::", "")[:44]


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = db.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
    # a forward call starts with prep_assemble_kernel (prep_kernel in older builds; a chain's pocket frame also runs
    # one prep_kernel, which is not a call boundary)
    marker = sys.argv[3] if len(sys.argv) > 3 else "prep_assemble_kernel"
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if not marks:
        marks = [i for i, r in enumerate(rows) if "prep_kernel" in r[0]]
    if len(marks) < back + 1:
        sys.exit("not enough forward calls in the trace")
    lo, hi = marks[-back - 1], marks[-back]
    seq = rows[lo:hi]
    span = seq[-1][2] - seq[0][1]
    busy = sum(r[2] - r[1] for r in seq)
    print(f"one forward call: {len(seq)} dispatches, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, "
          f"idle {100 * (1 - busy / span):.1f} %")
    print("| # | kernel | grid (wg) | us | gap us |")
    print("|---|---|---|---|---|")
    prev_end = seq[0][1]
    groups = OrderedDict()
    for i, (name, st, en, gx, gy, wx) in enumerate(seq):
        wg = f"{gx // max(wx, 1)}x{gy}"
        print(f"| {i} | `{short(name)}` | {wg} | {(en - st) / 1e3:.1f} | {(st - prev_end) / 1e3:.1f} |")
        prev_end = en
        g = groups.setdefault((short(name), wg), [0, 0])
        g[0] += 1
        g[1] += en - st
    print()
    print("| kernel | grid (wg) | calls | total us | avg us | % of span |")
    print("|---|---|---|---|---|---|")
    for (nm, wg), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{nm}` | {wg} | {n} | {t / 1e3:.1f} | {t / n / 1e3:.1f} | {100 * t / span:.1f} |")


if __name__ == "__main__":
    main()
