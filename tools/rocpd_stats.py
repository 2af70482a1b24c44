#!/usr/bin/env python
"""Per-kernel summary (count / total / avg / min / max / share) from a rocprofv3
rocpd sqlite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
DIR/NAME_results.db on ROCm 7.2).  Usage: rocpd_stats.py results.db [top_n]
The launches of the message-stage kernel differ in size (every stage runs on a prefix of the level-ordered edge
list): a second table splits `edge_wave_kernel<H, 0, ...>` into its largest launches (duration >= 0.9 x the
90th percentile: the ones bench.py brackets with HIP events) and the rest."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    s, e = cur.execute("select min(start), max(end) from kernels").fetchone()
    print(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches; "
          f"first-to-last span {(e - s) / 1e6:.2f} ms")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr+agpr | LDS B | grid x wg |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        nm = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:64]
        print(f"| `{nm}` | {r[1]} | {r[2] / 1e6:.2f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | "
              f"{100 * r[2] / tot:.1f} | {r[6]}+{r[7]} | {r[8]} | {r[9]}x{r[10]} |")


    # size classes of the message-stage kernel
    ed = cur.execute("select name, end-start from kernels where name like '%edge_wave_kernel<%, 0, %'").fetchall()
    if ed:
        d = sorted(x[1] for x in ed)
        ref = d[min(len(d) - 1, int(0.9 * len(d)))]
        big = [x for x in d if x >= 0.9 * ref]
        rest = [x for x in d if x < 0.9 * ref]
        print()
        print("| message-stage launches (`edge_wave_kernel<H, 0, ...>`) | calls | avg us | min us | max us |")
        print("|---|---|---|---|---|")
        for label, v in (("largest (the timed ones: rows of the largest radius)", big), ("all others (smaller prefixes, block 0 split)", rest)):
            if v:
                print(f"| {label} | {len(v)} | {sum(v) / len(v) / 1e3:.1f} | {v[0] / 1e3:.1f} | {v[-1] / 1e3:.1f} |")


if __name__ == "__main__":
    main()
