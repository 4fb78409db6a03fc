#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME -- cmd` -> DIR/NAME_results.db).

    python tools/rocpd_kernels.py DB [--runs] [--match SUBSTR]

default: one line per kernel (calls, total, mean, min in us), sorted by total time;  --runs: one line per RUN of consecutive launches of the
same kernel with the same grid (micro-benchmarks that sweep a parameter launch such runs one after the other)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None
    rows = list(db.cursor().execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
    rows = [(short(n), s, e, (gx // max(wx, 1), gy, gz)) for n, s, e, gx, gy, gz, wx in rows if match is None or match in n]
    if "--runs" in sys.argv:
        i = 0
        while i < len(rows):
            j = i
            while j < len(rows) and rows[j][0] == rows[i][0] and rows[j][3] == rows[i][3]:
                j += 1
            d = [(e - s) / 1e3 for _, s, e, _ in rows[i:j]]
            print(f"{rows[i][0]:60s} grid {str(rows[i][3]):18s} x{j - i:4d}  mean {sum(d) / len(d):8.1f} us  min {min(d):8.1f}")
            i = j
        return
    agg = {}
    for n, s, e, g in rows:
        a = agg.setdefault(n, [0, 0.0, 1e30])
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2] = min(a[2], (e - s) / 1e3)
    tot = sum(a[1] for a in agg.values())
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:70s} calls {a[0]:6d}  total {a[1] / 1e3:9.3f} ms  mean {a[1] / a[0]:8.1f} us  min {a[2]:8.1f}  {100 * a[1] / tot:5.1f} %")


if __name__ == "__main__":
    main()
