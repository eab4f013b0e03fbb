#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite database (--kernel-trace)."""
import re
import sqlite3
import sys


def main(path, top=40, skip=0.0):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id"
    agg = {}
    rows = sorted(c.execute(q), key=lambda r: r[1])
    rows = rows[int(len(rows) * skip):]          # skip > 0: only the tail of the timeline (e.g. the graph-replay phase of a run)
    for name, s, e in rows:
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name)
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'share':>7}")
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:90]:<90} {n:>7} {ns / 1e6:>10.3f} {ns / n / 1e3:>10.2f} {100.0 * ns / tot:>6.2f}%")
    print(f"{'TOTAL':<90} {sum(v[0] for v in agg.values()):>7} {tot / 1e6:>10.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
