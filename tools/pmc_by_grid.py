#!/usr/bin/env python
"""Like pmc_stats.py, but one line per (kernel, grid size): a micro-benchmark that launches one kernel on several shapes gives one row per
shape.  python tools/pmc_by_grid.py <rocpd .db> [kernel-name-substring]"""
import re
import sqlite3
import sys


def main(path, pat=""):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tables if x.startswith(p))      # noqa: E731
    disp, sym, pmc, info = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    dcols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    gx = "grid_size_x" if "grid_size_x" in dcols else next(x for x in dcols if "grid" in x)
    q = (f"select s.{name_col}, i.name, p.value, d.{gx}, d.end - d.start from {pmc} p join {disp} d on p.event_id = d.event_id "
         f"join {sym} s on d.kernel_id = s.id join {info} i on p.pmc_id = i.id")
    agg = {}
    for kname, cname, val, grid, dur in c.execute(q):
        kname = re.sub(r"\(.*", "", kname)
        if pat and pat not in kname:
            continue
        b = agg.setdefault((kname, grid, cname), [0, 0.0, 0.0])
        b[0] += 1; b[1] += val; b[2] += dur
    for (k, grid, cn), (n, v, dur) in sorted(agg.items()):
        print(f"{k[:70]:70s} grid {grid:>8}  {cn:<12} avg {v / n:14.1f}  n={n:4d}  avg_us {dur / n / 1e3:9.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
