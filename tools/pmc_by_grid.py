#!/usr/bin/env python
"""Like pmc_stats.py, but one line per (kernel, grid size): a micro-benchmark that launches one kernel on several shapes gives one row per
shape.  python tools/pmc_by_grid.py <rocpd .db> [kernel-name-substring]"""
import re
import sqlite3
import sys


def runs(path, pat, n):
    """Consecutive dispatches of the matching kernels in groups of n (tools/gemm_bench.py launches every shape 3 + iters times in a row)."""
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tables if x.startswith(p))      # noqa: E731
    disp, sym, pmc, info = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = (f"select s.{name_col}, i.name, p.value, d.start, d.end - d.start from {pmc} p join {disp} d on p.event_id = d.event_id "
         f"join {sym} s on d.kernel_id = s.id join {info} i on p.pmc_id = i.id order by d.start")
    rows = [(re.sub(r"\(.*", "", k), cn, v, dur) for k, cn, v, st, dur in c.execute(q) if pat in k]
    for g in range(0, len(rows), n):
        part = rows[g:g + n]
        print(f"run {g // n:3d}  {part[0][0][20:80]:60s} {part[0][1]:<11} avg {sum(r[2] for r in part) / len(part):12.1f} KB  x2 = "
              f"{2 * 1024 * sum(r[2] for r in part) / len(part) / 1e6:8.1f} MB  avg_us {sum(r[3] for r in part) / len(part) / 1e3:8.1f}")


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
    if len(sys.argv) > 3:
        runs(sys.argv[1], sys.argv[2], int(sys.argv[3]))
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
