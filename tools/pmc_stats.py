#!/usr/bin/env python
"""Per-kernel average of PMC counters from a rocprofv3 rocpd database (--pmc run)."""
import re
import sqlite3
import sys


def main(path, pat=""):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tables if x.startswith(p))
    disp, sym, pmc, info = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = (f"select s.{name_col}, i.name, p.value, d.id, d.end - d.start from {pmc} p join {disp} d on p.event_id = d.event_id "
         f"join {sym} s on d.kernel_id = s.id join {info} i on p.pmc_id = i.id")
    agg = {}
    for kname, cname, val, did, dur in c.execute(q):
        kname = re.sub(r"\(.*", "", kname)
        if pat and pat not in kname:
            continue
        a = agg.setdefault(kname, {})
        b = a.setdefault(cname, [0, 0.0])
        b[0] += 1
        b[1] += val
    for k, cs in agg.items():
        print(k[:110])
        for cn, (n, v) in sorted(cs.items()):
            print(f"    {cn:<28} avg/dispatch {v / n:16.1f}   (n={n})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
