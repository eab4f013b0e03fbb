#!/usr/bin/env python
"""GPU idle time between kernels from a rocprofv3 rocpd database (--kernel-trace): busy vs span of the dispatch timeline,
and the largest idle gaps with the kernels on either side.  usage: rocpd_gaps.py <db> [skip_first_fraction=0.5]"""
import re
import sqlite3
import sys


def main(path, skip=0.5):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    rows = rows[int(len(rows) * skip):]
    span = rows[-1][1] - rows[0][0]
    busy, cur_end, gaps = 0, rows[0][0], []
    prev = None
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n))
            busy += e - s
            cur_end = e
        else:
            busy += max(0, e - max(s, cur_end))
            cur_end = max(cur_end, e)
        prev = n
    short = lambda n: re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n or "")[:60]
    print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms ({100.0 * (span - busy) / span:.1f}%)")
    hist = {}
    for g, a, b in gaps:
        k = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
        h = hist.setdefault(k, [0, 0]); h[0] += 1; h[1] += g
    for k, (n, t) in hist.items():
        print(f"  gaps {k:>7}: {n:6d}  total {t / 1e6:8.3f} ms")
    for g, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
        print(f"  {g / 1e3:9.1f} us  after {short(a)}  before {short(b)}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
