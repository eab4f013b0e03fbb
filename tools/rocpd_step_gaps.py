#!/usr/bin/env python
"""Idle gaps of ONE training step (the last complete one between two AdamW launches) of a rocprofv3 --kernel-trace database, in
time order: offset from the step's first kernel, gap length, the kernels on either side.  usage: rocpd_step_gaps.py <db> [min_us=20]"""
import re
import sqlite3
import sys


def main(path, min_us=20.0):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    adam = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    # (round 4: the text tower's parameters are updated by a launch of their own on the text tower's stream - a step's optimizer launches
    #  lie within a fraction of a step of each other (the text tower finishes ~20 ms before the image tower at 512 pairs); the LAST of such a
    #  group delimits the step.  Group = launches closer than 0.4 of the median distance between every second launch.
    if len(adam) > 4:
        two = sorted(rows[adam[k + 2]][0] - rows[adam[k]][0] for k in range(len(adam) - 2))
        lim = 0.4 * two[len(two) // 2]
        adam = [i for k, i in enumerate(adam) if k + 1 == len(adam) or rows[adam[k + 1]][0] - rows[i][1] > lim]
    if len(adam) < 3:
        print("fewer than three optimizer launches in the trace"); return
    # the step to show: the SHORTEST interval between two consecutive optimizer launches (bench.py runs further legs after the timed
    # steps - padded captions, the instrumented single-stream step - whose intervals contain host-side synchronisations), or the one
    # whose index is given as the third argument
    spans = [(rows[adam[i + 1]][1] - rows[adam[i]][1], i) for i in range(len(adam) - 1)]
    pick = int(sys.argv[3]) if len(sys.argv) > 3 else min(spans)[1]
    print("intervals between optimizer launches (ms): " + " ".join(f"{s_ / 1e6:.1f}" for s_, _ in spans) + f"   -> showing #{pick}")
    lo, hi = adam[pick], adam[pick + 1]
    step = rows[lo:hi + 1]           # from the previous step's AdamW to this step's
    short = lambda n: re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", re.sub(r"^_ZN2at6native\d*", "at::", n or ""))[:70]
    t0 = step[0][1]
    print(f"step: {len(step) - 1} kernels, {(step[-1][1] - t0) / 1e6:.3f} ms from the end of the previous AdamW to the end of this one")
    cur_end, prev, idle = step[0][1], step[0][2], 0
    for s, e, n in step[1:]:
        if s > cur_end:
            g = s - cur_end
            idle += g
            if g >= min_us * 1e3:
                print(f"  +{(cur_end - t0) / 1e6:8.3f} ms  idle {g / 1e3:8.1f} us  after {short(prev)}  before {short(n)}")
        if e > cur_end:
            cur_end, prev = e, n
    print(f"idle inside the step: {idle / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0)
