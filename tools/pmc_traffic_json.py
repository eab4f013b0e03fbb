#!/usr/bin/env python
"""profiles/<round>_pmc_fetch_size.txt + <round>_pmc_write_size.txt (tools/pmc_stats.py output of the two rocprofv3 --pmc passes of
bench.py) -> profiles/<round>_pmc_traffic.json keyed by the GEMM kinds bench.py reports."""
import json
import re
import sys


def parse(path, counter):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("_Z") or line.startswith("__"):
            cur = line.strip()
        elif counter in line:
            out[cur] = float(line.split("avg/dispatch")[1].split()[0])
    return out


def kind_of(name):
    name = name.replace("DF16_", "DF16b")          # the fp16 flavour's mangled 16-bit type (the headline arithmetic since round 5)
    m = re.search(r"gemm_kernelI(DF16b|f)(DF16b|f)Lb(\d)ELb(\d)E", name)
    suffix = ""
    if not m:
        m2 = re.search(r"gemm_(large|pp|pp2)_kernelI(DF16b|f)Lb(\d)ELb(\d)E", name)
        if not m2:
            return None
        tin, tout, ta, tb, suffix = "DF16b", m2.group(2), m2.group(3), m2.group(4), {"large": "_L", "pp": "_P", "pp2": "_Q"}[m2.group(1)]
    else:
        tin, tout, ta, tb = m.groups()
    k = ("bf16" if tin == "DF16b" else "f32") + "_" + ("t" if ta == "1" else "n") + ("n" if tb == "1" else "t")
    k += "_o16" if tout == "DF16b" else "_o32"
    return k + suffix


def main(prefix="profiles/r1"):
    f, w = parse(prefix + "_pmc_fetch_size.txt", "FETCH_SIZE"), parse(prefix + "_pmc_write_size.txt", "WRITE_SIZE")
    res = {}
    for name, fetch in f.items():
        k = kind_of(name)
        if k:
            res[k] = {"kernel": name, "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": w.get(name),
                      "hbm_bytes_per_launch": int((2 * fetch + (w.get(name) or 0)) * 1024)}
    import hashlib
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = open(os.path.join(repo, "simseg_amd", "csrc", "gemm.hip"), "rb").read()
    blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()      # = git rev-parse HEAD:simseg_amd/csrc/gemm.hip when committed
    try:
        commit = subprocess.run(["git", "-C", repo, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    json.dump({"commit": commit, "gemm_hip_blob": blob, "command": "SIMSEG_BENCH_FP16=0 SIMSEG_AMD_TWO_STREAMS=0 rocprofv3 --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-seg  (the fp16 headline arithmetic; tools/profile_round.sh)",
               "note": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a "
                       "wide coalesced read); WRITE_SIZE is uncalibrated there", "per_kind": res}, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    for k, v in sorted(res.items()):
        print(f"{k:18s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:])
