#!/usr/bin/env python
"""Register / scratch use of the kernels in a built object, from the code-object metadata (what the loader sees, not a compiler remark):
    python tools/kernel_resources.py simseg_amd/build/gemm.o [name-substring]
prints  vgprs  spilled-vgprs  scratch-bytes  lds-bytes  kernel  for every kernel of the gfx950 device image."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(obj):
    """-> [(demangled-ish name, vgpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size)]"""
    tmp = tempfile.mkdtemp(prefix="ss_co_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, capture_output=True, check=True)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not cos:
            raise RuntimeError(f"{obj}: no gfx950 device image found")
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, cos[0])], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        f = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", k).group(1))      # noqa: E731
        out.append((name, f("vgpr_count"), f("vgpr_spill_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(n for n, *_ in out), capture_output=True, text=True).stdout.splitlines()
    return [(d,) + r[1:] for d, r in zip(names, out)]


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'vgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>7}  kernel")
    for name, vg, sp, sc, lds in kernel_resources(sys.argv[1]):
        if sub in name:
            print(f"{vg:5d} {sp:5d} {sc:7d} {lds:7d}  {name[:150]}")
