"""Build-container check: `python oracle/make_golden.py` (which imports the reference from /root/reference) reproduces every
committed fixture under tests/golden/ bit-for-bit.  Skipped where the reference is absent (the GPU box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference/simseg"), reason="the reference checkout is only present in the build container")
def test_fixture_recipe_is_a_noop(tmp_path):
    env = dict(os.environ, SIMSEG_GOLDEN_OUT=str(tmp_path), PYTHONPATH="")
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "make_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    names = sorted(f for f in os.listdir(GOLD) if f.endswith((".npz", ".json")))
    assert sorted(os.listdir(tmp_path)) == names
    for f in names:
        if f.endswith(".json"):
            assert json.load(open(tmp_path / f)) == json.load(open(os.path.join(GOLD, f))), f
            continue
        new, old = np.load(tmp_path / f), np.load(os.path.join(GOLD, f))
        assert set(new.files) == set(old.files), f
        for k in old.files:
            assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape, (f, k)
            assert np.array_equal(new[k], old[k], equal_nan=True), (f, k)
