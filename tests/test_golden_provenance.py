"""CPU: the committed `tests/golden/reference_*.npz` ARE what the committed generator writes.

Runs only where the reference checkout exists (the authoring container; never on the GPU box): re-executes
tests/golden/make_reference_golden.py into a temporary directory and compares every array of every file -- dtype, shape and
bytes -- with the committed one.  The generator imports the reference's modules by file path and compiles its `train`
closures in that process; nothing of the reference is stored."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF = os.environ.get("RLX_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rl_x")), reason="the reference checkout is not present here")
def test_every_reference_fixture_regenerates_bit_for_bit(tmp_path):
    env = dict(os.environ, RLX_GOLDEN_OUT=str(tmp_path), RLX_REFERENCE=REF)
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_reference_golden.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    committed = sorted(glob.glob(os.path.join(GOLD, "reference_*.npz")))
    fresh = sorted(glob.glob(os.path.join(str(tmp_path), "reference_*.npz")))
    assert [os.path.basename(p) for p in committed] == [os.path.basename(p) for p in fresh] and len(committed) == 13
    for a, b in zip(committed, fresh):
        za, zb = np.load(a), np.load(b)
        assert sorted(za.files) == sorted(zb.files), os.path.basename(a)
        for k in za.files:
            x, y = za[k], zb[k]
            assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), (os.path.basename(a), k)
