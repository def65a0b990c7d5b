"""CPU: the library's device code holds no packed-f32 VALU instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).

Why this is a test: on MI355X a wave whose v_pk_*_f32 result feeds the next instruction occasionally gets the HIGH half of its
last 16 lanes wrong when the SIMD is shared with a wave of ANOTHER kernel that streams v_mfma_f32_32x32x16_bf16 (DESIGN.md
section 4, "co-residency hazard"; found with tools/debug/l1fwd_victim.py).  The policy and critic chains of the update overlap
exactly such kernels, so rl-x_amd/build.py compiles with -fno-slp-vectorize -fno-vectorize.  This test recompiles every source
to assembly with build.py's own flags (hipcc cross-compiles without a GPU) and fails if a packed-f32 instruction comes back,
whether through changed flags or through hand-written vector arithmetic."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import build as rlx_build  # noqa: E402

PACKED = re.compile(r"^\s*v_pk_(mul|fma|add)_f32\b", re.M)


def _asm(src, outdir):
    out = os.path.join(outdir, src[:-4] + ".s")
    cmd = [rlx_build.HIPCC] + rlx_build.CFLAGS + ["-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                                                  os.path.join(rlx_build.CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return src, len(PACKED.findall(open(out).read()))


@pytest.mark.skipif(not os.path.exists(rlx_build.HIPCC), reason="hipcc not installed")
def test_no_packed_f32_instructions_in_the_device_code():
    assert "-fno-slp-vectorize" in rlx_build.CFLAGS and "-fno-vectorize" in rlx_build.CFLAGS
    srcs = sorted(f for f in os.listdir(rlx_build.CSRC) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as td, ThreadPoolExecutor(max_workers=8) as ex:
        counts = dict(ex.map(lambda s: _asm(s, td), srcs))
    offenders = {k: v for k, v in counts.items() if v}
    assert not offenders, f"packed-f32 VALU instructions in the device code: {offenders}"
