"""Build librlxhip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache: the .so
lands in rl-x_amd/lib/ so it travels with the source snapshot to the GPU box.

    python rl-x_amd/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# RLX_BUILD_TAG=<tag> (experiments): objects in build_<tag>/, library lib/librlxhip_<tag>.so -- select it at run time with
# RLX_HIP_LIBRARY=<path>; combine with RLX_EXTRA_DEFINES for an in-process-free A/B of compile-time variants on one GPU box
_TAG = os.environ.get("RLX_BUILD_TAG", "")
OBJ_DIR = os.path.join(HERE, "build" + ("_" + _TAG if _TAG else ""))
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librlxhip" + ("_" + _TAG if _TAG else "") + ".so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "rlx_hip.h")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -fno-slp-vectorize -fno-vectorize: no compiler-made packed-f32 VALU instructions (v_pk_mul/fma/add_f32).  MEASURED on MI355X
# (tools/probes/l1fwd_victim.py): a wave whose v_pk_*_f32 results feed the next instruction gets the HIGH half of the last 16
# lanes wrong (stale) now and then when the SIMD is shared with a wave of another kernel that streams v_mfma_f32_32x32x16_bf16
# -- one 64-byte piece of one LayerNorm row off by ~1e-2, never when the kernels run alone.  The policy and critic chains of
# the update overlap exactly such kernels.  (Packed f32 is also an anti-lever next to MFMAs: MI355X_MICROARCH.md.)
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-ffp-contract=fast", "-fno-slp-vectorize", "-fno-vectorize"]
# Reproducer of the hazard ONLY (never for a build that trains): RLX_REPRO_PACKED_F32=1 python rl-x_amd/build.py --force, then
# on an MI355X `python tools/probes/l1fwd_victim.py 4` reports rows of k_l1fwd_mfma's output that change from run to run while a
# split-fp16 GEMM runs on a second stream (0 differing with the default flags).  tests/test_isa_device_code.py guards the default.
if os.environ.get("RLX_EXTRA_DEFINES"):          # experiments: e.g. RLX_EXTRA_DEFINES="-DRLX_WS_MIN_WAVES=4"
    CFLAGS += os.environ["RLX_EXTRA_DEFINES"].split()
# RLX_REPRO_PACKED_F32_FILES="sac.hip,mlp.hip" restricts the vectorized flags to those sources (bisection, docs/PACKED_F32_HAZARD.md)
_VEC_FILES = [f for f in os.environ.get("RLX_REPRO_PACKED_F32_FILES", "").split(",") if f]
_NOVEC = ("-fno-slp-vectorize", "-fno-vectorize")
if os.environ.get("RLX_REPRO_PACKED_F32") == "1" and not _VEC_FILES:
    CFLAGS = [f for f in CFLAGS if f not in _NOVEC]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = os.path.getmtime(HEADER)
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    return m


def _compile(src, force, verbose):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(spath)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj, False
    flags = [f for f in CFLAGS if f not in _NOVEC] if src in _VEC_FILES else CFLAGS
    cmd = [HIPCC] + flags + ["-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or force or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(p)
