#!/usr/bin/env python
"""Sustained shader clock under the update's dominant kernel: runs k_gemm_fwd at the layer-2 shape (32768 x 256 x 512,
random operands, then all-zero operands) back to back for a couple of seconds while a thread samples the shader clock
(rocm-smi / sysfs pp_dpm_sclk / amd-smi, whatever the box offers).  The f32 MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s)
assumes the 2.4 GHz boost clock; what the kernel can reach scales with the clock the power management sustains."""
import glob
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch  # noqa: E402
from rlx_amd.hip import Ctx  # noqa: E402


def read_sclk():
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    m = re.search(r"(\d+)\s*M[Hh]z", line)
                    if m:
                        return float(m.group(1))
        except OSError:
            pass
    for cmd in (["rocm-smi", "--showclocks"], ["/opt/rocm/bin/rocm-smi", "--showclocks"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*M[Hh]z", out)
            if m:
                return float(m.group(1))
        except Exception:
            pass
    return None


def run(label, A, B, C, aux, M, N, K, seconds=2.0):
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            v = read_sclk()
            if v:
                samples.append(v)
            time.sleep(0.05)
    th = threading.Thread(target=sampler)
    for _ in range(5):
        ctx.dbg_gemm(0, A, B, C, aux, M, N, K, 1)
    torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            ctx.dbg_gemm(0, A, B, C, aux, M, N, K, 1)
        torch.cuda.synchronize()
        n += 200
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    tf = 2.0 * M * N * K * n / dt / 1e12
    clk = (sum(samples) / len(samples)) if samples else float("nan")
    print(f"{label}: {1e6 * dt / n:.1f} us/launch back to back, {tf:.1f} TFLOP/s; sclk samples {len(samples)}: "
          f"mean {clk:.0f} MHz, min {min(samples) if samples else float('nan'):.0f}, max {max(samples) if samples else float('nan'):.0f}"
          + (f" -> clock-scaled f32 MFMA peak {157.3 * clk / 2400:.1f} TFLOP/s, kernel at {tf / (157.3 * clk / 2400):.2f} of it" if samples else ""))


dev = torch.device("cuda:0")
ctx = Ctx(0)
M, N, K = 32768, 256, 512
print("idle sclk:", read_sclk())
A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * 0.05
C, aux = torch.empty(M, N, device=dev), torch.zeros(N, device=dev)
run("random operands", A, B, C, aux, M, N, K)
run("zero operands  ", torch.zeros_like(A), torch.zeros_like(B), C, aux, M, N, K)
