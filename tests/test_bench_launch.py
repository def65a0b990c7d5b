"""CPU: `python bench.py --gpus N` (N > 1, no WORLD_SIZE) turns itself into the launcher the driver would otherwise be --
torch.distributed.run with one rank per GPU on 127.0.0.1 -- instead of exiting.  The ranks themselves
need a GPU (tests/test_gpu_multiprocess.py::test_bench_self_launches_two_ranks); here the command line is checked."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_torchrun_builds_the_launch_command():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RLX_BENCH_DRY_LAUNCH"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
