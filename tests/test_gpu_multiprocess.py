"""GPU: the data-parallel path with REAL processes.  The GPU box has one device, so two ranks share it and talk over
gloo (RLX_DIST_BACKEND=gloo; production uses nccl = RCCL): the code path (env sharding by global env id, replicated
key, side-stream permutation prefetch, per-rank row restriction, batched statistics, one gradient all-reduce per net and
update, clip+Adam -- all inside rlx_ppo_update_dist_f32) is the one `bench.py --gpus N` runs, with the library's
all-reduce hook routed to gloo instead of its RCCL communicator."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, script_args, extra_env=None, timeout=600):
    env = dict(os.environ, RLX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    if nproc == 1:
        cmd = [sys.executable, *script_args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr=127.0.0.1", f"--master-port={_free_port()}", *script_args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:   # the ranks' tracebacks come before torchrun's summary
        tb = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or l.startswith("  File")]
        raise AssertionError("\n".join(tb[-40:]) + "\n" + r.stdout[-1500:])
    return r.stdout


def test_two_ranks_match_one_rank(tmp_path):
    """Same global problem (256 envs, 16 steps, minibatch 1024) on 1 rank (fused single-GPU update) and on 2 ranks:
    identical key / optimizer count, parameters equal up to fp32 reduction order."""
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    outs = []
    for nproc in (1, 2):
        out = str(tmp_path / f"w{nproc}.npz")
        _launch(nproc, [worker, out, "256", "16", "1024", "3"])
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["key"], b["key"]) and int(a["opt_count"]) == int(b["opt_count"]) == 3 * 2 * 4
    for k in ("pparams", "cparams"):
        d = np.abs(a[k] - b[k])
        # 24 Adam steps of lr 4e-4: a parameter whose gradient is rounding noise may step differently
        assert (d <= 2e-5 + 1e-3 * np.abs(a[k])).mean() > 0.995, (k, d.max(), (d > 2e-5).mean())
        assert d.max() <= 2 * 4e-4 * 24
    np.testing.assert_allclose(a["metrics"][:, :5], b["metrics"][:, :5], rtol=2e-3, atol=2e-4)


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two HIP devices: one rank per GPU over RCCL (the 1-GPU box runs the gloo twin above)")
def test_two_ranks_over_rccl_match_one_rank(tmp_path):
    """The PRODUCTION collective path: two ranks on two GPUs, backend nccl = RCCL -- the communicator owned by the library's
    context (rlx_ctx_create_dist, ncclCommInitRank from the broadcast unique id), every all-reduce of the update issued by the
    library on its communication stream (advantage statistics, one per network and update, metrics), no Python hook.
    Same global problem as the 1-rank fused update: identical key / optimizer count, parameters equal up to fp32 reduction
    order.  Skipped on a 1-GPU box; the driver's multi-GPU box exercises it."""
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    outs = []
    for nproc in (1, 2):
        out = str(tmp_path / f"r{nproc}.npz")
        _launch(nproc, [worker, out, "256", "16", "1024", "3"], extra_env={"RLX_DIST_BACKEND": "nccl"})
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["key"], b["key"]) and int(a["opt_count"]) == int(b["opt_count"]) == 3 * 2 * 4
    for k in ("pparams", "cparams"):
        d = np.abs(a[k] - b[k])
        assert (d <= 2e-5 + 1e-3 * np.abs(a[k])).mean() > 0.995, (k, d.max(), (d > 2e-5).mean())
        assert d.max() <= 2 * 4e-4 * 24
    np.testing.assert_allclose(a["metrics"][:, :5], b["metrics"][:, :5], rtol=2e-3, atol=2e-4)
    assert str(b["collectives"]) == "rccl"


def test_bench_two_ranks_json():
    """`bench.py --gpus 2`: the headline run keeps the reference's minibatch of 32768 rows GLOBAL (SURVEY.md 8(d) row 3:
    2 x more, 2 x smaller updates), the per-GPU-minibatch variant rides along as a labelled secondary object."""
    out = _launch(2, ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1"])
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["finite"] is True
    assert j["config"]["nr_envs_global"] == 8192 and j["config"]["minibatch_size_global"] == 32768
    assert j["config"]["updates_per_step"] == 320
    assert j["value"] > 0 and "cpu_baseline" not in j and "secondary_configs" not in j
    v = j["per_gpu_minibatch_variant"]
    assert v["minibatch_size_global"] == 65536 and v["updates_per_step"] == 160 and v["value"] > 0 and v["finite"] is True


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the way the driver starts N = 1): bench.py becomes the launcher
    itself, one rank per GPU (here: two ranks sharing the one device over gloo), rank 0 prints the single JSON line."""
    out = _launch(1, ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-secondary"])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["finite"] is True and j["value"] > 0
    assert j["config"]["nr_envs_global"] == 8192 and j["config"]["updates_per_step"] == 320
    assert j["multi_gpu"]["world_size"] == 2 and j["multi_gpu"]["backend"] == "gloo" and j["multi_gpu"]["rccl_comm_ranks"] == 0
    assert "NOT MEASURED" in j["multi_gpu"]["scaling_curve"] and "roofline" in j


def test_runner_two_ranks(tmp_path):
    """The reference-style entry point under torchrun: the Runner joins the process group itself."""
    script = tmp_path / "experiment.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\nfrom rlx_amd.runner.runner import Runner\n"
                      "if __name__ == '__main__':\n    m = Runner().run()\n"
                      "    assert m.world == 2 and m.nr_envs_local == 64 and m.opt_count == 16, (m.world, m.opt_count)\n"
                      "    import numpy as np\n    assert all(np.isfinite(v) for v in m.last_metrics.values())\n"
                      % os.path.join(ROOT, "rl-x_amd"))
    _launch(2, [str(script), "--algorithm.name=ppo.hip", "--environment.name=synthetic.random_obs", "--runner.mode=train",
                "--environment.nr_envs=128", "--algorithm.nr_steps=8", "--algorithm.minibatch_size=256",
                "--algorithm.nr_epochs=2", "--algorithm.total_timesteps=2048", "--runner.track_console=false"])


@pytest.mark.parametrize("arch", ["flax", "full_jit"])
def test_sac_two_ranks_stay_replicated(tmp_path, arch):
    """sac.hip on 2 ranks (gloo through the library's hook on the 1-GPU box): the env columns and the replay ring are sharded
    (32 of 64 envs per rank, different observations), every update all-reduces [gradients | loss sums] once, and the replicas end
    with BIT-identical parameters, targets, entropy coefficient, key and optimizer count."""
    worker = os.path.join(ROOT, "tests", "dist_worker_sac.py")
    out = str(tmp_path / "sac")
    _launch(2, [worker, out, arch])
    a, b = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    assert int(a["opt_count"]) == int(b["opt_count"]) == 14 - 4 and np.array_equal(a["key"], b["key"])
    for k in ("pparams", "qparams", "qtarget", "log_alpha"):
        assert np.array_equal(a[k], b[k]), k
    assert np.all(np.isfinite(a["metrics"])) and np.array_equal(a["metrics"], b["metrics"])
    assert int(a["ring_cols"]) == 32 and int(a["ring_rows"]) == 40           # capacity rows of the GLOBAL job, this rank's columns
    assert not np.array_equal(a["first_obs"], b["first_obs"])               # the ranks simulate different envs


def _check_lstm(out):
    a, b = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    for k in ("P", "C", "met"):
        assert np.array_equal(a[k], b[k]), k                                  # the replicas stay bit-identical
    assert np.array_equal(a["key"], b["key"]) and np.array_equal(a["key"], a["key_ref"]) and int(a["cnt"]) == int(b["cnt"]) == 8
    # against the one-device run over the same minibatches: same per-row terms, gradients summed in another fp32 order
    for k, ref in (("P", "P_ref"), ("C", "C_ref")):
        d = np.abs(a[k] - a[ref])
        assert (d <= 2e-6 + 1e-4 * np.abs(a[ref])).mean() > 0.995 and d.max() <= 2 * 3e-4 * 8, (k, d.max(), (d > 2e-6).mean())
    cols = [0, 1, 2, 3, 5, 6, 7, 8, 9]     # pg loss, critic loss, entropy, KL, advantage mean / std, policy std, gradient norms (4: clip fraction, a count)
    np.testing.assert_allclose(a["met"][:, cols], a["met_ref"][:, cols], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(a["met"][:4, cols], a["met_ref"][:4, cols], rtol=1e-4, atol=2e-6)   # before the parameters drift apart
    return a


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_recurrent_ppo_two_ranks_match_the_one_device_minibatches(tmp_path, cell):
    """rlx_ppo_lstm_update_f32 on 2 ranks (gloo through the library's hook on the 1-GPU box): env columns sharded 16 + 16, each
    rank gives 4 of the 8 envs of every sequence minibatch, statistics / gradients / metrics all-reduced by the library.  Equal to
    the one-device update over the same minibatches (given explicitly as env index sets) up to fp32 summation order."""
    worker = os.path.join(ROOT, "tests", "dist_worker_lstm.py")
    out = str(tmp_path / "lstm")
    _launch(2, [worker, out, cell])
    a = _check_lstm(out)
    assert int(a["n_collectives"]) == 1 + 2 * 8 + 1       # advantage sums, one per network and minibatch, metrics


@pytest.mark.skipif(_device_count() < 2, reason="needs two HIP devices: one rank per GPU over RCCL")
def test_recurrent_ppo_and_sac_two_ranks_over_rccl(tmp_path):
    """The same two checks with the library's own RCCL communicator (two GPUs; the driver's multi-GPU box runs it)."""
    out = str(tmp_path / "lstm_rccl")
    _launch(2, [os.path.join(ROOT, "tests", "dist_worker_lstm.py"), out, "lstm"], extra_env={"RLX_DIST_BACKEND": "nccl"})
    _check_lstm(out)
    out = str(tmp_path / "sac_rccl")
    _launch(2, [os.path.join(ROOT, "tests", "dist_worker_sac.py"), out, "flax"], extra_env={"RLX_DIST_BACKEND": "nccl"})
    a, b = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    for k in ("pparams", "qparams", "qtarget", "log_alpha", "key"):
        assert np.array_equal(a[k], b[k]), k


def test_recurrent_plugin_two_ranks_stay_replicated(tmp_path):
    """ppo_lstm.hip end to end on 2 ranks: sharded env (16 + 16), rank-offset acting noise, the data-parallel update -- replicas
    bit-identical, different env columns (carries differ)."""
    out = str(tmp_path / "lstm_plugin")
    _launch(2, [os.path.join(ROOT, "tests", "dist_worker_sac.py"), out, "ppo_lstm"])
    a, b = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    for k in ("pparams", "cparams", "key", "metrics"):
        assert np.array_equal(a[k], b[k]), k
    assert int(a["opt_count"]) == int(b["opt_count"]) == 2 * 2 * 4 and np.all(np.isfinite(a["metrics"]))
    assert a["carry"].shape == (16, 64) and not np.array_equal(a["carry"], b["carry"])
