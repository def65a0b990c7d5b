"""Worker for the multi-process data-parallel SAC test (launched by torch.distributed.run): a short sac.hip training run on a
sharded synthetic env; EVERY rank dumps its parameters so the test can check that the replicas stayed bit-identical.
Not a test module (no test_ prefix)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))


def main():
    out, arch = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = min(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count() - 1)
    torch.cuda.set_device(local)
    backend = os.environ.get("RLX_DIST_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.sac.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    if arch == "ppo_lstm":
        return _recurrent_plugin(out, rank, world, dist)
    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config("sac.hip")
    config.environment = get_environment_config("synthetic.random_obs")
    config.environment.nr_envs = 64
    config.environment.obs_dim, config.environment.act_dim = 24, 4
    config.environment.horizon = 16
    config.algorithm.network_architecture = arch
    config.algorithm.batch_size = 256
    config.algorithm.buffer_size = 64 * 40
    config.algorithm.learning_starts = 64 * 4
    config.algorithm.total_timesteps = 64 * 14
    config.algorithm.logging_frequency = 64 * 5
    train_env, eval_env = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    model = get_algorithm_model_class("sac.hip")(config, train_env, eval_env, "/tmp/rlx_dist_worker_sac", None)
    model.train()
    torch.cuda.synchronize()
    np.savez(out + f".rank{rank}.npz", pparams=model.pparams.cpu().numpy(), qparams=model.qparams.cpu().numpy(),
             qtarget=model.qtarget.cpu().numpy(), log_alpha=model.log_alpha.cpu().numpy(), key=model.key, opt_count=model.opt_count,
             ring_rows=model.ring[0].shape[0], ring_cols=model.ring[0].shape[1], metrics=np.array([model.last_metrics.get(k, np.nan) for k in
                      ("loss/q_loss", "loss/policy_loss", "entropy/alpha")]), first_obs=model.ring[0][0, 0].cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _recurrent_plugin(out, rank, world, dist):
    """ppo_lstm.hip on a sharded env: two training iterations; every rank dumps its replicas."""
    import torch
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.ppo_lstm.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config("ppo_lstm.hip")
    config.environment = get_environment_config("synthetic.random_obs")
    config.environment.nr_envs = 32
    config.environment.horizon = 12
    config.algorithm.nr_steps = 8
    config.algorithm.minibatch_size = 64
    config.algorithm.nr_epochs = 2
    config.algorithm.total_timesteps = 32 * 8 * 2
    config.algorithm.evaluation_and_save_frequency = 32 * 8 * 2
    train_env, eval_env = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    model = get_algorithm_model_class("ppo_lstm.hip")(config, train_env, eval_env, "/tmp/rlx_dist_worker_lstm", None)
    model.train()
    torch.cuda.synchronize()
    np.savez(out + f".rank{rank}.npz", pparams=model.pparams.cpu().numpy(), cparams=model.cparams.cpu().numpy(), key=model.key,
             opt_count=model.opt_count, carry=model.carry_h.cpu().numpy(),
             metrics=np.array([model.last_metrics.get(k, np.nan) for k in ("loss/policy_gradient_loss", "loss/critic_loss")]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
