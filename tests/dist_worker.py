"""Worker for the multi-process data-parallel test (launched by torch.distributed.run, or directly for world 1):
runs a few ppo.hip training iterations on the SAME global problem and dumps parameters + metrics from rank 0.
Not a test module (no test_ prefix)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))


def main():
    out, nr_envs, nr_steps, mb, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
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
    import rlx_amd.algorithms.ppo.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config("ppo.hip")
    config.environment = get_environment_config("synthetic.random_obs")
    config.environment.nr_envs = nr_envs
    config.environment.horizon = 16
    config.algorithm.nr_steps = nr_steps
    config.algorithm.minibatch_size = mb
    config.algorithm.nr_epochs = 2
    config.algorithm.total_timesteps = nr_envs * nr_steps * iters
    config.algorithm.force_distributed_update = os.environ.get("RLX_FORCE_DIST", "0") == "1"
    train_env, eval_env = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    model = get_algorithm_model_class("ppo.hip")(config, train_env, eval_env, "/tmp/rlx_dist_worker", None)
    batch = model._alloc_batch()
    metrics = torch.zeros(model.nr_epochs * model.nr_minibatches, 10, device=model.device)
    state, _ = train_env.reset()
    for _ in range(iters):
        state = model.train_iteration(batch, state, metrics)
    torch.cuda.synchronize()
    if rank == 0:
        # which path carried the collectives: the library's own RCCL communicator, or the all-reduce hook (gloo tests)
        how = "none" if world == 1 else ("rccl" if model.ctx.rank_world() == (rank, world) and backend == "nccl" else "hook")
        np.savez(out, pparams=model.pparams.cpu().numpy(), cparams=model.cparams.cpu().numpy(), key=model.key,
                 opt_count=model.opt_count, metrics=metrics.cpu().numpy(), collectives=how)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
