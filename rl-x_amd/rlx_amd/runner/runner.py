"""Runner -- the entry point of the reference kept as an interface (rl_x/runner/runner.py), so that
`experiments/experiment.py` (`Runner().run()`, experiment.py:4-6) works unchanged.  Behaviour mirrored:

  * `--algorithm.name= / --environment.name= / --runner.mode=` select the plugins; they are consumed from sys.argv
    before the typed flag parser sees the rest, and therefore need the `=` form (runner.py:206-229);
  * a plugin is found by importing `<pkg>.environments.<name>` / `<pkg>.algorithms.<name>` from the first
    implementation package that has it; packages without it are skipped (runner.py:232-247);
  * observation-space, action-space and data-interface types must be compatible (runner.py:86-91); a torch algorithm
    on a torch-resident simulator must share its device (runner.py:116-128);
  * modes train / test / show_config (runner.py:250-263); an exception inside `model.train()` / `model.test()` is logged
    and the environments are still closed (runner.py:340-352).

Differences: flags are parsed by rlx_amd.runner.config_dict (absl / ml_collections are not dependencies); wandb /
TensorBoard sinks are used only if importable; under torch.distributed.run the runner joins the process group first
(one rank per GPU).
"""
import importlib
import logging
import logging.handlers
import os
import sys

from rlx_amd.algorithms import algorithm_manager
from rlx_amd.algorithms.deep_learning_framework_type import DeepLearningFrameworkType
from rlx_amd.environments import environment_manager
from rlx_amd.environments.simulation_type import SimulationType
from rlx_amd.runner.config_dict import ConfigDict, apply_flag_overrides
from rlx_amd.runner.default_config import get_config as get_runner_config
from rlx_amd.runner.runner_mode import RunnerMode

DEFAULT_ALGORITHM = "ppo.hip"
DEFAULT_ENVIRONMENT = "synthetic.random_obs"
DEFAULT_RUNNER_MODE = "train"

# selector flag -> default value; consumed from sys.argv in this order
_SELECTORS = (("--algorithm.name=", DEFAULT_ALGORITHM), ("--environment.name=", DEFAULT_ENVIRONMENT),
              ("--runner.mode=", DEFAULT_RUNNER_MODE))
# (attribute of the environment's properties, attribute of the algorithm's properties, wording of the error)
_COMPATIBILITY = (("action_space_type", "action_space_types", "action space"),
                  ("observation_space_type", "observation_space_types", "observation space"),
                  ("data_interface_type", "data_interface_types", "data interface"))
_TORCH_RESIDENT_SIMULATORS = (SimulationType.ISAAC_LAB, SimulationType.MANISKILL, SimulationType.WARP)

rlx_logger = logging.getLogger("rl_x")


def _pop_selector(argv, prefix, default):
    """Value of the first `prefix<value>` argument, removed from argv in place; `default` when absent."""
    for position, argument in enumerate(argv):
        if argument.startswith(prefix):
            del argv[position]
            return argument[len(prefix):]
    return default


def _flag_or_default(argv, flag, config):
    """What `--<ns>.device=` will resolve to once the flags are parsed (the device check runs before that)."""
    prefix = f"--{flag}="
    given = [a[len(prefix):] for a in argv if a.startswith(prefix)]
    return given[0] if given else config.get("device")


def _setup_logging():
    """The "rl_x" logger: buffered console output that is flushed after a complete metrics table (log lines carry
    `flush=False` while a table is being written), as the plugins' `logger.info(msg, flush=...)` calls expect."""
    logger = logging.getLogger("rl_x")
    logger.setLevel(logging.INFO)
    logger.propagate = False
    if not logger.handlers:
        console = logging.StreamHandler(sys.stdout)
        console.setLevel(logging.INFO)
        console.setFormatter(logging.Formatter("[%(asctime)s] [%(filename)s:%(lineno)d] %(levelname)s - %(message)s",
                                               "%m-%d %H:%M:%S"))
        logger.addHandler(logging.handlers.MemoryHandler(100, logging.ERROR, console))
    buffered = logger.handlers[0]

    def info(message, *args, flush=True, **kwargs):
        if logger.isEnabledFor(logging.INFO):
            logger._log(logging.INFO, message, args, stacklevel=2, **kwargs)
        if flush:
            buffered.flush()
    logger.info = info


class Runner:
    def __init__(self, implementation_package_names=["rlx_amd"]):
        selected = self.parse_arguments()
        algorithm_name, environment_name = selected[:2]
        self._mode = selected[2]
        packages = list(implementation_package_names)
        self.import_environment(environment_name, packages)
        self.import_algorithm(algorithm_name, packages)
        env_props = self._registered(environment_manager.get_environment_general_properties, environment_name, "environment")
        alg_props = self._registered(algorithm_manager.get_algorithm_general_properties, algorithm_name, "algorithm")
        for env_attr, alg_attr, what in _COMPATIBILITY:
            offered, accepted = getattr(env_props, env_attr), getattr(alg_props, alg_attr)
            if offered not in accepted:      # membership on enum identity: plugins share the runner's enum classes
                raise ValueError(f"Incompatible {what} type. Environment: {offered}, Algorithm: {accepted}")

        self._namespaces = {"runner": get_runner_config(self._mode),
                            "algorithm": algorithm_manager.get_algorithm_config(algorithm_name),
                            "environment": environment_manager.get_environment_config(environment_name)}
        if (alg_props.deep_learning_framework_type == DeepLearningFrameworkType.TORCH
                and env_props.simulation_type in _TORCH_RESIDENT_SIMULATORS):
            devices = {ns: _flag_or_default(sys.argv, f"{ns}.device", self._namespaces[ns]) for ns in ("algorithm", "environment")}
            if None not in devices.values() and devices["algorithm"] != devices["environment"]:
                raise ValueError("Incompatible device types between algorithm and environment")

        self._model_class = algorithm_manager.get_algorithm_model_class(algorithm_name)
        self._create_train_and_eval_env = environment_manager.get_environment_create_train_and_eval_env(environment_name)
        self._explicit_flags = set()
        _setup_logging()

    # ------------------------------------------------------------------ plugin selection
    def parse_arguments(self):
        """-> (algorithm name, environment name, runner mode); the three selector flags leave sys.argv."""
        return tuple(_pop_selector(sys.argv, prefix, default) for prefix, default in _SELECTORS)

    @staticmethod
    def _import_plugin(kind, name, packages):
        for package in packages:
            try:
                importlib.import_module(f"{package}.{kind}.{name}")
                return package
            except ModuleNotFoundError:
                continue          # this package does not implement it: try the next one
        return None

    def import_environment(self, environment_name, implementation_package_names):
        return self._import_plugin("environments", environment_name, implementation_package_names)

    def import_algorithm(self, algorithm_name, implementation_package_names):
        return self._import_plugin("algorithms", algorithm_name, implementation_package_names)

    @staticmethod
    def _registered(lookup, name, kind):
        try:
            return lookup(name)
        except KeyError:
            raise ValueError(f"Unknown {kind}: {name}") from None

    # ------------------------------------------------------------------ modes
    def run(self):
        handlers = {RunnerMode.SHOW_CONFIG: self._show_config, RunnerMode.TRAIN: self._train, RunnerMode.TEST: self._test}
        if self._mode not in handlers:
            raise ValueError("Invalid mode")
        try:
            self._explicit_flags = apply_flag_overrides(self._namespaces, sys.argv)
            return handlers[self._mode](None)
        except KeyboardInterrupt:
            rlx_logger.warning("interrupted (KeyboardInterrupt)")

    def init_config(self):
        self._config = ConfigDict()
        for namespace in ("runner", "algorithm", "environment"):
            self._config[namespace] = self._namespaces[namespace]

    def _show_config(self, _):
        self.init_config()
        rlx_logger.info(f"\n{self._config}")
        return self._config

    def _run_path(self):
        r = self._config.runner
        return os.path.abspath(os.path.join("runs", r.project_name, r.exp_name, r.run_name))

    @staticmethod
    def _init_distributed():
        """One process per GPU when launched by torch.distributed.run (WORLD_SIZE > 1): bind the local device and
        join the process group (backend "nccl" = RCCL on ROCm).  The reference has no multi-GPU path; this is the
        num_envs data-parallel extension of the hot path."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1:
            return
        import torch
        import torch.distributed as dist
        local_rank = min(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count() - 1)
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("RLX_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)

    def _build_model(self, run_path, writer):
        self._init_distributed()
        train_env, eval_env = self._create_train_and_eval_env(self._config)
        args = (self._config, train_env, eval_env, run_path, writer)
        if self._config.runner.load_model:      # restore a checkpoint; flags given on the command line win over stored ones
            explicit = [flag for flag in self._explicit_flags if flag.startswith("algorithm.")]
            return self._model_class.load(*args, explicit), train_env, eval_env
        return self._model_class(*args), train_env, eval_env

    def _guarded(self, action, envs, cleanup=()):
        """Run `action`; log instead of propagating what it raises; always close the environments and the sinks."""
        try:
            action()
        except Exception:
            rlx_logger.error("Uncaught exception", exc_info=True)
        finally:
            for env in envs:
                env.close()
            for close in cleanup:
                close()

    def _train(self, _):
        self.init_config()
        r = self._config.runner
        run_path = self._run_path()
        if r.save_model or r.track_tb or r.track_wandb:
            os.makedirs(run_path, exist_ok=True)
        cleanup, writer = [], None
        if r.track_wandb:
            import wandb  # optional sink; ImportError if absent, like the reference
            wandb.init(entity=r.wandb_entity, project=r.project_name, group=r.exp_name, name=r.run_name, notes=r.notes,
                       config=self._config.to_dict())
            wandb.define_metric("*", step_metric="global_step")
            cleanup.append(wandb.finish)
        if r.track_tb:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(run_path)
            cleanup.insert(0, writer.close)
        model, train_env, eval_env = self._build_model(run_path, writer)
        self._guarded(model.train, (train_env, eval_env), cleanup)
        return model

    def _test(self, _):
        self.init_config()
        r = self._config.runner
        for enabled, what in ((r.track_wandb, "Wandb"), (r.track_tb, "Tensorboard"), (r.save_model, "Saving model")):
            if enabled:
                raise ValueError(f"{what} is not supported in test mode")
        model, train_env, eval_env = self._build_model(self._run_path(), None)
        self._guarded(lambda: model.test(r.nr_test_episodes), (train_env, eval_env))
        return model
