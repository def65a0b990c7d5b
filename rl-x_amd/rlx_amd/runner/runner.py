"""Runner -- the entry point kept from the reference (rl_x/runner/runner.py) so that
`experiments/experiment.py` (`Runner().run()`, experiment.py:4-6) works unchanged:

  * `--algorithm.name= / --environment.name= / --runner.mode=` are pre-parsed and removed
    from sys.argv by exact match (runner.py:206-229), so they need the `=` form;
  * plugins are discovered by importing `<pkg>.environments.<name>` / `<pkg>.algorithms.<name>`
    for each implementation package, silently skipping ModuleNotFoundError (runner.py:232-247);
  * the three type enums are checked for compatibility (runner.py:86-91) and, for torch
    algorithms on device-resident envs, `--algorithm.device` must equal `--environment.device`
    (runner.py:116-128);
  * modes train / test / show_config (runner.py:250-263); exceptions from `model.train()` are
    logged and cleanup still runs (runner.py:340-352).

Differences: flags are parsed by rlx_amd.runner.config_dict (absl / ml_collections are not
dependencies); wandb / TensorBoard sinks are used only if importable.
"""
import importlib
import logging
import logging.handlers
import os
import sys

from rlx_amd.algorithms.algorithm_manager import (get_algorithm_config, get_algorithm_general_properties,
                                                  get_algorithm_model_class)
from rlx_amd.algorithms.deep_learning_framework_type import DeepLearningFrameworkType
from rlx_amd.environments.environment_manager import (get_environment_config,
                                                      get_environment_create_train_and_eval_env,
                                                      get_environment_general_properties)
from rlx_amd.environments.simulation_type import SimulationType
from rlx_amd.runner.config_dict import ConfigDict, apply_flag_overrides
from rlx_amd.runner.default_config import get_config as get_runner_config
from rlx_amd.runner.runner_mode import RunnerMode

DEFAULT_ALGORITHM = "ppo.hip"
DEFAULT_ENVIRONMENT = "synthetic.random_obs"
DEFAULT_RUNNER_MODE = "train"

rlx_logger = logging.getLogger("rl_x")


class Runner:
    def __init__(self, implementation_package_names=["rlx_amd"]):
        algorithm_name, environment_name, self._mode = self.parse_arguments()

        self.import_environment(environment_name, implementation_package_names)
        try:
            environment_general_properties = get_environment_general_properties(environment_name)
        except KeyError:
            raise ValueError(f"Unknown environment: {environment_name}") from None

        # Compatibility check (runner.py:83-91)
        self.import_algorithm(algorithm_name, implementation_package_names)
        try:
            algorithm_general_properties = get_algorithm_general_properties(algorithm_name)
        except KeyError:
            raise ValueError(f"Unknown algorithm: {algorithm_name}") from None
        if environment_general_properties.action_space_type not in algorithm_general_properties.action_space_types:
            raise ValueError(f"Incompatible action space type. Environment: {environment_general_properties.action_space_type}, Algorithm: {algorithm_general_properties.action_space_types}")
        if environment_general_properties.observation_space_type not in algorithm_general_properties.observation_space_types:
            raise ValueError(f"Incompatible observation space type. Environment: {environment_general_properties.observation_space_type}, Algorithm: {algorithm_general_properties.observation_space_types}")
        if environment_general_properties.data_interface_type not in algorithm_general_properties.data_interface_types:
            raise ValueError(f"Incompatible data interface type. Environment: {environment_general_properties.data_interface_type}, Algorithm: {algorithm_general_properties.data_interface_types}")

        runner_default_config = get_runner_config(self._mode)
        algorithm_default_config = get_algorithm_config(algorithm_name)
        environment_default_config = get_environment_config(environment_name)

        algorithm_uses_torch = DeepLearningFrameworkType.TORCH == algorithm_general_properties.deep_learning_framework_type
        environment_uses_torch = environment_general_properties.simulation_type in (
            SimulationType.ISAAC_LAB, SimulationType.MANISKILL, SimulationType.WARP)
        if algorithm_uses_torch and environment_uses_torch:  # runner.py:116-128
            alg_device = [arg for arg in sys.argv if arg.startswith("--algorithm.device=")]
            alg_device = alg_device[0].split("=")[1] if alg_device else getattr(algorithm_default_config, "device", None)
            env_device = [arg for arg in sys.argv if arg.startswith("--environment.device=")]
            env_device = env_device[0].split("=")[1] if env_device else getattr(environment_default_config, "device", None)
            if alg_device and env_device and alg_device != env_device:
                raise ValueError("Incompatible device types between algorithm and environment")

        self._model_class = get_algorithm_model_class(algorithm_name)
        self._create_train_and_eval_env = get_environment_create_train_and_eval_env(environment_name)
        self._namespaces = {"runner": runner_default_config, "algorithm": algorithm_default_config,
                            "environment": environment_default_config}
        self._explicit_flags = set()

        # Logging (runner.py:183-203)
        logger = logging.getLogger("rl_x")
        logger.setLevel(logging.INFO)
        logger.propagate = False
        if not logger.handlers:
            console_handler = logging.StreamHandler(sys.stdout)
            console_handler.setLevel(logging.INFO)
            console_handler.setFormatter(logging.Formatter(
                "[%(asctime)s] [%(filename)s:%(lineno)d] %(levelname)s - %(message)s", "%m-%d %H:%M:%S"))
            logger.addHandler(logging.handlers.MemoryHandler(100, logging.ERROR, console_handler))

        def info(msg, flush=True, *args, **kwargs):
            if logger.isEnabledFor(logging.INFO):
                logger._log(logging.INFO, msg, args, stacklevel=2, **kwargs)
            if flush:
                logger.handlers[0].flush()
        logger.info = info

    def parse_arguments(self):
        algorithm_name = [arg for arg in sys.argv if arg.startswith("--algorithm.name=")]
        environment_name = [arg for arg in sys.argv if arg.startswith("--environment.name=")]
        runner_mode = [arg for arg in sys.argv if arg.startswith("--runner.mode=")]

        if algorithm_name:
            algorithm_name = algorithm_name[0].split("=")[1]
            del sys.argv[sys.argv.index("--algorithm.name=" + algorithm_name)]
        else:
            algorithm_name = DEFAULT_ALGORITHM

        if environment_name:
            environment_name = environment_name[0].split("=")[1]
            del sys.argv[sys.argv.index("--environment.name=" + environment_name)]
        else:
            environment_name = DEFAULT_ENVIRONMENT

        if runner_mode:
            runner_mode = runner_mode[0].split("=")[1]
            del sys.argv[sys.argv.index("--runner.mode=" + runner_mode)]
        else:
            runner_mode = DEFAULT_RUNNER_MODE

        return algorithm_name, environment_name, runner_mode

    def import_environment(self, environment_name, implementation_package_names):
        for implementation_library_name in implementation_package_names:
            try:
                importlib.import_module(f"{implementation_library_name}.environments.{environment_name}")
                break
            except ModuleNotFoundError:
                pass

    def import_algorithm(self, algorithm_name, implementation_package_names):
        for implementation_library_name in implementation_package_names:
            try:
                importlib.import_module(f"{implementation_library_name}.algorithms.{algorithm_name}")
                break
            except ModuleNotFoundError:
                pass

    def run(self):
        if self._mode == RunnerMode.SHOW_CONFIG:
            main_func = self._show_config
        elif self._mode == RunnerMode.TRAIN:
            main_func = self._train
        elif self._mode == RunnerMode.TEST:
            main_func = self._test
        else:
            raise ValueError("Invalid mode")
        try:
            self._explicit_flags = apply_flag_overrides(self._namespaces, sys.argv)
            return main_func(None)
        except KeyboardInterrupt:
            rlx_logger.warning("KeyboardInterrupt")

    def init_config(self):
        self._config = ConfigDict()
        self._config.runner = self._namespaces["runner"]
        self._config.algorithm = self._namespaces["algorithm"]
        self._config.environment = self._namespaces["environment"]

    def _show_config(self, _):
        self.init_config()
        rlx_logger.info("\n" + str(self._config))
        return self._config

    def _run_path(self):
        run_path = f"runs/{self._config.runner.project_name}/{self._config.runner.exp_name}/{self._config.runner.run_name}"
        return os.path.abspath(run_path)

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
        if self._config.runner.load_model:
            explicitly_set_algorithm_params = [p for p in self._explicit_flags if p.startswith("algorithm.")]
            model = self._model_class.load(self._config, train_env, eval_env, run_path, writer,
                                           explicitly_set_algorithm_params)
        else:
            model = self._model_class(self._config, train_env, eval_env, run_path, writer)
        return model, train_env, eval_env

    def _train(self, _):
        self.init_config()
        run_path = self._run_path()
        if self._config.runner.save_model or self._config.runner.track_tb or self._config.runner.track_wandb:
            os.makedirs(run_path, exist_ok=True)
        if self._config.runner.track_wandb:
            import wandb  # optional sink; raises ImportError if absent, like the reference would
            wandb.init(entity=self._config.runner.wandb_entity, project=self._config.runner.project_name,
                       group=self._config.runner.exp_name, name=self._config.runner.run_name,
                       notes=self._config.runner.notes, config=self._config.to_dict())
            wandb.define_metric("*", step_metric="global_step")
        writer = None
        if self._config.runner.track_tb:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(run_path)

        model, train_env, eval_env = self._build_model(run_path, writer)
        try:
            model.train()
        except Exception:
            rlx_logger.error("Uncaught exception", exc_info=True)
        finally:
            train_env.close()
            eval_env.close()
            if self._config.runner.track_tb:
                writer.close()
            if self._config.runner.track_wandb:
                wandb.finish()
        return model

    def _test(self, _):
        self.init_config()
        if self._config.runner.track_wandb:
            raise ValueError("Wandb is not supported in test mode")
        if self._config.runner.track_tb:
            raise ValueError("Tensorboard is not supported in test mode")
        if self._config.runner.save_model:
            raise ValueError("Saving model is not supported in test mode")
        model, train_env, eval_env = self._build_model(self._run_path(), None)
        try:
            model.test(self._config.runner.nr_test_episodes)
        except Exception:
            rlx_logger.error("Uncaught exception", exc_info=True)
        finally:
            train_env.close()
            eval_env.close()
        return model
