from rlx_amd.hip.lib import (  # noqa: F401
    load_library, library_path, RlxError, Ctx, MlpDesc, PpoHparams, SacHparams, mlp_desc, LnMlpDesc, FastSacHparams, lnmlp_desc,
    ACT_TANH, ACT_ELU, ACT_RELU, THREEFRY_LEGACY, THREEFRY_PARTITIONABLE,
)
