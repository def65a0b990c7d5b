"""CPU: the C-ABI shared library loads and exports every symbol include/rlx_hip.h declares;
host-only entry points compute; device entry points fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import nets, prng
from rlx_amd.hip import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "rlx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rlx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load_library()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"librlxhip.so does not export {s}"
    # and the ctypes binding covers the same set
    assert set(L.EXPORTED_SYMBOLS) == set(syms)


def test_version_and_param_count_without_gpu():
    lib = L.load_library()
    assert lib.rlx_version() >= 100
    for arch in "AB":
        for pol in (True, False):
            s = nets.make_spec(arch, 17, 6 if pol else 1, pol)
            d = L.mlp_desc(s.in_dim, s.hidden, s.out_dim, s.act, s.ln_first, s.has_logstd)
            assert lib.rlx_mlp_param_count(ctypes.byref(d)) == s.n_params
    # FastSAC's LayerNorm + SiLU networks: the library's layout and the oracle's agree (policy 2A-wide head, 101-atom critic)
    from oracle import fastsac as ofs
    for in_dim, hidden, out_dim in ((48, ofs.POLICY_HIDDEN, 24), (60, ofs.CRITIC_HIDDEN, 101)):
        d = L.lnmlp_desc(in_dim, hidden, out_dim)
        assert lib.rlx_lnmlp_param_count(ctypes.byref(d)) == ofs.param_count(in_dim, hidden, out_dim)
        assert sum(n for _, _, n in ofs.blocks(in_dim, hidden, out_dim)) == ofs.param_count(in_dim, hidden, out_dim)
    assert ctypes.sizeof(L.LnMlpDesc) == 28 and ctypes.sizeof(L.FastSacHparams) == 15 * 4 + 8      # include/rlx_hip.h


@pytest.mark.parametrize("scheme", [0, 1])
def test_host_split_matches_oracle(scheme):
    for seed in (0, 1, 42):
        assert np.array_equal(L.threefry_split(prng.prng_key(seed), 4, scheme), prng.split(prng.prng_key(seed), 4, bool(scheme)))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.RlxError):
        L.Ctx(0)
    with pytest.raises(L.RlxError):
        L._ptr(torch.zeros(4))          # host tensors are rejected by the binding


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rl-x_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
