"""CPU: rlx_amd/checkpoint.py -- the mapping between the native flat-vector checkpoints and the Flax / Optax state trees the
reference saves with orbax (rl_x/algorithms/ppo/flax/ppo.py:423-466, ppo/flax_full_jit/ppo.py:382-425), checked without orbax
against trees built by hand in the reference's module order (Dense_i kernels [in, out], LayerNorm_0 after the first Dense,
policy_logstd [1, A]; SAC: separate mean / log_std Dense heads, vmapped critics with a leading axis of 2)."""
import json

import numpy as np
import pytest

from oracle import nets, sac as osac
from rlx_amd import checkpoint as ck


def _hand_tree(rng, in_dim, hidden, out_dim, ln_first, has_logstd):
    """A flax variable dict built module by module + the flat vector the oracle's layout (= include/rlx_hip.h) gives it."""
    spec = nets.MLPSpec(in_dim, hidden, out_dim, nets.ACT_ELU, ln_first, has_logstd)
    flat = rng.standard_normal(spec.n_params).astype(np.float32)
    params = {}
    for li, L in enumerate(spec.layers):
        params[f"Dense_{li}"] = {"kernel": flat[L["W"]:L["W"] + L["in"] * L["out"]].reshape(L["in"], L["out"]),
                                 "bias": flat[L["b"]:L["b"] + L["out"]]}
        if "g" in L:
            params["LayerNorm_0"] = {"scale": flat[L["g"]:L["g"] + L["out"]], "bias": flat[L["be"]:L["be"] + L["out"]]}
    H = spec.head
    params[f"Dense_{len(hidden)}"] = {"kernel": flat[H["W"]:H["W"] + H["in"] * H["out"]].reshape(H["in"], H["out"]),
                                      "bias": flat[H["b"]:H["b"] + H["out"]]}
    if has_logstd:
        params["policy_logstd"] = flat[spec.logstd:spec.logstd + out_dim].reshape(1, out_dim)
    return spec, flat, {"params": params}


@pytest.mark.parametrize("in_dim,hidden,out_dim,ln,logstd", [(17, [512, 256, 128], 6, True, True), (17, [512, 256, 128], 1, True, False),
                                                             (11, [64, 64], 3, False, True), (376, [256, 256], 1, False, False)])
def test_flat_vector_and_flax_tree_are_the_same_numbers(in_dim, hidden, out_dim, ln, logstd):
    rng = np.random.default_rng(in_dim + out_dim)
    spec, flat, tree = _hand_tree(rng, in_dim, hidden, out_dim, ln, logstd)
    got = ck.flat_to_flax(flat, in_dim, hidden, out_dim, ln, logstd)
    assert sorted(got["params"]) == sorted(tree["params"])
    for mod, leaves in tree["params"].items():
        if isinstance(leaves, dict):
            for leaf, arr in leaves.items():
                assert np.array_equal(got["params"][mod][leaf], arr), (mod, leaf)
        else:
            assert np.array_equal(got["params"][mod], leaves) and got["params"][mod].shape == (1, out_dim)
    back, i, o = ck.flax_to_flat(tree, hidden, ln, logstd)
    assert np.array_equal(back, flat) and (i, o) == (in_dim, out_dim)
    assert ck.infer_in_dim(flat.size, hidden, out_dim, ln, logstd) == in_dim
    # the tree evaluates like the flat vector: Dense -> (LayerNorm) -> act chain on the kernels taken from the tree
    x = rng.standard_normal((5, in_dim))
    exp, _ = nets.forward(spec, flat.astype(np.float64), x)
    h = x
    for li in range(len(hidden)):
        d = tree["params"][f"Dense_{li}"]
        h = h @ d["kernel"].astype(np.float64) + d["bias"]
        if ln and li == 0:
            mu, var = h.mean(1, keepdims=True), h.var(1, keepdims=True)
            h = (h - mu) / np.sqrt(var + 1e-6) * tree["params"]["LayerNorm_0"]["scale"] + tree["params"]["LayerNorm_0"]["bias"]
        h = np.where(h > 0, h, np.expm1(np.minimum(h, 0)))
    d = tree["params"][f"Dense_{len(hidden)}"]
    np.testing.assert_allclose(h @ d["kernel"].astype(np.float64) + d["bias"], exp, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("arch", ["full_jit", "flax"])
def test_ppo_checkpoint_round_trip_through_the_reference_tree(arch):
    rng = np.random.default_rng(3)
    O, A = 17, 6
    hidden, ln = ([512, 256, 128], True) if arch == "full_jit" else ([64, 64], False)
    n_p = nets.MLPSpec(O, hidden, A, 0, ln, True).n_params
    n_c = nets.MLPSpec(O, hidden, 1, 0, ln, False).n_params
    cfg = {"learning_rate": 4e-4, "network_architecture": arch, "nr_hidden_units": 64, "nr_steps": 128}
    npz = {k: rng.standard_normal(n_p if k[0] == "p" else n_c).astype(np.float32) for k in ("pparams", "pm", "pv", "cparams", "cm", "cv")}
    npz.update(opt_count=480, config_algorithm=json.dumps(cfg))
    tree, cfg_out = ck.ppo_npz_to_tree(npz)                          # dimensions inferred from the vector sizes
    assert cfg_out == cfg and set(tree) == {"policy", "critic"}
    pol = tree["policy"]
    assert int(pol["step"]) == 480 and pol["params"]["params"]["policy_logstd"].shape == (1, A)
    assert pol["params"]["params"]["Dense_0"]["kernel"].shape == (O, hidden[0])
    adam = pol["opt_state"]["1"]["inner_state"]["0"]                  # chain(clip, inject_hyperparams(adam)): ScaleByAdamState
    assert int(adam["count"]) == 480 and adam["mu"]["params"]["Dense_1"]["kernel"].shape == (hidden[0], hidden[1])
    assert float(pol["opt_state"]["1"]["hyperparams"]["learning_rate"]) == pytest.approx(4e-4)
    # optax >= 0.2.6 (the reference's pin): InjectStatefulHyperparamsState has exactly these four fields
    assert set(pol["opt_state"]["1"]) == {"count", "hyperparams", "hyperparams_states", "inner_state"}
    assert pol["opt_state"]["1"]["hyperparams_states"] == {}
    back = ck.ppo_tree_to_npz(tree, cfg)
    for k in ("pparams", "pm", "pv", "cparams", "cm", "cv"):
        assert np.array_equal(back[k], npz[k]), k
    assert back["opt_count"] == 480 and (back["policy_obs_dim"], back["critic_obs_dim"], back["act_dim"]) == (O, O, A)
    # an optimizer chain nested differently is still found (the Adam state is located by its mu / nu fields)
    odd = {"policy": dict(pol, opt_state=[{}, {"inner_state": [adam]}]), "critic": tree["critic"]}
    assert np.array_equal(ck.ppo_tree_to_npz(odd, cfg)["pm"], npz["pm"])


def test_sac_heads_and_vmapped_critics():
    rng = np.random.default_rng(5)
    O, A, H = 11, 3, 32
    ps, qs = osac.make_specs(O, A, H)
    pp = rng.standard_normal(ps.n_params).astype(np.float32)
    qq = rng.standard_normal(2 * qs.n_params).astype(np.float32)
    pol = ck.sac_policy_flat_to_flax(pp, O, [H, H], A)["params"]
    head = pp[ps.head["W"]:ps.head["W"] + H * 2 * A].reshape(H, 2 * A)
    assert np.array_equal(pol["Dense_2"]["kernel"], head[:, :A]) and np.array_equal(pol["Dense_3"]["kernel"], head[:, A:])   # mean | log_std
    assert np.array_equal(ck.sac_policy_flax_to_flat({"params": pol}, [H, H]), pp)
    cri = ck.sac_critic_flat_to_flax(qq, O + A, [H, H])["params"]["VmapCritic_0"]
    assert cri["Dense_0"]["kernel"].shape == (2, O + A, H) and cri["Dense_2"]["bias"].shape == (2, 1)
    L0 = qs.layers[0]
    assert np.array_equal(cri["Dense_0"]["kernel"][1], qq[qs.n_params + L0["W"]:qs.n_params + L0["W"] + (O + A) * H].reshape(O + A, H))
    assert np.array_equal(ck.sac_critic_flax_to_flat({"params": {"VmapCritic_0": cri}}, [H, H]), qq)


def test_orbax_io_fails_loudly_without_orbax(tmp_path):
    try:
        import orbax.checkpoint  # noqa: F401
        pytest.skip("orbax is installed here")
    except Exception:
        pass
    with pytest.raises(RuntimeError, match="orbax"):
        ck.from_reference_model(str(tmp_path / "x.model"), str(tmp_path / "x.npz"))
