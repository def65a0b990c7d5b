"""Checkpoint interchange between this package's native `.npz` files and the reference's Flax/Optax state trees.

The reference saves `{"policy": TrainState, "critic": TrainState}` with orbax's PyTreeCheckpointer, zips the directory together
with `config_algorithm.json` and renames the archive to `<name>.model` (rl_x/algorithms/ppo/flax/ppo.py:423-436, restore
:440-466; the fully jitted flavour: ppo/flax_full_jit/ppo.py:382-425; the runner hands the path over as `--runner.load_model`,
rl_x/runner/runner.py:334-336).  This package's plugins keep ONE flat fp32 vector per network (layout: include/rlx_hip.h) plus
the Adam moments in an `.npz`.

Two layers:
  * the TREE mapping -- flat vector <-> the nested dict orbax restores without a target (`params/Dense_i/{kernel,bias}`,
    `LayerNorm_0/{scale,bias}`, `policy_logstd`; `opt_state/1/inner_state/0/{count,mu,nu}` of
    optax.chain(clip_by_global_norm, inject_hyperparams(adam)), `step`) -- is pure numpy and tested without orbax against
    hand-built trees (tests/test_checkpoint_convert.py);
  * the orbax / zip I/O around it (`to_reference_model`, `from_reference_model`) needs `orbax.checkpoint` and runs on the
    reference side only (it is not installable in the build container; the archive handling itself is plain zipfile).

    python -m rlx_amd.checkpoint to-reference   best.model out_dir/best.model   [--obs-dim O --act-dim A]
    python -m rlx_amd.checkpoint from-reference ref/best.model  out.npz
"""
import io
import json
import os
import zipfile

import numpy as np


# ----------------------------------------------------------------------------------------------------------- one network
def _dense_names(n_hidden, ln_first):
    """(flax module name, kind) in the order of the flat layout: Dense_0, [LayerNorm_0], Dense_1, ..., Dense_n (head)."""
    out = []
    for li in range(n_hidden):
        out.append((f"Dense_{li}", "dense"))
        if ln_first and li == 0:
            out.append(("LayerNorm_0", "ln"))
    out.append((f"Dense_{n_hidden}", "dense"))
    return out


def flat_to_flax(flat, in_dim, hidden, out_dim, ln_first, has_logstd, logstd_name="policy_logstd"):
    """Flat parameter vector (include/rlx_hip.h layout) -> flax variable dict {"params": {...}} (kernels [in, out])."""
    flat = np.asarray(flat)
    params, off, d = {}, 0, int(in_dim)
    widths = [int(h) for h in hidden] + [int(out_dim)]
    wi = 0
    for name, kind in _dense_names(len(hidden), ln_first):
        if kind == "dense":
            h = widths[wi]
            params[name] = {"kernel": flat[off:off + d * h].reshape(d, h).copy(), "bias": flat[off + d * h:off + d * h + h].copy()}
            off += d * h + h
            d, wi = h, wi + 1
        else:
            params[name] = {"scale": flat[off:off + d].copy(), "bias": flat[off + d:off + 2 * d].copy()}
            off += 2 * d
    if has_logstd:
        params[logstd_name] = flat[off:off + out_dim].reshape(1, out_dim).copy()
        off += out_dim
    if off != flat.size:
        raise ValueError(f"flat vector has {flat.size} entries, the layout ({in_dim}, {list(hidden)}, {out_dim}) needs {off}")
    return {"params": params}


def flax_to_flat(variables, hidden, ln_first, has_logstd, logstd_name="policy_logstd", dtype=np.float32):
    """Inverse of flat_to_flax.  Returns (flat, in_dim, out_dim)."""
    params = variables["params"] if "params" in variables else variables
    parts, in_dim, out_dim = [], None, None
    for name, kind in _dense_names(len(hidden), ln_first):
        leaf = params[name]
        if kind == "dense":
            k = np.asarray(leaf["kernel"])
            in_dim = k.shape[0] if in_dim is None else in_dim
            out_dim = k.shape[1]
            parts += [k.reshape(-1), np.asarray(leaf["bias"]).reshape(-1)]
        else:
            parts += [np.asarray(leaf["scale"]).reshape(-1), np.asarray(leaf["bias"]).reshape(-1)]
    if has_logstd:
        parts.append(np.asarray(params[logstd_name]).reshape(-1))
    return np.concatenate(parts).astype(dtype), int(in_dim), int(out_dim)


def infer_in_dim(n_params, hidden, out_dim, ln_first, has_logstd):
    """Observation width of a flat vector whose other dimensions are known."""
    rest, d = 0, None
    for li, h in enumerate(hidden):
        if li > 0:
            rest += d * h
        rest += h + (2 * h if ln_first and li == 0 else 0)
        d = h
    rest += d * out_dim + out_dim + (out_dim if has_logstd else 0)
    in_dim, rem = divmod(n_params - rest, hidden[0])
    if rem or in_dim <= 0:
        raise ValueError("parameter count does not fit the architecture")
    return in_dim


# --------------------------------------------------------------------------------------------------------- train states
def train_state_tree(flat, m, v, opt_count, learning_rate, **arch):
    """{"step", "params", "opt_state"} of flax.training.train_state.TrainState with
    tx = optax.chain(clip_by_global_norm, inject_hyperparams(adam)) (ppo/flax/ppo.py:84-100), as orbax restores it without a
    target: tuples become {"0": ..., "1": ...}, named tuples dicts of their fields.  The reference pins optax >= 0.2.6, whose
    inject_hyperparams state is InjectStatefulHyperparamsState(count, hyperparams, hyperparams_states, inner_state); a constant
    learning rate has no stateful schedule, so `hyperparams_states` is the empty dict."""
    count = np.asarray(int(opt_count), dtype=np.int32)
    adam = {"count": count, "mu": flat_to_flax(m, **arch), "nu": flat_to_flax(v, **arch)}
    return {"step": count,
            "params": flat_to_flax(flat, **arch),
            "opt_state": {"0": {},                                     # clip_by_global_norm: EmptyState
                          "1": {"count": count, "hyperparams": {"learning_rate": np.asarray(learning_rate, dtype=np.float32)},
                                "hyperparams_states": {},
                                "inner_state": {"0": adam, "1": {}}}}}  # adam = chain(scale_by_adam, scale_by_learning_rate)


def _find_adam(node):
    """The dict holding Adam's `mu` / `nu` / `count`, wherever the optimizer chain nests it."""
    if isinstance(node, dict):
        if "mu" in node and "nu" in node:
            return node
        for child in node.values():
            hit = _find_adam(child)
            if hit is not None:
                return hit
    elif isinstance(node, (list, tuple)):
        for child in node:
            hit = _find_adam(child)
            if hit is not None:
                return hit
    return None


def train_state_flat(tree, hidden, ln_first, has_logstd):
    """-> (flat, m, v, opt_count, in_dim, out_dim) of one restored TrainState tree."""
    flat, in_dim, out_dim = flax_to_flat(tree["params"], hidden, ln_first, has_logstd)
    adam = _find_adam(tree.get("opt_state", {}))
    if adam is None:
        m, v, count = np.zeros_like(flat), np.zeros_like(flat), int(np.asarray(tree.get("step", 0)))
    else:
        m = flax_to_flat(adam["mu"], hidden, ln_first, has_logstd)[0]
        v = flax_to_flat(adam["nu"], hidden, ln_first, has_logstd)[0]
        count = int(np.asarray(adam.get("count", tree.get("step", 0))))
    return flat, m, v, count, in_dim, out_dim


# ------------------------------------------------------------------------------------------------------------------ PPO
def _ppo_arch(config_algorithm):
    arch = config_algorithm.get("network_architecture", "full_jit" if "nr_hidden_units" not in config_algorithm else "flax")
    if arch == "full_jit":
        return [512, 256, 128], True
    h = int(config_algorithm.get("nr_hidden_units", 256))
    return [h, h], False


def ppo_npz_to_tree(ckpt, obs_dim=None, act_dim=None):
    """Native ppo.hip checkpoint (mapping with pparams, pm, pv, cparams, cm, cv, opt_count, config_algorithm) ->
    ({"policy": TrainState tree, "critic": TrainState tree}, config_algorithm dict)."""
    cfg = json.loads(str(ckpt["config_algorithm"]))
    hidden, ln = _ppo_arch(cfg)
    cparams = np.asarray(ckpt["cparams"])
    c_in = int(ckpt["critic_obs_dim"]) if "critic_obs_dim" in ckpt else infer_in_dim(cparams.size, hidden, 1, ln, False)
    p_in = int(ckpt["policy_obs_dim"]) if "policy_obs_dim" in ckpt else int(obs_dim or c_in)
    pparams = np.asarray(ckpt["pparams"])
    if act_dim is None:
        act_dim = int(ckpt["act_dim"]) if "act_dim" in ckpt else None
    if act_dim is None:                                               # n = (trunk with p_in) + h_last * A + 2 A
        trunk = p_in * hidden[0] + hidden[0] + (2 * hidden[0] if ln else 0) + sum(a * b + b for a, b in zip(hidden[:-1], hidden[1:]))
        act_dim, rem = divmod(pparams.size - trunk, hidden[-1] + 2)
        if rem:
            raise ValueError("cannot infer the action dimension: pass act_dim")
    count, lr = int(ckpt["opt_count"]), float(cfg.get("learning_rate", 0.0))
    pol = dict(in_dim=p_in, hidden=hidden, out_dim=act_dim, ln_first=ln, has_logstd=True)
    cri = dict(in_dim=c_in, hidden=hidden, out_dim=1, ln_first=ln, has_logstd=False)
    tree = {"policy": train_state_tree(pparams, ckpt["pm"], ckpt["pv"], count, lr, **pol),
            "critic": train_state_tree(cparams, ckpt["cm"], ckpt["cv"], count, lr, **cri)}
    return tree, cfg


def ppo_tree_to_npz(tree, config_algorithm):
    """Restored reference checkpoint -> the arrays of a native ppo.hip `.npz`."""
    hidden, ln = _ppo_arch(config_algorithm)
    pp, pm, pv, pc, p_in, act_dim = train_state_flat(tree["policy"], hidden, ln, True)
    cp, cm, cv, cc, c_in, _ = train_state_flat(tree["critic"], hidden, ln, False)
    cfg = dict(config_algorithm)
    cfg.setdefault("network_architecture", "full_jit" if ln else "flax")
    return dict(pparams=pp, pm=pm, pv=pv, cparams=cp, cm=cm, cv=cv, opt_count=max(pc, cc), policy_obs_dim=p_in, critic_obs_dim=c_in,
                act_dim=act_dim, config_algorithm=json.dumps(cfg))


# ------------------------------------------------------------------------------------------------------------------ SAC
def sac_policy_flat_to_flax(flat, in_dim, hidden, act_dim):
    """sac/flax/policy.py:22-41: the mean and log_std heads are two Dense modules (Dense_n, Dense_n+1); this package keeps them
    as the two column blocks of one [H, 2A] head."""
    both = flat_to_flax(flat, in_dim, hidden, 2 * act_dim, False, False)["params"]
    n = len(hidden)
    head = both.pop(f"Dense_{n}")
    both[f"Dense_{n}"] = {"kernel": head["kernel"][:, :act_dim].copy(), "bias": head["bias"][:act_dim].copy()}
    both[f"Dense_{n + 1}"] = {"kernel": head["kernel"][:, act_dim:].copy(), "bias": head["bias"][act_dim:].copy()}
    return {"params": both}


def sac_policy_flax_to_flat(variables, hidden):
    params = dict(variables["params"])
    n = len(hidden)
    mean, ls = params.pop(f"Dense_{n}"), params.pop(f"Dense_{n + 1}")
    params[f"Dense_{n}"] = {"kernel": np.concatenate([np.asarray(mean["kernel"]), np.asarray(ls["kernel"])], axis=1),
                            "bias": np.concatenate([np.asarray(mean["bias"]), np.asarray(ls["bias"])])}
    return flax_to_flat({"params": params}, hidden, False, False)[0]


def sac_critic_flat_to_flax(flat2, in_dim, hidden):
    """sac/flax/critic.py:36-53: nn.vmap over two Critic modules -> every leaf carries a leading axis of size 2 under
    "VmapCritic_0"; this package keeps Q0's vector followed by Q1's."""
    n = flat2.size // 2
    q = [flat_to_flax(flat2[k * n:(k + 1) * n], in_dim, hidden, 1, False, False)["params"] for k in range(2)]
    return {"params": {"VmapCritic_0": {name: {leaf: np.stack([q[0][name][leaf], q[1][name][leaf]]) for leaf in q[0][name]}
                                        for name in q[0]}}}


def sac_critic_flax_to_flat(variables, hidden):
    stacked = variables["params"]["VmapCritic_0"]
    return np.concatenate([flax_to_flat({"params": {name: {leaf: np.asarray(arr)[k] for leaf, arr in mod.items()}
                                                    for name, mod in stacked.items()}}, hidden, False, False)[0] for k in range(2)])


# --------------------------------------------------------------------------------------------- archive + orbax (reference side)
def _orbax():
    try:
        import orbax.checkpoint as ocp
        return ocp
    except Exception as e:                                            # not installable in the build container
        raise RuntimeError("orbax.checkpoint is needed for the reference's on-disk format (install it on the reference side); "
                           "the tree mapping itself (ppo_npz_to_tree / ppo_tree_to_npz) needs numpy only") from e


def to_reference_model(npz_path, model_path, obs_dim=None, act_dim=None):
    """native `.npz` -> `<name>.model` as ppo/flax/ppo.py:423-436 writes it (orbax PyTree + config_algorithm.json, zipped)."""
    import shutil
    import tempfile
    ocp = _orbax()
    tree, cfg = ppo_npz_to_tree(np.load(npz_path, allow_pickle=False), obs_dim, act_dim)
    tmp = tempfile.mkdtemp()
    try:
        ocp.PyTreeCheckpointer().save(os.path.join(tmp, "ckpt"), tree)
        with open(os.path.join(tmp, "ckpt", "config_algorithm.json"), "w") as f:
            json.dump(cfg, f)
        shutil.make_archive(model_path, "zip", os.path.join(tmp, "ckpt"))
        os.replace(model_path + ".zip", model_path)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def from_reference_model(model_path, npz_path):
    import shutil
    import tempfile
    ocp = _orbax()
    tmp = tempfile.mkdtemp()
    try:
        with zipfile.ZipFile(model_path) as z:
            z.extractall(tmp)
        cfg = json.load(open(os.path.join(tmp, "config_algorithm.json")))
        tree = ocp.PyTreeCheckpointer().restore(tmp)
        save_npz(npz_path, ppo_tree_to_npz(tree, cfg))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def save_npz(path, arrays):
    buf = io.BytesIO()
    np.savez(buf, **arrays)
    with open(path, "wb") as f:
        f.write(buf.getvalue())


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("to-reference")
    a.add_argument("npz")
    a.add_argument("model")
    a.add_argument("--obs-dim", type=int)
    a.add_argument("--act-dim", type=int)
    b = sub.add_parser("from-reference")
    b.add_argument("model")
    b.add_argument("npz")
    args = ap.parse_args(argv)
    if args.cmd == "to-reference":
        to_reference_model(args.npz, args.model, args.obs_dim, args.act_dim)
    else:
        from_reference_model(args.model, args.npz)


if __name__ == "__main__":
    main()
