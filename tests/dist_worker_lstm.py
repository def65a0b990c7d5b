"""Worker for the multi-process data-parallel PPO+LSTM test (launched by torch.distributed.run).  Every rank builds the SAME
global rollout from a seed, keeps its env columns, and runs rlx_ppo_lstm_update_f32 on a context that takes part in
collectives.  Rank 0 also computes the one-device equivalent on a plain context: the same minibatches given explicitly as env
index sets (union over the ranks of each rank's share), one rlx_ppo_lstm_minibatch_fwd_bwd_f32 + two rlx_clip_adam_step_f32
per minibatch.  Not a test module (no test_ prefix)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
sys.path.insert(0, ROOT)


def main():
    out, cell = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from oracle import nets, ppo_lstm as ol, prng
    from rlx_amd.hip import Ctx, PpoHparams, mlp_desc
    from rlx_amd.hip import lib as hiplib
    from rlx_amd.hip.lib import lstm_policy_desc
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = min(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count() - 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("RLX_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    T, NG, O, A, E = 8, 32, 11, 3, 2
    NE_G = 8                                            # envs per global minibatch -> minibatch_size = T * NE_G
    NL, NE_L = NG // world, NE_G // world
    rng = np.random.default_rng(12)
    spec = ol.LstmPolicySpec(O, A, 128, 64, (512, 256, 128), False, cell, "concat")
    p = (ol.init_params(spec, rng, 1.0) + 0.03 * rng.standard_normal(spec.n_params)).astype(np.float32)
    cs = nets.make_spec("B", O, 1, False)
    cp = (nets.init_params(cs, rng, 1.0) + 0.03 * rng.standard_normal(cs.n_params)).astype(np.float32)
    ldesc = lstm_policy_desc(spec.O, spec.A, spec.E, spec.H, spec.torso, spec.share, 1 if cell == "gru" else 0, 0)
    cdesc = mlp_desc(cs.in_dim, cs.hidden, cs.out_dim, cs.act, cs.ln_first, cs.has_logstd)
    g = dict(states=rng.standard_normal((T, NG, O)), actions=rng.standard_normal((T, NG, A)), log_probs=rng.standard_normal((T, NG)) * 0.1 - 4,
             returns=rng.standard_normal((T, NG)), advantages=rng.standard_normal((T, NG)) * 2 + 0.5,
             dones=(rng.random((T, NG)) < 0.15).astype(np.float64), c0=rng.standard_normal((NG, 64)) * 0.3, h0=rng.standard_normal((NG, 64)) * 0.3)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    sl = slice(rank * NL, (rank + 1) * NL)
    loc = {k: tt(v[sl] if k in ("c0", "h0") else v[:, sl]) for k, v in g.items()}
    hp = PpoHparams(0.2, 0.01, 0.5, 0.5, 0.9, 0.999, 1e-5)
    key = prng.prng_key(21)
    n_upd = E * (NL // NE_L)
    lr = np.linspace(3e-4, 1e-4, n_upd).astype(np.float32)

    # ---- the data-parallel update
    if backend == "nccl":
        ids = [hiplib.nccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx = Ctx(local, rank, world, ids[0])
    else:
        ctx = Ctx(local)
        ctx.set_rank(rank, world)
        side = ctx.side_stream()

        class _Buf:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        calls = []

        def hook(ptr, n, dtype, on_side):
            buf = torch.as_tensor(_Buf(ptr, n, "<f8" if dtype else "<f4"), device=dev)
            calls.append((n, dtype, on_side))
            if on_side:
                with torch.cuda.stream(side):
                    dist.all_reduce(buf)
            else:
                dist.all_reduce(buf)
        ctx.set_allreduce_hook(hook)
    P, C = tt(p), tt(cp)
    pm, pv, cm, cv = (torch.zeros_like(x) for x in (P, P, C, C))
    met = torch.zeros(n_upd, 10, device=dev)
    new_key, cnt = ctx.ppo_lstm_update(ldesc, P, pm, pv, cdesc, C, cm, cv, loc["states"], loc["actions"], loc["log_probs"], loc["returns"],
                                       loc["advantages"], loc["dones"], loc["c0"], loc["h0"], E, T * NE_G, key, 0, lr, hp, met, 1)
    torch.cuda.synchronize()
    res = dict(P=P.cpu().numpy(), C=C.cpu().numpy(), met=met.cpu().numpy(), key=new_key, cnt=cnt)
    if backend != "nccl":
        res["n_collectives"] = len(calls)

    # ---- rank 0: the one-device equivalent with the minibatches given explicitly
    if rank == 0:
        ref = Ctx(local)
        perm = torch.empty(E * NL, dtype=torch.int32, device=dev)
        key_ref = ref.permutation(key, perm, E, NL, 1)                  # what every rank permutes: its LOCAL env indices
        glob = {k: tt(v) for k, v in g.items()}
        P2, C2 = tt(p), tt(cp)
        pm2, pv2, cm2, cv2 = (torch.zeros_like(x) for x in (P2, P2, C2, C2))
        pg, cg = torch.zeros_like(P2), torch.zeros_like(C2)
        met2 = torch.zeros(n_upd, 10, device=dev)
        for u in range(n_upd):
            share = perm[u * NE_L:(u + 1) * NE_L]
            env_idx = torch.cat([share + r * NL for r in range(world)]).contiguous()   # union over the ranks
            ref.ppo_lstm_minibatch_fwd_bwd(ldesc, P2, pg, cdesc, C2, cg, met2[u], glob["states"], glob["actions"], glob["log_probs"],
                                           glob["returns"], glob["advantages"], glob["dones"], glob["c0"], glob["h0"], env_idx, hp)
            ref.clip_adam_step(C2, cg, cm2, cv2, u + 1, float(lr[u]), hp.max_grad_norm, hp.adam_b1, hp.adam_b2, hp.adam_eps, met2[u, 9:10])
            ref.clip_adam_step(P2, pg, pm2, pv2, u + 1, float(lr[u]), hp.max_grad_norm, hp.adam_b1, hp.adam_b2, hp.adam_eps, met2[u, 8:9])
        torch.cuda.synchronize()
        res.update(P_ref=P2.cpu().numpy(), C_ref=C2.cpu().numpy(), met_ref=met2.cpu().numpy(), key_ref=key_ref)
    np.savez(out + f".rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
