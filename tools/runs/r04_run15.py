import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x"]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import runpy
g = runpy.run_path(os.path.join(ROOT, "tools/sac_bench.py"), run_name="bench")
m, env, torch = g["m"], g["env"], g["torch"]
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
K = 300
torch.cuda.synchronize()
for _ in range(K):
    t0 = time.perf_counter(); views = tuple(x[m.pos] for x in m.ring); tick("ring views", t0)
    t0 = time.perf_counter(); views[0].copy_(env.obs); tick("obs copy_", t0)
    t0 = time.perf_counter(); po = m.policy_obs(env.obs); tick("policy_obs", t0)
    t0 = time.perf_counter()
    m.key = m.ctx.sac_act(m.pdesc, m.pparams, po, m.key, views[2], m.log_std_min, m.log_std_max, scheme=m.scheme,
                          processed=(m._low, m._half_range, m._processed), row_offset=0, n_global=m.nr_envs)
    tick("sac_act", t0)
    t0 = time.perf_counter(); env.step_into(m._processed, views[1], views[3], views[4]); tick("env.step_into", t0)
    m.pos = (m.pos + 1) % m.capacity; m.size = min(m.size + 1, m.capacity)
    t0 = time.perf_counter(); i1, i2 = m._host_indices(m.batch_size, m.nr_envs); tick("_host_indices", t0)
    t0 = time.perf_counter(); m.ctx.sac_replay_sample(m.ring, i1, i2, m.batch); tick("replay_sample", t0)
    t0 = time.perf_counter(); hp = m.hparams(); tick("hparams", t0)
    t0 = time.perf_counter()
    m.key, m.opt_count = m.ctx.sac_update(m.pdesc, m.pparams, m.pm, m.pv, m.qdesc, m.qparams, m.qm, m.qv, m.qtarget, m.log_alpha,
                                          m.am, m.av, m.batch, m.key, m.opt_count, hp, m.metrics_dev, m.scheme)
    tick("sac_update", t0)
torch.cuda.synchronize()
for k, v in T.items():
    print(f"{k:16s} {1e6 * v / K:7.1f} us")
print(f"{'sum':16s} {1e6 * sum(T.values()) / K:7.1f} us")
