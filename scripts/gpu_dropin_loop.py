"""The reference's single-environment training loop (plen_td3.py:83-157) on this package's drop-in surfaces, timed: env-steps/s while exploring (random actions, no
updates) and while training (select_action + env.step + replay add + one train() of batch 100 per step), with the time of each call.
usage: python scripts/gpu_dropin_loop.py [train_steps]   -> gpurun_out/r04_dropin_loop.json"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PLEN_QUIET", "1")
import numpy as np, torch
from plen_ml_walk_amd import plen_td3 as D

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
out = {}
for name, fused in (("fused train()", None), ("autograd train()", False)):
    run = D._Run(tempfile.mkdtemp(), 0, 0, 0, True)
    run.agent.fused_train = fused
    env = run.env
    obs = env.reset()
    tm = {"select_action": 0.0, "env.step": 0.0, "buffer.add": 0.0, "train": 0.0}
    ep_len = 0

    def loop(n, exploring, train):
        global obs, ep_len
        t0 = time.perf_counter()
        for t in range(n):
            a0 = time.perf_counter()
            action = run.behaviour_action(obs, exploring=exploring, noise_scale=0.1)
            a1 = time.perf_counter()
            obs2, r, done, _ = env.step(action)
            a2 = time.perf_counter()
            ep_len += 1
            run.buffer.add((obs, action, obs2, r, float(done) if ep_len < 500 else 0))
            a3 = time.perf_counter()
            obs = obs2
            if train:
                run.agent.train(run.buffer, 100)
            a4 = time.perf_counter()
            if train:
                tm["select_action"] += a1 - a0; tm["env.step"] += a2 - a1; tm["buffer.add"] += a3 - a2; tm["train"] += a4 - a3
            if done:
                obs = env.reset(); ep_len = 0
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)
    rate_explore = loop(1500, True, False)
    loop(100, False, True)
    for k in tm:
        tm[k] = 0.0
    rate_train = loop(n_train, False, True)
    out[name] = {"explore_env_steps_per_s": rate_explore, "train_env_steps_per_s": rate_train, "us_per_call": {k: v / n_train * 1e6 for k, v in tm.items()}}
    print(name, json.dumps(out[name]), flush=True)
    env.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_dropin_loop.json"), "w"), indent=1)
