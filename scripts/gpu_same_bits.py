"""Development: does a kernel change leave every bit alone?  40 steps x 512 envs of random actions through the in-tree library and through csrc/variants/<lib>
(a copy of the build before the change), both arithmetics: outputs and final states compared bit for bit -- once in the reference configuration and once with
joint_act targets of up to +-2.6 rad, which drive joints past their +-1.7 rad limits (the limit flavours of the solver-loop copies; the script reports how many
substeps had a violated limit at the end).   usage: python scripts/gpu_same_bits.py prev.so"""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from plen_ml_walk_amd.vec_env import PlenVecEnv\n"
        "out = []\n"
        "g = torch.Generator().manual_seed(5); acts = (torch.rand(40, 512, 18, generator=g) * 2 - 1).float().cuda()\n"
        "lim = sys.argv[3] == 'limits'\n"
        "if lim: acts = acts * 2.6\n"
        "env = PlenVecEnv(512, dtype=getattr(torch, sys.argv[2]), joint_act=lim, auto_reset=(sys.argv[3] != 'fallen')); env.reset()\n"
        "for t in range(40):\n"
        "    o, r, d, _ = env.step(acts[t]); out.append(torch.cat([o, r[:, None], d.to(o.dtype)[:, None]], 1).cpu().numpy().copy())\n"
        "np.save(sys.argv[1], np.array(out)); np.save(sys.argv[1] + '.state.npy', env.get_state().cpu().numpy())\n" % ROOT)
# fallen: no auto-reset, the robots fall and stay down -- links other than the feet on the ground, contact slots lent to their box corners (ports on another limb's support)
for dt, mode in (("float32", "reference"), ("float64", "reference"), ("float32", "limits"), ("float64", "limits"), ("float32", "fallen"), ("float64", "fallen")):
    res = {}
    for tag in ("-", sys.argv[1]):
        env = dict(os.environ)
        if tag != "-": env["PLENVEC_LIB"] = os.path.join(ROOT, "plen_ml_walk_amd/csrc/variants", tag)
        p = "/tmp/sb_%s_%s.npy" % (dt, tag.replace(".so", ""))
        subprocess.run([sys.executable, "-c", code, p, dt, mode], check=True, env=env)
        res[tag] = (np.load(p), np.load(p + ".state.npy"))
    a, b = res["-"], res[sys.argv[1]]
    q = a[1][:, 13:31]
    print(dt, mode, "outputs bitwise equal:", np.array_equal(a[0], b[0], equal_nan=True), " states:", np.array_equal(a[1], b[1], equal_nan=True),
          " envs with a joint beyond +-1.7 rad at the end: %d of %d; torso below 0.08 m: %d" % (int((np.abs(q) > 1.7).any(1).sum()), q.shape[0], int((a[1][:, 2] < 0.08).sum())), flush=True)
