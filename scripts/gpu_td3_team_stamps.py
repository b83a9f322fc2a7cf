"""Where one k_critic_team launch (batch 100) spends its time: builds csrc/variants/td3_stamps.so (-DTEAM_STAMPS: workgroup 0 records the shader clock after
every workgroup barrier), runs updates and prints the phase durations (100 MHz constant clock -> us).
usage: python scripts/gpu_td3_team_stamps.py [extra -D flags ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd import build as Bd, td3_fused as F, td3 as T

out = os.path.join(Bd.CSRC, "variants", "td3_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([Bd.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTEAM_STAMPS"] + sys.argv[1:] + ["-o", out, Bd.TD3_SRC], cwd=Bd.CSRC)
F.LIB_PATH = out
torch.manual_seed(0)
ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
fz = F.FusedTD3(ag, seed=1, team=True)
fz.enable_flat_adam()
made = []
def alloc(*shape):
    t = torch.zeros(*shape, device="cuda"); made.append(t); return t
fz._alloc = alloc
data = torch.randn(100000, 72, device="cuda")
tot = torch.tensor(100000, dtype=torch.long, device="cuda")
names = ["gather", "at1", "at2", "at3 + action (wave 0) | c14 (waves 1-7)", "ct14", "ct2/5 + c2/5 heads", "y, loss, dq", "dh2", "dh1"]
rows = []
for k in range(40):
    del made[:]
    fz.update(data, 100, with_policy=(k % 2 == 1), all_reduce=False, total=tot)
    torch.cuda.synchronize()
    t1 = [t for t in made if t.shape == (100, 512)][1]          # allocation order in critic_backward_rows: t0, t1, c1, c2, dh2, dh1
    st = t1[0, 256 + 8:256 + 8 + 2 * 12].cpu().numpy().view(np.uint64).astype(np.int64)
    if k >= 8:
        rows.append(np.diff(st[:len(names) + 1]))
d = np.median(np.array(rows), 0) / 100.0          # s_memtime ticks at 100 MHz
for n, v in zip(names, d):
    print("%-24s %6.2f us" % (n, v))
print("%-24s %6.2f us" % ("total (to last barrier)", d.sum()))
