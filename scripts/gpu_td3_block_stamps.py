"""Where k_critic_block spends its time: a -DBLK_STAMPS build of libplentd3 (csrc/variants/) stamps the shader clock of workgroup 0 at every phase boundary.
usage: [BLK_CRITIC_NW=4|8] [TD3_LIGHT_HANDOFF=0|1] python scripts/gpu_td3_block_stamps.py [batch] -> gpurun_out/td3_block_stamps_nw<4|8>_l<0|1>.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd import td3_fused as F, td3 as T
from plen_ml_walk_amd.build import CSRC, hipcc_path
NW, LIGHT = os.environ.get("BLK_CRITIC_NW", "8"), os.environ.get("TD3_LIGHT_HANDOFF", "1")
so = os.path.join(CSRC, "variants", "td3_stamps_nw%s_l%s.so" % (NW, LIGHT))
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in ("td3_block.hip", "td3_kernels.hip")):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DBLK_STAMPS", "-DBLK_CRITIC_NW=" + NW, "-DTD3_LIGHT_HANDOFF=" + LIGHT, "-o", so, os.path.join(CSRC, "td3_kernels.hip")], cwd=CSRC)
F.LIB_PATH = so
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
fz = F.FusedTD3(ag, seed=1, rows=False, team=False, block=True)
data = torch.randn(100000, 72, device="cuda"); data[:, 71] = 1.0
tot = torch.tensor(100000, dtype=torch.long, device="cuda")
names = ["gather", "ta1", "ta2", "a2 split-k", "a2 sum", "th1a", "th2a+th1b", "th2b+c1a", "c2a+head", "dh2a", "dh1a", "flush+c1b", "c2b+head", "dh2b", "dh1b", "loss sums"]
rows = []
for it in range(6):
    fz._zeroed = {"critic": True}
    fz.critic_backward(data, B, total=tot)
    torch.cuda.synchronize()
    nb = (B + 15) // 16
    st = fz._partials[4 * nb:4 * nb + 64].view(torch.int64).cpu().numpy()
    d = [int(st[i + 1] - st[i]) for i in range(len(names))]
    rows.append(d)
d = rows[-1]
tot_c = sum(d)
for n, c in zip(names, d):
    print("%-12s %7d cycles  %5.1f %%" % (n, c, 100.0 * c / tot_c))
print("total %d cycles (workgroup 0)" % tot_c)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"batch": B, "waves_per_workgroup": int(NW), "light_handoff": int(LIGHT), "phases": names, "cycles_last_run": d, "cycles_all_runs": rows}, open(os.path.join(ROOT, "gpurun_out", "td3_block_stamps_nw%s_l%s.json" % (NW, LIGHT)), "w"), indent=1)
