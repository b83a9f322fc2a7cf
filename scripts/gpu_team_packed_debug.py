"""Development: where do the packed and the parked small-batch paths part?  Per update: loss and parameter checksums of both, from equal seeds."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd import td3 as T
from plen_ml_walk_amd import td3_fused as F
data = torch.randn(5000, 72, device="cuda")
data[:, 70] = torch.rand(5000, device="cuda"); data[:, 71] = (torch.rand(5000, device="cuda") > 0.1).float()
tot = torch.tensor(5000, dtype=torch.long, device="cuda")
MODS = sys.argv[1] if len(sys.argv) > 1 else ""
print("mods:", MODS)
seq = {}
for packed in ("0", "1", "0b"):
    os.environ["PLEN_TD3_TEAM_PACKED"] = packed[0]
    torch.manual_seed(41)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = F.FusedTD3(ag, seed=9)
    fz.enable_flat_adam()
    rows = []
    for k in range(12):
        if k == 5 and "5" in MODS:
            with torch.no_grad():
                ag.actor.fc2.weight.mul_(1.001)
        if k == 8 and "8" in MODS:
            fz.fuse_adam = False
            fz.update(data, 100, with_policy=True, all_reduce=False, total=tot)
            fz.fuse_adam = True
        loss = fz.update(data, (100 if k % 3 else 64) if "b" in MODS else 100, with_policy=(k % 2 == 1), all_reduce=False, total=tot).clone()
        torch.cuda.synchronize()
        rows.append((float(loss), ag._critic_flat.flat.double().sum().item(), ag._actor_flat.flat.double().sum().item(), ag._critic_target_flat.flat.double().sum().item(),
                     [t.clone() for t in (ag._critic_flat.flat, ag._actor_flat.flat, ag._critic_target_flat.flat, ag._actor_target_flat.flat)], loss))
    seq[packed] = rows
for k in range(12):
    a, b, c = seq["0"][k], seq["1"][k], seq["0b"][k]
    print(k, "loss equal 0/1:", torch.equal(a[5], b[5]), " 0/0b:", torch.equal(a[5], c[5]), " params equal 0/1:", [torch.equal(x, y) for x, y in zip(a[4], b[4])],
          " 0/0b:", [torch.equal(x, y) for x, y in zip(a[4], c[4])], "max diff 0/1:", [float((x - y).abs().max()) for x, y in zip(a[4], b[4])])
