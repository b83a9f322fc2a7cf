"""Profiling build (-DPGS_STAMPS, variants/stamps.so): distribution of wave lifetimes inside one 4096-env launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = build_variant("stamps", ["-DPGS_STAMPS"])
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
n = 4096
dtype = torch.float64 if "f64" in sys.argv else torch.float32          # usage: python scripts/gpu_wave_times.py [f64]
env = PlenVecEnv(n, dtype=dtype); env.reset()
print('balance:', 'off' if os.environ.get('PLENVEC_NO_BALANCE') else 'on')
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.rand(60, n, 18, device="cuda", generator=g) * 2 - 1
for t in range(60):
    _, _, done, _ = env.step(acts[t])
    if t in (5, 20, 40, 59):
        aux = env.get_aux().cpu().numpy()
        d = aux[:, 6].astype(np.float64); ok = d > 1000; est = aux[:, 7].astype(np.float64)          # envs that auto-reset this step carry the cached aux (0)
        d = d[ok]; c = (aux[ok, 4] + aux[ok, 5]); est = est[ok]
        A = np.stack([est, np.ones_like(est)], 1); coef = np.linalg.lstsq(A, d, rcond=None)[0]; resid = d - A @ coef
        print('   cycles ~ %.2f * estimate + %.0f, residual rms %.0f (%.1f %% of mean); corr %.3f' % (coef[0], coef[1], resid.std(), 100 * resid.std() / d.mean(), np.corrcoef(est, d)[0, 1]))
        full = np.zeros(n); full[ok] = d; sums = full.reshape(4, 1024).sum(0)
        slots = 2048 if dtype == torch.float64 else 4096
        print('   sum of wave cycles / resident wave slots (%d): %.0f cycles = the launch time of a perfectly packed schedule at this contention' % (slots, d.sum() / slots))
        print('   per-SIMD sum of wave cycles: mean %.0f  max %.0f  (max/mean %.3f)' % (sums.mean(), sums.max(), sums.max() / sums.mean()))
        print("step %2d: waves %d  cycles min %.0f  p10 %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f | feet on ground 0/1/2: %s | median cycles by feet: %s" % (
            t, len(d), d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max(),
            [int((c == k).sum()) for k in range(3)], [int(np.median(d[c == k])) if (c == k).any() else -1 for k in range(3)]))
env.close()
