"""bench.py's td3_reference leg alone (the reference's recipe: 64 envs, batch 100, one update per env-step) -> one JSON object."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
print(json.dumps(bench.td3_reference_leg(None, torch.device("cuda:0"), 0, 1, None)))
