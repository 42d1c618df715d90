import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch
import bench, zkp_hip as zkp
zkp.init()
print(bench.bench_plonk(zkp, torch, torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 16,
                        expand=("auto" if sys.argv[2] == "auto" else int(sys.argv[2])) if len(sys.argv) > 2 else 0))
