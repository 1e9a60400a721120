import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from cellector_amd import Cellector
N, L, d = 20000, 500000, 0.015
g = Cellector(0, stream=torch.cuda.current_stream().cuda_stream)
g.set_option("keep_coo", 0)
g.load_synthetic(L, N, d, seed=4, minority_fraction=0.05, min_alt=4, min_ref=4)
for _ in range(23):
    g.em_iteration(5.0)
torch.cuda.synchronize()
g.close()
