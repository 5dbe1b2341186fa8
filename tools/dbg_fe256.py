import sys, os, json
sys.path.insert(0, '/root/repo')
import torch, bench
import vplines_slam_amd as v
dev = torch.device("cuda:0")
imgs = v.workload.frame_stream(64)
r = bench.frontend_config4(torch, v, dev, steps=3, warm=1, n=256, imgs=imgs)
print(round(r["value"]), r["ms_per_batch"], {k: round(x, 3) for k, x in r["kernels_ms_per_batch"].items()})
