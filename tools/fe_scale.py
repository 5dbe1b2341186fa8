"""Line front-end throughput against the number of frames in flight (bench.py's config-4 measurement at n = 64 .. 1024):
the routing stage runs one wave per frame, so throughput grows until the CUs are full.  GPU only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vplines_slam_amd as v
dev = torch.device("cuda:0")
imgs = v.workload.frame_stream(64)
for n in (64, 256, 512, 1024):
    try:
        r = bench.frontend_config4(torch, v, dev, steps=3, warm=1, n=n, imgs=imgs)
        print(n, round(r["value"]), "frames/s", round(r["ms_per_batch"], 2), "ms/batch", "detect", round(r["device_ms_detect"], 2), "match", round(r["device_ms_match"], 2), flush=True)
    except Exception as e:
        print(n, "failed", repr(e)[:200], flush=True)
        break
