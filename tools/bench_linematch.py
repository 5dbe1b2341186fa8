"""Secondary metric: frame pairs/s of the KLT line matcher on 63 consecutive pairs of a 64-frame batch
(752x480), inputs resident in HBM, plus the CPU oracle on one host core.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
import torch
import vplines_slam_amd as v
from bench_edlines import frames

def main():
    n, steps, warm = 64, 20, 3
    dev = torch.device("cuda", 0)
    imgs = frames(n)
    fe = v.frontend.FrontendContext(device=0, max_images=n, width=752, height=480, max_lines=256,
                                    stream=torch.cuda.current_stream(dev).cuda_stream)
    lines = fe.detect_batch(imgs)
    lines = [l[np.argsort(-l[:, 9])][:256] for l in lines]
    pairs = [(i, i + 1) for i in range(n - 1)]
    fe.match_reserve(len(pairs), 4096)
    fe.match_upload(pairs, [lines[a] for a, _ in pairs], [lines[b] for _, b in pairs])
    for _ in range(warm):
        fe.match_run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fe.match_run()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    r2c, ok = fe.match_download()
    nk = [len(fe.match_debug_kps(i)["status"]) for i in range(len(pairs))]
    import oracle_api as o
    tc = time.perf_counter()
    same = 0
    for i in range(8):
        a, b = pairs[i]
        _, ro, _ = o.line_match(imgs[a], imgs[b], lines[a], lines[b])
        same += int(np.array_equal(ro, r2c[i]))
    tc = time.perf_counter() - tc
    print(json.dumps({"metric": "KLT line matching pairs/s (752x480, 63 pairs)", "value": len(pairs) * steps / dt,
                      "unit": "pairs/s", "ms_per_batch": 1e3 * dt / steps, "mean_keypoints_per_pair": float(np.mean(nk)),
                      "mean_matches_per_pair": float(np.mean([(r >= 0).sum() for r in r2c])),
                      "cpu_oracle_pairs_per_s_1thread": 8 / tc, "identical_matches_on_8_pairs": same}))
    fe.close()

if __name__ == "__main__":
    main()
