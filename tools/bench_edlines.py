"""Secondary metric: frames/s of the EDLines extractor on a batch of 64 frames 752x480,
inputs resident in HBM, plus the CPU oracle on the host cores for reference.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import vplines_slam_amd as v

def frames(n, seed=0x5EED0000 + 4000):
    """the two MH_04 fixtures plus warped + noisy variants"""
    rng = np.random.default_rng(seed)
    base = [np.load(os.path.join(ROOT, "tests", "golden", "mh04_%d.npy" % i)) for i in (1, 2)]
    out = []
    for i in range(n):
        im = base[i % 2].astype(np.float32)
        sx, sy = rng.integers(-6, 7), rng.integers(-4, 5)
        im = np.roll(np.roll(im, sx, axis=1), sy, axis=0)
        im = im + rng.normal(0, 2.0, im.shape)
        out.append(np.clip(np.rint(im), 0, 255).astype(np.uint8))
    return np.stack(out)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64   # SURVEY 8d config 4 is 64; larger batches fill more of the 256 CUs
    steps, warm = 20, 3
    dev = torch.device("cuda", 0)
    imgs = frames(n)
    fe = v.frontend.FrontendContext(device=0, max_images=n, width=752, height=480, max_lines=1024,
                                    stream=torch.cuda.current_stream(dev).cuda_stream)
    fe.upload(imgs)
    for _ in range(warm):
        fe.detect()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fe.detect()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    lines = fe.download()
    rs = fe.route_stats(0)
    na = len(fe.debug_stage(0)["anchors"])
    import oracle_api as o
    tc = time.perf_counter()
    ref = [o.edlines(imgs[i]) for i in range(16)]
    tc = time.perf_counter() - tc
    same = sum(len(ref[i]) == len(lines[i]) for i in range(16))
    px = 752 * 480
    print(json.dumps({"metric": "EDLines frames/s (752x480, batch %d)" % n, "value": n * steps / dt, "unit": "frames/s",
                      "ms_per_batch": 1e3 * dt / steps, "mean_lines_per_frame": float(np.mean([len(l) for l in lines])),
                      "gradient_stage_algorithmic_bytes_per_frame": px * 8,
                      "cpu_oracle_frames_per_s_1thread": 16 / tc, "line_count_match_on_16_frames": same, "route_stats_frame0": rs, "anchors_frame0": na}))
    fe.close()

if __name__ == "__main__":
    main()
