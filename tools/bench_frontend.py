"""Secondary metric of SURVEY 8(d), config 4: frames/s of the whole line front-end -- EDLines on 64 frames 752x480 plus KLT
line matching of the 63 consecutive pairs -- with the frames resident in HBM.  The detected lines go from the detector to the matcher on the
device (vpl_match_from_detected) and are downloaded, with the matches, at the end of the batch (VPL_FE_HOST_HANDOVER=1: through
the host between the two stages, as rounds 1-3 measured it); all of it inside the timed region.
With --prep the raw frames first go through the undistortion remap + CLAHE(3.0, 8x8) of LineFeatureTracker::readImage
(EuRoC cam0 maps) on the device, inside the timed region.  With --vp the vanishing-point stage runs on the lines of every
frame after the match (hypotheses from all lines of the frame), also inside the timed region.  Prints one JSON line."""
import json
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
import torch
import vplines_slam_amd as v
from bench_edlines import frames as _old_frames


def frames(n):
    return v.workload.frame_stream(n)


def main():
    n, steps, warm = 64, 10, 2
    dev = torch.device("cuda", 0)
    imgs = frames(n)
    fe = v.frontend.FrontendContext(device=0, max_images=n, width=752, height=480, max_lines=256,
                                    stream=torch.cuda.current_stream(dev).cuda_stream)
    prep = "--prep" in sys.argv
    vp = "--vp" in sys.argv
    vp_out, vp_ms = [], []
    fe.match_reserve(n - 1, 8192 if prep else 4096)
    if prep:
        from test_preproc import euroc_maps, oracle_clahe, oracle_remap
        mx, my = euroc_maps()
        fe.set_maps(mx, my)
        fe.pre_upload(imgs)
    else:
        fe.upload(imgs)
    pairs = [(i, i + 1) for i in range(n - 1)]

    def step():
        if prep:
            fe.pre_run(True, 3.0, (8, 8))
        fe.detect()
        if os.environ.get("VPL_FE_HOST_HANDOVER"):      # rounds 1-3: the lines went through the host between detector and matcher
            fe.synchronize()
            lines = fe.download()
            lines = [l[:256] for l in lines]
            fe.match_upload(pairs, [lines[a] for a, _ in pairs], [lines[b] for _, b in pairs])
        else:
            fe.match_from_detected(pairs, 256)          # device to device (vpl_match_from_detected)
        fe.match_run()
        fe.synchronize()
        if not os.environ.get("VPL_FE_HOST_HANDOVER"):
            lines = [l[:256] for l in fe.download()]
        out = fe.match_download()
        if vp:
            t = time.perf_counter()
            vp_out.append(fe.vp_detect(lines, lines, 458.654, 376.0, 240.0, np.arange(n) + 7, np.zeros(n, np.int32)))
            vp_ms.append(1e3 * (time.perf_counter() - t))
        return lines, out

    for _ in range(warm):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        lines, (r2c, ok) = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    # device-only share: the kernels of one batch, timed with events
    ep, e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    ep.record()
    if prep:
        fe.pre_run(True, 3.0, (8, 8))
    e0.record(); fe.detect(); e1.record(); fe.match_run(); e2.record()
    torch.cuda.synchronize(dev)
    if prep:
        imgs = np.stack([oracle_clahe(oracle_remap(imgs[i], mx, my)) for i in range(5)])
    import oracle_api as o
    tc = time.perf_counter()
    same = 0
    for i in range(4):
        la, lb = o.edlines(imgs[i]), o.edlines(imgs[i + 1])
        _, ro, _ = o.line_match(imgs[i], imgs[i + 1], la, lb)
        same += int(len(la) == len(lines[i]) and np.array_equal(ro, r2c[i]))
    tc = (time.perf_counter() - tc) / 4
    print(json.dumps({"metric": "line front-end frames/s (EDLines + KLT matching, 752x480, batch 64)", "value": n / dt,
                      "unit": "frames/s", "prep": prep, "vp": vp, "vp_stage_ms_per_batch_incl_transfers": float(np.mean(vp_ms[2:])) if vp else None,
                      "vp_classified_lines_per_frame": float(np.mean([(i < 3).sum() for i in vp_out[-1][1]])) if vp else None, "ms_per_batch": 1e3 * dt, "device_ms_prep": ep.elapsed_time(e0), "device_ms_detect": e0.elapsed_time(e1),
                      "device_ms_match": e1.elapsed_time(e2), "host_round_trip_ms": 1e3 * dt - ep.elapsed_time(e2),
                      "mean_lines_per_frame": float(np.mean([len(l) for l in lines])),
                      "mean_matches_per_pair": float(np.mean([(r >= 0).sum() for r in r2c])),
                      "cpu_oracle_frames_per_s_1thread": 1.0 / tc, "identical_on_4_pairs": same}))
    fe.close()


if __name__ == "__main__":
    main()
