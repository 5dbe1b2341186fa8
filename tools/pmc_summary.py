"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) of `bench.py` and of the
calibration program tools/pmc_calib into profiles/<round>/pmc_traffic.json: HBM bytes per launch of every BA kernel,
corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KB and counts half of the bytes of coalesced reads on
gfx950 -- confirmed for this path's 8-byte-per-lane reads by the calibration kernels; WRITE_SIZE is exact).
usage: pmc_summary.py <fetch_counter_csv> <write_counter_csv> <calib_fetch_csv> <calib_write_csv> <out_json>"""
import json, sys
import pandas as pd

def per_kernel(path, counter):
    d = pd.read_csv(path)
    d = d[d["Counter_Name"] == counter]
    d["k"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("vpl::", "")
    return d.groupby("k")["Counter_Value"].agg(["mean", "max", "count"])

def main():
    fetch, write, cf, cw, out = sys.argv[1:6]
    F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    CF, CW = per_kernel(cf, "FETCH_SIZE"), per_kernel(cw, "WRITE_SIZE")
    gib = float(1 << 30)
    fetch_scale = gib / (CF.loc["read_f64", "mean"] * 1024.0)      # bytes really read / bytes reported
    write_scale = gib / (CW.loc["write_f64", "mean"] * 1024.0)
    res = {"unit": "bytes per launch", "fetch_correction": fetch_scale, "write_correction": write_scale,
           "calibration": "tools/pmc_calib: 1 GiB coalesced 8-B/lane read -> FETCH_SIZE %.0f KB; 1 GiB write -> WRITE_SIZE %.0f KB"
                          % (CF.loc["read_f64", "mean"], CW.loc["write_f64", "mean"]), "kernels": {}}
    for k in F.index:
        if not k.startswith("k_"):
            continue
        rd = F.loc[k, "mean"] * 1024.0 * fetch_scale
        wr = (W.loc[k, "mean"] if k in W.index else 0.0) * 1024.0 * write_scale
        rdx = F.loc[k, "max"] * 1024.0 * fetch_scale
        wrx = (W.loc[k, "max"] if k in W.index else 0.0) * 1024.0 * write_scale
        # mean over all launches of the run (light and heavy ones) and the launch with the most traffic (every window active)
        res["kernels"][k] = {"read": rd, "write": wr, "total": rd + wr, "read_max": rdx, "write_max": wrx,
                             "total_max": rdx + wrx, "launches_sampled": int(F.loc[k, "count"])}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

main()
