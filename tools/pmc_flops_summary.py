"""FP64 operation counts per kernel launch from rocprofv3 PMC passes of bench.py (SQ instruction counters; separate run,
kernel trace only): flops = 64 x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 x FMA_F64) + 512 x MFMA_MOPS_F64 (the SQ counts VALU
instructions per wave; an MFMA_MOPS unit is 512 operations).  Writes profiles/<round>/pmc_flops.json.
usage: pmc_flops_summary.py <out_json> <counter_csv> [<counter_csv> ...]"""
import json, sys
import pandas as pd

NAMES = ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64"]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    d = pd.concat([pd.read_csv(f) for f in files])
    d = d[d["Counter_Name"].isin(NAMES)]
    d["k"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("vpl::", "")
    g = d.groupby(["k", "Counter_Name"])["Counter_Value"].agg(["mean", "max", "count"]).reset_index()
    res = {"unit": "FP64 operations per launch (mean over the launches of the run; max = the launch in which every window works)",
           "formula": "64 * (ADD + MUL + TRANS + 2 * FMA) + 512 * MFMA_MOPS", "kernels": {}}
    for k in sorted(set(g["k"])):
        if not k.startswith("k_"):
            continue
        c = {r["Counter_Name"]: r for _, r in g[g["k"] == k].iterrows()}
        def val(n, w):
            return float(c[n][w]) if n in c else 0.0
        def fl(w):
            return 64.0 * (val(NAMES[0], w) + val(NAMES[1], w) + val(NAMES[2], w) + 2.0 * val(NAMES[3], w)) + 512.0 * val(NAMES[4], w)
        res["kernels"][k] = {"flops": fl("mean"), "flops_max": fl("max"), "mfma_share": 512.0 * val(NAMES[4], "mean") / max(fl("mean"), 1.0),
                             "launches_sampled": int(max(r["count"] for r in c.values())) if c else 0}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


main()
