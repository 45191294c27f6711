#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass) into profiles/: per-kernel mean counter value and the
HBM traffic of the dominant kernel (gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE under-reports wide
coalesced reads by 2x; WRITE_SIZE is exact; both in KB).

    python scripts/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE --tag r01b
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def summarise(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[r["Counter_Name"]][f"{name}|grid{r['Grid_Size']}"].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--kernel", default="lsh_attn_bwd_kernel<128, true, true>")
    ap.add_argument("--suffix", default="lsh_kernels", help="file name suffix of the per-counter CSVs")
    args = ap.parse_args()
    means = {}
    for d in args.dirs:
        for counter, kernels in summarise(d).items():
            out = os.path.join(ROOT, "profiles", f"{args.tag}_pmc_{counter}_{args.suffix}.csv")
            with open(out, "w") as fh:
                fh.write("kernel|grid,launches,mean_counter_value_KB\n")
                for k, v in kernels.items():
                    fh.write(f"{k},{len(v)},{sum(v) / len(v):.1f}\n")
                    if k.startswith(args.kernel):
                        means[counter] = sum(v) / len(v)
            print("wrote", out)
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        fetch, write = means["FETCH_SIZE"] * 1024 * 2, means["WRITE_SIZE"] * 1024
        js = dict(kernel=args.kernel, shape="B*H=96, T=1024, bucket 128, 8 rounds (decoder layer of config/baseline.yml, B=12)",
                  FETCH_SIZE_KB=round(means["FETCH_SIZE"], 1), WRITE_SIZE_KB=round(means["WRITE_SIZE"], 1),
                  fetch_bytes_corrected=int(fetch), write_bytes=int(write), traffic_bytes=int(fetch + write),
                  correction="gfx950: FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact",
                  how="two separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -- python3 scripts/kbench.py --only fwd,bwd --iters 3")
        out = os.path.join(ROOT, "profiles", f"{args.tag}_pmc_lsh_attn_bwd.json" if args.suffix == "lsh_kernels" else f"{args.tag}_pmc_{args.suffix}.json")
        if args.suffix != "lsh_kernels":
            js["shape"] = "see the kernel|grid column of the per-counter CSVs"
            js["how"] = js["how"].replace("scripts/kbench.py --only fwd,bwd --iters 3", "scripts/gemm_nt_once.py")
        json.dump(js, open(out, "w"), indent=1)
        print("wrote", out, js["traffic_bytes"])


if __name__ == "__main__":
    main()
