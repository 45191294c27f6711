#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/: per-kernel mean counter values (one CSV per counter) and
profiles/<tag>_pmc_traffic.json -- HBM bytes per launch of the LSH kernels, which bench.py's ``roofline.traffic`` reads IF the
kernel name and shape it launches match an entry (a changed kernel must not inherit old counters).

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x (doubled here); WRITE_SIZE
is exact; both in KB.  FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots), SQ counters a third:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r03_FETCH_SIZE -- python3 scripts/kbench.py --only fwd,bwd,hash --iters 3
    rocprofv3 --pmc WRITE_SIZE ... ; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY ... (8 SQ slots)
    python scripts/pmc_summary.py gpurun_out/pmc_r03_FETCH_SIZE gpurun_out/pmc_r03_WRITE_SIZE gpurun_out/pmc_r03_SQ --tag r03
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kbench.py's three shapes: grid size (threads) -> shape key of bench.py's lookup.  Grid = workgroups x threads per workgroup.
KBENCH_SHAPES = {"dec": {"BH": 96, "T": 1024, "bucket": 128, "rounds": 8}, "enc": {"BH": 96, "T": 256, "bucket": 64, "rounds": 8},
                 "long": {"BH": 32, "T": 4096, "bucket": 64, "rounds": 8}}


def shape_of(kernel: str, grid: int):
    """Which kbench shape a (kernel, grid) pair belongs to: the hash / sort kernels by their bucket-count template argument
    (kbench's three shapes have 8, 4 and 64 buckets), the attention kernels by bucket size and launch geometry."""
    import re
    m = re.match(r"lsh_hash_rounds_kernel<(\d+)>", kernel) or re.match(r"lsh_hash_sort_kernel<(\d+),", kernel)
    if m:
        return {4: ("dec", KBENCH_SHAPES["dec"]), 2: ("enc", KBENCH_SHAPES["enc"]), 32: ("long", KBENCH_SHAPES["long"])}.get(int(m.group(1)), (None, None))
    m = re.match(r"lsh_sort_ids_kernel<(\d+),", kernel)
    if m:
        return {3: ("dec", KBENCH_SHAPES["dec"]), 2: ("enc", KBENCH_SHAPES["enc"]), 6: ("long", KBENCH_SHAPES["long"])}.get(int(m.group(1)), (None, None))
    for name, sh in KBENCH_SHAPES.items():
        chunks = sh["BH"] * sh["rounds"] * sh["T"] // sh["bucket"]
        threads = sh["bucket"] * 4
        cands = {chunks // r * threads for r in (1, 2, 4, 8)}               # one workgroup per chunk, or runs of r chunks
        if f"<{sh['bucket']}," in kernel and grid in cands:
            return name, sh
    return None, None


def summarise(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[r["Counter_Name"]][(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--tag", default="r03")
    args = ap.parse_args()
    means = collections.defaultdict(dict)
    for d in args.dirs:
        for counter, kernels in summarise(d).items():
            out = os.path.join(ROOT, "profiles", f"{args.tag}_pmc_{counter}_lsh_kernels.csv")
            with open(out, "w") as fh:
                fh.write("kernel|grid,launches,mean_counter_value\n")
                for (k, g), v in sorted(kernels.items()):
                    fh.write(f"{k}|grid{g},{len(v)},{sum(v) / len(v):.1f}\n")
                    means[(k, g)][counter] = sum(v) / len(v)
            print("wrote", out)
    entries = []
    for (k, g), m in sorted(means.items()):
        if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m or "lsh" not in k:
            continue
        name, sh = shape_of(k, g)
        if sh is None:
            continue
        fetch, write = m["FETCH_SIZE"] * 1024 * 2, m["WRITE_SIZE"] * 1024
        e = dict(kernel=k, grid=g, kbench_shape=name, shape=sh, FETCH_SIZE_KB=round(m["FETCH_SIZE"], 1), WRITE_SIZE_KB=round(m["WRITE_SIZE"], 1),
                 fetch_bytes_corrected=int(fetch), write_bytes=int(write), traffic_bytes=int(fetch + write))
        sq = {c: v for c, v in m.items() if c.startswith("SQ_")}
        if sq:
            e["sq"] = {c: round(v, 1) for c, v in sorted(sq.items())}
            if sq.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_conflict_share"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"], 4)
        entries.append(e)
    doc = dict(entries=entries,
               correction="gfx950: FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact; KB",
               how="separate passes: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc <8 SQ counters> --output-format csv -- "
                   "python3 scripts/kbench.py --only fwd,bwd,hash --iters 3 (the program directly behind --)",
               sq_units="SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed over waves (resp. SEs); "
                        "SQ_VALU_MFMA_BUSY_CYCLES counts cycles; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE are LDS-array cycles")
    out = os.path.join(ROOT, "profiles", f"{args.tag}_pmc_traffic.json")
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, [(e["kernel"][:40], e["kbench_shape"], e["traffic_bytes"]) for e in entries])


if __name__ == "__main__":
    main()
