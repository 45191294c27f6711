#!/usr/bin/env python3
"""Per-class and per-kernel summary of a rocprofv3 --kernel-trace --stats run of bench.py: scripts/kstats.py <kernel_stats.csv> [n]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
steps = [int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]][0]
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def cls(n):
    if "gemm_nt_kernel" in n:
        return "gemm_nt"
    if "gemm_tn" in n or "slab_reduce" in n:
        return "wgrad"
    if n.startswith("Cijk"):
        return "hipblaslt"
    if "lsh_attn_bwd" in n:
        return "lsh_bwd"
    if "lsh_attn_fwd" in n:
        return "lsh_fwd"
    if "lsh_bwd_reduce" in n:
        return "lsh_bwd_reduce"
    if "lsh_" in n:
        return "lsh_other"
    if "xattn" in n or "sum_slabs" in n:
        return "xattn"
    if "at::native" in n or "rocclr" in n:
        return "aten"
    if "adamw" in n or "sumsq" in n or "clip" in n:
        return "optim"
    if any(k in n for k in ["residual", "ln_", "colsum", "cast_"]):
        return "rows"
    return "edges"


agg = collections.defaultdict(lambda: [0.0, 0.0])
for r in rows:
    c = cls(r["Name"])
    agg[c][0] += float(r["TotalDurationNs"]) / steps / 1e3
    agg[c][1] += int(r["Calls"]) / steps
print(f"steps {steps}: {tot / steps / 1e6:.3f} ms of kernel time per step, {sum(v[1] for v in agg.values()):.0f} launches per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:16s} {v[0]:8.1f} us/step  {v[1]:6.1f} launches/step")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step {int(r['Calls']) / steps:5.1f}x avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:120]}")
