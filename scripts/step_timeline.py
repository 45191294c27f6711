#!/usr/bin/env python3
"""Timeline of one replayed training step from a rocprofv3 --kernel-trace CSV: per hardware queue the busy time and span,
the gaps on every queue, and where the queues run side by side.
    python scripts/step_timeline.py gpurun_out/<tag>/kernel_trace.csv [step index, default 3] [--list]"""
import csv
import sys


def main():
    path = sys.argv[1]
    k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    rows = list(csv.DictReader(open(path)))
    r = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0].replace("void ", "")[:56], x["Queue_Id"])
               for x in rows)
    ad = [i for i, x in enumerate(r) if x[2].startswith("adamw")]
    seg = r[ad[k] + 1:ad[k + 1] + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    print(f"step {k}: wall {(t1 - t0) / 1e3:.1f} us, {len(seg)} kernels, sum of kernel time {sum(e - s for s, e, _, _ in seg) / 1e3:.1f} us")
    queues = {}
    for s, e, n, q in seg:
        queues.setdefault(q, []).append((s, e, n))
    for q, v in queues.items():
        print(f"  queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e3:.1f} us, first at {(v[0][0] - t0) / 1e3:.1f}, last ends {(v[-1][1] - t0) / 1e3:.1f}")
        for a, b in zip(v, v[1:]):
            g = b[0] - a[1]
            if g > 20000:
                print(f"      gap {g / 1e3:7.1f} us at {(a[1] - t0) / 1e3:8.1f}: after {a[2][:40]} | before {b[2][:40]}")
    # time with >= 2 queues busy
    ev = []
    for s, e, _, _ in seg:
        ev += [(s, 1), (e, -1)]
    ev.sort()
    busy2 = busy1 = 0
    cur, last = 0, t0
    for t, d in ev:
        if cur >= 2:
            busy2 += t - last
        elif cur == 1:
            busy1 += t - last
        cur += d
        last = t
    print(f"  two queues busy {busy2 / 1e3:.1f} us, one {busy1 / 1e3:.1f} us, none {(t1 - t0 - busy1 - busy2) / 1e3:.1f} us")
    if "--list" in sys.argv:
        for s, e, n, q in seg:
            print(f"q{q} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {n}")


if __name__ == "__main__":
    main()
