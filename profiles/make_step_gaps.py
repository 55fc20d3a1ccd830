#!/usr/bin/env python3
"""Per-stream gaps of the training step from ONE rocprofv3 --kernel-trace of bench.py (program directly after `--`).

  python profiles/make_step_gaps.py <kernel_trace.csv> profiles/step_gaps_r03.json

A step is delimited by the last AdamW launch of consecutive steps (the update closes a step).  Reported for the LAST full step
of the run and as the mean over all full steps but the first (warm-up): wall time, the union of all kernel intervals (GPU busy
with at least one kernel), idle = wall - union (no kernel resident on any queue: launch gaps nobody covers), the sum of kernel
durations, and per hardware queue (= HIP stream): launches, busy time, span, the sum of the gaps between consecutive kernels
of that queue and the number of gaps above 5 us.  Kernels of different queues do overlap in the trace (it records real start
/ end timestamps); the tool itself slows the run (~30 ms per step under the tracer against ~23.5 ms without), so the absolute
gap sums are upper bounds."""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
if not ad:
    raise SystemExit("no AdamW launches in the trace")
# consecutive AdamW launches closer than 5 ms belong to one update
ends, last = [], ad[0]
for i in ad[1:]:
    if rows[i]["s"] - rows[last]["e"] > 5_000_000:
        ends.append(last)
    last = i
ends.append(last)


def describe(seg):
    t0, t1 = seg[0]["s"], max(r["e"] for r in seg)
    iv = sorted((r["s"], r["e"]) for r in seg)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    out = {"launches": len(seg), "wall_ms": (t1 - t0) / 1e6, "busy_union_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6,
           "kernel_time_sum_ms": sum(r["e"] - r["s"] for r in seg) / 1e6, "queues": {}}
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append(r)
    for q, v in sorted(byq.items()):
        v.sort(key=lambda r: r["s"])
        gaps = [max(0, b["s"] - a["e"]) for a, b in zip(v, v[1:])]
        out["queues"][q] = {"launches": len(v), "busy_ms": sum(r["e"] - r["s"] for r in v) / 1e6, "span_ms": (v[-1]["e"] - v[0]["s"]) / 1e6,
                            "gap_sum_ms": sum(gaps) / 1e6, "gaps_over_5us": sum(1 for g in gaps if g > 5000),
                            "largest_gap_us": max(gaps, default=0) / 1e3}
    return out


steps = [describe(rows[a + 1:b + 1]) for a, b in zip(ends, ends[1:])]
if len(steps) < 2:
    raise SystemExit("fewer than two full steps in the trace")
mean = lambda key: sum(s[key] for s in steps[1:]) / len(steps[1:])
out = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " "), "source": sys.argv[1].split("/")[-1], "full_steps": len(steps),
       "mean_over_steps": {k: round(mean(k), 4) for k in ("launches", "wall_ms", "busy_union_ms", "idle_ms", "kernel_time_sum_ms")},
       "last_step": steps[-1]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["mean_over_steps"]), {q: (v["launches"], round(v["busy_ms"], 2), round(v["gap_sum_ms"], 2)) for q, v in steps[-1]["queues"].items()})
