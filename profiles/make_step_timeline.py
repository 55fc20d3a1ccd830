#!/usr/bin/env python3
"""Which kernels are EXPOSED in the training step: from one rocprofv3 --kernel-trace of bench.py.

  python profiles/make_step_timeline.py <kernel_trace.csv> profiles/step_timeline_r04.json

For the last full step (delimited by AdamW launches, as make_step_gaps.py does) the wall time is cut at every kernel start / end
and each slice is attributed by the number of kernels resident in it: 0 (idle), 1 (exposed: the step waits for exactly that
kernel - the other stream has nothing to run) or >= 2 (overlapped).  Reported: the three totals, the exposed time per kernel
name (top 30) and the same per coarse section of the step in launch order (encoders forward, routing forward, routing backward,
encoders backward, optimiser), found from the first / last launch of the routing module's kernels."""
import collections
import csv
import json
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends, last = [], ad[0]
for i in ad[1:]:
    if rows[i]["s"] - rows[last]["e"] > 5_000_000:
        ends.append(last)
    last = i
ends.append(last)
seg = rows[ends[-2] + 1:ends[-1] + 1]


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN?(?:12_GLOBAL__N_1)?(\d+)", n)
    if m:
        k = int(m.group(1))
        rest = n[m.end():]
        tail = rest[k:]
        lay = re.search(r"Li(\d)E", tail)
        return rest[:k] + ("<%s>" % lay.group(1) if lay else "")
    return n.split("(")[0][:60]


ev = []
for i, r in enumerate(seg):
    ev.append((r["s"], 1, i))
    ev.append((r["e"], -1, i))
ev.sort()
live, t_prev = set(), ev[0][0]
tot = collections.Counter()
exposed = collections.Counter()
pair = collections.Counter()
slices = []  # (t0, t1, n_live, name if exposed)
for t, d, i in ev:
    if t > t_prev:
        n = len(live)
        tot[min(n, 2)] += t - t_prev
        if n == 1:
            nm = short(seg[next(iter(live))]["Kernel_Name"])
            exposed[nm] += t - t_prev
            slices.append((t_prev, t, nm))
        elif n >= 2:
            for j in live:
                pair[short(seg[j]["Kernel_Name"])] += t - t_prev
    if d == 1:
        live.add(i)
    else:
        live.discard(i)
    t_prev = t

t0 = seg[0]["s"]
route = [r for r in seg if re.search(r"xattn|route_aggregate|router_pool|saf_gate|meanpool", r["Kernel_Name"])]
marks = {}
if route:
    rs = sorted(route, key=lambda r: r["s"])
    gaps = [(b["s"] - a["e"], k) for k, (a, b) in enumerate(zip(rs, rs[1:]))]
    marks["routing_first_ms"] = (rs[0]["s"] - t0) / 1e6
    marks["routing_last_ms"] = (max(r["e"] for r in rs) - t0) / 1e6
wall = max(r["e"] for r in seg) - t0
sect = collections.Counter()
if route:
    a, b = rs[0]["s"], max(r["e"] for r in rs)
    for s, e, nm in slices:
        sect["encoders forward" if e <= a else "routing module fwd+bwd, head, loss" if s < b else "encoders backward + optimiser"] += e - s
out = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " "), "source": sys.argv[1].split("/")[-1], "launches": len(seg),
       "wall_ms": wall / 1e6, "idle_ms": tot[0] / 1e6, "exposed_ms": tot[1] / 1e6, "overlapped_ms": tot[2] / 1e6, "marks": marks,
       "exposed_ms_by_section": {k: round(v / 1e6, 3) for k, v in sect.items()},
       "exposed_ms_by_kernel": {k: round(v / 1e6, 3) for k, v in exposed.most_common(30)},
       "overlapped_ms_by_kernel": {k: round(v / 1e6, 3) for k, v in pair.most_common(12)}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("wall_ms", "idle_ms", "exposed_ms", "overlapped_ms", "marks", "exposed_ms_by_section")}))
for k, v in list(out["exposed_ms_by_kernel"].items())[:30]:
    print("  %-60s %7.3f ms" % (k, v))

if len(sys.argv) > 3:  # optional: the launches of the last step as text (start us, duration us, queue, name), for reading a section by eye
    with open(sys.argv[3], "w") as f:
        for r in seg:
            f.write("%9.1f %7.1f q%s %s\n" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"], short(r["Kernel_Name"])))
