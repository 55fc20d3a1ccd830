"""GPU-busy analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals vs wall time, per stream/queue, and the
largest idle gaps.  Usage: python tests/probes/trace_busy.py <kernel_trace.csv> [t_skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip  # analyse the steady-state tail only
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n, q in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
print(f"window {(t1 - t0) / 1e6:.2f} ms, {len(ev)} kernels; GPU busy (union) {busy / 1e6:.2f} ms = {100 * busy / (t1 - t0):.1f} %; "
      f"sum of kernel durations {tot / 1e6:.2f} ms (overlap factor {tot / busy:.2f})")
gaps.sort(reverse=True)
print("idle total %.2f ms in %d gaps; gaps > 20 us: %d (%.2f ms); > 5 us: %d (%.2f ms)" % (
    sum(g for g, _ in gaps) / 1e6, len(gaps), sum(1 for g, _ in gaps if g > 20000), sum(g for g, _ in gaps if g > 20000) / 1e6,
    sum(1 for g, _ in gaps if g > 5000), sum(g for g, _ in gaps if g > 5000) / 1e6))
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:12]])
byq = collections.Counter()
for s, e, n, q in ev:
    byq[q] += e - s
print("busy per queue (ms):", {k: round(v / 1e6, 2) for k, v in byq.items()})
