#!/usr/bin/env python3
"""Aggregates the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the counters do not fit one
pass on gfx950) of `bench.py --steps 2 --warmup 1` into per-kernel-family HBM traffic per d2r_gemm launch.

  python profiles/make_pmc_summary.py gpurun_out/pmc/fetch/f_counter_collection.csv \
         gpurun_out/pmc/write/w_counter_collection.csv profiles/pmc_traffic_r01.json

Units / corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request of a wide coalesced read, i.e. HALF the bytes -> doubled here; WRITE_SIZE is exact for
16-B-per-lane stores.  A d2r_gemm launch of the TN (weight-gradient) family is the GEMM kernel plus, when split-K is
used, its splitk_reduce_kernel, listed as its own row (gemm_splitk_reduce)."""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name: str):
    """Kernel name -> (row name, counts as a launch).  GEMM rows are per KERNEL, named as bench.py names them:
    gemm_<dtype>_<layout>[_grouped]_<variant>, variant = ldsdma128x64 | ldsdma128x128w4 | ldsdma128x128w8 (+p: pipelined K-loop) |
    ldsdma128x128 (grouped weight gradients) | tiles (register-staged generic kernel; tiles64x64 when grouped) | skinny.
    rocprofv3 leaves names with a __bf16 / _Float16 template argument mangled (_Z16gemm_glds_kernelIDF16bLi2ELi128E...) or
    demangles them wrongly ("<bool _Accum, int, E, 64, ...>" for <__bf16, 1 (NN), 64, ...>)."""
    lay = ("NT", "NN", "TN")

    def glds(layout, bn, nwn, pipe, wgrad):
        if wgrad:
            return "gemm_bf16_TN_grouped_ldsdma128x128"
        v = "ldsdma128x64" if bn == 64 else ("ldsdma128x128w4" if nwn == 2 else "ldsdma128x128w8")
        return "gemm_bf16_%s_%s%s" % (lay[layout], v, "p" if pipe else "")

    if "splitk_reduce" in name:
        return "gemm_splitk_reduce", True
    # round 4: the 256 x 256 deep-pipelined kernels (gemm8.hip) and the grouped launches of the 128-wide kernel (same tiles: same row)
    m = re.search(r"gemm8_wgrad_kernelIDF16([b_])", name)
    if m:
        return "gemm_%s_TN_grouped_ldsdma256x256" % ("bf16" if m.group(1) == "b" else "f16"), True
    m = re.search(r"gemm8_fwd_kernelIDF16([b_])Li(\d)E", name)
    if m:
        return "gemm_%s_%s_ldsdma256x256" % ("bf16" if m.group(1) == "b" else "f16", lay[int(m.group(2))]), True
    m = re.search(r"gemm_glds_group_kernelIDF16([b_])Li(\d)E", name)
    if m:
        return "gemm_%s_%s_ldsdma128x128w8" % ("bf16" if m.group(1) == "b" else "f16", lay[int(m.group(2))]), True
    # round 3: <E, LAYOUT, BN, NWN, PIPE, MODE> (MODE 1 grouped weight gradients, 2 grouped + batched 16-bit output); E = bf16 (b) / fp16 (_)
    m = re.search(r"gemm_glds_kernelIDF16([b_])Li(\d)ELi(\d+)ELi(\d)ELi(\d)ELi(\d)E", name)
    if m:
        dt = "bf16" if m.group(1) == "b" else "f16"
        mode = int(m.group(6))
        if mode == 1:
            return "gemm_%s_TN_grouped_ldsdma128x128" % dt, True
        if mode == 2:
            return "gemm_%s_TN_grouped_batched16" % dt, True
        return glds(int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), False).replace("bf16", dt), True
    m = re.search(r"gemm_skinny_h16_kernelIDF16([b_])Li(\d)E", name)
    if m:
        return "gemm_%s_%s_skinny" % ("bf16" if m.group(1) == "b" else "f16", lay[int(m.group(2))]), True
    m = re.search(r"gemm_kernelIDF16_Li(\d)ELi\d+ELi\d+ELi\dELi\dELi\dELb([01])E", name)
    if m:
        return "gemm_f16_" + lay[int(m.group(1))] + ("_grouped_tiles64x64" if m.group(2) == "1" else "_tiles"), True
    # third-generation single-head attention: the backward op = query-side kernel + product kernel (one launch of the op each)
    if "xattn3_fwd" in name:
        return "xattn_core_fwd", True
    if "xattn3_bwd" in name:
        return "xattn_core_bwd", True
    if "xattn3_dkv" in name:
        return "xattn_core_bwd", False
    if "nonfinite_kernel" in name:
        return "d2r_grad_nonfinite", True
    m = re.search(r"gemm_glds_kernelIDF16[b_]Li(\d)ELi(\d+)ELi(\d)ELi(\d)ELb([01])E", name)  # <E, LAYOUT, BN, NWN, PIPE, WGRAD>
    if m:
        return glds(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5) == "1"), True
    m = re.search(r"gemm_glds_kernel<[^,]*, (\d), (\d+), (\d), (\d), (true|false)>", name)
    if m:
        return glds(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5) == "true"), True
    m = re.search(r"gemm_glds_kernel<bool _Accum, int, E, (\d+), (\d), (\d), (true|false)>", name)  # garbled <__bf16, 1 (NN), ...>
    if m:
        return glds(1, int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true"), True
    m = re.search(r"gemm_glds_kernel<(\d), (\d+)", name)  # (round-1 spelling without the element type)
    if m:
        return "gemm_bf16_%s_ldsdma128x%s" % (lay[int(m.group(1))], m.group(2)), True
    m = re.search(r"gemm_kernelIDF16[b_]Li(\d)ELi\d+ELi\d+ELi\dELi\dELi\dELb([01])E", name)  # <T, LAYOUT, BM, BN, WM, WN, NBUF, GROUPED>
    if m:
        return "gemm_bf16_" + lay[int(m.group(1))] + ("_grouped_tiles64x64" if m.group(2) == "1" else "_tiles"), True
    if "gemm_kernel<bool _Accum" in name:
        return "gemm_bf16_NN_tiles", True
    m = re.search(r"gemm_kernel<float, (\d), \d+, \d+, \d, \d, \d, (true|false)>", name)
    if m:
        return "gemm_f32_" + lay[int(m.group(1))] + ("_grouped_tiles64x64" if m.group(2) == "true" else "_tiles"), True
    m = re.search(r"gemm_skinny_f32_kernel(?:ILi|<)(\d)", name)
    if m:
        return "gemm_f32_%s_skinny" % lay[int(m.group(1))], True
    for key, fam in (("mha_long", "mha_core_long"), ("mha_fwd", "mha_core_fwd"), ("mha_bwd", "mha_core_bwd"), ("adamw", "d2r_adamw_step"),
                     ("xattn2_fwd", "xattn_core_fwd"), ("xattn_bwd", "xattn_core_bwd"), ("xattn_fwd", "xattn_core_fwd"),
                     ("agg_fwd", "route_aggregate_fwd"), ("agg_bwd", "route_aggregate_bwd"), ("meanpool_fwd", "d2r_meanpool_fwd"),
                     ("layernorm_bwd", "d2r_layernorm_bwd"), ("ln_sum_partials", "d2r_layernorm_bwd_sum"), ("layernorm_fwd", "d2r_layernorm_fwd")):
        if key in name:
            return fam, True
    return None, False


def load(path, counter):
    tot, launches = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            fam, counts = family(row["Kernel_Name"])
            if fam is None:
                continue
            tot[fam] += float(row["Counter_Value"]) * 1024.0
            launches[fam] += 1 if counts else 0
    return tot, launches


fetch, n_f = load(sys.argv[1], "FETCH_SIZE")
write, n_w = load(sys.argv[2], "WRITE_SIZE")
# steps of the profiled command: 1 warm-up + 2 timed + 4 of the fwd+bwd-only leg (bench.py --steps 2 --warmup 1) = 7
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 7
out = {"_doc": "HBM bytes from rocprofv3 --pmc (FETCH_SIZE x2 on gfx950, WRITE_SIZE exact); bench.py --steps 2 --warmup 1, C2 "
               "workload (%d steps in the run).  *_per_launch: per GEMM kernel of the REAL step (a grouped weight-gradient launch "
               "holds 6-16 problems); hbm_bytes_per_step: family total per training step - bench.py divides it by the launches "
               "per step of its per-kernel pass to compare with the algorithmic bytes per launch." % STEPS,
       "_steps": STEPS}
for fam in sorted(fetch):
    n = max(n_f[fam], 1)
    rd, wr = 2.0 * fetch[fam] / n, write.get(fam, 0.0) / max(n_w.get(fam, 0), 1)
    out[fam] = {"launches_profiled": n_f[fam], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                "hbm_bytes_per_launch": round(rd + wr), "hbm_bytes_per_step": round((rd + wr) * n / STEPS)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
