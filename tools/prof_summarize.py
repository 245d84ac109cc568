#!/usr/bin/env python3
"""Summarise rocprofv3 CSV outputs (kernel trace stats + PMC passes) into small text/JSON files.
Usage: prof_summarize.py <prof_dir> <out_prefix>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof, out = sys.argv[1], sys.argv[2]
summary = {}
# kernel trace: per-kernel count / avg / total duration
for f in glob.glob(os.path.join(prof, "trace", "**", "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(lambda: [0, 0.0])
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = agg[r["Kernel_Name"]]
        a[0] += 1; a[1] += d
    tot = sum(a[1] for a in agg.values())
    summary["kernel_trace"] = [
        {"kernel": k[:100], "calls": a[0], "total_ms": a[1] / 1e6, "avg_us": a[1] / a[0] / 1e3, "pct": 100 * a[1] / tot}
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])][:12]
    if rows:
        r0 = sorted((r for r in rows if "ss_env_kernel" in r["Kernel_Name"]), key=lambda r: float(r["Start_Timestamp"]))
        # bench.py launch sequence: reset(all), then per step: ss_step launch, masked autoreset launch
        if len(r0) >= 3:
            dur = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 for r in r0]
            st, rs = dur[1::2], dur[2::2]
            if rs and sum(rs) / len(rs) > 0.3 * sum(st) / len(st):   # fused autoreset: every launch after the first is a step
                st, rs = dur[1:], []
            summary["launches_alternate_step_reset"] = bool(rs)
            summary["step_launches"] = {"n": len(st), "avg_us": sum(st) / len(st), "min_us": min(st), "max_us": max(st)}
            if rs:
                summary["autoreset_launches"] = {"n": len(rs), "avg_us": sum(rs) / len(rs)}
            summary["initial_reset_us"] = dur[0]
        if r0:
            summary["step_kernel_resources"] = {k: r0[0].get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
# PMC passes: per-kernel mean of each counter (sum over dispatch dims as reported)
for d in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        rows = [r for r in csv.DictReader(open(f)) if "ss_env_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        alt = summary.get("launches_alternate_step_reset", True)
        kind = {d: ("initial_reset" if i == 0 else ("step" if (i % 2 == 1 or not alt) else "autoreset")) for i, d in enumerate(ids)}
        for r in rows:
            a = agg[kind[int(r["Dispatch_Id"])] + " launches of " + r["Kernel_Name"][:40]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
        for kname, cs in agg.items():
            summary.setdefault("pmc", {}).setdefault(kname, {}).update({c: {"dispatches": a[0], "mean_per_dispatch": a[1] / a[0]} for c, a in cs.items()})
# HBM traffic of one step launch for bench.py's roofline.traffic: FETCH_SIZE (KiB, x2 on gfx950 per
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KiB), separate PMC passes, mean over the step launches
for kname, cs in summary.get("pmc", {}).items():
    if kname.startswith("step ") and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, wr = cs["FETCH_SIZE"]["mean_per_dispatch"], cs["WRITE_SIZE"]["mean_per_dispatch"]
        summary["hbm_traffic"] = {"workload": os.environ.get("WORKLOAD", "smpl"), "fetch_size_kib_raw": f, "write_size_kib": wr,
                                  "bytes_per_step_launch": (2.0 * f + wr) * 1024.0,
                                  "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), mean over step launches"}
        json.dump(summary["hbm_traffic"], open(out + "_hbm_traffic.json", "w"), indent=1)
json.dump(summary, open(out + ".json", "w"), indent=1)
with open(out + ".txt", "w") as fo:
    for e in summary.get("kernel_trace", []):
        fo.write(f"{e['pct']:6.2f}%  calls={e['calls']:5d}  avg={e['avg_us']:10.1f} us  total={e['total_ms']:9.2f} ms  {e['kernel']}\n")
    fo.write(json.dumps(summary.get("step_kernel_resources", {})) + "\n")
    for key in ("step_launches", "autoreset_launches", "initial_reset_us"):
        if key in summary:
            fo.write(f"{key}: {json.dumps(summary[key])}\n")
    for kname, cs in summary.get("pmc", {}).items():
        fo.write(f"PMC {kname}\n")
        for c, a in sorted(cs.items()):
            fo.write(f"   {c:28s} mean/dispatch={a['mean_per_dispatch']:.6g}  (n={a['dispatches']})\n")
print(open(out + ".txt").read())
