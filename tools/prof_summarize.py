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
# PROF_SKIP_STEPS: step launches after the initial reset that are warm-up and do not count (getup / imitation: the profiled launches
# must be the TIMED workload — Fall resets and re-initialisations running inside them — not the first launches after a fresh reset)
SKIP = int(os.environ.get("PROF_SKIP_STEPS", "0"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smplsim_amd._lib import source_hash   # noqa: E402  (the tree the counters were taken on: bench.py refuses a summary of another tree)
SRC_HASH = source_hash()
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
            if SKIP and not rs and len(st) > SKIP:
                st = st[SKIP:]
            summary["skipped_warmup_step_launches"] = SKIP
            summary["launches_alternate_step_reset"] = bool(rs)
            summary["step_launches"] = {"n": len(st), "avg_us": sum(st) / len(st), "min_us": min(st), "max_us": max(st)}
            if rs:
                summary["autoreset_launches"] = {"n": len(rs), "avg_us": sum(rs) / len(rs)}
            summary["initial_reset_us"] = dur[0]
        if r0:
            first = r0[0]
            summary["step_kernel_resources"] = {k: first.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
            summary["step_kernel_resources"]["Workgroup_Size"] = first.get("Workgroup_Size") or first.get("Workgroup_Size_X")
            summary["step_kernel_resources"]["Grid_Size"] = first.get("Grid_Size") or first.get("Grid_Size_X")
            summary["kernel_trace_columns"] = list(first.keys())
# PMC passes: per-kernel mean of each counter (sum over dispatch dims as reported)
for d in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        rows = [r for r in csv.DictReader(open(f)) if "ss_env_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        alt = summary.get("launches_alternate_step_reset", True)
        kind = {d: ("initial_reset" if i == 0 else ("warmup" if (not alt and i <= SKIP) else ("step" if (i % 2 == 1 or not alt) else "autoreset"))) for i, d in enumerate(ids)}
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
# limiter figures of the step launch for bench.py's roofline object (profiles/pmc_summary_<workload>.json):
#   valu_issue_frac   = 2 cycles x SQ_INSTS_VALU / (SIMDs x launch cycles): a wave64 VALU instruction occupies its SIMD-32 for 2
#                       cycles (MI355X_MICROARCH.md); launch cycles = GRBM_GUI_ACTIVE / 8 XCDs
#   lds_wait_frac     = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES; lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
#   wave_slot_occupancy = 4 x SQ_WAVE_CYCLES (quad-cycles) / (launched waves x launch cycles): how much of the launch the
#                       persistent waves are alive (the rest is the straggler tail)
for kname, cs in summary.get("pmc", {}).items():
    if not kname.startswith("step "):
        continue
    g = lambda c: cs[c]["mean_per_dispatch"] if c in cs else None
    res = summary.get("step_kernel_resources", {})
    lim = {"workload": os.environ.get("WORKLOAD", "smpl"), "envs_per_gpu": int(os.environ.get("ENVS_PER_GPU", "4096")), "src_hash": SRC_HASH,
           "profile": os.environ.get("TAG", "prof"), "step_launch_avg_us": summary.get("step_launches", {}).get("avg_us")}
    cyc = g("GRBM_GUI_ACTIVE") / 8.0 if g("GRBM_GUI_ACTIVE") else None
    n_simd = 1024
    if cyc and g("SQ_INSTS_VALU"):
        lim["launch_cycles"] = cyc
        lim["valu_issue_frac"] = 2.0 * g("SQ_INSTS_VALU") / (n_simd * cyc)
        lim["valu_wave_instructions"] = g("SQ_INSTS_VALU")
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_WAIT_INST_LDS") is not None:
            lim["lds_wait_frac"] = g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")
        if g("SQ_ACTIVE_INST_ANY") is not None:
            lim["wave_active_frac"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
        if cyc and g("SQ_WAVES"):
            lim["wave_slot_occupancy"] = 4.0 * g("SQ_WAVE_CYCLES") / (g("SQ_WAVES") * cyc)
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        lim["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if "hbm_traffic" in summary:
        lim["traffic"] = summary["hbm_traffic"]["bytes_per_step_launch"]
        lim["traffic_method"] = summary["hbm_traffic"]["method"]
    # (LDS_Block_Size is not reported: the kernel's LDS is dynamic and the trace field shows 0.  VGPR_Count: rocprofv3 decodes the
    # kernel descriptor's granulated count with the pre-gfx90a granule of 4 — (20 + 1) x 4 = 84 for this kernel; on gfx950 the
    # granule is 8: (20 + 1) x 8 = 168 = .amdhsa_next_free_vgpr = hipFuncGetAttributes().numRegs = ss_launch_info's value)
    for k_src, k_dst in (("Scratch_Size", "scratch_bytes_per_lane"), ("VGPR_Count", "vgprs_trace_field_granule4"), ("Workgroup_Size", "workgroup_size")):
        try:
            lim[k_dst] = int(res.get(k_src))
        except (TypeError, ValueError):
            pass
    if lim.get("vgprs_trace_field_granule4"):
        lim["vgprs"] = 2 * lim["vgprs_trace_field_granule4"]
    if lim.get("workgroup_size"):
        lim["waves_per_cu"] = lim["workgroup_size"] // 64
    summary["limiters"] = lim
    json.dump(lim, open(out + "_pmc_summary.json", "w"), indent=1)
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
