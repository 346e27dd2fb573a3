#!/usr/bin/env python3
"""Summarise a tools/profile_pmc.sh output directory.

Per kernel: average duration (kernel trace) and the per-launch average of every collected counter.
Only the steady-state dispatches are averaged: the last `tail` fraction of each kernel's dispatches
(LunarLander needs hundreds of steps to reach its flight/contact mix).  FETCH_SIZE gets the gfx950
correction of MI355X_MICROARCH.md (x2 for wide coalesced reads; reported in KiB), WRITE_SIZE is exact.
Derived per kernel: VALU instructions per wave, VALU busy share, mean active lanes per VALU
instruction, f32 flop/s from the counted FMA/MUL/ADD instructions."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
res = {"kernels": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        res["kernels"].setdefault(row["Name"], {})["avg_ns_all"] = float(row["AverageNs"])
        res["kernels"][row["Name"]]["calls"] = int(row["Calls"])
# steady-state duration from the per-dispatch trace when it is still there
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    per = defaultdict(list)
    for row in csv.DictReader(open(f)):
        per[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in per.items():
        t = v[int(len(v) * (1 - tail)):]
        res["kernels"].setdefault(k, {})["avg_ns_steady"] = sum(t) / max(len(t), 1)
        res["kernels"][k]["steady_dispatches"] = len(t)
for grp in ("sqA", "sqB", "sqC", "fetch", "write"):
    per = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(out, grp, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = dict(grid=int(row["Grid_Size"]), wg=int(row["Workgroup_Size"]), vgpr=int(row["VGPR_Count"]),
                           sgpr=int(row["SGPR_Count"]), scratch=int(row["Scratch_Size"]), lds=int(row["LDS_Block_Size"]))
    for k, ctrs in per.items():
        rec = res["kernels"].setdefault(k, {})
        rec.setdefault("launch", meta[k])
        for c, v in ctrs.items():
            t = v[int(len(v) * (1 - tail)):]
            rec.setdefault("pmc", {})[c] = sum(t) / max(len(t), 1)
CLK = 2.4e9
for k, rec in res["kernels"].items():
    p = rec.get("pmc", {})
    d = {}
    if "SQ_WAVES" in p and p["SQ_WAVES"]:
        d["valu_insts_per_wave"] = p.get("SQ_INSTS_VALU", 0) / p["SQ_WAVES"]
        d["salu_insts_per_wave"] = p.get("SQ_INSTS_SALU", 0) / p["SQ_WAVES"]
    if p.get("SQ_ACTIVE_INST_VALU"):
        d["mean_active_lanes_per_valu_inst"] = p.get("SQ_THREAD_CYCLES_VALU", 0) / p["SQ_ACTIVE_INST_VALU"]
    if p.get("SQ_WAVE_CYCLES"):
        d["valu_active_share_of_wave_cycles"] = p.get("SQ_ACTIVE_INST_VALU", 0) / p["SQ_WAVE_CYCLES"]
        d["wait_inst_any_share"] = p.get("SQ_WAIT_INST_ANY", 0) / p["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in p:
        d["fetch_bytes_x2"] = p["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in p:
        d["write_bytes"] = p["WRITE_SIZE"] * 1024
    ns = rec.get("avg_ns_steady") or rec.get("avg_ns_all")
    if ns and "SQ_INSTS_VALU_FMA_F32" in p:
        # counters count wave-level instructions; x64 lanes (upper bound: exec-masked lanes included)
        flop = 64 * (2 * p["SQ_INSTS_VALU_FMA_F32"] + p.get("SQ_INSTS_VALU_MUL_F32", 0) + p.get("SQ_INSTS_VALU_ADD_F32", 0))
        d["f32_flop_per_launch_lanes64"] = flop
        d["f32_tflops_lanes64"] = flop / ns / 1e3
    if ns and p.get("SQ_INSTS_VALU"):
        d["valu_inst_issue_rate_G_per_s"] = p["SQ_INSTS_VALU"] / ns
    rec["derived"] = d
for k, rec in sorted(res["kernels"].items(), key=lambda kv: -(kv[1].get("avg_ns_all", 0) * kv[1].get("calls", 1))):
    if "pmc" not in rec and rec.get("calls", 0) < 4:
        continue
    print(f"== {k[:120]}")
    print("   ", {a: rec[a] for a in ("calls", "avg_ns_all", "avg_ns_steady", "steady_dispatches", "launch") if a in rec})
    for c, v in sorted(rec.get("pmc", {}).items()):
        print(f"    {c:32s} {v:16.1f}")
    for c, v in rec.get("derived", {}).items():
        print(f"    -> {c:40s} {v:.4g}")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
