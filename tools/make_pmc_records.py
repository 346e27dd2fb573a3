#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from tools/profile_pmc.sh summaries (gpurun_out/pmc_<tag>/summary.json).

usage: tools/make_pmc_records.py <round-tag> key=tag [key=tag ...]      e.g.  r02 cartpole:1048576=r02_cp_final
Each record names the kernel sources it was measured on (sha256 prefix over the listed files): bench.py reports a
record only while those files are unchanged, so a kernel edit silently retires the counters taken on the old kernel."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "modurl_gym_amd", "csrc")
SRC = {"cartpole": ["cartpole.hip", "cartpole_step.h", "cartpole_math.h", "mgym_math.h", "philox.h", "common.h"],
       "mountain_car": ["mountain_car.hip", "mgym_math.h", "philox.h", "common.h"],
       "lunar_lander": ["lunar_lander.hip", "ll_roll.h", "ll_env.h", "ll_free.h", "ll_world.h", "ll_b2.h", "mgym_math.h", "philox.h", "common.h"]}


def sha16(files):
    sys.path.insert(0, ROOT)
    from modurl_gym_amd._srchash import kernel_source_sha16
    return kernel_source_sha16(files)


def main():
    rnd = sys.argv[1]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}   # records of the other workloads stay
    for kv in sys.argv[2:]:
        key, tag = kv.split("=")
        fam = key.split(":")[0].replace("_cont", "").split("_rollout")[0]
        summ = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "summary.json")))["kernels"]
        files = SRC[fam]
        rec = {"src_files": files, "src_sha16": sha16(files),
               "source": f"profiles/{rnd}_*/ ({tag}): rocprofv3 --pmc in separate passes (tools/profile_pmc.sh), per-launch averages over the "
                         f"last quarter of the dispatches; FETCH_SIZE x2 (gfx950 correction for wide coalesced reads), WRITE_SIZE exact"}
        if fam in ("cartpole", "mountain_car"):
            name = next(k for k in summ if ("cartpole_step_kernel" in k or "mountaincar_step" in k) and "pmc" in summ[k])
            d = summ[name]["derived"]
            rec.update(kernel=name, hbm_bytes_per_launch=d["fetch_bytes_x2"] + d["write_bytes"], fetch_bytes_x2=d["fetch_bytes_x2"],
                       write_bytes=d["write_bytes"], kernel_avg_ns=summ[name].get("avg_ns_steady"),
                       valu_insts_per_wave=d.get("valu_insts_per_wave"))
        else:
            lanes64 = active = hbm = 0.0
            per = {}
            steps = max([v.get("calls", 0) for k, v in summ.items() if "ll_free_kernel" in k or "ll_step_kernel" in k] + [1])  # one step launch (or one free-flight launch) per step
            if "_rollout" in key:   # one persistent launch per K steps: per-step figures = per-launch figures / K
                kroll = int(key.split(":")[0].split("_rollout")[1])
                steps = max([v.get("calls", 0) for k, v in summ.items() if "ll_rollout_kernel" in k] + [1]) * kroll
            for k, v in summ.items():
                if "mgym::ll_" not in k or "pmc" not in v or "f32_flop_per_launch_lanes64" not in v["derived"] or v.get("calls", 0) < 10:
                    continue   # (one-shot kernels — the initial reset of the whole population — are not part of a step)
                d = v["derived"]
                per_step = v.get("calls", steps) / steps   # the contact kernel runs twice per step in the overlapped order
                f64 = d["f32_flop_per_launch_lanes64"] * per_step
                act = f64 * d.get("mean_active_lanes_per_valu_inst", 64.0) / 64.0
                lanes64 += f64
                active += act
                hbm += (d.get("fetch_bytes_x2", 0.0) + d.get("write_bytes", 0.0)) * per_step
                per[k.split("(")[0].replace("void mgym::", "")] = {"avg_ns": v.get("avg_ns_steady"), "launches_per_step": per_step, "f32_flop_lanes64": f64, "f32_flop_active_lanes": act,
                                                                   "mean_active_lanes": d.get("mean_active_lanes_per_valu_inst"),
                                                                   "valu_busy_share_of_wave_cycles": d.get("valu_active_share_of_wave_cycles"),
                                                                   "wait_any_share_of_wave_cycles": (v["pmc"]["SQ_WAIT_ANY"] / v["pmc"]["SQ_WAVE_CYCLES"]) if v["pmc"].get("SQ_WAVE_CYCLES") else None,
                                                                   "vgprs": v.get("launch", {}).get("vgpr"), "scratch_bytes_per_lane": v.get("launch", {}).get("scratch"),
                                                                   "lds_bytes_per_block": v.get("launch", {}).get("lds"),
                                                                   "waves_per_simd": (1 if (v.get("launch", {}).get("vgpr") or 0) > 128 else 2 if (v.get("launch", {}).get("vgpr") or 0) > 96 else None)}
            rec.update(f32_flop_per_step_lanes64=lanes64, f32_flop_per_step_active_lanes=active, hbm_bytes_per_step=hbm, kernels=per,
                       note="per step (launches_per_step x the per-launch average; the kernels of one step overlap in time); flop = 64 x (2 x SQ_INSTS_VALU_FMA_F32 + MUL_F32 + ADD_F32), weighted by "
                            "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (mean active lanes per VALU instruction) for the active-lane figure")
        out[key] = rec
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a not in ("kernels", "src_files", "source", "note")} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
