"""CPU-side checks of bench.py's bookkeeping (no GPU): counter records are reported only for the kernel sources they were
measured on, the CPU-baseline thread count comes from affinity / cgroup, and the shard plan the bench uses."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_counter_records_are_dropped_when_the_kernel_sources_changed(tmp_path, monkeypatch):
    recs = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key in ("cartpole:1048576", "cartpole:33554432", "mountain_car:1048576", "lunar_lander:262144"):
        assert key in recs and len(recs[key]["src_sha16"]) == 16 and recs[key]["src_files"]
    # the committed records match the committed sources (a kernel edit without a new profile fails here, on purpose:
    # re-run tools/profile_pmc.sh + tools/make_pmc_records.py, or accept that bench.py will omit `traffic`)
    cp = bench.pmc_record("cartpole:1048576")
    if cp is not None:
        assert abs(cp["hbm_bytes_per_launch"] / (50 * 1048576) - 1.0) < 0.02   # within 2 % of the algorithmic bytes
    # a record whose hash does not match is not reported
    fake = dict(recs)
    fake["cartpole:1048576"] = dict(recs["cartpole:1048576"], src_sha16="0" * 16)
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    (root / "profiles" / "pmc_traffic.json").write_text(json.dumps(fake))
    os.symlink(os.path.join(ROOT, "modurl_gym_amd"), root / "modurl_gym_amd")
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.pmc_record("cartpole:1048576") is None
    assert bench.pmc_record("no-such-key") is None


def test_source_hash_ignores_comments_but_not_code():
    from modurl_gym_amd._srchash import code_only
    a = "int f(int x) {  // adds one\n    /* really\n       it does */\n    return x + 1;\n}\n\n"
    b = "int f(int x) {\n    return x + 1;   // a different remark\n}\n"
    c = "int f(int x) {\n    return x + 2;\n}\n"
    assert code_only(a) == code_only(b) != code_only(c)


def test_launch_mode_per_family():
    # --launch auto: hipGraph replay for the launch-bound families, eager launches for LunarLander
    assert bench.launch_mode("auto", "cartpole") == "graph" and bench.launch_mode("auto", "mountain_car_cont") == "graph"
    assert bench.launch_mode("auto", "lunar_lander") == "eager"
    assert bench.launch_mode("graph", "lunar_lander") == "graph" and bench.launch_mode("eager", "cartpole") == "eager"


def test_host_cores_follows_affinity_and_override(monkeypatch):
    n = bench.host_cores()
    assert 1 <= n <= 64 and n <= len(os.sched_getaffinity(0))
    monkeypatch.setenv("MGYM_BENCH_CORES", "1")
    assert bench.host_cores() == 1


def test_lunar_roofline_fields():
    r = bench.lunar_roofline(262144, 1.95e-3)
    assert r["bound"] == "valu" and r["peak"] == 157.3 and r["unit"] == "TFLOP/s"
    assert r["hbm_for_reference"]["frac"] < 0.05
    if r["achieved"] is not None:   # counters present and current
        assert 0.001 < r["frac"] < 0.2 and r["frac"] < r["frac_if_all_64_lanes_counted"]
