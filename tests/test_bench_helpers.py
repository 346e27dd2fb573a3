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
    # --launch auto: hipGraph replay for every family — a LunarLander step is two launches on one stream (single_launch), replayed
    # exactly as issued; only the multi-stream order of the largest populations stays eager
    assert bench.launch_mode("auto", "cartpole") == "graph" and bench.launch_mode("auto", "mountain_car_cont") == "graph"
    assert bench.launch_mode("auto", "lunar_lander", "single_launch") == "graph"
    assert bench.launch_mode("auto", "lunar_lander", "overlapped") == "eager"
    assert bench.launch_mode("graph", "lunar_lander", "overlapped") == "graph" and bench.launch_mode("eager", "cartpole") == "eager"


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    # `python bench.py --gpus 2` with WORLD_SIZE unset (how the driver calls N = 1; the reference is single-threaded and !Send,
    # lunar_lander.rs:240-249, so the multi-process split is this build's): bench.py starts torch.distributed.run as a CHILD process
    # before torch / HIP are touched; --dry-run-launch rehearses exactly that plumbing on the CPU (gloo) and prints one JSON line.
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-launch"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout          # ONE JSON line on stdout, from rank 0
    d = json.loads(lines[0])
    assert d["dry_run_launch"] is True and d["n_gpus"] == 2 and (d["steps"], d["warmup"]) == (3, 1)   # the same arguments reached the ranks
    assert sorted((x["RANK"], x["LOCAL_RANK"], x["WORLD_SIZE"]) for x in d["ranks"]) == [(0, 0, 2), (1, 1, 2)]
    # a failing child's exit code is propagated (an unknown flag makes every rank exit 2 -> torchrun fails -> non-zero)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_package_asks_for_the_hardware_queues_before_hip_initialises():
    # LunarLander's three streams only overlap on distinct hardware queues; the package owns the knob
    import subprocess
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import modurl_gym_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "8", r.stderr[-1000:]
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '6'; import modurl_gym_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.stdout.strip() == "6"            # an embedder's own setting wins
    assert "GPU_MAX_HW_QUEUES" not in open(os.path.join(ROOT, "bench.py")).read().split("def parse")[1]   # bench.py no longer sets it itself


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
