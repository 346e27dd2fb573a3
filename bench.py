#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched step()/reset() hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE).
A "step" is one pass of the hot path over one batch: every environment of the rank's shard
advances by one Gym::step, and every episode that finished is reset (so the population stays
in-distribution).  Default workload = BASELINE.json configs[1]: CartPole-v1, 1 048 576 envs per
GPU (weak scaling: rank r owns global env ids [r*n, (r+1)*n), no data-path collective).
W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + torch.cuda.synchronize()
on both sides; MAX over ranks; rank 0 prints ONE JSON line.

Inputs are synthetic and already resident in HBM when the timed region starts: states come from
the engine's own reset, actions from a ring of 16 pre-generated uniform columns.

Besides the contract fields the line carries
  roofline     — dominant kernel (the step kernel): algorithmic bytes per launch / average launch
                 duration from HIP events on the launch stream over the timed region, vs 8 TB/s.
  cpu_baseline — the CPU oracle (C restatement of the Rust reference; the reference itself cannot
                 be built here) timed on this box's host cores on a bounded sample, rank 0, N=1 only.
  extra        — the other BASELINE configs measured after the timed region (not the headline).
"""
import argparse
import json
import os
import sys
import time

# (The hardware-queue setting the LunarLander helper streams need — GPU_MAX_HW_QUEUES — is owned by the package:
# modurl_gym_amd/__init__.py sets it before anything initialises HIP; bench.py imports the package before torch touches the GPU.)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RING = 16  # pre-generated action columns (actions are not cache-resident constants)
HBM_PEAK = 8.0e12
ALG_BYTES = {"cartpole": 50, "mountain_car": 26, "mountain_car_cont": 26}  # SURVEY §8d, DESIGN.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cartpole",
                    choices=["cartpole", "mountain_car", "mountain_car_cont", "lunar_lander", "mixed"])
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default: BASELINE size)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="auto: hipGraph replay (every family; LunarLander populations that use the multi-stream launch order stay eager)")
    ap.add_argument("--reset", default="fused", choices=["fused", "separate"],
                    help="fused: auto-reset inside the step kernel; separate: step + mgym_reset_done launch")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --envs (default: BASELINE size) per GPU; strong: --total-envs fixed for the node, index-sharded over the ranks")
    ap.add_argument("--total-envs", type=int, default=8 << 20, help="node total for --scaling strong (default 8 388 608, BASELINE configs[4])")
    ap.add_argument("--ll-rollout", type=int, default=0, metavar="K",
                    help="LunarLander: step through mgym_rollout, K steps per persistent launch (0: mgym_step); --steps / --warmup are rounded down to multiples of K")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="rehearse the multi-process launch without a GPU: every rank joins a gloo group, rank 0 prints one JSON line "
                         "listing the RANK/LOCAL_RANK/WORLD_SIZE each rank saw")
    return ap.parse_args()


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.

    The reference is single-threaded and `!Send` (src/box_2d/lunar_lander.rs:240-249), so the split over GPUs is this build's to
    own.  Runs BEFORE torch is imported or HIP is touched, starts `python -m torch.distributed.run` as a CHILD process (never an
    exec: a process that has initialised the GPU must not be replaced), lets the child's rank 0 write the single JSON line to the
    stdout we share with it, and returns the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as s:   # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")   # (torchrun sets it anyway and says so on stderr)
    return subprocess.run(cmd, env=env).returncode


def dry_run_launch(args, rank, local_rank, world, real_stdout):
    """--dry-run-launch: the launch plumbing without a GPU (gloo): every rank reports the env it was given."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.tensor([rank, local_rank, world, int(os.environ.get("WORLD_SIZE", "-1"))], dtype=torch.int64)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        line = {"dry_run_launch": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ranks": [{"RANK": int(t[0]), "LOCAL_RANK": int(t[1]), "WORLD_SIZE": int(t[3])} for t in seen]}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())


def launch_mode(requested, family, launch_order="single_launch"):
    """`--launch auto`: hipGraph replay for every family.  CartPole / MountainCar steps are ~10 us (launch-bound); a LunarLander step is
    two launches on the caller's stream (`launch_order` of mgym_get_info = single_launch), which a graph replays exactly as eager
    launches run (profiles/r03_lunarlander/launch_modes.txt).  Only populations that select the multi-stream order (64-lane contact
    blocks, from 376 832 envs) stay eager: a graph executor serialises its side branches."""
    if requested != "auto":
        return requested
    return "eager" if (family == "lunar_lander" and launch_order != "single_launch") else "graph"


class Stepper:
    """One env family on this rank: device buffers, an action ring and a (graph-captured) step."""

    def __init__(self, mg, torch, kind_name, n, device, seed, base, stream, reset_mode, launch, roll_k=0):
        kinds = {"cartpole": (mg.CARTPOLE, 2), "mountain_car": (mg.MOUNTAINCAR, 3),
                 "mountain_car_cont": (mg.MOUNTAINCAR_CONT, 0), "lunar_lander": (mg.LUNARLANDER, 4)}
        kind, nact = kinds[kind_name]
        self.name, self.n, self.reset_mode = kind_name, n, reset_mode
        extra = dict(enable_wind=True) if kind_name == "lunar_lander" else {}
        self.env = mg.VecEnv(kind, n, device=device, seed=seed, env_id_base=base, auto_reset=(reset_mode == "fused"),
                             stream=stream.cuda_stream, **extra)
        self.launch = launch_mode(launch, kind_name, self.env.info().get("launch_order", "") if kind_name == "lunar_lander" else "")
        g = torch.Generator(device=f"cuda:{device}")
        g.manual_seed(seed & 0x7FFFFFFF)
        if nact:
            self.actions = torch.randint(0, nact, (RING, n), generator=g, device=f"cuda:{device}", dtype=torch.int32)
        else:
            self.actions = torch.rand((RING, n), generator=g, device=f"cuda:{device}", dtype=torch.float32) * 2 - 1
        self.reward = torch.empty(n, device=f"cuda:{device}", dtype=torch.float32)
        self.done = torch.zeros(n, device=f"cuda:{device}", dtype=torch.uint8)
        self.trunc = torch.zeros(n, device=f"cuda:{device}", dtype=torch.uint8)
        # the tensors above were filled on torch's current stream; the engine launches on `stream`: order them
        stream.wait_stream(torch.cuda.current_stream(device))
        self.env.reset_device(None, None)
        self.graph = None
        self.tail_graphs = {}
        self.timed = None
        self.t = 0
        self.roll_k = roll_k if kind_name == "lunar_lander" else 0
        if self.roll_k:   # mgym_rollout: a [K][n] action table and [K][n] outputs, one persistent launch per K steps
            self.launch = "rollout"
            self.roll_actions = torch.randint(0, nact, (self.roll_k, n), generator=g, device=f"cuda:{device}", dtype=torch.int32)
            self.roll_out = (torch.empty((self.roll_k, n), device=f"cuda:{device}", dtype=torch.float32),
                             torch.zeros((self.roll_k, n), device=f"cuda:{device}", dtype=torch.uint8), torch.zeros((self.roll_k, n), device=f"cuda:{device}", dtype=torch.uint8))
            stream.wait_stream(torch.cuda.current_stream(device))

    def one(self, k):
        a = self.actions[k % RING]
        # obs_out = NULL: the policy reads the engine-owned observation (mgym_observation), zero-copy
        self.env.step_device(a, None, self.reward, self.done, self.trunc)
        if self.reset_mode == "separate":
            self.env.reset_done_device(self.done, self.trunc, None)

    def build_graph(self, steps=None):
        """Capture + instantiate, before anything is timed, the graphs that `run(steps)` will replay: one of RING step
        launches and — when `steps` is not a multiple of RING — one of exactly the remainder, so a timed region of K
        steps is graph replays only (the same K step launches, no eager tail) however small K is."""
        if self.graph is None:
            self.graph = self.env.graph_capture(lambda: [self.one(k) for k in range(RING)])
        rem = (steps or 0) % RING
        if rem and rem not in self.tail_graphs:
            self.tail_graphs[rem] = self.env.graph_capture(lambda: [self.one(k) for k in range(rem)])

    def build_timed_graph(self, steps):
        """ONE graph of exactly `steps` step launches: a short timed region is then a single hipGraphLaunch (no eager
        tail, no second replay).  (Event records captured INTO the graph cannot be timed on this HIP: hipEventElapsedTime
        returns hipErrorInvalidHandle for them; the records stay on the stream, around the launch.)"""
        self.timed = (steps, self.env.graph_capture(lambda: [self.one(k) for k in range(steps)]))

    def run_timed(self, steps):
        assert self.timed is not None and self.timed[0] == steps
        self.env.graph_launch(self.timed[1])

    def run(self, steps):
        """issue exactly `steps` steps on the stream"""
        if self.roll_k:
            for _ in range(steps // self.roll_k):
                self.env.rollout_device(self.roll_actions, self.roll_k, None, *self.roll_out)
            return
        if self.launch != "graph":
            for k in range(steps):
                self.one(self.t + k)
            self.t += steps
            return
        self.build_graph(steps)   # no-op when the graphs exist already (they do for the timed region)
        for _ in range(steps // RING):
            self.env.graph_launch(self.graph)
        if steps % RING:
            self.env.graph_launch(self.tail_graphs[steps % RING])

    def close(self):
        for g in [self.graph] + list(self.tail_graphs.values()) + ([self.timed[1]] if self.timed else []):
            if g is not None:
                self.env.graph_destroy(g)
        self.env.close()


def host_cores():
    """CPU share actually available to this process (affinity mask, then the cgroup quota)."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    c = min(c, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    c = min(c, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    if os.environ.get("MGYM_BENCH_CORES"):   # explicit override; otherwise what affinity + cgroup quota allow
        c = min(c, int(os.environ["MGYM_BENCH_CORES"]))
    return max(1, min(c, 64))   # `cores` reports the threads actually used; 64 bounds the OpenMP team on a big shared host


def kernel_source_sha16(files):
    from modurl_gym_amd._srchash import kernel_source_sha16 as f  # comments and blank lines do not count
    return f(files)


def pmc_record(key):
    """Counter figures of profiles/pmc_traffic.json (rocprofv3 --pmc runs of this same bench.py, collected by
    tools/profile_pmc.sh).  An entry names the kernel sources it was measured on (sha256 prefix): a record taken on other
    sources than the ones this run executes is stale and is not reported."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(key)
        if rec and rec.get("src_sha16") == kernel_source_sha16(rec["src_files"]):
            return rec
    except Exception:
        pass
    return None


def lunar_roofline(n, step_s, variant=""):
    """LunarLander is bound by f32 VALU issue / dependent chains (180 Gauss-Seidel sweeps per step), not by HBM:
    achieved = counted f32 FMA (x2) + MUL + ADD lane-operations per step (SQ_INSTS_VALU_{FMA,MUL,ADD}_F32 x mean active
    lanes per VALU instruction, summed over the step's kernels; profiles/pmc_traffic.json) / this run's step time,
    against the 157.3 TFLOP/s f32 vector peak.  The HBM fraction is given for reference."""
    alg = (107 * 4 * 2 + 46) * n   # state words always touched, R + W, + API traffic
    out = {"bound": "valu", "peak": 157.3, "unit": "TFLOP/s", "achieved": None, "frac": None, "traffic": None,
           "kernel": "ll_step_kernel<32> (contact path, free-flight path, reset preparation as block roles of one launch) + ll_epilogue_kernel; from 376 832 envs: ll_contact_kernel<64> beside ll_free_kernel + ll_epilogue_kernel", "avg_step_us": step_s * 1e6,
           "hbm_for_reference": {"alg_bytes_per_step": alg, "GBps": alg / step_s / 1e9, "frac": alg / step_s / HBM_PEAK}}
    rec = pmc_record(f"lunar_lander{variant}:{n}")
    if variant:
        out["kernel"] = ("ll_rollout_kernel<32> (one persistent launch per K steps: free-flight residents, touching / light-contact / sub-step / reset batches through device queues) "
                         "+ ll_rollout_free_kernel<32> (free-flight helper waves beside it; the counter passes serialise kernels, so there the main waves did the helpers' steps too: same work)")
    if rec:
        out["achieved"] = rec["f32_flop_per_step_active_lanes"] / step_s / 1e12
        out["frac"] = out["achieved"] / 157.3
        out["flop_per_step_active_lanes"] = rec["f32_flop_per_step_active_lanes"]
        out["flop_per_step_all_64_lanes"] = rec["f32_flop_per_step_lanes64"]
        out["frac_if_all_64_lanes_counted"] = rec["f32_flop_per_step_lanes64"] / step_s / 157.3e12
        out["traffic"] = rec.get("hbm_bytes_per_step")   # FETCH_SIZE x2 + WRITE_SIZE summed over the step's launches (the bound is VALU: for reference)
        out["counter_source"] = rec.get("source")
        # why the fraction is what it is, from the same counter records (dominant kernel of the step)
        dom = max(rec.get("kernels", {}).items(), key=lambda kv: (kv[1].get("avg_ns") or 0) * kv[1].get("launches_per_step", 1), default=(None, None))
        if dom[0]:
            out["dominant_kernel"] = dom[0]
            for k_out, k_in in (("active_lanes_per_valu_inst", "mean_active_lanes"), ("valu_busy_share_of_wave_cycles", "valu_busy_share_of_wave_cycles"),
                                ("wait_share", "wait_any_share_of_wave_cycles"), ("waves_per_simd", "waves_per_simd"), ("scratch_bytes_per_lane", "scratch_bytes_per_lane"),
                                ("vgprs", "vgprs"), ("lds_bytes_per_block", "lds_bytes_per_block")):
                if dom[1].get(k_in) is not None:
                    out[k_out] = dom[1][k_in]
    return out


def cpu_baseline(workload, n, seed):
    """The oracle on this box's host cores: `for env in envs { env.step(a) }` over a bounded sample."""
    import numpy as np

    from oracle import oracle as ora  # test infrastructure; timed here as the reported CPU baseline

    kinds = {"cartpole": (ora.CARTPOLE, 2), "mountain_car": (ora.MOUNTAINCAR, 3), "mountain_car_cont": (ora.MOUNTAINCAR_CONT, 0),
             "lunar_lander": (ora.LUNARLANDER, 4)}
    kind, nact = kinds[workload]
    if workload == "lunar_lander":
        n = min(n, 4096)  # LunarLander CPU sample is 4 096 envs (scaled comparison, stated in `sample`)
    cores = host_cores()
    rng = np.random.default_rng(0)
    acts = np.stack([(rng.integers(0, nact, n).astype(np.uint32) if nact else rng.uniform(-1, 1, n).astype(np.float32))
                     for _ in range(4)])
    out = {}
    for label, threads, budget in (("1core", 1, 4.0), ("allcores", cores, 8.0)):
        env = ora.OracleVec(kind, n, seed=seed, **({"enable_wind": True} if workload == "lunar_lander" else {}))
        env.reset(nthreads=threads)
        chunk = 4 if workload != "lunar_lander" else 2   # steps per call into the C loop (step + reset-on-finish)
        steps, t0 = 0, time.perf_counter()
        while True:
            env.run(acts, chunk, nthreads=threads)
            steps += chunk
            el = time.perf_counter() - t0
            if el > budget or steps >= 2000:
                break
        out[label] = (n * steps / el, steps, el)
    c1 = None
    if workload == "cartpole":
        # BASELINE configs[0]: CartPole-v1, ONE env, CPU step() loop (plumbing case): 1e6 steps inside the C library
        one = ora.OracleVec(kind, 1, seed=seed)
        one.reset()
        t0 = time.perf_counter()
        one.run(rng.integers(0, 2, (16, 1)).astype(np.uint32), 1_000_000)
        c1 = (time.perf_counter() - t0) / 1e6 * 1e9
    v, steps, el = out["allcores"]
    res = {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{workload} {n} envs x {steps} steps (step + masked reset), {el:.1f} s, OpenMP index shards over {cores} threads; "
                      f"oracle = C restatement of the Rust reference (cargo/rustc absent: reference unbuildable)",
            "value_1core": out["1core"][0]}
    if c1 is not None:
        res["configs0_one_env_ns_per_step"] = c1   # step + reset-on-finish, 1 env, 1 core, 1e6 steps
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))   # no launcher around us: start the ranks as a child torchrun (before torch / HIP are touched)
    # The contract is ONE JSON line on stdout.  Native libraries print banners there (RCCL's version block at communicator
    # creation): point fd 1 at stderr for the run and keep the real stdout for the result line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world   # the launcher's world size wins (N > 1 without a launcher never gets here: self_launch)
    if args.dry_run_launch:
        dry_run_launch(args, rank, local_rank, world, real_stdout)
        return

    import modurl_gym_amd as mg   # first: it asks for the hardware queues before anything initialises HIP
    import torch

    if not torch.cuda.is_available() or mg.device_count() == 0:
        print("bench.py: no MI355X visible — the engine has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("MGYM_FORCE_DIST") == "1":  # the env var rehearses the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")   # (only the MGYM_FORCE_DIST=1 rehearsal comes here without a launcher's rendezvous)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    def barrier():
        if dist is not None:
            dist.barrier()

    stream = torch.cuda.Stream(device=local_rank)
    n_default = {"cartpole": 1 << 20, "mountain_car": 1 << 20, "mountain_car_cont": 1 << 20, "lunar_lander": 1 << 18,
                 "mixed": 1 << 20}
    n = args.envs or n_default[args.workload]
    # this rank's handles: global env ids [family][rank][local index] (weak) or contiguous blocks of a fixed node total (strong)
    plan = mg.population_plan(args.workload, world, rank, args.scaling, n_per_gpu=n, n_total=args.total_envs)
    pop = {name: cnt for name, cnt, _ in plan}
    if args.scaling == "strong" and args.workload != "mixed":
        n = plan[0][1]
    # one stream per family: the mixed batch's three step pipelines are independent and overlap on the device
    streams = [stream] + [torch.cuda.Stream(device=local_rank) for _ in plan[1:]]
    steppers = [Stepper(mg, torch, name, cnt, local_rank, args.seed, gbase, st, args.reset, args.launch, roll_k=args.ll_rollout)
                for (name, cnt, gbase), st in zip(plan, streams)]
    if args.ll_rollout and any(s_.roll_k for s_ in steppers):
        args.steps = max(args.ll_rollout, args.steps // args.ll_rollout * args.ll_rollout)
        args.warmup = args.warmup // args.ll_rollout * args.ll_rollout
    lead = steppers[0]

    def run(steps):
        for s in steppers:
            s.run(steps)

    # short timed regions (the driver runs --steps 20): ONE graph of exactly K launches with the event records inside
    one_graph = args.steps <= 256
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for s in steppers:
        if s.launch == "graph":
            s.build_graph(args.warmup)   # capture + instantiate before anything is timed, whatever the warm-up length
            if one_graph:
                s.build_timed_graph(args.steps)
            else:
                s.build_graph(args.steps)
    run(args.warmup)
    for s in steppers:
        s.env.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # HIP events on the LAUNCH stream of the lead family (torch events recorded on that stream, not on torch's current one); the second
    # one is only recorded inside the wall-clock window — waiting for it and reading it happen after t1, so the window holds the K
    # launches, one event record and the device synchronisation, nothing else
    ev_a.record(streams[0])
    t0 = time.perf_counter()
    for s in steppers:
        if s.launch == "graph" and one_graph:
            s.run_timed(args.steps)   # EXACTLY K steps: one hipGraphLaunch
        else:
            s.run(args.steps)         # EXACTLY K steps (graph replays, or eager launches)
    ev_b.record(streams[0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ev_ms = ev_a.elapsed_time(ev_b)
    barrier()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])
    for s in steppers:
        s.env.sync()                  # surfaces any sticky device-side error

    total_envs = sum(pop.values()) * world if args.scaling == "weak" else (args.total_envs if args.workload != "mixed" else sum(mg.mixed_population(args.total_envs).values()))
    value = total_envs * args.steps / elapsed
    result = {
        "metric": "env-steps/sec (whole node), CartPole 1M envs, at 1/2/4/8 MI355X",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"cartpole": "CartPole-v1, 1048576 envs per GPU, f32 SoA (BASELINE configs[1])",
                                "mixed": "Mixed CartPole+MountainCar+LunarLander, 1048576 envs per GPU (BASELINE configs[4] per-GPU load)"}
                   .get(args.workload, args.workload),
                   "n_envs_per_gpu": sum(pop.values()), "n_envs_total": total_envs, "launch": ", ".join(sorted({(((f"graph (one hipGraph of {args.steps} step launches)" if one_graph else f"graph (hipGraph of {RING} steps, replayed)") if s_.launch == "graph" else "eager")
                                                if s_.launch != "rollout" else f"mgym_rollout, {s_.roll_k} steps per persistent launch")
                                                + (f" [{s_.name}]" if len(steppers) > 1 else "") for s_ in steppers})),
                   "reset": "fused auto-reset in the step kernel" if args.reset == "fused" else "separate mgym_reset_done launch per step",
                   "parallelism": f"index-sharded x{world}, no data-path collective", "action_ring": RING,
                   "scaling": ("weak: fixed envs per GPU" if args.scaling == "weak" else f"strong: {total_envs} envs fixed for the node, contiguous index blocks per rank")},
    }
    if args.workload in ALG_BYTES:
        # dominant kernel = the step kernel; with --reset fused the timed region is K launches of it
        # (graph-replayed), so its average launch duration is event_time / K (includes the ~1.5 us
        # inter-kernel boundary: conservative against rocprof's pure kernel duration)
        alg = ALG_BYTES[args.workload] * n
        dur = ev_ms * 1e-3 / args.steps
        result["roofline"] = {"bound": "hbm", "achieved": alg / dur / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                              "frac": alg / dur / HBM_PEAK, "traffic": None,
                              "kernel": f"{args.workload}_step_kernel<4>", "alg_bytes_per_launch": alg,
                              "avg_launch_us": dur * 1e6,
                              "note": "algorithmic bytes/env-step x envs per launch / HIP-event time per launch; at 1Mi envs the working set "
                                      "(~50 MB) is Infinity-Cache resident, see DESIGN.md for the >256 MiB run"}
        if args.steps < 256 and lead.launch == "graph":
            # A short timed region (the driver runs --steps 20) starts on an idle stream, so its event interval carries
            # the host's graph-launch latency (10-20 us) and the clock ramp.  Report, beside it and AFTER the timed
            # region, the same kernel's steady per-launch time: HIP events around 30 replays of the 16-step graph.
            lead.build_graph(480)
            lead.run(64)
            lead.env.sync()
            lead.env.timer_start()
            lead.run(480)
            sms = lead.env.timer_stop()
            sdur = sms * 1e-3 / 480
            result["roofline"]["steady"] = {"avg_launch_us": sdur * 1e6, "achieved": alg / sdur / 1e9, "frac": alg / sdur / HBM_PEAK, "steps": 480,
                                            "note": "same kernel, HIP events around 480 graph-replayed launches after the timed region"}
        rec = pmc_record(f"{args.workload}:{n}")
        if rec:
            result["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
            result["roofline"]["traffic_source"] = rec.get("source")

    if args.workload == "lunar_lander":
        result["roofline"] = lunar_roofline(n, ev_ms * 1e-3 / args.steps, "_rollout%d" % args.ll_rollout if args.ll_rollout else "")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and (args.workload in ALG_BYTES or args.workload == "lunar_lander"):
        result["cpu_baseline"] = cpu_baseline(args.workload, n, args.seed)

    mixed_rec = None
    if not args.no_extra and args.workload == "cartpole" and args.envs is None:
        # BASELINE configs[4] at this N (north_star: "steps/sec reported at 1/2/4/8 GPUs on the 8M-env mixed batch"):
        # 524288 CartPole + 262144 MountainCar + 262144 LunarLander per GPU, measured on EVERY rank after the
        # headline region (never part of `value`).  Every rank reaches both collectives even if its run raised.
        msteps, mwarm, m_el, m_err = 64, 320, -1.0, None
        try:
            mplan = mg.population_plan("mixed", world, rank, "weak", n_per_gpu=1 << 20)
            mstreams = [stream] + [torch.cuda.Stream(device=local_rank) for _ in mplan[1:]]   # the three families overlap on the device
            mst = [Stepper(mg, torch, name, cnt, local_rank, args.seed + 3, gbase, st_, "fused", args.launch)
                   for (name, cnt, gbase), st_ in zip(mplan, mstreams)]
            for st in mst:
                st.run(mwarm)          # LunarLander needs a few hundred steps to reach its steady contact mix
            for st in mst:
                st.env.sync()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            m_err = repr(e)
        barrier()
        m_el_roll = -1.0
        if m_err is None:
            try:
                t0 = time.perf_counter()
                for st in mst:
                    st.run(msteps)
                torch.cuda.synchronize()
                m_el = time.perf_counter() - t0
                for st in mst:
                    st.env.sync()
                # the same mixed batch with LunarLander stepped through mgym_rollout (K = 16: the action ring IS a [16][n] table), CartPole and
                # MountainCar as before; a persistent LunarLander launch holds every SIMD, so the other two families run between the launches
                ll = next(st for st in mst if st.name == "lunar_lander")
                rw = torch.empty((RING, ll.n), device=f"cuda:{local_rank}", dtype=torch.float32)
                dn = torch.zeros((RING, ll.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
                tr = torch.zeros((RING, ll.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
                torch.cuda.synchronize()
                ll.env.rollout_device(ll.actions, RING, None, rw, dn, tr)   # (untimed first call)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for st in mst:
                    if st is ll:
                        for _ in range(msteps // RING):
                            ll.env.rollout_device(ll.actions, RING, None, rw, dn, tr)
                    else:
                        st.run(msteps)
                torch.cuda.synchronize()
                m_el_roll = time.perf_counter() - t0
                for st in mst:
                    st.env.sync()
                    st.close()
            except Exception as e:  # noqa: BLE001
                m_err, m_el, m_el_roll = repr(e), -1.0, -1.0
        if dist is not None:
            t = torch.tensor([m_el, -m_el, m_el_roll, -m_el_roll], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            m_el = float(t[0]) if float(t[1]) < 0 else -1.0   # any rank failing (its -m_el = +1) voids the figure
            m_el_roll = float(t[2]) if float(t[3]) < 0 else -1.0
        if m_el > 0:
            mixed_rec = {"env_steps_per_s": (1 << 20) * world * msteps / m_el, "ms_per_step": 1e3 * m_el / msteps,
                         "n_envs_total": (1 << 20) * world, "n_gpus": world, "steps": msteps, "warmup": mwarm,
                         "per_gpu": "524288 CartPole + 262144 MountainCar + 262144 LunarLander (wind on), fused auto-reset, one stream per family",
                         "scaling": "weak", "note": "max over ranks, barrier before; LunarLander dominates the step time"}
            if m_el_roll > 0:
                mixed_rec["lunar_lander_through_mgym_rollout_K16"] = {"env_steps_per_s": (1 << 20) * world * msteps / m_el_roll, "ms_per_step": 1e3 * m_el_roll / msteps,
                                                                      "note": "same batch, LunarLander stepped by mgym_rollout (K = 16 per launch: one persistent launch in which environments advance "
                                                                              "independently), CartPole and MountainCar by graph-replayed mgym_step"}
        else:
            mixed_rec = {"error": m_err or "failed on another rank"}

    if rank == 0 and world == 1 and not args.no_extra and args.workload == "cartpole" and args.envs is None:
        # the other BASELINE configs, measured AFTER the timed region (never the headline `value`)
        extra = {}
        for name, cnt, k in (("mountain_car", 1 << 20, 400), ("mountain_car_cont", 1 << 20, 400), ("lunar_lander", 1 << 18, 128),
                             ("cartpole_32Mi_envs_hbm_regime", 1 << 25, 96), ("mountain_car_32Mi_envs_hbm_regime", 1 << 25, 96)):
            wl = "cartpole" if name.startswith("cartpole") else ("mountain_car" if name.startswith("mountain_car_32Mi") else name)
            st = Stepper(mg, torch, wl, cnt, local_rank, args.seed + 7, 0, stream, args.reset, args.launch)
            st.run(640 if wl == "lunar_lander" else RING * 2)   # LunarLander: reach the steady mix of flight / contact / resets
            st.env.sync()
            torch.cuda.synchronize()
            st.env.timer_start()
            st.run(k)
            ms = st.env.timer_stop()
            st.env.sync()
            rec = {"env_steps_per_s": cnt * k / (ms * 1e-3), "us_per_step": ms * 1e3 / k, "n_envs": cnt, "steps": k, "launch": st.launch}
            if wl in ALG_BYTES:
                rec["alg_GBps"] = ALG_BYTES[wl] * cnt * k / (ms * 1e-3) / 1e9
                rec["hbm_frac"] = rec["alg_GBps"] * 1e9 / HBM_PEAK
                # the same object the headline carries: dominant kernel, algorithmic bytes per launch / HIP-event time per launch, counter traffic
                alg1 = ALG_BYTES[wl] * cnt
                rec["roofline"] = {"bound": "hbm", "achieved": rec["alg_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": rec["hbm_frac"], "traffic": None,
                                   "kernel": ("cartpole_step_kernel<4>" if wl == "cartpole" else "mountaincar_step4_kernel<%s, true>" % ("true" if wl.endswith("cont") else "false")),
                                   "alg_bytes_per_launch": alg1, "avg_launch_us": ms * 1e3 / k}
                prec = pmc_record(f"{wl}:{cnt}")
                if prec:
                    rec["roofline"]["traffic"] = prec["hbm_bytes_per_launch"]
                    rec["roofline"]["traffic_source"] = prec.get("source")
                if cnt >= (1 << 25):
                    # memory-side reference on THIS box, in this run: a plain device-to-device copy with the same read + write footprint as one
                    # launch (what the box's HBM delivers to a kernel with no arithmetic at all; boxes of the pool differ by ~15 % here)
                    nbytes = alg1 // 2
                    src = torch.empty(nbytes, device=f"cuda:{local_rank}", dtype=torch.uint8)
                    dst = torch.empty_like(src)
                    with torch.cuda.stream(stream):
                        for _ in range(3):
                            dst.copy_(src)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        for _ in range(20):
                            dst.copy_(src)
                        e1.record(stream)
                    e1.synchronize()
                    cms = e0.elapsed_time(e1) / 20
                    rec["roofline"]["copy_reference"] = {"GBps": 2 * nbytes / (cms * 1e-3) / 1e9, "us": cms * 1e3, "bytes_read_plus_written": 2 * nbytes,
                                                         "frac_of_copy": rec["alg_GBps"] / (2 * nbytes / (cms * 1e-3) / 1e9),
                                                         "note": "torch device-to-device copy of the same footprint, same box, same run"}
                    del src, dst
            if wl == "lunar_lander":
                rec["roofline"] = lunar_roofline(cnt, ms * 1e-3 / k)
                # mgym_rollout: K steps in ONE persistent launch in which every environment advances as soon as it is ready (SURVEY §8f-1;
                # modurl_gym_amd/csrc/ll_roll.h) — same handle, same steady population, the same per-environment results as K mgym_step calls
                for K_roll, reps in ((8, 6), (16, 4), (64, 2)):
                    acts_k = torch.randint(0, 4, (K_roll, cnt), device=f"cuda:{local_rank}", dtype=torch.int32)
                    rw = torch.empty((K_roll, cnt), device=f"cuda:{local_rank}", dtype=torch.float32)
                    dn = torch.zeros((K_roll, cnt), device=f"cuda:{local_rank}", dtype=torch.uint8)
                    tr = torch.zeros((K_roll, cnt), device=f"cuda:{local_rank}", dtype=torch.uint8)
                    stream.wait_stream(torch.cuda.current_stream(local_rank))
                    st.env.rollout_device(acts_k, K_roll, None, rw, dn, tr)   # (first call: untimed)
                    st.env.sync()
                    st.env.timer_start()
                    for _ in range(reps):
                        st.env.rollout_device(acts_k, K_roll, None, rw, dn, tr)
                    rms = st.env.timer_stop()
                    st.env.sync()
                    extra[f"lunar_lander_rollout_K{K_roll}"] = {
                        "env_steps_per_s": cnt * K_roll * reps / (rms * 1e-3), "us_per_step": rms * 1e3 / (K_roll * reps), "n_envs": cnt, "steps": K_roll * reps,
                        "launches_per_rollout": 4, "vs_mgym_step": (ms / k) / (rms / (K_roll * reps)),
                        "note": "mgym_rollout: one persistent launch per K steps (counter reset, main waves, free-flight helper waves beside them, next step's contact list), "
                                "environments advance independently through device queues; "
                                "word-for-word equal to K mgym_step calls (tests/test_gpu_lunar_rollout.py); the launch ends with its last environments' chains, "
                                "so longer rollouts amortise better"}
                    del acts_k, rw, dn, tr
            extra[name] = rec
            st.close()
        # fused K-step rollout (mgym_rollout): same semantics as K steps, state stays in registers
        st = Stepper(mg, torch, "cartpole", 1 << 20, local_rank, args.seed + 9, 0, stream, "fused", "eager")
        rw = torch.empty((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.float32)
        dn = torch.zeros((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
        tr = torch.zeros((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
        for _ in range(4):
            st.env.rollout_device(st.actions, RING, None, rw, dn, tr)
        st.env.sync()
        st.env.timer_start()
        reps = 50
        for _ in range(reps):
            st.env.rollout_device(st.actions, RING, None, rw, dn, tr)
        ms = st.env.timer_stop()
        st.env.sync()
        ksteps = reps * RING
        extra["cartpole_rollout_K16"] = {"env_steps_per_s": st.n * ksteps / (ms * 1e-3), "us_per_step": ms * 1e3 / ksteps, "n_envs": st.n,
                                         "steps": ksteps, "alg_bytes_per_env_step": 10 + 40 / RING,
                                         "note": "mgym_rollout: K=16 steps per launch, bit-identical to 16 mgym_step calls"}
        st.close()
        # ... and under the in-kernel linear policy (mgym_rollout_linear): the policy loop of `cartpole_torch_policy_loop` below without leaving the registers
        st = Stepper(mg, torch, "cartpole", 1 << 20, local_rank, args.seed + 10, 0, stream, "fused", "eager")
        rw = torch.empty((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.float32)
        dn = torch.zeros((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
        tr = torch.zeros((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
        stream.wait_stream(torch.cuda.current_stream(local_rank))
        pol = [0.1, 0.5, 1.0, 1.0, 0.0]
        for _ in range(4):
            st.env.rollout_linear_device(pol, RING, None, None, rw, dn, tr)
        st.env.sync()
        st.env.timer_start()
        reps = 50
        for _ in range(reps):
            st.env.rollout_linear_device(pol, RING, None, None, rw, dn, tr)
        ms = st.env.timer_stop()
        st.env.sync()
        extra["cartpole_rollout_linear_policy_K16"] = {"env_steps_per_s": st.n * reps * RING / (ms * 1e-3), "us_per_step": ms * 1e3 / (reps * RING), "n_envs": st.n, "steps": reps * RING,
                                                       "note": "mgym_rollout_linear: policy a = (w . obs + b > 0) evaluated in the rollout kernel (no action table, no policy launch); "
                                                               "equal to a stepping loop with the same weights and to the oracle (tests/test_gpu_classic.py)"}
        st.close()
        # BASELINE configs[2] as SURVEY §8d C3 defines it: MountainCar-v0 and MountainCarContinuous-v0, 1 048 576 envs each, separate handles on
        # separate streams, both in flight at once (graph replay on each stream, common start, the later end counts)
        s_a, s_b = stream, torch.cuda.Stream(device=local_rank)
        pa = Stepper(mg, torch, "mountain_car", 1 << 20, local_rank, args.seed + 13, 0, s_a, args.reset, "graph")
        pb = Stepper(mg, torch, "mountain_car_cont", 1 << 20, local_rank, args.seed + 14, 0, s_b, args.reset, "graph")
        pair_steps = 25 * RING
        for p_ in (pa, pb):
            p_.build_timed_graph(pair_steps)    # ONE graph of pair_steps launches per handle: two hipGraphLaunch calls in all
            p_.run(2 * RING)
            p_.env.sync()
        torch.cuda.synchronize()
        e_go, e_a, e_b = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e_go.record(s_a)
        s_b.wait_event(e_go)
        pa.run_timed(pair_steps)
        pb.run_timed(pair_steps)
        e_a.record(s_a)
        e_b.record(s_b)
        torch.cuda.synchronize()
        pms = max(e_go.elapsed_time(e_a), e_go.elapsed_time(e_b))
        pa.env.sync(), pb.env.sync()
        pair_bytes = (ALG_BYTES["mountain_car"] + ALG_BYTES["mountain_car_cont"]) * (1 << 20)
        extra["mountain_car_pair"] = {
            "env_steps_per_s": 2 * (1 << 20) * pair_steps / (pms * 1e-3), "us_per_step_of_both": pms * 1e3 / pair_steps, "n_envs": 2 << 20, "steps": pair_steps,
            "roofline": {"bound": "hbm", "achieved": pair_bytes * pair_steps / (pms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": pair_bytes * pair_steps / (pms * 1e-3) / HBM_PEAK, "traffic": None, "alg_bytes_in_flight_per_step": pair_bytes,
                         "kernel": "mountaincar_step4_kernel<false, true> beside mountaincar_step4_kernel<true, true>"},
            "note": "MountainCar-v0 + MountainCarContinuous-v0, 1 048 576 envs each, two handles on two streams, graph-replayed at once; time from the common start "
                    "to the later stream's end"}
        pa.close(), pb.close()
        # fused K-step rollout for MountainCar (table form of mgym_rollout: the state stays in registers across the K steps)
        st = Stepper(mg, torch, "mountain_car", 1 << 20, local_rank, args.seed + 15, 0, stream, "fused", "eager")
        rw = torch.empty((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.float32)
        dn = torch.zeros((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
        tr = torch.zeros((RING, st.n), device=f"cuda:{local_rank}", dtype=torch.uint8)
        stream.wait_stream(torch.cuda.current_stream(local_rank))
        for _ in range(4):
            st.env.rollout_device(st.actions, RING, None, rw, dn, tr)
        st.env.sync()
        st.env.timer_start()
        reps = 50
        for _ in range(reps):
            st.env.rollout_device(st.actions, RING, None, rw, dn, tr)
        ms = st.env.timer_stop()
        st.env.sync()
        ksteps = reps * RING
        extra["mountain_car_rollout_K16"] = {"env_steps_per_s": st.n * ksteps / (ms * 1e-3), "us_per_step": ms * 1e3 / ksteps, "n_envs": st.n, "steps": ksteps,
                                             "alg_bytes_per_env_step": 10 + 16 / RING,
                                             "note": "mgym_rollout: K=16 steps per launch (action 4 R + reward 4 W + flags 2 W per step, state 8 R + 8 W per launch), bit-identical to 16 mgym_step calls"}
        # ... and under the in-kernel linear policy (mgym_rollout_linear, Discrete(3): the index of the largest of three scores; here: push with the velocity)
        pol = [0.0, -30.0, 0.0, 0.0, 0.0, 1e-4, 0.0, 30.0, 0.0]
        for _ in range(4):
            st.env.rollout_linear_device(pol, RING, None, None, rw, dn, tr)
        st.env.sync()
        st.env.timer_start()
        for _ in range(reps):
            st.env.rollout_linear_device(pol, RING, None, None, rw, dn, tr)
        ms = st.env.timer_stop()
        st.env.sync()
        extra["mountain_car_rollout_linear_policy_K16"] = {"env_steps_per_s": st.n * ksteps / (ms * 1e-3), "us_per_step": ms * 1e3 / ksteps, "n_envs": st.n, "steps": ksteps,
                                                           "note": "mgym_rollout_linear: a = argmax_j (w_j . obs + b_j) evaluated in the rollout kernel (no action table); equal to a stepping loop "
                                                                   "with the same weights and to the oracle (tests/test_gpu_classic.py)"}
        st.close()
        # the step either side of the path: a torch policy produces the actions on the same
        # stream, the engine steps, the next observation feeds the policy — no host synchronisation in the loop
        tenv = mg.TorchVecEnv(mg.CARTPOLE, 1 << 20, device=local_rank, seed=args.seed + 11, auto_reset=True)
        w = torch.tensor([0.1, 0.5, 1.0, 1.0], device=tenv.device)
        with torch.cuda.stream(stream):
            obs = tenv.reset()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ksteps = 400
            for t in range(ksteps + 40):
                if t == 40:
                    ev0.record(stream)
                obs = tenv.step((torch.mv(obs.t(), w) > 0).to(torch.int32))[0]
            ev1.record(stream)
        ev1.synchronize()
        tenv.check()
        ms = ev0.elapsed_time(ev1)
        extra["cartpole_torch_policy_loop"] = {"env_steps_per_s": tenv.n * ksteps / (ms * 1e-3), "us_per_step": ms * 1e3 / ksteps,
                                               "n_envs": tenv.n, "steps": ksteps,
                                               "note": "per step: torch linear policy (mv, >, cast: 3 eager kernels) + mgym_step, all on one stream"}
        tenv.close()
        result["extra"] = extra
    if mixed_rec is not None and rank == 0:
        result.setdefault("extra", {})["mixed_configs4_this_n"] = mixed_rec

    if dist is not None and args.workload in ("cartpole", "mixed") and not args.no_extra:
        # the optional collective: all-gather of the CartPole observation shards into a learner tensor on every
        # rank, RCCL over xGMI; off the step path, reported on its own (rehearsable on one GPU with MGYM_FORCE_DIST=1)
        from modurl_gym_amd.shard import all_gather_observations, all_reduce_episode_count
        from modurl_gym_amd.torch_env import _DeviceSpan
        # Every rank reaches the SAME collectives in the same order whatever happens locally: a rank that fails before a collective
        # would leave the others blocked in it (and the job without its JSON line).  Local failures and unequal shard sizes
        # (--scaling strong with a population the world size does not divide) are agreed on first, through one all-reduce.
        ag, episodes, ag_err, view, local_count = None, None, None, None, 0
        try:
            ptr, stride = lead.env.observation_device()
            view = torch.as_tensor(_DeviceSpan(ptr, (4, lead.n), (4 * stride, 4), lead), device=f"cuda:{local_rank}")
            local_count = lead.env.episode_count()
        except Exception as e:  # noqa: BLE001
            ag_err = repr(e)
        agree = torch.tensor([1.0 if ag_err else 0.0, float(lead.n), -float(lead.n)], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(agree, op=dist.ReduceOp.MAX)
        any_failed, equal_shards = float(agree[0]) > 0, float(agree[1]) == -float(agree[2])
        if not any_failed and equal_shards:
            with torch.cuda.stream(stream):
                for _ in range(3):
                    all_gather_observations(view, world)
                torch.cuda.synchronize()
                t0g = time.perf_counter()
                for _ in range(20):
                    all_gather_observations(view, world)
                torch.cuda.synchronize()
                ag = (time.perf_counter() - t0g) / 20
        elif not any_failed:
            ag_err = "shard sizes differ across ranks: all-gather skipped (it takes equal shards)"
        else:
            ag_err = ag_err or "failed on another rank"
        episodes = all_reduce_episode_count(local_count, device=f"cuda:{local_rank}")   # (0 from a rank that failed locally)
        if ag_err:
            episodes = {"sum_over_ranks": episodes, "note": ag_err}
        if rank == 0:
            result.setdefault("extra", {})["all_gather_observations"] = {
                "ms": None if ag is None else ag * 1e3, "bytes_per_rank": lead.n * 16, "world": world,
                "episodes_finished_all_ranks": episodes,
                "note": "torch.distributed all_gather (nccl = RCCL) of the [4][n] CartPole observation shard + 8-byte all-reduce of the "
                        "finished-episode counters; never inside the timed region"}
    for s in steppers:
        s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
