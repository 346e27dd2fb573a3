"""bench.py's torch.distributed (nccl = RCCL) path on ONE GPU, as the driver's 8-GPU launch would execute it on each rank: process group,
barriers, the MAX-over-ranks reductions, the mixed configs[4] section with its agreement all-reduce, the optional all-gather of observation
shards.  Run as a CHILD process (a fresh interpreter; never a re-exec of the test process).  The reference is single-threaded and `!Send`
(/root/reference src/box_2d/lunar_lander.rs:240-249): the split over GPUs is this build's to prove."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_rccl_path_runs_on_one_gpu_as_a_child_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MGYM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # the contract: ONE JSON line on stdout (RCCL's banner goes to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 1e10 and line["scaling"] == "weak"
    ag = line["extra"]["all_gather_observations"]
    assert isinstance(ag["ms"], float) and ag["ms"] > 0 and ag["world"] == 1 and ag["bytes_per_rank"] == (1 << 20) * 16
    assert isinstance(ag["episodes_finished_all_ranks"], int) and ag["episodes_finished_all_ranks"] > 0
    mixed = line["extra"]["mixed_configs4_this_n"]
    assert "error" not in mixed and mixed["env_steps_per_s"] > 5e8 and mixed["n_gpus"] == 1
