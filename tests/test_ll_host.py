"""No-GPU check of the LunarLander KERNEL SOURCE: modurl_gym_amd/csrc/ll_{b2,world,env}.h are compiled
for the host (tests/native/ll_host_check.cpp supplies a shim for the HIP qualifiers) and stepped
side by side with the CPU oracle on identical seeds, actions and Philox dispersion draws.
Also covers the device math header against the container's glibc (tests/native/math_host_check.cpp)."""
import os
import re
import subprocess

import pytest

from oracle import oracle as ora

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "native", "_build")


def _build(src, out, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, out)
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-o", exe, os.path.join(ROOT, "tests", "native", src),
           *extra, "-lm"]
    subprocess.run(cmd, check=True)
    return exe


@pytest.fixture(scope="module")
def ll_host():
    ora.build()
    odir = os.path.join(ROOT, "oracle", "_build")
    return _build("ll_host_check.cpp", "ll_host_check", [f"-L{odir}", "-loracle", f"-Wl,-rpath,{odir}"])


@pytest.mark.parametrize("n,steps,wind,det,toi_staged,physics", [(64, 300, 0, 1, 1, ()), (256, 400, 0, 0, 1, ()), (256, 400, 1, 0, 1, ()), (256, 300, 1, 0, 0, ()),
                                                                  (192, 500, 1, 0, 1, (-3.5, 19.5, 1.95)), (192, 300, 1, 0, 1, (-11.9, 0.5, 0.1))])
def test_kernel_source_matches_oracle_on_cpu(ll_host, n, steps, wind, det, toi_staged, physics):
    """toi_staged = 0: the contact cache's TOI word stays in its column (the 64-lane blocks' layout); in every case the
    velocity constraints beyond the fourth go through the far-workspace path.  physics = (gravity, wind_power, turbulence_power) near the
    ends of the ranges the builder accepts: the sweep loops' early exits (fixed point, cycle, settled velocity) are proofs about the
    constraint arithmetic and must give the full 180 / 60 iterations' words — which the oracle runs — under any of them."""
    r = subprocess.run([ll_host, str(n), str(steps), str(wind), str(det), str(toi_staged), *map(str, physics)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    m = re.search(r"mismatches=(\d+) exact_words=(\d+)/(\d+) episodes_done=(\d+) overflow=(\d+)", r.stdout)
    assert m, r.stdout
    mism, exact, total, done, overflow = map(int, m.groups())
    assert mism == 0 and overflow == 0
    assert exact == total          # same operation order: every observation word is bit-identical
    assert done > n                # the contact / TOI / crash paths ran many times


def test_device_math_header_matches_glibc():
    exe = _build("math_host_check.cpp", "math_host_check")
    out = subprocess.run([exe, "997"], capture_output=True, text=True, check=True).stdout
    rows = {l.split()[0]: dict(kv.split("=") for kv in l.split()[1:]) for l in out.strip().splitlines()}
    for fn in ("sinf", "cosf", "tanhf", "sincosf_self", "sincosf_uniform_vs_sincosf"):
        assert int(rows[fn]["bad_fast"]) == 0, out      # |x| <= 16: every angle the environments produce
    assert int(rows["sinf"]["bad_large"]) == 0 and int(rows["cosf"]["bad_large"]) == 0
    # 16 < |x| < 120: glibc's FMA ifunc variant may round a handful of inputs differently
    assert int(rows["sinf"]["bad_mid"]) + int(rows["cosf"]["bad_mid"]) <= 8, out
    assert int(rows["tanhf"]["bad_mid"]) == 0 and int(rows["tanhf"]["bad_large"]) == 0


def test_kernel_source_is_clean_under_asan_and_ubsan():
    """GPU AddressSanitizer is not available on the pool, and an out-of-bounds index in a kernel can take a GPU
    down: run the same kernel source on the host under ASan + UBSan (contact cache, constraint tables, worklists)."""
    ora.build()
    odir = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "ll_host_check_san")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-o", exe, os.path.join(ROOT, "tests", "native", "ll_host_check.cpp"),
                    f"-L{odir}", "-loracle", f"-Wl,-rpath,{odir}", "-lm"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe, "128", "350", "1", "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert re.search(r"mismatches=0 ", r.stdout), r.stdout[-2000:]
