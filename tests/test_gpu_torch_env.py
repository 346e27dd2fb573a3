"""Torch-ROCm interop: a policy loop that never leaves the GPU or torch's stream,
checked against the CPU oracle driven with the same actions."""
import numpy as np
import pytest

import modurl_gym_amd as mg
from oracle import oracle as ora

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def linear_policy(w):
    return lambda obs: (torch.einsum("k,kn->n", w, obs) > 0).to(torch.int32)


def test_policy_loop_on_torch_stream_matches_oracle():
    n, T = 40000, 120
    env = mg.TorchVecEnv(mg.CARTPOLE, n, seed=9, auto_reset=True)
    ref = ora.OracleVec(ora.CARTPOLE, n, seed=9)
    w = torch.tensor([0.1, 0.5, -1.0, -1.0], device=env.device)  # pushes the wrong way: episodes end quickly
    policy = linear_policy(w)
    side = torch.cuda.Stream(env.device)
    obs = env.reset()
    exp = ref.reset()
    assert np.array_equal(obs.cpu().numpy(), exp)
    acts, outs = [], []
    with torch.cuda.stream(side):  # everything below is stream-ordered; no host sync inside the loop
        side.wait_stream(torch.cuda.default_stream(env.device))
        for t in range(T):
            a = policy(obs)
            obs, rew, done, trunc = env.step(a)
            acts.append(a.clone())
            outs.append((obs.clone(), rew.clone(), done.clone(), trunc.clone()))
    side.synchronize()
    env.check()
    finished = 0
    for t in range(T):
        a = acts[t].cpu().numpy().astype(np.uint32)
        e_obs, e_rew, e_done, e_trunc = ref.step(a)
        g_obs, g_rew, g_done, g_trunc = (x.cpu().numpy() for x in outs[t])
        assert np.array_equal(g_rew, e_rew) and np.array_equal(g_done, e_done.astype(bool)) and np.array_equal(g_trunc, e_trunc.astype(bool)), t
        ref.reset(mask=e_done | e_trunc)
        assert np.array_equal(g_obs, ref.get_state()[:4]), f"step {t}"
        finished += int((e_done | e_trunc).sum())
    assert finished > 0  # the fused auto-reset was exercised


def test_lunar_lander_on_alternating_torch_streams_matches_a_plain_handle():
    """LunarLander forks helper streams from whatever stream the caller steps on and prepares resets beside later steps:
    stepping alternately on two torch streams (ordered by the caller with wait_stream, no host sync) must give the words a
    plain handle on its own stream gives."""
    n, T = 8192, 150
    env = mg.TorchVecEnv(mg.LUNARLANDER, n, seed=13, enable_wind=True, auto_reset=True)
    ref = mg.VecEnv(mg.LUNARLANDER, n, seed=13, enable_wind=True, auto_reset=True)
    obs0 = env.reset()
    assert np.array_equal(obs0.cpu().numpy(), ref.reset())
    gen = torch.Generator(device="cpu").manual_seed(5)
    acts = torch.randint(0, 4, (T, n), generator=gen, dtype=torch.int32).to(env.device)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(env.device), torch.cuda.Stream(env.device)]
    outs = []
    prev = torch.cuda.current_stream(env.device)
    for t in range(T):
        s = streams[t % 2]
        s.wait_stream(prev)          # the caller's ordering between its own streams
        with torch.cuda.stream(s):
            obs, rew, done, trunc = env.step(acts[t])
            outs.append((obs.clone(), rew.clone(), done.clone()))
        prev = s
    torch.cuda.synchronize()
    env.check()
    finished = 0
    for t in range(T):
        e_obs, e_rew, e_done, _ = ref.step(acts[t].cpu().numpy().astype(np.uint32))
        g_obs, g_rew, g_done = (x.cpu().numpy() for x in outs[t])
        assert np.array_equal(g_done, e_done.astype(bool)), t
        assert np.array_equal(g_rew.view(np.uint32), e_rew.view(np.uint32)), t
        assert np.array_equal(g_obs.view(np.uint32), e_obs.view(np.uint32)), t
        finished += int(e_done.sum())
    assert finished > n // 2   # auto-resets (prepared ahead on the helper stream) happened throughout


def test_aos_layout_and_zero_copy_view():
    n = 3000
    env = mg.TorchVecEnv(mg.LUNARLANDER, n, seed=3, auto_reset=False, obs_layout="aos", enable_wind=True)
    obs = env.reset()
    assert tuple(obs.shape) == (n, 8) and obs.is_contiguous()
    g = torch.Generator(device="cpu").manual_seed(1)
    for _ in range(5):
        a = torch.randint(0, 4, (n,), generator=g).to(env.device)  # int64: converted by the wrapper
        obs, rew, done, trunc = env.step(a)
    view = env.observation_view()
    assert tuple(view.shape) == (8, n) and view.data_ptr() == env.env.observation_device()[0]
    env.check()
    assert torch.equal(view.t().contiguous(), obs)
    assert done.dtype == torch.bool and rew.dtype == torch.float32


def test_rollout_tensor_equals_steps():
    n, K = 5000, 12
    a = torch.randint(0, 3, (K, n), dtype=torch.int32, device="cuda")
    fused = mg.TorchVecEnv(mg.MOUNTAINCAR, n, seed=2, auto_reset=True)
    plain = mg.TorchVecEnv(mg.MOUNTAINCAR, n, seed=2, auto_reset=True)
    fused.reset(), plain.reset()
    obs, rew, done, trunc = fused.rollout(a)
    for k in range(K):
        o, r, d, t = plain.step(a[k])
        assert torch.equal(obs[k], o) and torch.equal(rew[k], r) and torch.equal(done[k], d) and torch.equal(trunc[k], t)
    fused.check(), plain.check()


def test_argument_checks():
    env = mg.TorchVecEnv(mg.CARTPOLE, 64, seed=1)
    env.reset()
    with pytest.raises(ValueError):
        env.step(torch.zeros(63, dtype=torch.int32, device=env.device))
    with pytest.raises(ValueError):
        env.step(torch.zeros(64, dtype=torch.int32))  # host tensor
    with pytest.raises(TypeError):
        env.step(torch.zeros(64, dtype=torch.float32, device=env.device))
    with pytest.raises(mg.InvalidActionError):  # cartpole.rs:252
        env.step(torch.full((64,), 2, dtype=torch.int32, device=env.device), check=True)


def test_lunar_lander_steps_captured_by_torch_cuda_graph_at_the_multi_stream_size(monkeypatch):
    """A capture the engine did not begin itself: `torch.cuda.graph` around `TorchVecEnv.step` at 425 984 envs, a population that
    runs the multi-stream order (64-lane contact kernel beside the free-flight kernel, helper streams forked from and joined to the capturing
    stream by events, staged resets prepared beside later steps).  Replays interleaved with eager steps must give the words of a plain handle
    that computes every reset when the episode ends (MGYM_LL_STAGED_RESET=0) — the pending reset preparations are handed over between eager
    steps and replays in both directions, and no env that finishes inside a replay may stay unreset.  (Semantics: the reference resets a done
    env through `reset()` incl. the implicit `step(0)`, /root/reference src/box_2d/lunar_lander.rs:727-917.)"""
    n, ring = 425984, 4
    env = mg.TorchVecEnv(mg.LUNARLANDER, n, seed=77, enable_wind=True, auto_reset=True)
    monkeypatch.setenv("MGYM_LL_STAGED_RESET", "0")
    ref = mg.VecEnv(mg.LUNARLANDER, n, seed=77, enable_wind=True, auto_reset=True)
    monkeypatch.delenv("MGYM_LL_STAGED_RESET")
    assert env.env.info()["launch_order"] == "overlapped" and env.env.info()["staged_resets"] == "1" and ref.info()["staged_resets"] == "0"
    gen = torch.Generator(device="cpu").manual_seed(3)
    side = torch.cuda.Stream(env.device)
    static_a = torch.zeros((ring, n), dtype=torch.int32, device=env.device)
    keep = [torch.empty(n, dtype=torch.float32, device=env.device) for _ in range(ring)]   # reward of each captured step, copied inside the graph
    finished = 0

    def eager(steps):
        nonlocal finished
        for _ in range(steps):
            a = torch.randint(0, 4, (n,), generator=gen, dtype=torch.int32)
            with torch.cuda.stream(side):
                obs, rew, done, _ = env.step(a.to(env.device, non_blocking=False))
            side.synchronize()
            e_obs, e_rew, e_done, _ = ref.step(a.numpy().astype(np.uint32))
            assert np.array_equal(done.cpu().numpy(), e_done.astype(bool)) and np.array_equal(rew.cpu().numpy().view(np.uint32), e_rew.view(np.uint32))
            assert np.array_equal(obs.cpu().numpy().view(np.uint32), e_obs.view(np.uint32))
            finished += int(e_done.sum())

    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.default_stream(env.device))
        obs0 = env.reset()
    side.synchronize()
    assert np.array_equal(obs0.cpu().numpy(), ref.reset())
    eager(70)                                  # until contacts, crashes and staged resets are frequent; binds the engine to `side`
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for k in range(ring):
            _, rew, _, _ = env.step(static_a[k])
            keep[k].copy_(rew)
    # the capture itself advanced nothing
    assert np.array_equal(env.state_numpy().view(np.uint32), ref.get_state().view(np.uint32))
    for rnd in range(4):
        for rep in range(2):
            a = torch.randint(0, 4, (ring, n), generator=gen, dtype=torch.int32)
            static_a.copy_(a.to(env.device))
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            for k in range(ring):
                _, e_rew, e_done, _ = ref.step(a[k].numpy().astype(np.uint32))
                assert np.array_equal(keep[k].cpu().numpy().view(np.uint32), e_rew.view(np.uint32)), f"round {rnd} replay {rep} step {k}"
                finished += int(e_done.sum())
        env.check()                            # a finished env without a reset would have raised the sticky internal error
        assert np.array_equal(env.state_numpy().view(np.uint32), ref.get_state().view(np.uint32)), f"round {rnd}: state blob after the replays"
        eager(3)
    assert finished > n // 4
    env.close(), ref.close()
