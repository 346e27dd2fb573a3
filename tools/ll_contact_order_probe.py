"""How much does the ONE documented deviation of the Box2D restatement matter?  (CPU only; oracle = test infrastructure.)

Box2D's order of newly created contacts follows its dynamic-tree traversal; the oracle and the kernels both use "moved proxies in
body-list order, partner edges by ascending id"  — a common-mode choice no GPU-vs-oracle test can see.  The order
only decides in which order contacts created in the SAME FindNewContacts call enter the contact list (hence the solver).  This
script runs whole episodes from reset under the product's order and under the opposite orders (partners by descending id, moved
proxies reversed, both) with identical seeds and actions, and reports, per episode, when the first difference appears and how large
the observation / reward differences are at that step and at the end of the episode.
usage: python tools/ll_contact_order_probe.py [n_envs=4096] [steps=500] [wind=1]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as ora  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
wind = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
lib = ora.lib()
rng = np.random.default_rng(3)
acts = rng.integers(0, 4, (steps, n)).astype(np.uint32)
acts[:, ::4] = np.where(rng.random((steps, (n + 3) // 4)) < 0.5, 2, 0)[:, : acts[:, ::4].shape[1]]   # every fourth env brakes half the time: softer touch-downs, more landings


def run(variant):
    lib.ora_set_contact_order_variant(variant)
    env = ora.OracleVec(ora.LUNARLANDER, n, seed=99, enable_wind=wind)
    env.reset(nthreads=8)
    obs = np.zeros((steps, 8, n), np.float32); rew = np.zeros((steps, n), np.float32); done = np.zeros((steps, n), np.uint8)
    for t in range(steps):
        o, r, d, _ = env.step(acts[t], nthreads=8)   # no reset: a finished env keeps being stepped, only its first episode is compared
        obs[t], rew[t], done[t] = o, r, d
    lib.ora_set_contact_order_variant(0)
    return obs, rew, done


base = run(0)
first_done = np.where(base[2].any(0), base[2].argmax(0), steps - 1)        # last step of each env's first episode
landed = base[1][first_done, np.arange(n)] == 100.0
print(f"{n} episodes from reset, {steps} steps, wind={int(wind)}: {int(base[2].any(0).sum())} end within the horizon, {int(landed.sum())} of them asleep on the ground (+100)")
for variant, name in ((1, "partner edges by DESCENDING id"), (2, "moved proxies in REVERSE order"), (3, "both")):
    obs, rew, done = run(variant)
    t_idx = np.arange(steps)[:, None]
    in_episode = t_idx <= first_done[None, :]
    diff = np.abs(obs - base[0]).max(1)                                     # [steps, n] max over the 8 observation words
    diff = np.where(in_episode, diff, 0.0)
    differs = diff > 0
    any_diff = differs.any(0)
    first = np.where(any_diff, differs.argmax(0), 0)
    at_first = diff[first, np.arange(n)][any_diff]
    at_end = diff[first_done, np.arange(n)][any_diff]
    rew_diff = np.where(in_episode, np.abs(rew - base[1]), 0.0).max(0)
    flags = (np.where(in_episode, done != base[2], False)).any(0)
    print(f"variant {variant} ({name}): {int(any_diff.sum())} of {n} episodes differ anywhere ({100.0 * any_diff.mean():.2f} %)")
    if any_diff.any():
        print(f"    observation difference at the first differing step: median {np.median(at_first):.3g}, max {at_first.max():.3g};"
              f" at the end of the episode: median {np.median(at_end):.3g}, max {at_end.max():.3g}")
        print(f"    largest reward difference inside an episode {rew_diff.max():.3g}; episodes whose done flag differs at some step: {int(flags.sum())}"
              f"; steps into the episode at which the first difference appears: median {int(np.median(first[any_diff]))} (episode length median {int(np.median(first_done))})")
