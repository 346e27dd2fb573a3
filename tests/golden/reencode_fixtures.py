#!/usr/bin/env python3
"""Re-encode the reference's golden trajectories into compact fixtures.

Provenance: /root/reference/python_tests/{cartpole,mountain_car,lunar_lander}/
{inputs,output}.json — numeric outputs of gymnasium CartPole-v1 / MountainCar-v0 /
LunarLander-v3 recorded by the reference's generator scripts (never run here:
gymnasium/Box2D are not installed).  Only DATA is copied: the 100 actions and the
100 expected (observation, reward, done, truncated[, info.raw_*]) rows per env.
The reference repository carries no LICENSE file; the numbers are facts about
gymnasium's behaviour.

Run only in the authoring container (the GPU box has no /root/reference):
    python tests/golden/reencode_fixtures.py
"""
import json
import os

SRC = "/root/reference/python_tests"
DST = os.path.dirname(os.path.abspath(__file__))
KEEP = ("leg0_contact", "leg1_contact", "lander_awake", "game_over", "prev_shaping", "helipad_y")


def main():
    for env in ("cartpole", "mountain_car", "lunar_lander"):
        actions = json.load(open(f"{SRC}/{env}/inputs.json"))
        outputs = json.load(open(f"{SRC}/{env}/output.json"))
        rows = []
        for e in outputs:
            r = {k: e[k] for k in ("observation", "reward", "done", "truncated")}
            if env == "lunar_lander":
                info = e.get("info") or {}
                r["info"] = {k: v for k, v in info.items() if k.startswith("raw_") or k in KEEP}
            rows.append(r)
        with open(f"{DST}/{env}.json", "w") as f:
            json.dump({"actions": actions, "expected": rows}, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
