"""Decode tests/golden/cases.json (written by tests/golden/gen_golden.py) back into numpy."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases.json")


def dec(x):
    if isinstance(x, dict) and "hex" in x:
        bits = np.array([int(h, 16) for h in x["hex"]], dtype=np.uint64)
        return bits.view(np.float64).reshape(x["shape"])
    if isinstance(x, dict) and "u8" in x:
        return np.array(x["u8"], dtype=np.uint8)
    if isinstance(x, dict):
        return {k: dec(v) for k, v in x.items()}
    if isinstance(x, list):
        return [dec(v) for v in x]
    return x


def load_golden():
    with open(_PATH) as f:
        raw = json.load(f)
    return [dec(r) for r in raw]
