#!/usr/bin/env python3
"""More seeds of the randomised NaN-input and tall-column tests of tests/test_gpu_random.py (a one-off sweep on a GPU
box: python tests/devtools/random_sweep_nan_tall.py [first_seed] [count]); prints one line per seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_random as t

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    t0 = time.time()
    for name, fn in (("nan", t.test_random_problems_with_nan_inputs), ("tall", t.test_random_tall_columns)):
        try:
            with np.errstate(all="ignore"):
                fn(seed)
            res = "ok"
        except AssertionError as exc:
            res = f"VIOLATION {exc}"
            bad += 1
        print(f"seed {seed} {name}: {res} ({time.time() - t0:.1f} s)", flush=True)
print(f"{count} seeds, {bad} violations")
