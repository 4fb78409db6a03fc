"""Seeded input generators shared by oracle/gen_golden.py (which needs /root/reference) and the tests (which must not): TEST INFRASTRUCTURE."""
from __future__ import annotations

import numpy as np


def a5_random_batches(n_cases=120, seed=0):
    """Seeded random ragged batches for the splice: 1-4 samples of 2-29 tokens, 0-3 <image> sentinels each (never first, never adjacent), the
    token before a sentinel is <image_start> (an answer image) half of the time, T in {1, 4, 16}, max length in {12, 24, 40, 4096}, every
    fifth case left-padded.  Returns [(id rows, label rows, T, max_len, padding side)]; inputs are never stored, only this seed."""
    rng = np.random.default_rng(seed)
    cases = []
    for trial in range(n_cases):
        B = int(rng.integers(1, 5))
        Timg = int(rng.choice([1, 4, 16]))
        max_len = int(rng.choice([12, 24, 40, 4096]))
        rows, labs = [], []
        for b in range(B):
            n = int(rng.integers(2, 30))
            ids = rng.integers(3, 1000, size=n).tolist()
            lab = [int(x) if rng.random() < 0.5 else -100 for x in ids]
            for _ in range(int(rng.integers(0, 4))):
                q = int(rng.integers(1, len(ids)))
                if ids[q - 1] == -200 or (q < len(ids) and ids[q] == -200):
                    continue
                ids.insert(q, -200)
                lab.insert(q, -200)
                if rng.random() < 0.5:
                    lab[q - 1] = 128256
            rows.append(ids)
            labs.append(lab)
        cases.append((rows, labs, Timg, max_len, "left" if trial % 5 == 0 else "right"))
    return cases
