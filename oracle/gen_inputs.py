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


_WORDS = "a red bird sat on the mat what is shown here draw me one two three blue cat please describe it now and then yes no".split()


def n2_random_sources(n=40, seed=7):
    """Seeded random single-sample conversations for the batch producer: 1-4 rounds, 0-6 words per turn (empty answers included), <image>
    at the start / middle / end of a human or gpt turn (0-2 per conversation), sometimes a leading gpt turn the template must skip."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        rounds = int(rng.integers(1, 5))
        conv = []
        if rng.random() < 0.15:
            conv.append({"from": "gpt", "value": "ignored greeting"})
        budget = int(rng.integers(0, 3))
        for r in range(rounds):
            for who in ("human", "gpt"):
                words = [str(rng.choice(_WORDS)) for _ in range(int(rng.integers(0 if who == "gpt" else 1, 7)))]
                if budget and rng.random() < 0.35:
                    words.insert(int(rng.integers(0, len(words) + 1)), "<image>")
                    budget -= 1
                conv.append({"from": who, "value": " ".join(words)})
        out.append([conv])
    return out
