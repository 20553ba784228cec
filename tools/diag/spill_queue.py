#!/usr/bin/env python3
"""Steady-state backlog of a row of the quad scatter, as a multiple of its single-round overflow (CPU, numpy).

A row with `c` slots receives Poisson(rho * c) items per round, `c` leave with the record, the rest is carried to the
next round: q' = max(0, q + X - c).  Prints mean(q) / E[max(X - c, 0)] -- the table `ratio` in
kpal_amd/csrc/kpal_quads.hip (quad_expected_backlog), which the host uses to pick the tile size of a feed."""
from math import exp
import numpy as np

rng = np.random.default_rng(7)


def overflow(lam, c):
    p, acc = exp(-lam), 0.0
    for x in range(1, int(lam + 12 * lam ** 0.5 + 40)):
        p *= lam / x
        if x > c:
            acc += (x - c) * p
    return acc


def mean_queue(lam, c, rounds, rows, burn):
    q, tot, n = np.zeros(rows), 0.0, 0
    for t in range(rounds):
        q = np.maximum(0, q + rng.poisson(lam, rows) - c)
        if t >= burn:
            tot += q.mean()
            n += 1
    return tot / n


if __name__ == '__main__':
    rhos = [0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9, 0.925, 0.95, 0.975]
    print('rho  ', rhos)
    for c in (16, 32, 64, 128):
        row = []
        for r in rhos:
            heavy = r >= 0.95
            m = mean_queue(r * c, c, 30000 if heavy else 6000, 1500 if heavy else 4000, 10000 if heavy else 2000)
            e = overflow(r * c, c)
            row.append(max(1.0, m / e) if e > 1e-9 else 1.0)
        print(c, [round(x, 2) for x in row])
