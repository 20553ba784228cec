#!/usr/bin/env python3
"""What a THRESHOLD flush of the k = 12 scatter's rows would give (docs/NOTEBOOK.md, round 6 item 9; not built).

Today every row (2048 rows of 20 three-byte items) leaves LDS as one padded 64-byte record at the end of every tile; a row
receives Poisson(m) items per tile (m = 14.8 at the eight-step tile of uniform reads), what does not fit rides in the spill list
(2048 entries) and is placed first in the next tile.  Here a row is written out only when it holds at least T items at the end
of a tile and is carried otherwise: the records get fuller, at the price of shorter tiles (m small enough that a carried row
rarely overflows) -- i.e. more barriers per input byte -- and of per-(row, workgroup) cursors instead of a round number.

Prints, per (m, T): the fill of the records written, the share of the items that ride in the spill list, the list's mean / 99 %
length, the share the list cannot hold, and the rows flushed per tile.
"""
import numpy as np

ROWS, SLOTS, CAP = 2048, 20, 2048


def simulate(m, T, tiles=400, seed=1):
    rng = np.random.default_rng(seed)
    fill = np.zeros(ROWS, int)
    spill = np.zeros(ROWS, int)
    records = items_out = direct = 0
    backlog = []
    for _ in range(tiles):
        total = fill + spill + rng.poisson(m, ROWS)          # the carried items are placed first
        over = np.maximum(total - SLOTS, 0)
        fill = np.minimum(total, SLOTS)
        b = int(over.sum())
        backlog.append(b)
        if b > CAP:                                          # what the list cannot hold is counted on the spot
            keep = np.floor(over * (CAP / b)).astype(int)
            direct += b - int(keep.sum())
            over = keep
        spill = over
        out = fill >= T
        records += int(out.sum())
        items_out += int(fill[out].sum())
        fill[out] = 0
    backlog = np.array(backlog)
    return {'m': m, 'T': T, 'fill': items_out / (records * SLOTS), 'spilled': backlog.mean() / (m * ROWS),
            'list_mean': backlog.mean(), 'list_p99': float(np.percentile(backlog, 99)),
            'direct': direct / (m * ROWS * tiles), 'rows_per_tile': records / tiles}


if __name__ == '__main__':
    print('   m   T   fill  spilled  list mean / p99   direct  rows flushed per tile')
    for m, T in [(14.8, 0), (16.6, 0), (14.8, 6), (14.8, 10), (7.4, 0), (7.4, 13), (7.4, 14), (7.4, 15), (5.5, 14), (5.5, 16),
                 (3.7, 16), (3.7, 17), (1.85, 17), (1.85, 18)]:
        r = simulate(m, T)
        print('%5.2f %3d  %5.3f  %6.4f   %7.0f / %5.0f   %6.4f   %6.0f' % (r['m'], r['T'], r['fill'], r['spilled'], r['list_mean'],
                                                                       r['list_p99'], r['direct'], r['rows_per_tile']))
