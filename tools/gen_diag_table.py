"""Work split of the shared-read coordinate searches (pcm1_bin_device.h, search_pcm1_data; pcm16_bin_device.h, search_pcm16_data): the N x N candidate grid
(`parts` of them one behind the other) is walked along its anti-diagonals (row + col = d: the candidates of one width), a lane takes a run of consecutive
cells of that order.  A run costs `per_cell` reads per cell plus `per_stretch` per diagonal it touches (the reads a stretch starts with).  Prints the 65 run
boundaries that minimise the largest cost.
usage: gen_diag_table.py N [per_cell per_stretch parts]      PCM-1: 25 1 2 1 (grid step of one pixel)   PCM-16x0: 21 2 1 3 (step of two pixels, three parts)"""
import sys
N = int(sys.argv[1])
per_cell = int(sys.argv[2]) if len(sys.argv) > 2 else 1
per_stretch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cells = [(p, d, r) for p in range(parts) for d in range(2 * N - 1) for r in range(max(0, d - N + 1), min(N - 1, d) + 1)]
def split(limit):
    bounds = [0]; cost = 0; cur = None
    for i, (p, d, r) in enumerate(cells):
        add = per_cell + (per_stretch if (p, d) != cur else 0)
        if cost + add > limit:
            bounds.append(i); cost = per_cell + per_stretch
        else:
            cost += add
        cur = (p, d)
    bounds.append(len(cells))
    return bounds
lim = 1
while len(split(lim)) - 1 > 64: lim += 1
b = split(lim)
b += [len(cells)] * (65 - len(b))
print("max cost", lim, "runs", len(set(b)) - 1, "cells", len(cells))
print(", ".join(str(x) for x in b))
