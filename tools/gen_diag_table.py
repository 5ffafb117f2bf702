"""Work split of the shared-read coordinate search (pcm1_bin_device.h, search_pcm1_data; pcm16_bin_device.h): the N x N candidate grid is walked
along its anti-diagonals (row + col = d: the candidates of one width), a lane takes a run of consecutive cells of that order.  A run costs its cells
plus `border` extra reads per diagonal it touches (the neighbours of its first and last cell).  Prints the 65 run boundaries that minimise the largest
cost.  usage: gen_diag_table.py N [border]"""
import sys
N = int(sys.argv[1]); border = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cells = [(d, r) for d in range(2 * N - 1) for r in range(max(0, d - N + 1), min(N - 1, d) + 1)]
def split(limit):
    bounds = [0]; cost = 0; cur_d = None
    for i, (d, r) in enumerate(cells):
        add = 1 + (border if d != cur_d else 0)
        if cost + add > limit:
            bounds.append(i); cost = 1 + border; cur_d = d
        else:
            cost += add; cur_d = d
    bounds.append(len(cells))
    return bounds
lim = 1
while len(split(lim)) - 1 > 64: lim += 1
b = split(lim)
b += [len(cells)] * (65 - len(b))
print("max cost", lim, "runs", len(set(b)) - 1, "cells", len(cells))
print(", ".join(str(x) for x in b))
