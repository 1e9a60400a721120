#!/usr/bin/env python3
"""Simulation of the LDS bank conflicts of k_tile_ll's table lookups for different orders of the entries inside the rows of a
slice (tile-build time), table strides 18 / 19 doubles per locus.  A lookup step of a half-wave = 32 lanes each reading 8 bytes:
cycles = max over the 32 bank pairs of the number of DISTINCT addresses on it (MI355X_MICROARCH.md, LDS).  Two lookups per entry:
log-pmf at slot*S + code, expected term at slot*S + 14 + (n-1)."""
import sys
import numpy as np

rng = np.random.default_rng(1)
NSLOT, DENS = 639, 0.01


def make_tile():
    """rows (cells) of a 1024-cell x 639-locus tile: lists of (slot, code, n-1), ascending slot"""
    rows = []
    p_l = rng.uniform(0.05, 0.95, NSLOT)
    for _ in range(1024):
        slots = np.nonzero(rng.random(NSLOT) < DENS)[0]
        n = rng.geometric(0.7, len(slots))
        keep = n <= 4
        slots, n = slots[keep], n[keep]
        ref = rng.binomial(n, 1 - p_l[slots])
        code = n * (n + 1) // 2 - 1 + ref
        rows.append(np.stack([slots, code, n - 1], axis=1))
    return rows


def slices_of(rows):
    order = sorted(range(1024), key=lambda i: len(rows[i]))  # stable
    out = []
    for s in range(16):
        rs = [rows[i] for i in order[s * 64:(s + 1) * 64]]
        K = max(len(r) for r in rs) | 1
        out.append((rs, K))
    return out


def cost_half(ent, S):
    """ent: [32][K] arrays of (slot, code, nm1) or None (padding) in lookup order -> total cycles of the 2K lookup instructions"""
    K = len(ent[0])
    tot = 0
    for k in range(K):
        for which in (0, 1):
            per_bank = {}
            for lane in range(32):
                e = ent[lane][k]
                if e is None:
                    addr = NSLOT * S + (0 if which == 0 else 14)
                else:
                    addr = e[0] * S + (e[1] if which == 0 else 14 + e[2])
                per_bank.setdefault(addr % 32, set()).add(addr)
            tot += max(len(v) for v in per_bank.values())
    return tot


def pad(rs, K):
    return [[tuple(r[i]) if i < len(r) else None for i in range(K)] for r in rs]


def greedy_seq(rs, K, S, both=True):
    """step by step; lanes in turn pick the remaining entry whose bank pairs are least loaded in this step"""
    rem = [[tuple(e) for e in r] for r in rs]
    out = [[None] * K for _ in rs]
    for k in range(K):
        load_a, load_b = {}, {}
        # lanes with the fewest remaining choices first
        for lane in sorted(range(len(rs)), key=lambda i: len(rem[i])):
            if not rem[lane]:
                continue
            # must place ALL remaining entries in the remaining K - k steps: a lane with as many entries as steps must pick now
            slack = (K - k) - len(rem[lane])
            best, bc = None, None
            for idx, e in enumerate(rem[lane]):
                a, b = (e[0] * S + e[1]), (e[0] * S + 14 + e[2])
                ca = len(load_a.get(a % 32, set()) - {a})
                cb = len(load_b.get(b % 32, set()) - {b}) if both else 0
                c = ca + cb
                if bc is None or c < bc:
                    best, bc = idx, c
            if slack > 0 and bc > 0 and rng.random() < 0.0:
                continue
            e = rem[lane].pop(best)
            out[lane][k] = e
            a, b = (e[0] * S + e[1]), (e[0] * S + 14 + e[2])
            load_a.setdefault(a % 32, set()).add(a)
            load_b.setdefault(b % 32, set()).add(b)
    return out


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    res = {}
    for S in (18, 19):
        base = greedy = free = 0
        for _ in range(n_tiles):
            rows = make_tile()
            for rs, K in slices_of(rows):
                for h in (0, 32):
                    half = rs[h:h + 32]
                    base += cost_half(pad(half, K), S)
                    greedy += cost_half(greedy_seq(half, K, S), S)
                    free += 2 * K  # conflict-free: one cycle per lookup instruction and half-wave
        res[S] = (base, greedy, free)
        print(f"stride {S}: file order {base / free:.2f} cycles per lookup group, sequential greedy {greedy / free:.2f} (conflict-free = 1.00)")


if __name__ == "__main__":
    main()


def step_cost(col, S):
    """cycles of the two lookup instructions of one step: col = list of 32 entries (or None)"""
    tot = 0
    for which in (0, 1):
        per_bank = {}
        for e in col:
            addr = (NSLOT * S + (0 if which == 0 else 14)) if e is None else (e[0] * S + (e[1] if which == 0 else 14 + e[2]))
            per_bank.setdefault(addr % 32, set()).add(addr)
        tot += max(len(v) for v in per_bank.values())
    return tot


def local_search(ent, S, rounds=3):
    """after the greedy: swap two entries of ONE lane between two steps whenever the two steps' cycles go down"""
    n, K = len(ent), len(ent[0])
    cols = [[ent[l][k] for l in range(n)] for k in range(K)]
    costs = [step_cost(c, S) for c in cols]
    for _ in range(rounds):
        improved = False
        for k1 in sorted(range(K), key=lambda k: -costs[k]):
            if costs[k1] <= 2:
                continue
            for lane in range(n):
                for k2 in range(K):
                    if k2 == k1 or cols[k1][lane] is None or cols[k2][lane] is None:
                        continue
                    a, b = cols[k1][lane], cols[k2][lane]
                    cols[k1][lane], cols[k2][lane] = b, a
                    c1, c2 = step_cost(cols[k1], S), step_cost(cols[k2], S)
                    if c1 + c2 < costs[k1] + costs[k2]:
                        costs[k1], costs[k2] = c1, c2
                        improved = True
                    else:
                        cols[k1][lane], cols[k2][lane] = a, b
        if not improved:
            break
    return sum(costs)


def main2():
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for S in (18, 19):
        base = greedy = ls = free = 0
        for _ in range(n_tiles):
            rows = make_tile()
            for rs, K in slices_of(rows):
                for h in (0, 32):
                    half = rs[h:h + 32]
                    base += cost_half(pad(half, K), S)
                    g = greedy_seq(half, K, S)
                    greedy += cost_half(g, S)
                    ls += local_search(g, S)
                    free += 2 * K
        print(f"stride {S}: file order {base / free:.2f}, greedy {greedy / free:.2f}, greedy + swaps {ls / free:.2f}")


if __name__ == "__main__" and len(sys.argv) > 2:
    main2()


def greedy_par(rs, K, S, rounds=4):
    """the parallel form a GPU would run: in every round all unplaced lanes propose their cheapest remaining entry given the loads
    of the lanes already placed in this step; of the proposers that share a bank pair the lowest lane is placed, the others try
    again; after the last round whoever is left is placed as proposed"""
    rem = [[tuple(e) for e in r] for r in rs]
    out = [[None] * K for _ in rs]
    for k in range(K):
        load_a, load_b = {}, {}
        todo = [l for l in range(len(rs)) if rem[l]]
        for rd in range(rounds):
            props = {}
            for lane in todo:
                best, bc = None, None
                for idx, e in enumerate(rem[lane]):
                    a, b = (e[0] * S + e[1]), (e[0] * S + 14 + e[2])
                    c = len(load_a.get(a % 32, set()) - {a}) + len(load_b.get(b % 32, set()) - {b})
                    if bc is None or c < bc:
                        best, bc = idx, c
                props[lane] = best
            last = rd == rounds - 1
            won_a, won_b, placed = set(), set(), []
            for lane in todo:  # ascending lane: the lowest proposer of a bank pair wins it
                e = rem[lane][props[lane]]
                a, b = (e[0] * S + e[1]) % 32, (e[0] * S + 14 + e[2]) % 32
                if last or (a not in won_a and b not in won_b):
                    won_a.add(a); won_b.add(b)
                    placed.append(lane)
            for lane in placed:
                e = rem[lane].pop(props[lane])
                out[lane][k] = e
                a, b = (e[0] * S + e[1]), (e[0] * S + 14 + e[2])
                load_a.setdefault(a % 32, set()).add(a)
                load_b.setdefault(b % 32, set()).add(b)
            todo = [l for l in todo if l not in placed]
            if not todo:
                break
    return out


def main3():
    n_tiles = 1
    for S in (18, 19):
        for rounds in (2, 4, 8):
            base = par = free = 0
            rows = make_tile()
            for rs, K in slices_of(rows):
                for h in (0, 32):
                    half = rs[h:h + 32]
                    base += cost_half(pad(half, K), S)
                    par += cost_half(greedy_par(half, K, S, rounds), S)
                    free += 2 * K
            print(f"stride {S} rounds {rounds}: file order {base / free:.2f}, parallel proposals {par / free:.2f}")


if __name__ == "__main__" and len(sys.argv) > 3:
    main3()


def greedy_par_w(rs, K, S, rounds=4, window=99):
    rem = [[tuple(e) for e in r] for r in rs]
    out = [[None] * K for _ in rs]
    for k in range(K):
        load_a, load_b = {}, {}
        todo = [l for l in range(len(rs)) if rem[l]]
        for rd in range(rounds):
            props = {}
            for lane in todo:
                best, bc = None, None
                for idx, e in enumerate(rem[lane][:window]):
                    a, b = (e[0] * S + e[1]) % 32, (e[0] * S + 14 + e[2]) % 32
                    c = load_a.get(a, 0) + load_b.get(b, 0)   # (counts, not distinct addresses: what the GPU version does)
                    if bc is None or c < bc:
                        best, bc = idx, c
                props[lane] = best
            last = rd == rounds - 1
            won_a, won_b, placed = set(), set(), []
            for lane in todo:
                e = rem[lane][props[lane]]
                a, b = (e[0] * S + e[1]) % 32, (e[0] * S + 14 + e[2]) % 32
                if last or (a not in won_a and b not in won_b):
                    won_a.add(a); won_b.add(b)
                    placed.append(lane)
            for lane in placed:
                e = rem[lane].pop(props[lane])
                out[lane][k] = e
                a, b = (e[0] * S + e[1]) % 32, (e[0] * S + 14 + e[2]) % 32
                load_a[a] = load_a.get(a, 0) + 1
                load_b[b] = load_b.get(b, 0) + 1
            todo = [l for l in todo if l not in placed]
            if not todo:
                break
    return out


def main4():
    rows = make_tile()
    sl = slices_of(rows)
    for S in (18,):
        for rounds, window in ((4, 99), (4, 4), (4, 3), (3, 99), (3, 4), (2, 4), (6, 99)):
            base = par = free = 0
            for rs, K in sl:
                for h in (0, 32):
                    half = rs[h:h + 32]
                    base += cost_half(pad(half, K), S)
                    par += cost_half(greedy_par_w(half, K, S, rounds, window), S)
                    free += 2 * K
            print(f"stride {S} rounds {rounds} window {window}: file order {base / free:.2f}, parallel proposals {par / free:.2f}")


if __name__ == "__main__" and len(sys.argv) > 4:
    main4()
