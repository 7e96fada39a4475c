"""Search LDS tile strides (RS row, PS plane, CS cell-slot; in doubles) minimising bank-conflict
cycles of the pencil kernel's three access patterns on gfx950 (MI355X_MICROARCH.md, LDS):
ds_read_b64: 2 groups of 32 lanes, bank = double index mod 32;
ds_write_b64: 4 groups of 16 contiguous lanes, bank = double index mod 16."""
import itertools, sys

def cost(addrs_by_lane, group, mod):
    tot = 0
    for g0 in range(0, 64, group):
        banks = {}
        for l in range(g0, g0 + group):
            a = addrs_by_lane[l]
            if a is None: continue
            banks.setdefault(a % mod, set()).add(a)
        tot += max([len(v) for v in banks.values()], default=0)
    return tot

def evaluate(n, LPC, RS, PS, CS, nfields=3):
    n2 = n * n
    CPT = 64 // LPC
    pats = []
    for kind in ("z", "y", "x"):
        addrs = []
        for t in range(64):
            c, ab = divmod(t, LPC)
            if ab >= n2 or c >= CPT: addrs.append(None); continue
            a, b = ab % n, ab // n
            if kind == "z": ad = c * CS + b * RS + a
            elif kind == "y": ad = c * CS + b * PS + a
            else: ad = c * CS + b * PS + a * RS
            addrs.append(ad)
        pats.append(addrs)
    rd = [cost(p, 32, 32) for p in pats]
    wr = [cost(p, 16, 16) for p in pats]
    return rd, wr

def search(n, LPC):
    best = None
    for RS in (n, n + 1):
        for PS in range(n * RS, n * RS + 12):
            for CS in range(3 * n * PS, 3 * n * PS + 33):
                rd, wr = evaluate(n, LPC, RS, PS, CS)
                # weights: general kernel: z: 2n wr+2n rd ; y: 5n rd+5n wr ; x: 6n rd/wr (approx equal)
                score = sum(rd) * 2 + sum(wr) * 3 + (CS - 3 * n * n * n) * 0.001
                if best is None or score < best[0]:
                    best = (score, RS, PS, CS, rd, wr)
    return best

def evaluate_block(n, LPC, RS, PS, SLOT):
    """block kernel, cells that span waves (n^2 lanes per cell, 256 // LPC cells per 256-thread workgroup): all four waves count"""
    n2, CPT = n * n, 256 // LPC
    rd, wr = [0, 0, 0], [0, 0, 0]
    for w in range(4):
        for ki, kind in enumerate("zyx"):
            addrs = []
            for t in range(64 * w, 64 * w + 64):
                c, ab = divmod(t, LPC)
                if ab >= n2 or c >= CPT: addrs.append(None); continue
                a, b = ab % n, ab // n
                addrs.append(c * SLOT + (b * RS + a if kind == "z" else b * PS + a if kind == "y" else b * PS + a * RS))
            rd[ki] += cost(addrs, 32, 32)
            wr[ki] += cost(addrs, 16, 16)
    return rd, wr

def block_score(rd, wr):
    # sequential-tile pass (BlockPass::run): writes z x2, y x5, x x3; reads z x2, y x5, x x3
    return 2 * rd[0] + 5 * rd[1] + 3 * rd[2] + 2 * wr[0] + 5 * wr[1] + 3 * wr[2]

def search_block(n, LPC, tiles):
    """tiles = 2: two tiles per cell slot (BlockPass::PP); SLOT >= tiles * n * PS"""
    res = []
    for RS in (n, n + 1, n + 2):
        for PS in range(n * RS, n * RS + 20):
            for SLOT in range(tiles * n * PS, tiles * n * PS + 49):
                rd, wr = evaluate_block(n, LPC, RS, PS, SLOT)
                res.append((block_score(rd, wr), SLOT, RS, PS, rd, wr))
    res.sort(key=lambda x: (x[0], x[1]))
    return res[:3]

if __name__ == "__main__":
    if "--block" in sys.argv:   # BlockLayout specialisations of csrc/bp5_kernels.hpp
        for n, LPC, tiles in ((9, 81, 2), (3, 9, 2), (6, 36, 1)):
            for r in search_block(n, LPC, tiles):
                print(f"n={n} LPC={LPC} tiles={tiles}: weighted cycles {r[0]} SLOT={r[1]} RS={r[2]} PS={r[3]} read(z,y,x)={r[4]} write={r[5]}")
        sys.exit(0)
    for n in range(2, 10):
        n2 = n * n
        opts = sorted({min(64, n2), *( [32] if n2 <= 32 else []), *([16] if n2 <= 16 else [])})
        for LPC in opts:
            if LPC < n2: continue
            b = search(n, LPC)
            ideal_rd = 2 * 3; 
            print(f"n={n} LPC={LPC}: RS={b[1]} PS={b[2]} CS={b[3]} read-cycles(z,y,x)={b[4]} (ideal 2 each) write-cycles={b[5]} (ideal 4 each)")
