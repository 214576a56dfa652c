"""ds_read_b128 bank-conflict model of the conv3x3 k-loop (MI355X_MICROARCH.md, LDS table): a wave64 b128 read is served in
four 16-lane groups, one LDS cycle per group when the group's 16 addresses fall on 16 different 16-B slots of the 256-B bank row,
one extra cycle per extra distinct address on a busy slot.  Prints the conflict cycles per read for a swizzle candidate."""
import itertools, sys

GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]

def read_cycles(addrs):
    tot = 0
    for g in GROUPS:
        by_slot = {}
        for l in g:
            a = addrs[l]
            by_slot.setdefault((a >> 4) & 15, set()).add(a)
        tot += max(len(v) for v in by_slot.values())
    return tot

def conv_conflicts(W, TH, CB, S, MW, WM, key, zero_cols=1, verbose=False):
    """key(row, col) -> swizzle of a staged position (col == W: the zero position)."""
    WOUT = W // S
    SPP = CB // 16
    RS = (W + zero_cols) * CB
    MT = TH * WOUT
    KS = CB // 32
    total = reads = 0
    worst = 0
    for wm in range(WM):
        for i in range(MW):
            for dh in range(3):
                for dw in range(3):
                    for ks in range(KS):
                        addrs = []
                        for lane in range(64):
                            r, h = lane & 31, lane >> 5
                            m = min((wm * MW + i) * 32 + r, MT - 1)
                            ho, wo = divmod(m, WOUT)
                            row = ho * S + dh
                            col = wo * S + dw - 1
                            if col < 0:
                                col = W
                            c = 2 * ks + h
                            k = key(row, col)
                            addrs.append(row * RS + col * CB + (((c ^ k) & (SPP - 1)) << 4) + ((c & ~(SPP - 1)) << 4))
                        cyc = read_cycles(addrs)
                        total += cyc; reads += 1
                        worst = max(worst, cyc)
    return total / reads, worst

if __name__ == "__main__":
    cfgs = {"L1": (80, 8, 64, 1, 5, 4), "L2": (40, 8, 128, 1, 5, 4), "L3": (20, 8, 256, 1, 5, 1), "L4": (10, 17, 256, 1, 6, 1),
            "L2A": (80, 4, 64, 2, 3, 2), "L3A": (40, 4, 128, 2, 3, 1), "L4A": (20, 8, 128, 2, 3, 1)}
    for name, (W, TH, CB, S, MW, WM) in cfgs.items():
        SPP = CB // 16
        SWF = min(SPP, 16); SWSH = {4: 2, 8: 1}.get(SPP, 0)
        cur = lambda row, col: (col >> SWSH) & (SWF - 1)
        print(name, "current: mean cycles/read %.2f (ideal 4), worst %d" % conv_conflicts(W, TH, CB, S, MW, WM, cur))


def search_R(W, TH, CB, S, MW, WM, zero_cols=1, iters=4000, seed=0, pfun=None):
    """Local search for a per-staged-row XOR table R (key = P(col) ^ R[row]) that minimises the k-loop's LDS cycles."""
    import random
    rnd = random.Random(seed)
    SPP = CB // 16
    SWF = min(SPP, 16); SWSH = {4: 2, 8: 1}.get(SPP, 0)
    RIN = (TH - 1) * S + 3
    P = pfun or (lambda col: (col >> SWSH) & (SWF - 1))
    def cost(R):
        return conv_conflicts(W, TH, CB, S, MW, WM, lambda row, col: P(col) ^ R[row], zero_cols)[0]
    best = [0] * RIN
    bc = cost(best)
    for it in range(iters):
        cand = list(best)
        for _ in range(rnd.choice((1, 1, 2))):
            cand[rnd.randrange(RIN)] = rnd.randrange(SWF)
        c = cost(cand)
        if c <= bc:
            best, bc = cand, c
        if bc <= 4.0:
            break
    return bc, best


G_OF_R = {}
for _g, _lst in enumerate(GROUPS[:2]):
    for _j, _r in enumerate(_lst):
        G_OF_R[_r] = (_g, _j)


def grid_reads(W, TH, CB, MW, WM, GR, GC, rmask, rshift, csh, cmask):
    """GRID lane assignment (conv3x3.hip, block mode): 16-lane read group (wave row wm, M-tile i, group g) = the GR x GC block at
    block row 2*wm + g, block column i; image = rows of W positions + one trailing zero position (col -1 of a row is read
    from that row's own trailing position)."""
    SPP = CB // 16
    RS = (W + 1) * CB
    KS = CB // 32
    tot = n = worst = 0
    for wm in range(WM):
        for i in range(MW):
            for dh in range(3):
                for dw in range(3):
                    for ks in range(KS):
                        addrs = []
                        for lane in range(64):
                            r, h = lane & 31, lane >> 5
                            g, j = G_OF_R[r]
                            row = (2 * wm + g) * GR + j // GC + dh
                            col = i * GC + j % GC + dw - 1
                            key = ((row & rmask) << rshift) ^ ((col >> csh) & cmask)
                            pcol = W if col < 0 else col
                            c = 2 * ks + h
                            addrs.append(row * RS + pcol * CB + (((c ^ key) & (SPP - 1)) << 4) + ((c & ~(SPP - 1)) << 4))
                        cyc = read_cycles(addrs)
                        tot += cyc; n += 1; worst = max(worst, cyc)
    return tot / n, worst


def dense_reads(W, TH, CB, MW):
    """DENSE lane assignment: lanes enumerate the padded-width tile (W real columns + the zero column), key = index & 15."""
    SPP = CB // 16
    assert SPP == 16
    KS = CB // 32
    tot = n = worst = 0
    for i in range(MW):
        for dh in range(3):
            for dw in range(3):
                for ks in range(KS):
                    addrs = []
                    for lane in range(64):
                        r, h = lane & 31, lane >> 5
                        lm = min(32 * i + r, TH * (W + 1) - 1) + dh * (W + 1) + dw - 1 + 1   # +1: the leading zero position
                        c = 2 * ks + h
                        addrs.append(lm * CB + ((c ^ ((lm - 1) & 15)) << 4))
                    cyc = read_cycles(addrs)
                    tot += cyc; n += 1; worst = max(worst, cyc)
    return tot / n, worst


# ---- 16x16x32 MFMA operand reads: lane l holds position l & 15 of a 16-position tile and 16-B chunk (4 * s + (l >> 4)) ----------
def m16_grid(W, TH, CB, MW, WM, GR, GC, rmask, rshift, csh, cmask, lead):
    SPP = CB // 16
    RS = (W + 1) * CB
    KS32 = CB // 64
    img0 = CB if lead else 0
    tot = n = worst = 0
    for wm in range(WM):
        for t in range(2 * MW):
            br, bc = 2 * wm + t // MW, t % MW
            for dh in range(3):
                for dw in range(3):
                    for s in range(KS32):
                        addrs = []
                        for lane in range(64):
                            p, q = lane & 15, lane >> 4
                            row = br * GR + p // GC + dh
                            col = bc * GC + p % GC + dw - 1
                            key = ((row & rmask) << rshift) ^ ((col >> csh) & cmask)
                            if lead:
                                pos = row * (W + 1) + col
                            else:
                                pos = row * (W + 1) + (W if col < 0 else col)
                            c = 4 * s + q
                            addrs.append(img0 + pos * CB + (((c ^ key) & (SPP - 1)) << 4) + ((c & ~(SPP - 1)) << 4))
                        cyc = read_cycles(addrs)
                        tot += cyc; n += 1; worst = max(worst, cyc)
    return tot / n, worst


def m16_dense(W, TH, MT16, K, perm):
    """DENSE with 16-position tiles: lane p of tile t owns padded-image position 16 t + perm[p]; swizzle key K[(index) & 15]."""
    CB = 256
    tot = n = worst = 0
    for t in range(MT16):
        for dh in range(3):
            for dw in range(3):
                for s in range(4):
                    addrs = []
                    for lane in range(64):
                        p, q = lane & 15, lane >> 4
                        lm = 16 * t + perm[p] + dh * (W + 1) + dw - 1
                        c = 4 * s + q
                        addrs.append((lm + 1) * CB + ((c ^ K[lm & 15]) << 4))
                    cyc = read_cycles(addrs)
                    tot += cyc; n += 1; worst = max(worst, cyc)
    return tot / n, worst


# ---- stride-2 shapes on a PLANAR image with block lane orders (round 3) -----------------------------------------------------------
def s2_grid_conflicts(W, TH, CB, MW, WM, GR, GC, key, verbose=False):
    """Stride-2 3x3 conv, 32x32x16 MFMA (a lane = one output position, 16-B chunk 2 ks + h): staged row = [even columns | odd columns |
    zero position], lanes of read group g of M-tile i own the GR x GC block at block row wm, block column 2 i + g of the output tile.
    key(staged row, planar position) -> XOR on the chunk index.  Returns mean / worst LDS cycles per ds_read_b128 (ideal 4)."""
    WOUT, SPP, KS = W // 2, CB // 16, CB // 32
    RS = (W + 1) * CB
    NBC = WOUT // GC
    assert WOUT % GC == 0 and TH == WM * GR
    def planar(col):                      # column -1 and column W: the row's trailing zero position
        if col < 0 or col >= W:
            return W
        return (col & 1) * (W // 2) + (col >> 1)
    def jrel(r):
        g = 1 if (4 <= r < 12 or 16 <= r < 20 or r >= 28) else 0
        j = r - (0 if r < 4 else 4 if r < 12 else 8 if r < 20 else 12 if r < 28 else 16)
        return g, j
    total = reads = worst = 0
    for wm in range(WM):
        for i in range(MW):
            for dh in range(3):
                for dw in range(3):
                    for ks in range(KS):
                        addrs = []
                        for lane in range(64):
                            r, h = lane & 31, lane >> 5
                            g, j = jrel(r)
                            bc = min(2 * i + g, NBC - 1)
                            ho, wo = wm * GR + j // GC, bc * GC + j % GC
                            row, P = 2 * ho + dh, planar(2 * wo + dw - 1)
                            c = 2 * ks + h
                            addrs.append(row * RS + P * CB + (((c ^ key(row, P)) & (SPP - 1)) << 4))
                        cyc = read_cycles(addrs)
                        total += cyc; reads += 1; worst = max(worst, cyc)
    return total / reads, worst


def s2_report():
    # (W, TH, CB, MW, WM, GR, GC): L2A 32->64 ch W 80 (64-B positions), L3A 64->128 W 40, L4A 128->256 W 20 in 64-channel chunks (128 B)
    shapes = {"L2A": (80, 4, 64, 3, 2, 2, 8), "L3A": (40, 4, 128, 3, 1, 4, 4), "L4A": (20, 8, 128, 3, 1, 8, 2)}
    for name, (W, TH, CB, MW, WM, GR, GC) in shapes.items():
        SPP = CB // 16
        if SPP == 4:    # 4 positions per bank row: P mod 4 picks the quarter, the 2-bit key = (row pair bit, (P >> 2) & 1)
            key = lambda row, P: ((((row >> 1) & 1) << 1) | ((P >> 2) & 1))
        elif GC == 4:   # 2 positions per bank row: P & 1 picks the half, 3-bit key = (row pair & 3, (P >> 1) & 1)
            key = lambda row, P: ((((row >> 1) & 3) << 1) | ((P >> 1) & 1))
        else:           # 8 x 2 blocks: the two columns are the two halves, 3-bit key = row pair & 7
            key = lambda row, P: (row >> 1) & 7
        print(name, "planar image, %d x %d blocks: mean cycles/read %.2f, worst %d" % ((GR, GC) + s2_grid_conflicts(W, TH, CB, MW, WM, GR, GC, key)))


# ---- stride-2 shapes, planar image, 16x16x32 MFMA (round 4) ------------------------------------------------------------------------
def s2_m16_conflicts(W, TH, CB, WM, GR, GC, key, split_rows=True):
    """Stride-2 3x3 conv with v_mfma_f32_16x16x32_bf16: lane l = position l & 15 of a 16-position tile (= one GR x GC block of the
    output tile: block row wm, block column t) and 16-B chunk 4 s + (l >> 4).  A read group carries chunk q for tile positions
    {0-3, 12-15} and chunk q + 1 for {4-11}; with `split_rows` those two halves are the upper and the lower half of the block's rows
    (positions 0-3, 12-15 -> rows [0, GR/2), positions 4-11 -> rows [GR/2, GR)), so that the row part of the key separates them."""
    WOUT, SPP, KS32 = W // 2, CB // 16, CB // 64
    RS = (W + 1) * CB
    NBC = WOUT // GC
    assert WOUT % GC == 0 and TH == WM * GR
    def planar(col):
        if col < 0 or col >= W:
            return W
        return (col & 1) * (W // 2) + (col >> 1)
    def blockpos(p):
        if not split_rows:
            return p // GC, p % GC
        j = p if p < 4 else (p - 8 if p >= 12 else p - 4)           # index inside its half (0..7)
        half = 1 if 4 <= p < 12 else 0
        if split_rows == "parity":                                  # 8 x 2 blocks: even rows / odd rows
            return 2 * (j // GC) + half, j % GC
        return half * (GR // 2) + j // GC, j % GC
    total = reads = worst = 0
    for wm in range(WM):
        for t in range(NBC):
            for dh in range(3):
                for dw in range(3):
                    for s in range(KS32):
                        addrs = []
                        for lane in range(64):
                            p, q = lane & 15, lane >> 4
                            pr, pc = blockpos(p)
                            ho, wo = wm * GR + pr, t * GC + pc
                            row, P = 2 * ho + dh, planar(2 * wo + dw - 1)
                            c = 4 * s + q
                            addrs.append(row * RS + P * CB + (((c ^ key(row, P)) & (SPP - 1)) << 4))
                        cyc = read_cycles(addrs)
                        total += cyc; reads += 1; worst = max(worst, cyc)
    return total / reads, worst


def s2_m16_report():
    shapes = {"L2A": (80, 4, 64, 2, 2, 8), "L3A": (40, 4, 128, 1, 4, 4), "L4A": (20, 8, 128, 1, 8, 2)}
    for name, (W, TH, CB, WM, GR, GC) in shapes.items():
        SPP = CB // 16
        if SPP == 4:
            key = lambda row, P: ((((row >> 1) & 1) << 1) | ((P >> 2) & 1))
        elif GC == 4:
            key = lambda row, P: ((((row >> 1) & 3) << 1) | ((P >> 1) & 1))
        else:   # 8 x 2 blocks: 3-bit row key rotated left by one (key bit 0 = bit 2 of the row pair), halves = even / odd rows: two rows whose
                # keys differ in bit 0 alone are then 4 apart, i.e. in the same half, and the chunk difference of 1 between the halves cannot
                # map two of their positions onto one slot -- under the (dh = 2) shift of the row pair as well
            rot = lambda h: ((h << 1) & 7) | (h >> 2)
            key = lambda row, P: rot((row >> 1) & 7)
        for sr in ((False, True, "parity") if GR == 8 else (False, True)):
            print(name, "planar image, 16x16x32 operand reads, %d x %d blocks, halves = row halves: %s -> mean cycles/read %.2f, worst %d"
                  % ((GR, GC, sr) + s2_m16_conflicts(W, TH, CB, WM, GR, GC, key, sr)))
