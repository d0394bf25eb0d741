"""LDS placements for csrc/conv_wave.hip: the 16-byte slot of every (pixel, chunk) of a wave's quarter image (conv1's output, 16 channels = 2 chunks per
pixel and piece plane) and of its a2 image (32 channels = 4 chunks), found by simulated annealing so that

  hard   every ds_read_b128 of a conv2 / conv3 A operand touches 16 DISTINCT bank quadruples in each of the instruction's four 16-lane groups
         (MI355X_MICROARCH.md, LDS table: lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; a slot's bank quadruple
         is slot mod 16), for every (row tile, tap row) / (row tile, tap) the kernel reads;
  soft   the transposed conv1's 8-byte stores (16 lanes of one kb: 16 pixels, one chunk) fall on 8 slot classes mod 8 at most twice each (the floor:
         an 8-byte store group covers 16 of 32 banks), conv2's 4-byte epilogue stores at most twice per class, and the training copies' reads
         (lane L -> pixel L >> 1 or L >> 2 in order) conflict as little as possible.

    python tools/probe/wave_layout.py 5      -> CW_SLOT1 (64 entries: 32 pixel columns incl. the padding columns 25 .. 31), CW_SLOT2 (64)
    python tools/probe/wave_layout.py 7      -> 98 / 144 entries (stores of padding columns are masked there)

With rows in pixel order every such read takes 8 LDS cycles instead of 4 (d = 5: 8 extra cycles over the 8 read groups of conv2, 16 over conv3's 16)."""
import random
import sys

GL = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
      list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def solve(hard, soft, items, nslot, iters=600000, seed=0):
    """items -> distinct slots in [0, nslot).  hard: lists of items read together (cost 4 x (max multiplicity of slot mod 16 - 1));
    soft: (items, modulus, free) (cost max(0, max multiplicity - free))."""
    random.seed(seed)
    items = sorted(items)
    idx = {it: i for i, it in enumerate(items)}
    allg = [([idx[x] for x in g], 16, 1, 4.0) for g in hard] + [([idx[x] for x in g], m, f, 1.0) for g, m, f in soft]
    memb = [[] for _ in items]
    for gi, (g, m, f, w) in enumerate(allg):
        for x in set(g):
            memb[x].append(gi)
    slot = list(range(len(items)))
    random.shuffle(slot)
    used = {s: i for i, s in enumerate(slot)}

    def gcost(gi):
        g, m, f, w = allg[gi]
        cnt = {}
        for x in set(g):
            cnt[slot[x] % m] = cnt.get(slot[x] % m, 0) + 1
        return w * max(0, max(cnt.values()) - f)
    cost = [gcost(i) for i in range(len(allg))]
    tot = sum(cost)
    T, best = 2.0, (tot, list(slot))
    for _ in range(iters):
        if tot == 0:
            break
        x, ns = random.randrange(len(items)), random.randrange(nslot)
        os_ = slot[x]
        if ns == os_:
            continue
        y = used.get(ns, -1)
        slot[x] = ns
        if y >= 0:
            slot[y] = os_
        aff = set(memb[x] + (memb[y] if y >= 0 else []))
        newc = {gi: gcost(gi) for gi in aff}
        delta = sum(newc[gi] - cost[gi] for gi in aff)
        if delta <= 0 or random.random() < 2.718 ** (-delta / T):
            for gi, c in newc.items():
                cost[gi] = c
            tot += delta
            used[ns] = x
            if y >= 0:
                used[os_] = y
            else:
                del used[os_]
            if tot < best[0]:
                best = (tot, list(slot))
        else:
            slot[x] = os_
            if y >= 0:
                slot[y] = ns
        T = max(0.05, T * 0.99999)
    return best[0], {items[i]: best[1][i] for i in range(len(items))}


def problem(D):
    OW1, OW2, OW3 = D, D - 1, D - 2
    R1, R2, R3 = D * D, OW2 * OW2, OW3 * OW3
    T1, T2, T3 = (R1 + 15) // 16, (R2 + 15) // 16, (R3 + 15) // 16
    pad = D == 5                                                     # d = 5: the padding columns of conv1's last tile get slots of their own
    P1 = 16 * T1 if pad else R1
    hard1, hard2 = [], []
    for u in range(T2):
        for ky in range(2):
            for g in GL:
                s = []
                for l in g:
                    j, kb = l & 15, l >> 4
                    n = 16 * u + j if 16 * u + j < R2 else 0
                    y, x = divmod(n, OW2)
                    s.append(((y + ky) * OW1 + x + (kb >> 1), kb & 1))
                hard1.append(s)
    for u in range(T3):
        for tap in range(4):
            for g in GL:
                s = []
                for l in g:
                    j, kb = l & 15, l >> 4
                    n = 16 * u + j if 16 * u + j < R3 else 0
                    y, x = divmod(n, OW3)
                    s.append(((y + (tap >> 1)) * OW2 + x + (tap & 1), kb))
                hard2.append(s)
    soft1, soft2 = [], []
    for u in range(T1):
        for c in range(2):
            s = [(16 * u + j, c) for j in range(16) if 16 * u + j < P1]
            if len(s) > 1:
                soft1.append((s, 8, 2))
    for i in range((2 * R1 + 63) // 64):
        for g in GL:
            s = [((64 * i + l) >> 1, (64 * i + l) & 1) for l in g if 64 * i + l < 2 * R1]
            if len(s) > 1:
                soft1.append((s, 16, 1))
    for u in range(T2):
        for r in range(4):
            for hw in range(2):
                s = [(16 * u + 8 * hw + 4 * k + r, c) for k in range(2) for c in range(4) if 16 * u + 8 * hw + 4 * k + r < R2]
                if len(s) > 1:
                    soft2.append((s, 8, 2))
    for i in range((4 * R2 + 63) // 64):
        for g in GL:
            s = [((64 * i + l) >> 2, (64 * i + l) & 3) for l in g if 64 * i + l < 4 * R2]
            if len(s) > 1:
                soft2.append((s, 16, 1))
    items1 = set((p, c) for p in range(P1) for c in range(2))
    items2 = set((p, c) for p in range(R2) for c in range(4))
    n1 = 16 * ((len(items1) + 15) // 16)
    n2 = 16 * ((len(items2) + 15) // 16)
    return (hard1, soft1, items1, n1, P1, 2), (hard2, soft2, items2, n2, R2, 4)


def read_conflicts(hard, table, chunks):
    ex = 0
    for g in hard:
        cnt = {}
        for p, c in set(g):
            k = table[chunks * p + c] % 16
            cnt[k] = cnt.get(k, 0) + 1
        ex += max(cnt.values()) - 1
    return ex


if __name__ == "__main__":
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for name, (hard, soft, items, nslot, P, chunks) in zip(("SLOT1 (quarter image)", "SLOT2 (a2)"), problem(D)):
        best = None
        for seed in range(seeds):
            c, sol = solve(hard, soft, items, nslot, seed=seed)
            print(f"# d = {D} {name}: seed {seed} cost {c}", flush=True)
            if best is None or c < best[0]:
                best = (c, sol)
            if c == 0:
                break
        tab = [best[1][(p, c)] for p in range(P) for c in range(chunks)]
        ident = [chunks * p + c for p in range(P) for c in range(chunks)]
        print(f"# cost {best[0]}; extra LDS cycles over the {len(hard)} read groups: {read_conflicts(hard, tab, chunks)} (pixel order: {read_conflicts(hard, ident, chunks)}); {nslot} slots")
        print(f"CW{D}_{name.split()[0]} = {tab}", flush=True)
