"""LDS bank conflicts of winograd4_s2.hip's access patterns, computed from the layout with the per-instruction lane groups and bank
moduli of MI355X_MICROARCH.md (section LDS): cycles per wave-instruction against the conflict-free count.  CPU only.
    python3 tools/probe/lds_conflicts_wino4s2.py [variant]"""
import sys
from collections import Counter

RAWP, PC = 577, 33
R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
R128 = R128 + [[l + 32 for l in g] for g in R128]
G2x32 = [list(range(0, 32)), list(range(32, 64))]
G4x16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
G8x8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def cycles(addr_floats, dwords, groups, mod):
    tot = 0
    for g in groups:
        banks, seen = Counter(), set()
        for l in g:
            for w in range(dwords):
                word = addr_floats[l] + w
                if word not in seen:
                    seen.add(word)
                    banks[word % mod] += 1
        tot += max(banks.values())
    return tot


VARIANT = 0                                     # 0: the shipped layout, 1: the first version (lane = kg + 4 ttx + 32 tty, slot ^ 4 kg)


def lanes(wv):
    for l in range(64):
        tt = 64 * wv + l
        if VARIANT == 1:
            yield l, tt & 3, (tt >> 2) & 7, tt >> 5        # lane, kg, ttx, tty
        else:
            yield l, tt & 3, ((tt >> 2) & 1) | (((tt >> 4) & 3) << 1), (((tt >> 3) & 1) << 1) | ((tt >> 6) & 1)


def swz(kg):
    return (kg << 2) if VARIANT == 1 else 2 * kg


def report():
    # transform reads of the raw patch: thread (kg, ttx, tty) reads float2 of pixel (4 tty + r, 4 ttx + c), k-quad kg >> 1, half kg & 1
    tot = ideal = tot2 = 0
    for wv in (0, 1):
        for r in range(5):
            for c in range(5):
                ad = [0] * 64
                for l, kg, ttx, tty in lanes(wv):
                    ad[l] = ((kg >> 1) * RAWP + (4 * tty + r) * PC + 4 * ttx + c) * 4 + 2 * (kg & 1)
                tot += cycles(ad, 2, G2x32, 64); ideal += 2; tot2 += cycles(ad, 2, G4x16, 32)
    print('transform reads as ds_read_b64   : %.2f cycles per instruction (conflict-free 2); as one access of the ds_read2_b64 hipcc emits: %.2f (conflict-free 4)' % (2.0 * tot / ideal, 2.0 * tot2 / ideal))
    # V stores: [pos][kg][16 tile slots][tile half][2 k-steps], slot = ((tty & 1) * 8 + ttx) ^ swz(kg)
    for wv in (0, 1):
        ad = [0] * 64
        for l, kg, ttx, tty in lanes(wv):
            slot = ((tty & 1) * 8 + ttx) ^ swz(kg)
            ad[l] = (kg * 16 + slot) * 4 + (tty >> 1) * 2
        print('V stores (ds_write_b64), wave %d  : %d cycles (conflict-free 4)' % (wv, cycles(ad, 2, G4x16, 32)))
    # fragment reads: lane (kgl = lane >> 4, ml = lane & 15) reads float4 at (kgl * 16 + (ml ^ (kgl << 2))) * 4
    ad = [((l >> 4) * 16 + ((l & 15) ^ swz(l >> 4))) * 4 for l in range(64)]
    print('fragment reads (ds_read_b128)    : %d cycles (conflict-free 4)' % cycles(ad, 4, R128, 64))
    # raw patch stores: thread t: pixel (t >> 5) * 16 + (t & 15) + 128 q, k-quad (t >> 4) & 1
    for wv in range(4):
        ad = [0] * 64
        for l in range(64):
            t = 64 * wv + l
            ad[l] = (((t >> 4) & 1) * RAWP + (t >> 5) * 16 + (t & 15)) * 4
        print('raw patch stores (ds_write_b128), wave %d: %d cycles (conflict-free 8)' % (wv, cycles(ad, 4, G8x8, 32)))
    # drain: ow[g * 272 + (yy * 4 + x) * 16 + co16] b32 stores, float4 read-back at lane * 4
    ad = [(l >> 4) * 272 + (l & 15) for l in range(64)]
    print('drain stores (ds_write_b32)      : %d cycles (conflict-free 2; up to 4 costs nothing)' % cycles(ad, 1, G2x32, 32))
    ad = [l * 4 for l in range(64)]
    print('drain read-back (ds_read_b128)   : %d cycles (conflict-free 4)' % cycles(ad, 4, R128, 64))


if __name__ == '__main__':
    VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    report()
