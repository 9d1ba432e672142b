#!/usr/bin/env python3
"""Exhaustive bank-conflict check of the 64-byte-row LDS layouts of kp_conv_pair.hip (no GPU needed).

ds_read_b128 is serviced in four 16-lane groups (MI355X_MICROARCH.md, LDS): a group is conflict-free when its 16 lanes hit
16 distinct 16-byte slots of the 256-byte bank row.  Pixels: LDS row = patch pixel (18 columns per patch row), slot XORed
with (patch column >> 1) & 3; weights: LDS row = output channel, slot XORed with (row >> 2) & 3.  A fragment's 32 lanes of
one half are 2 patch rows x 16 columns (pixels) / 32 consecutive rows (weights)."""
PW = 18
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def pixels(f):
    bad = 0
    for row0 in range(18):
        for s in range(3):
            for g in GROUPS:
                for slot in range(4):
                    pos = set()
                    for fr in g:
                        row, col = row0 + (fr >> 4), (fr & 15) + s
                        pix = row * PW + col
                        pos.add(((pix % 4) * 4 + (slot ^ f(col))) % 16)
                    bad += len(pos) != 16
    return bad


def weights(f):
    bad = 0
    for base in range(0, 128, 32):
        for g in GROUPS:
            for slot in range(4):
                pos = {(((base + fr) % 4) * 4 + (slot ^ f(base + fr))) % 16 for fr in g}
                bad += len(pos) != 16
    return bad


if __name__ == '__main__':
    print('pixels  (col >> 1) & 3 :', pixels(lambda c: (c >> 1) & 3), 'conflicting groups')
    print('pixels  (col >> 2) & 3 :', pixels(lambda c: (c >> 2) & 3))
    print('pixels  no swizzle     :', pixels(lambda c: 0))
    print('weights (row >> 2) & 3 :', weights(lambda r: (r >> 2) & 3))
    print('weights (row >> 1) & 3 :', weights(lambda r: (r >> 1) & 3))
    print('weights no swizzle     :', weights(lambda r: 0))
