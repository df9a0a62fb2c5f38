#!/usr/bin/env python3
"""Enumerate the ds_read_b128 lane groups of the implicit-GEMM A-fragment read (MI355X_MICROARCH.md, LDS table: four
16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32); bank = 16-byte slot mod 16) for every tile shape and report the
extra LDS cycles per read with and without the row padding of igemm_kernel.h (row stride == BX * voxel pitch mod 16 slots
for the 16- / 8-wide tiles: the next row continues the previous row's slot sequence).  Pure arithmetic, no GPU: `python tools/lds_conflicts.py`."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def extra_cycles(slots_per_voxel, BX, row_slots):
    tot = 0
    for g in GROUPS:
        cnt = {}
        for lane in g:
            slot = ((lane // BX) * row_slots + (lane % BX) * slots_per_voxel) % 16
            cnt[slot] = cnt.get(slot, 0) + 1
        tot += max(cnt.values()) - 1
    return tot


if __name__ == "__main__":
    for name, P in (("fp32 (20 B x 4: 5 slots)", 5), ("bf16x6 (112 B: 7 slots)", 7), ("bf16 (48 B: 3 slots)", 3)):
        for KS in (3, 5):
            for BX in (32, 16, 8):
                HX = BX + KS - 1
                raw = HX * P
                padded = raw + ((BX * P - raw) % 16 if BX < 32 else 0)
                print(f"{name:28s} k{KS} BX={BX:2d}: row {raw:3d} slots -> extra cycles per 2 groups {extra_cycles(P, BX, raw)}; "
                      f"padded row {padded:3d} slots -> {extra_cycles(P, BX, padded)}")
