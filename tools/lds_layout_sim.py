"""Bank-conflict model (MI355X_MICROARCH.md, LDS section) of the Kronecker Gram kernel's LDS traffic: the lift's 16-byte reads of
power-table entries (ds_read_b128: four groups of 16 lanes over 64 banks) for candidate table layouts - entry stride P, row
stride RB (doubles) - on the poly-3 / 6-state dictionary and a few others.  Prints LDS cycles per snapshot-pair chunk of one
workgroup (ideal: 4 waves x 3 factors x 4 groups = 48).  Round 6: the in-loop table build moved the rows from 3 to 4 entries;
at the dense row stride (40) the reads took 77 cycles against 57 before - the row stride 62 restores 57."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koopman_realizations_amd import poly_exponent_table  # noqa: E402

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]


def columns(nz, deg):
    return [[(v, int(e)) for v, e in enumerate(r) if e] for r in poly_exponent_table(nz, deg)] + [[]]


def lift_read_cycles(cols, nz, m, P, RB, dense_ids_D=0):
    nzm, nfull = nz + m, len(cols)

    def addr(side, v, e):
        return ((side * nzm + v) * dense_ids_D + e - 1) * P if dense_ids_D else (side * nzm + v) * RB + (e - 1) * P
    const = 2 * nzm * (dense_ids_D * P if dense_ids_D else RB)
    thr = []
    for t in range(256):
        f = []
        if t < 2 * nfull:
            f = [addr(t // nfull, v, e) for v, e in cols[t % nfull]]
        elif t < 2 * nfull + 9 and not dense_ids_D:          # the Kronecker weights, lifted as columns (round 6)
            w, cnt = t - 2 * nfull + 1, 0
            for x in range(4):
                for y in range(x, 4):
                    if cnt == w:
                        f = [addr(0, nz + q - 1, 1) for q in (x, y) if q > 0]
                    cnt += 1
        thr.append(f + [const] * (3 - len(f)))
    tot = 0
    for wave in range(4):
        for fi in range(3):
            for g in G128:
                banks = collections.defaultdict(set)
                for l in g:
                    a = thr[wave * 64 + l][fi]
                    for b in range(4):
                        banks[(2 * a + b) % 64].add(a)
                tot += max(len(v) for v in banks.values())
    return tot


if __name__ == "__main__":
    shapes = {"nz6 deg3": (columns(6, 3), 6), "nz6 deg2": (columns(6, 2), 6), "nz4 deg3": (columns(4, 3), 4), "nz3 deg4": (columns(3, 4), 3)}
    print("layout", *shapes)
    print("rounds 1-5 (dense ids, D entries per row, P=10)", *[lift_read_cycles(c, nz, 3, 10, 0, dense_ids_D=max(e for col in c for _, e in col)) for c, nz in shapes.values()])
    for P, RB in ((10, 40), (10, 62), (10, 58), (12, 50), (8, 34)):
        print(f"P={P} RB={RB}", *[lift_read_cycles(c, nz, 3, P, RB) for c, nz in shapes.values()])
