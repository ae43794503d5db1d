"""LDS bank-conflict search for the z-marching split-operand kernel's operand reads (conv3d_x3_common.h: x3_row_stride,
x3_plane_stride, x3_group_stride, x3_pair_perm).

Model: a ds_read_b64 is served 32 lanes at a time = two lane quarters of 16 lanes x 8 contiguous bytes; 64 banks of 4 bytes.  A pass
is conflict-free iff no bank sees two DIFFERENT dword addresses.  Lane quarter q of K-slice s reads the operand pairs
(perm[s][2q], perm[s][2q+1]); quarters (0,1) and (2,3) share a pass.  The pairs of a packed slice are 8 consecutive tap-major
indices P = tap * ncgs + cg; their LDS offsets follow from the strides and the ring phase.  For every slice the best perfect
matching of its 8 pairs into 4 passes (summed over the three ring phases) is found by enumeration (105 matchings).

    python tools/x3_bank_search.py            # prints the table rows used in x3_pair_perm and the counts quoted in the header
"""
import sys

HY = 10


def conflicted(delta: int) -> bool:
    banks = {}
    for base in (0, delta):
        for w in range(32):
            a = base + 4 * w
            banks.setdefault((a // 4) % 64, set()).add(a)
    return max(len(v) for v in banks.values()) > 1


def offsets(ncgs, rs, plpad, cgpad, ring):
    pls = HY * rs + plpad
    cgs = 3 * pls * 8 + cgpad
    return [(P % ncgs) * cgs + (((ring + (P // ncgs) // 9) % 3) * pls + (((P // ncgs) // 3) % 3) * rs + (P // ncgs) % 3) * 8
            for P in range(27 * ncgs)]


def matchings(lst):
    if not lst:
        yield []
        return
    a = lst[0]
    for i in range(1, len(lst)):
        for m in matchings(lst[1:i] + lst[i + 1:]):
            yield [(a, lst[i])] + m


def best(ncgs, rs, plpad, cgpad):
    npairs = 27 * ncgs
    nsl = (npairs + 7) // 8
    offs = [offsets(ncgs, rs, plpad, cgpad, r) for r in range(3)]

    def cost(a, b):
        if a >= npairs or b >= npairs:
            return 0          # a padding pair reads its partner's address
        return sum(conflicted(offs[r][b] - offs[r][a]) for r in range(3))

    packed = permuted = 0
    rows = []
    for s in range(nsl):
        ps = list(range(8 * s, 8 * s + 8))
        packed += cost(ps[0], ps[2]) + cost(ps[1], ps[3]) + cost(ps[4], ps[6]) + cost(ps[5], ps[7])
        c, m = min(((sum(cost(a, b) for a, b in m), m) for m in matchings(ps)), key=lambda t: t[0])
        permuted += c
        rows.append([p - 8 * s for p in (m[0][0], m[1][0], m[0][1], m[1][1], m[2][0], m[3][0], m[2][1], m[3][1])])
    return packed, permuted, nsl * 4 * 3, rows


if __name__ == "__main__":
    for (ncgs, rs, plpad, cgpad) in ((1, 49, 0, 0), (1, 49, 9, 0), (2, 34, 0, 160), (3, 34, 0, 0), (3, 34, 0, 160), (4, 34, 0, 160),
                                     (5, 34, 0, 160), (6, 34, 0, 160)):
        packed, permuted, n, rows = best(ncgs, rs, plpad, cgpad)
        print(f"ncgs {ncgs} row stride {rs} plane pad {plpad} rec group pad {cgpad} B: conflicted passes as packed {packed}, permuted {permuted} of {n}")
        if "-v" in sys.argv or ncgs in (3, 5):
            print("   {" + ", ".join("{" + ",".join(map(str, r)) + "}" for r in rows) + "}")
