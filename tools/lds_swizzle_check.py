"""Bank-conflict check of the swizzled [row][64 float] LDS images of the 16-row-tile attention kernels (csrc/attention16.hip).
16-byte slot p of row R holds logical slot p ^ f(R), f(4a + b) = 4b + a (the two 2-bit fields of R & 15 swapped).
ds_read_b128 is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); a group is conflict free when its 16
lanes touch 16 different slots of the 256-byte line.  Reads checked:
  rows-as-A   lane (m = l & 15, g = l >> 4) reads row R0 + m, logical slot 4 g + i        (K for S^T, Q / dO rows in the backward)
  cols-as-A   lane (m, g) reads row R0 + 4 g + r, logical slot m                          (V for O^T, Q / dO for dK^T / dV^T)
Run: python tools/lds_swizzle_check.py"""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def f(row):
    return 4 * (row & 3) + ((row >> 2) & 3)


def check(name, addr):
    worst = 1
    for grp in GROUPS:
        slots = [addr(l) for l in grp]
        worst = max(worst, max(slots.count(s) for s in set(slots)))
    print(f"{name:40s} {'conflict free' if worst == 1 else str(worst) + '-way conflict'}")
    return worst == 1


ok = True
for R0 in (0, 16, 192):
    for i in range(4):
        ok &= check(f"rows-as-A  R0={R0} i={i}", lambda l: ((4 * (l >> 4) + i) ^ f(R0 + (l & 15))) & 15)
    for r in range(4):
        ok &= check(f"cols-as-A  R0={R0} r={r}", lambda l: ((l & 15) ^ f(R0 + 4 * (l >> 4) + r)) & 15)
# the fill: LDS-DMA writes 1 KiB pieces linearly (lane l of piece p -> physical slot 64 p + l); its SOURCE is the logical slot
for phys in range(0, 64 * 4):
    row, ps = phys >> 4, phys & 15
    ls = ps ^ f(row)
    assert (ls ^ f(row)) == ps                      # involution: reader and filler agree
print("all conflict free" if ok else "CONFLICTS")
