"""LDS bank-conflict model of the single-wave element kernel's pencil re-orientations (device/sumfact_fast.hpp: FastCfg::PS).

A point (c, b, a) = (x, y, z index) of a field group lives at 16-byte unit b * PS + a * M + c of its buffer; a stage reads or writes one
pencil per lane, i.e. for k = 0 .. M-1 one unit per lane with the lanes running over two of the three indices (first fastest).  A
ds_read_b128 / ds_write_b128 is served 16 lanes per pass, conflict-free when the 16 unit indices are distinct modulo 16; otherwise the
pass is repeated max-multiplicity times.  Prints LDS passes per (element, field group, three orientations) for the plane strides PS.

    python tools/lds_bank_model.py            # all M = 2 .. 8, PS = M*M .. M*M + 11
"""


def passes(units):
    tot = 0
    for g in range(4):
        cnt = {}
        for u in units[16 * g:16 * g + 16]:
            if u is not None:
                cnt[u % 16] = cnt.get(u % 16, 0) + 1
        if cnt:
            tot += max(cnt.values())
    return tot


def cost(M, PS, team_stride_units):
    team, ew = M * M, max(64 // (M * M), 1)

    def lane(l):
        t, r = divmod(l, team)
        return (t, r % M, r // M) if t < ew else None

    res = {}
    for name, f in (("x-pencils (lanes y,z)", lambda p, q, k: p * PS + q * M + k), ("y-pencils (lanes x,z)", lambda p, q, k: k * PS + q * M + p),
                    ("z-pencils (lanes x,y)", lambda p, q, k: q * PS + k * M + p)):
        tot = 0
        for k in range(M):
            units = []
            for l in range(64):
                L = lane(l)
                units.append(None if L is None else L[0] * team_stride_units + f(L[1], L[2], k))
            tot += passes(units)
        res[name] = tot
    return res


if __name__ == "__main__":
    NG = 2  # field groups of Diffusion3D (U = 4, F = 0): buffers A and B of NG groups, + 13 units of vertices per team
    for M in range(2, 9):
        print(f"M = {M} (order {M - 1}), {max(64 // (M * M), 1)} element(s) per wave; ideal {M * ((min(64, M * M * max(64 // (M * M), 1)) + 15) // 16)} passes per orientation")
        for PS in range(M * M, M * M + 12):
            c = cost(M, PS, 2 * NG * PS * M + 13)
            print(f"   PS = {PS:3d}: {sum(c.values()):4d}  {c}")
