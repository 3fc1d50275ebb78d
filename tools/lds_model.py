#!/usr/bin/env python
"""Bank-conflict model of the LDS accesses of the fused InvBottleneck kernels (MI355X_MICROARCH.md, LDS table):
an instruction is serviced in fixed lane groups, one LDS cycle per group; every extra distinct address on a busy
bank inside a group costs one more cycle.

    ds_read_b128       4 groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, (+32), 64 banks
    ds_read_b64        2 groups of 32 lanes, 64 banks
    ds_read2_b64       two accesses, each 4 CONTIGUOUS groups of 16 lanes, 32 banks
    ds_write_b64       4 contiguous groups of 16 lanes, 32 banks
    ds_write_b128      8 contiguous groups of 8 lanes, 32 banks

    python tools/lds_model.py

Round 3: the model reproduces the 240 conflict cycles per wave and chunk that round 2's PMC pass charged
mb16_kernel with (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 31 %) and puts 192 of them on ONE instruction: hipcc
narrows the two half-used outer slots of a depthwise row to 8-byte reads and pairs them into ds_read2_b64, whose lane
groups are not the ones the quad -> row tables were built for.  keep_b128() (csrc/split3.h) keeps those slots whole.
The layouts of mbt_kernel and mbt_s2_kernel (mbtile_kernels.hip) were checked here before they first ran."""


def groups_b128():
    g = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
         list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    return g + [[lane + 32 for lane in x] for x in g]


G16 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
G8 = [list(range(i, i + 8)) for i in range(0, 64, 8)]
G32 = [list(range(0, 32)), list(range(32, 64))]


def cyc(addrs, width, groups, nbanks):
    """(LDS cycles, conflict-free cycles) of one wave instruction; addrs = byte address per lane."""
    tot = 0
    for g in groups:
        bank = {}
        for lane in g:
            for d in range(width // 4):
                dw = addrs[lane] // 4 + d
                bank.setdefault(dw % nbanks, set()).add(dw)
        tot += max(len(s) for s in bank.values())
    return tot, len(groups)


def total(instrs):
    a = b = 0
    for addrs, width, groups, nb in instrs:
        x, y = cyc(addrs, width, groups, nb)
        a += x
        b += y
    return a, b


# ------------------------------------------------------------------------------------------ mb16_kernel
def mb16():
    RS, PAIR = 22, 22 * 22 * 2 + 4
    tab = 0x7623673254014510

    def lane_geo(lane):
        q, strip = lane >> 2, lane & 3
        return (q >> 2) & 1, (tab >> (4 * q)) & 15, strip
    out = {}
    ins = []
    for R in range(8):
        for q in range(6):
            ad = []
            for lane in range(64):
                pair, rp, strip = lane_geo(lane)
                ad.append((pair * PAIR + (2 * rp * RS + strip * 4) * 2 + R * RS * 2 + 4 * q) * 4)
            ins.append((ad, 16, groups_b128(), 64))
    out['depthwise rows as 48 ds_read_b128'] = total(ins)
    ins = []
    for R in range(8):
        for q, sub in ((0, 2), (5, 0)):
            ad = []
            for lane in range(64):
                pair, rp, strip = lane_geo(lane)
                ad.append((pair * PAIR + (2 * rp * RS + strip * 4) * 2 + R * RS * 2 + 4 * q + sub) * 4)
            ins.append((ad, 8, G16, 32))
    out['the outer half-slots as 8 ds_read2_b64 (what hipcc built)'] = total(ins)
    ins = []
    for k in range(4):
        ad = []
        for lane in range(64):
            pair, rp, strip = lane_geo(lane)
            ad.append((pair * PAIR + ((2 * rp + 3) * RS + 4 + strip * 4) * 2 + [0, 4, RS * 2, RS * 2 + 4][k]) * 4)
        ins.append((ad, 16, G8, 32))
    out['depthwise result, 4 ds_write_b128'] = total(ins)
    ins = []
    for ks2 in range(2):
        for j in range(4):
            ad = []
            for lane in range(64):
                half, pl = lane >> 5, lane & 31
                cell = (((pl >> 4) + 3) * RS + (pl & 15) + 4) * 2
                ad.append(((8 * ks2 + 4 * half + j) * PAIR + cell) * 4)
            ins.append((ad, 8, G32, 64))
    out['project operands, 8 ds_read_b64'] = total(ins)
    return out


# ------------------------------------------------------------------------------------------ mbt_kernel
def mbt():
    RS, PAIR = 26, 22 * 26 * 2 + 4
    tab = 0x6732673245104510
    ins = []
    for R in range(8):
        for q in range(6):
            ad = []
            for lane in range(64):
                dq, strip = lane >> 2, lane & 3
                pair, rp = (dq >> 2) & 1, (tab >> (4 * dq)) & 15
                ad.append((pair * PAIR + (2 * rp * RS + strip * 4) * 2 + R * RS * 2 + 4 * q) * 4)
            ins.append((ad, 16, groups_b128(), 64))
    return {'depthwise rows as 48 ds_read_b128': total(ins)}


# ------------------------------------------------------------------------------------------ mbt_s2_kernel
def mbt_s2():
    RS, ODD = 44, 22
    PAIRC = 21 * RS
    ins = []
    for R in range(9):
        for plane, q in ((0, 0), (0, 1), (0, 2), (ODD, 0), (ODD, 1)):
            ad = []
            for lane in range(64):
                pair, rp, cp = lane >> 5, (lane >> 3) & 3, lane & 7
                ad.append((pair * PAIRC + (4 * rp + R) * RS + plane + 2 * cp + 2 * q) * 8)
            ins.append((ad, 16, groups_b128(), 64))
    out = {'depthwise rows (even / odd planes), 45 ds_read_b128': total(ins)}
    ins = []
    for a_ in range(2):
        ad = []
        for lane in range(64):
            pair, rp, cp = lane >> 5, (lane >> 3) & 3, lane & 7
            ad.append((pair * PAIRC + (2 * rp + a_) * 16 + 2 * cp) * 8)
        ins.append((ad, 16, G8, 32))
    out['depthwise result, 2 ds_write_b128'] = total(ins)
    ins = []
    for j in range(4):
        ad = []
        for lane in range(64):
            half, pl = lane >> 5, lane & 31
            ad.append(((4 * half + j) * PAIRC + pl) * 8)
        ins.append((ad, 8, G32, 64))
    out['project operands, 4 ds_read_b64'] = total(ins)
    return out


# ------------------------------------------------------------------------------------------ mbtb_kernel (bf16 path)
def mbtb():
    """The depthwise result of mbtb_kernel: one dword (two bf16 channels) per pixel, [16 pairs][264 dwords]; a row pair
    is 32 dwords and its two rows swap places when the row pair is odd (mbtile_bf16.hip)."""
    DP = 264
    tab = 0x6732673245104510
    out = {}
    for swz in (0, 1):
        ins = []
        for r in range(2):
            ad = []
            for lane in range(64):
                dq, strip = lane >> 2, lane & 3
                pair, rp = (dq >> 2) & 1, (tab >> (4 * dq)) & 15
                slot = (rp & 1) * 16 if swz else 0
                row_off = (slot if r == 0 else 16 - slot) if swz else 16 * r
                ad.append((pair * DP + rp * 32 + row_off + 4 * strip) * 4)
            ins.append((ad, 16, G8, 32))
        out['depthwise result, 2 ds_write_b128, rows %s' % ('swapped on odd row pairs' if swz else 'in order')] = total(ins)
    ins = []
    for wave in range(8):
        for j in range(4):
            ad = []
            for lane in range(64):
                half, pl = lane >> 5, lane & 31
                dcell = 32 * wave + (((pl >> 4) ^ (wave & 1)) << 4) + (pl & 15)
                ad.append(((4 * half + j) * DP + dcell) * 4)
            ins.append((ad, 4, G32, 32))
    out['project operands of the 8 waves, 32 ds_read_b32'] = total(ins)
    return out


if __name__ == '__main__':
    for name, fn in (('mb16_kernel', mb16), ('mbt_kernel', mbt), ('mbt_s2_kernel', mbt_s2), ('mbtb_kernel', mbtb)):
        print(name)
        for k, (c, base) in fn().items():
            print('  %-62s %4d LDS cycles per wave (conflict-free: %d)' % (k, c, base))
