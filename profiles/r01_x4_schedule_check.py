#!/usr/bin/env python3
"""Offline check of the static schedule of fa_bf16_x4_kernel.h: rebuilds the slot sequence, the VALU unit list and the
cost-weighted dealing table exactly as the constexpr code does, and verifies every dependency (a pack is dealt out before
the first MFMA that reads its fragment; the lane maxima of a block start after its last K.Q^T).  Run after any change to
x4_slot / x4_make_units / x4_weight_end."""
def pv(blk, j):
    if j == 2: return (2, blk, 0)
    if j == 5: return (2, blk, 1)
    return (1, blk, j if j < 2 else j - 1)
def slot(i):
    if i < 8: return (0, i % 2, i // 2)
    if i < 14: return pv(0, i - 8)
    if i < 18: return (0, 2 + (i - 14) % 2, (i - 14) // 2)
    if i < 24: return pv(1, i - 18)
    if i < 28: return (0, 2 + (i - 24) % 2, 2 + (i - 24) // 2)
    if i < 34: return pv(2, i - 28)
    return pv(3, i - 34)
def build(opt):
    units, pending = [], None
    for b in range(4):
        for e in range(16):
            units.append(('exp', b, e, 12))
            if e == 1 and pending is not None: units.append(pending); pending = None
            if e == 9: units.append(('pack', b, 0, 16))
        pending = ('pack', b, 1, 16)
    if opt:
        units.append(pending)
        return units
    for b in range(4):
        for m, c in ((0, 12), (1, 12), (2, 8)):
            units.append(('max', b, m, c))
            if b == 0 and m == 1 and pending is not None: units.append(pending); pending = None
    units.append(('test', 0, 0, 20))
    return units
w = [0]
for i in range(40): w.append(w[-1] + (1 if slot(i)[0] == 2 else 2))
for opt, wend in ((False, w[40]), (True, w[37])):
    units = build(opt)
    tot = sum(u[3] for u in units)
    cum = [0]
    for u in units: cum.append(cum[-1] + u[3])
    ub = []
    for i in range(41):
        target = tot * min(w[i], wend) // wend + 6
        n = 0
        while n < len(units) and cum[n + 1] <= target: n += 1
        ub.append(n)
    ub[40] = len(units)
    ok = True
    lastqk = {}
    for i in range(40):
        k, b, idx = slot(i)
        if k == 0: lastqk[b] = i
        if k in (1, 2):
            f = (0 if idx < 2 else 1) if k == 1 else idx
            need = [j for j, u in enumerate(units) if u[0] == 'pack' and u[1] == b and u[2] == f][0]
            if ub[i] <= need: print('VIOLATED: slot', i, 'needs unit', need, 'dealt', ub[i]); ok = False
            elif need >= ub[i - 1]: print('  note: slot', i, 'reads a fragment packed in the slot before it -> s_nop 1 (x4_needs_pad)')
    for j, u in enumerate(units):
        if u[0] == 'max' and u[2] == 0:
            s0 = [i for i in range(40) if ub[i] <= j < ub[i + 1]][0]
            if s0 < lastqk[u[1]] + 5: print('VIOLATED: maxima of block', u[1], 'at slot', s0, 'last K.Q^T slot', lastqk[u[1]]); ok = False
    print('optimistic' if opt else 'rescaled  ', len(units), 'units', tot, 'issue cycles, table', ub, 'OK' if ok else 'BROKEN')
