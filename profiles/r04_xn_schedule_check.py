#!/usr/bin/env python3
"""Offline model of the static step schedule of csrc/fa_bf16_xn_kernel.h (slots, VALU units, cost-weighted dealing table) for any
(D, NB, PF).  The header checks the same dependencies at compile time (xn_schedule_ok); this script exists to SEARCH: it prints, for
the NB = 2 shapes, which limits kWend of XShape keep every dependency, and shows the table of a given configuration.

    python profiles/r04_xn_schedule_check.py            # all instantiated shapes
    python profiles/r04_xn_schedule_check.py 64 2 3     # D NB PF: search the dealing limit

Round 4: PF = 3 (bf16 hi + bf16 lo): the sixteen v_dot2c of a block ('lo', one unit behind the second fragment's pack) and the four packs of
each lo fragment ('lopk') are separate units (a dot result may be read by another VALU instruction three wait states later at the earliest);
slots that hold dots take DOT_SLOT_EXTRA more units and emit their dots last (D >= 64).
"""
import sys

# dealing limits of the PF = 3 schedules: (D, NB, 3, optimistic) -> kWend (the FA_WEND_* defaults of csrc/fa_bf16_xn_kernel.h)
PF3_HI_FIRST = True   # the P.V group of a block runs (hi, hi, lo, lo) for PF = 3
DOT_SLOT_EXTRA = 1  # FA_PB2_DOT_SLOT_EXTRA of the header
WEND3 = {(64, 4, 3, True): 96, (64, 4, 3, False): 107, (32, 2, 3, True): 25, (32, 2, 3, False): 27, (64, 2, 3, True): 47, (64, 2, 3, False): 49,
         (128, 2, 3, True): 89, (128, 2, 3, False): 92}


def shape(D, NB, PF):
    KS, DB = D // 16, D // 32
    NV, NT = 2 * DB, (2 if PF >= 2 else 1)
    GRP = NT * (NV + 2)
    return dict(KS=KS, DB=DB, NV=NV, NT=NT, GRP=GRP, kSlots=NB * (KS + GRP), PF=PF)


def pv_group(S, blk, j):
    H = S['DB'] + 1
    run, w = divmod(j, H)
    if PF3_HI_FIRST and S['NT'] == 2 and S.get('PF') == 3:
        term, tt = divmod(run, 2)      # (tt0 hi)(tt1 hi)(tt0 lo)(tt1 lo)
    else:
        tt, term = divmod(run, S['NT'])
    if w == S['DB']:
        return (2, blk, tt, term)
    return (1, blk, tt * S['DB'] + w, term)


def slot(S, NB, i):
    G = S['GRP']
    if NB == 4:
        if i < 8: return (0, i % 2, i // 2, 0)
        if i < 8 + G: return pv_group(S, 0, i - 8)
        if i < 12 + G: return (0, 2 + (i - 8 - G) % 2, (i - 8 - G) // 2, 0)
        if i < 12 + 2 * G: return pv_group(S, 1, i - 12 - G)
        if i < 16 + 2 * G: return (0, 2 + (i - 12 - 2 * G) % 2, 2 + (i - 12 - 2 * G) // 2, 0)
        if i < 16 + 3 * G: return pv_group(S, 2, i - 16 - 2 * G)
        return pv_group(S, 3, i - 16 - 3 * G)
    if i < 2 * S['KS']: return (0, i % 2, i // 2, 0)
    return pv_group(S, (i - 2 * S['KS']) // G, (i - 2 * S['KS']) % G)


def units(NB, PF, opt):
    u, pend = [], []
    for b in range(NB):
        for e in range(16):
            u.append(('exp', b, e, 12))
            if pend and 1 <= e <= 5: u.append(pend.pop(0))
            if e == 9: u.append(('pack', b, 0, 16))
            if PF == 2 and e == 10: u.append(('lo', b, 0, 16))
            if PF == 2 and e == 11: u.append(('lo', b, 1, 16))
        if PF == 3:    # the block's sixteen dots in one unit behind the second fragment's pack, then the packs of each lo fragment
            pend = [('pack', b, 1, 16), ('lo', b, 3, 64), ('lopk', b, 1, 16), ('lopk', b, 3, 16)]
        else:
            pend = [('pack', b, 1, 16)] + ([('lo', b, 2, 16), ('lo', b, 3, 16)] if PF == 2 else [])
    if opt:
        return u + pend
    for b in range(NB):
        for m, c in ((0, 12), (1, 12), (2, 8)):
            u.append(('max', b, m, c))
            if m >= 1 and pend: u.append(pend.pop(0))
    u += pend
    u.append(('test', 0, 0, 20))
    return u


def table(D, NB, PF, opt, wend=None):
    S = shape(D, NB, PF)
    kS = S['kSlots']
    w = [0]
    for i in range(kS): w.append(w[-1] + (1 if slot(S, NB, i)[0] == 2 else 2))
    if wend is None:
        wend = (w[16 + 3 * S['GRP'] + S['DB'] + 1] if opt else w[kS]) if NB == 4 else None
    un = units(NB, PF, opt)
    tot = sum(x[3] for x in un)
    cum = [0]
    for x in un: cum.append(cum[-1] + x[3])
    ub = []
    for i in range(kS + 1):
        target = tot * min(w[i], wend) // wend + 6
        n = 0
        while n < len(un) and cum[n + 1] <= target: n += 1
        ub.append(n)
    ub[kS] = len(un)
    if PF == 3 and D >= 64 and DOT_SLOT_EXTRA > 0:   # a slot that holds a dot unit takes more (non-dot) units from the slot behind it
        for i in range(kS - 1):
            if not any(x[0] == 'lo' for x in un[ub[i]:ub[i + 1]]): continue
            for _ in range(DOT_SLOT_EXTRA):
                u = ub[i + 1]
                if u >= ub[i + 2] or u >= len(un) or un[u][0] in ('lo', 'lopk'): break
                ub[i + 1] += 1
    return S, un, ub, w


def check(D, NB, PF, opt, wend=None, verbose=False):
    S, un, ub, w = table(D, NB, PF, opt, wend)
    ok, lastqk, pads = True, {}, 0
    for i in range(S['kSlots']):
        k, b, idx, term = slot(S, NB, i)
        if k == 0:
            lastqk[b] = i
            continue
        f = idx // S['DB'] if k == 1 else idx
        need = [j for j, x in enumerate(un) if (x[0] == 'pack' and term == 0 and x[1] == b and x[2] == f) or
                (x[0] == ('lopk' if PF == 3 else 'lo') and term == 1 and x[1] == b and x[2] == 2 * f + 1)][0]
        if ub[i] <= need:
            ok = False
            if verbose: print('  VIOLATED: slot', i, slot(S, NB, i), 'needs unit', need, un[need], 'dealt', ub[i])
        elif need >= ub[i - 1]:
            pads += 1
    for j, x in enumerate(un):
        if x[0] == 'lopk':
            dots = [i for i, y in enumerate(un) if y[0] == 'lo' and y[1] == x[1]][0]
            sd = [i for i in range(S['kSlots']) if ub[i] <= dots < ub[i + 1]][0]
            sp = [i for i in range(S['kSlots']) if ub[i] <= j < ub[i + 1]][0]
            if j < dots + 1 or (D >= 64 and sp <= sd):   # (a slot's dots are emitted behind its other units)
                ok = False
                if verbose: print('  VIOLATED: packs of lo half', x, 'right behind their dots / in their slot')
        if x[0] == 'max' and x[2] == 0:
            s0 = [i for i in range(S['kSlots']) if ub[i] <= j < ub[i + 1]][0]
            if s0 < lastqk[x[1]] + 5:
                ok = False
                if verbose: print('  VIOLATED: maxima of block', x[1], 'at slot', s0, 'last K.Q^T slot', lastqk[x[1]])
    return ok, pads, (S, un, ub, w)


if __name__ == '__main__':
    if len(sys.argv) == 4:
        D, NB, PF = map(int, sys.argv[1:])
        S = shape(D, NB, PF)
        total_w = table(D, NB, PF, False, 10 ** 6)[3][-1]
        for opt in ((False,) if PF in (1, 2) else (False, True)):
            good = [we for we in range(8, total_w + 1) if check(D, NB, PF, opt, we)[0]]
            print('D', D, 'NB', NB, 'PF', PF, 'opt' if opt else 'rsc', 'half-slots in the step', total_w, 'valid kWend:', good)
        sys.exit(0)
    wends = {(128, 0, True): 58, (64, 0, True): 30, (32, 0, True): 16, (128, 0, False): 64, (64, 0, False): 34, (32, 0, False): 18,
             (128, 2, False): 99, (64, 2, False): 52, (32, 2, False): 29}
    wends.update(WEND3)
    for D, NB in ((64, 4), (32, 2), (64, 2), (128, 2)):
        for PF in (0, 1, 2, 3):
            for opt in (False, True):
                if opt and PF in (1, 2): continue
                we = wends[(D, NB, 3, opt)] if PF == 3 else None if NB == 4 else wends[(D, 2 if PF == 2 else 0, opt)]
                ok, pads, (S, un, ub, w) = check(D, NB, PF, opt, we, verbose=True)
                print(f'D={D} NB={NB} PF={PF} {"optimistic" if opt else "rescaled  "} {len(un)} units, {sum(x[3] for x in un)} issue cycles, '
                      f'{S["kSlots"]} slots, {pads} padded slots:', 'OK' if ok else 'BROKEN')
