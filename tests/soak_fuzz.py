"""Randomized soak of the product dispatch on one GPU.  By hand (minutes):

    python tests/soak_fuzz.py --cases 800 --seed 7 [--out gpurun_out/soak.txt]

and, since round 5, a bounded slice of the same generator inside the suite the driver runs (tests/test_gpu_adversarial.py::test_soak_slice:
2 seeds x 60 cases, rows up to 4500 keys).

Every case draws (bh, n, d, causal, scale, dtype, data family) and compares FA_KERNEL_AUTO -- bf16 tensors with bf16 and with fp32
output, fp32 tensors -- with (a) the rung-0 kernel (one thread per query row, fp32) on every slab and (b) the FP64 ORACLE on sampled query
rows of one sampled slab (tests/adversarial.py: rows_f64), with the tolerances of tests/test_gpu_parity.py.  Lengths are drawn around the
tiling boundaries (multiples of 32 .. 512, +-1), the data from several families: N(0,1); wide logits (x3); planted dominant keys; a
constant V; zero Q; values at the bf16 / fp16 range ends for V; and the coherent-rounding families of tests/adversarial.py (constant-
component rows, few-valued rows, a broadcast token, quantised + offset: VERDICT r04 weak #1) at logit widths up to 89.  Exit code 1 on
the first mismatch or NaN.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flashattention_c_amd as fa  # noqa: E402
import adversarial as adv  # noqa: E402

TOL_F32 = 1e-3
TOL_ACC = 5e-4         # kernel="split" for bf16 tensors (hi + lo bf16 terms of P and Q')
TOL_PB2 = 2e-4         # kernel="pb2" (= FA_KERNEL_AUTO with an fp32 output): P as two bf16 terms; see tests/test_gpu_parity.py


def bf16_tol(scale, out_f32, causal, n):
    if not out_f32:
        return 2.5e-2
    if scale >= 0.5:
        return 1.2e-2
    return 1e-3 if (not causal and n >= 1000) else 4e-3


def draw_n(rng):
    kind = rng.integers(0, 4)
    if kind == 0:
        return int(rng.integers(1, 200))
    if kind == 1:
        base = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048, 4096]))
        return max(1, base * int(rng.integers(1, 5)) + int(rng.integers(-1, 2)))
    if kind == 2:
        return int(rng.integers(200, 3000))
    return int(rng.integers(3000, 9000))


COHERENT = ("const_two_keys", "const_many_keys", "few_valued", "broadcast_token", "quantised_offset")
V_OFFSET = 6 + len(COHERENT)          # family 11: V = offset + N(0, 1) (the fp32 default centres V; fp32 accumulation itself is relative to |v|)
N_FAMILIES = V_OFFSET + 1


def make_data(rng, g, family, bh, n, d, case_seed=0):
    q, k, v = (torch.randn(bh, n, d, generator=g) for _ in range(3))
    vmag = 1.0
    if family == 1:      # wide logits
        q *= 3.0
    elif family == 2:    # planted dominant keys
        for _ in range(min(8, n)):
            b, r, c = int(rng.integers(0, bh)), int(rng.integers(0, n)), int(rng.integers(0, n))
            k[b, c] = float(rng.uniform(8.0, 16.0)) * q[b, r] / q[b, r].norm()
    elif family == 3:    # constant V: the output is that constant whatever the weights
        v[:] = 1.25
    elif family == 4:    # zero Q: uniform weights
        q.zero_()
    elif family == 5:    # large V (round 3's fp16-P chain had to hand |v| >= 2^16 to the split kernel; two bf16 terms of P take any V)
        vmag = float(rng.choice([300.0, 7.0e4]))
        v *= vmag
    elif family == V_OFFSET:
        v += float(rng.choice([100.0, 1000.0, -2500.0]))
    elif family >= 6:    # coherent rounding residuals (tests/adversarial.py) at a logit width the round-4 guard let through
        width = float(rng.uniform(20.0, 89.0))
        q, k, v = (torch.from_numpy(t) for t in adv.make(COHERENT[family - 6], d, width, n=n, bh=bh, seed=case_seed))
    return q, k, v, vmag


def bf16_p_bound(q, k, v, rows, causal, scale, out_bf16):
    """The worst case of a bf16-P kernel on this slab, from the data (tests/adversarial.py: p_rounding_bound): every weight off by 2^-8
    with the worst signs, the mass the optimistic mix may flush (<= 2^-10 of a row), the output's own rounding, fp32 noise."""
    b, o = adv.p_rounding_bound(q, k, v, rows, causal, scale, rel=2.0 ** -8 + 2.0 ** -10)
    return b + (2.0 ** -8 * float(np.abs(o).max()) if out_bf16 else 0.0) + 1e-4


def run(cases=400, seed=1, out="", only=-1, max_n=0, verbose=True):
    """Returns (worst, routes): worst[key] = (error, tolerance, description).  Exits the process with code 1 on the first failure when run
    from the command line; raises AssertionError when imported (the pytest slice)."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    worst = {}
    routes = {}
    lines = []
    stats0 = fa.stats()

    def note(key, err, tol, desc):
        if err / tol > (worst[key][0] / worst[key][1] if key in worst else -1.0):
            worst[key] = (err, tol, desc)
        if not (err < tol):
            msg = f"FAIL {key}: {err:.3e} >= {tol:.1e}  {desc}"
            print(msg, flush=True)
            raise AssertionError(msg)

    def coverage(key, got, ref, qn, kn, vn, causal, scale, simple_tol, contract_tol, chain, vmag, desc, lse=False):
        """Rung 0 (fp32 arithmetic on the device) covers EVERY slab and row, where the fp64 oracle saw ~200 rows of one slab: unwritten rows
        and wrong tiles show up here.  It is itself a rounding fp32 chain, though: on coherent or wide inputs it can sit several 1e-3 off
        fp64 on a slab the sample did not look at (seed 51, case 48: 7.4e-3 between the two on 5-key rows at d = 128).  So a difference
        above the simple threshold is ADJUDICATED: the three rows where the two disagree most are recomputed in fp64, and the kernel
        under test must hold its own contract there (max(tolerance, the emulated chain's error on that row) when `chain`)."""
        diff = (got - ref).abs()
        if not lse:
            diff = diff.amax(dim=-1)
        worst_diff = float(diff.max()) / (1.0 if lse else vmag)
        if worst_diff < simple_tol:
            note(key, worst_diff, simple_tol, desc)
            return
        flat = torch.topk(diff.flatten().nan_to_num(nan=float("inf")), k=min(3, diff.numel())).indices.cpu().numpy()
        for idx in flat:
            sl, row = int(idx) // diff.shape[1], int(idx) % diff.shape[1]
            o64, l64 = adv.rows_f64(qn[sl], kn[sl], vn[sl], [row], causal, scale)
            want = l64 if lse else o64
            mine = got[sl, row].cpu().numpy()
            err = float(np.abs(mine - want).max()) / (1.0 if lse else vmag)
            tol = contract_tol
            if chain:
                oc, lc = adv.rows_f64(qn[sl], kn[sl], vn[sl], [row], causal, scale, chain=True)
                tol = max(tol, float(np.abs((lc if lse else oc) - want).max()) / (1.0 if lse else vmag))
            note(key + " -- adjudicated by fp64 at the rows of largest disagreement", err, tol, desc + f" slab {sl} row {row}: |kernel - rung 0| {worst_diff:.2e}")

    for case in range(cases):
        d = int(rng.choice([32, 64, 128]))
        n = draw_n(rng)
        if max_n:
            n = min(n, max_n)
        bh = int(rng.integers(1, 9)) if n > 3000 else int(rng.integers(1, 49))
        causal = bool(rng.integers(0, 2))
        scale = float(rng.choice([1.0, 0.5, d ** -0.5]))
        family = int(rng.integers(0, N_FAMILIES))
        if 6 <= family < V_OFFSET:
            scale = 1.0          # the families are built for a logit width at the reference's scale
            bh = min(bh, 8)
        g = torch.Generator(device="cpu").manual_seed(seed * 100003 + case)
        q, k, v, vmag = make_data(rng, g, family, bh, n, d, case_seed=seed * 1000 + case)
        desc = f"case {case} bh={bh} n={n} d={d} causal={int(causal)} scale={scale:.4g} family={family}"
        sb = int(rng.integers(0, bh))                                                    # the slab and the rows the fp64 oracle looks at
        rows = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, size=min(n, 192))]))
        pack_nh = int(rng.integers(1, 5)) if (case % 8 == 0 and d in (32, 64, 128)) else 0
        if only >= 0 and case != only:
            continue
        if only >= 0:
            qd, kd, vd = (t.to(dev) for t in (q, k, v))
            ref, lse_ref = fa.forward(qd, kd, vd, causal, scale=scale, kernel="naive", return_lse=True)
            o64, l64 = adv.rows_f64(q[sb].numpy(), k[sb].numpy(), v[sb].numpy(), rows, causal, scale)
            print(desc, "workspace", fa.workspace_bytes(bh, n, d, causal))
            for kern in ("auto", "split", "exact"):
                o, l = fa.forward(qd, kd, vd, causal, scale=scale, kernel=kern, return_lse=True)
                print(f"  {kern:6s} route {fa.last_forward_route()}  max|O - naive| {float((o - ref).abs().max()):.3e}  max|lse - naive| {float((l - lse_ref).abs().max()):.3e}"
                      f"  slab {sb} rows vs fp64: {np.abs(o[sb].cpu().numpy()[rows] - o64).max():.3e} / {np.abs(l[sb].cpu().numpy()[rows] - l64).max():.3e}")
            oc, lc = adv.rows_f64(q[sb].numpy(), k[sb].numpy(), v[sb].numpy(), rows, causal, scale, chain=True)
            print(f"  the fp32 FMA chain's own error on those rows: {np.abs(oc - o64).max():.3e} / {np.abs(lc - l64).max():.3e}")
            return worst, routes
        # ---- fp32 tensors
        qd, kd, vd = (t.to(dev) for t in (q, k, v))
        ref, lse_ref = fa.forward(qd, kd, vd, causal, scale=scale, kernel="naive", return_lse=True)
        res, lse = fa.forward(qd, kd, vd, causal, scale=scale, return_lse=True)
        r = fa.last_forward_route()
        routes[("f32", r)] = routes.get(("f32", r), 0) + 1
        if torch.isnan(res).any() or torch.isnan(ref).any():
            raise AssertionError("FAIL NaN fp32 " + desc)
        qs, ks, vs = q[sb].numpy(), k[sb].numpy(), v[sb].numpy()
        o64, l64 = adv.rows_f64(qs, ks, vs, rows, causal, scale)
        tol_o = tol_l = TOL_F32          # (also for the coherent families: the kernel works on centred keys -- observed <= 3e-4 there)
        ref_o = ref_l = TOL_F32          # what rung 0 (fp32 arithmetic itself) may be off by on this input
        if family in (1, 2) or 6 <= family < V_OFFSET:
            # wide or coherent logits: the reference's OWN arithmetic (a rounding fp32 FMA chain, flashattention.cu:236-252) may leave more
            # than 1e-3 against fp64 there; the general contract is max(1e-3, that) (tests/test_gpu_adversarial.py)
            oc, lc = adv.rows_f64(qs, ks, vs, rows, causal, scale, chain=True)
            ref_o = max(TOL_F32, float(np.abs(oc - o64).max()) / vmag)
            ref_l = max(TOL_F32, float(np.abs(lc - l64).max()))
            if family in (1, 2):
                tol_o, tol_l = ref_o, ref_l
        if family == V_OFFSET:   # values are centred: what is left is the output's own fp32 rounding at the offset's magnitude ...
            tol_o = 2e-4 + 2.0 ** -21 * float(v.abs().max())
            # ... while rung 0 accumulates w_j v_j at the offset's magnitude in fp32 (6e-3 off fp64 at V = 1000 + N(0, 1) over 8617 keys,
            # seed 42 case 14): its error MEASURED on the sampled rows is the yardstick of the coverage check below
            ref_o = max(tol_o, float(np.abs(ref[sb].cpu().numpy()[rows] - o64).max()))
        note("fp32 tensors vs fp64 (sampled rows)", float(np.abs(res[sb].cpu().numpy()[rows] - o64).max()) / vmag, tol_o, desc)
        note("fp32 tensors, LSE vs fp64 (sampled rows)", float(np.abs(lse[sb].cpu().numpy()[rows] - l64).max()), tol_l, desc)
        # rung 0 is fp32 arithmetic itself (its own error against fp64 is the FMA chain's), and it is compared on EVERY slab and row where the
        # fp64 sample above saw ~200 rows of one: a coverage check (unwritten rows, wrong tiles), at five times the chain's error on the sample
        wide = family in (1, 2) or 6 <= family < V_OFFSET
        qn, kn, vn = q.numpy(), k.numpy(), v.numpy()
        coverage("fp32 tensors vs rung 0", res, ref, qn, kn, vn, causal, scale, 5.0 * ref_o, tol_o, wide or family == V_OFFSET, vmag, desc)
        coverage("fp32 tensors, LSE vs rung 0", lse, lse_ref, qn, kn, vn, causal, scale, 5.0 * ref_l, tol_l, wide, vmag, desc, lse=True)
        if pack_nh:    # the llm.c entry: packed (B, T, 3C) fp32, causal, 1/sqrt(d)
            nh = pack_nh
            B, T = max(1, bh // nh), min(n, 2048)
            inp = torch.randn(B, T, 3 * nh * d, generator=g).to(dev)
            got = fa.forward_packed_qkv(inp, nh)
            qq, kk, vv = (inp[:, :, i * nh * d:(i + 1) * nh * d].reshape(B, T, nh, d).permute(0, 2, 1, 3).reshape(B * nh, T, d).contiguous() for i in range(3))
            want = fa.forward(qq, kk, vv, True, scale=d ** -0.5, kernel="naive").reshape(B, nh, T, d).permute(0, 2, 1, 3).reshape(B, T, nh * d)
            note("packed QKV (llm.c layout)", float((got - want).abs().max()), TOL_F32, desc + f" B={B} T={T} NH={nh}")
        # ---- bf16 tensors
        if family == V_OFFSET:
            vmag = float(v.abs().max())   # bf16 tensors: V of magnitude 1000 has an ulp of 4, and fp32 accumulation is relative to |v|: relative figures
        qb, kb, vb = (t.to(torch.bfloat16).to(dev) for t in (q, k, v))
        refb, lse_refb = fa.forward(qb.float(), kb.float(), vb.float(), causal, scale=scale, kernel="naive", return_lse=True)
        ob, lse_b = fa.forward(qb, kb, vb, causal, scale=scale, return_lse=True)                                  # bf16 out
        of, lse_f = fa.forward(qb, kb, vb, causal, scale=scale, out_dtype=torch.float32, return_lse=True)          # fp32 out: accurate P
        r = fa.last_forward_route()
        routes[("bf16->f32", r)] = routes.get(("bf16->f32", r), 0) + 1
        if torch.isnan(ob.float()).any() or torch.isnan(of).any():
            raise AssertionError("FAIL NaN bf16 " + desc)
        qsb, ksb, vsb = (t[sb].float().cpu().numpy() for t in (qb, kb, vb))
        ob64, lb64 = adv.rows_f64(qsb, ksb, vsb, rows, causal, scale)
        note("bf16 tensors, fp32 out vs fp64 (sampled rows)", float(np.abs(of[sb].cpu().numpy()[rows] - ob64).max()) / vmag, TOL_PB2, desc)
        note("bf16 tensors, fp32 out, LSE vs fp64 (sampled rows)", float(np.abs(lse_f[sb].cpu().numpy()[rows] - lb64).max()), 1e-3, desc)
        # the bf16-P kernels against their WORST CASE computed from the data (every weight off by 2^-8 with the worst signs) ...
        rb = np.arange(0, len(rows), 3)                                          # (a third of the rows: the bound costs rows x keys x d)
        note("bf16 tensors, bf16 out vs its data-derived bound (sampled rows)", float(np.abs(ob[sb].float().cpu().numpy()[rows[rb]] - ob64[rb]).max()),
             bf16_p_bound(qsb, ksb, vsb, rows[rb], causal, scale, True), desc)
        # ... and against the regression thresholds of seeded random data (wide logits sharpen the softmax: the scale-1 figure whatever the
        # nominal scale; coherent families put two keys at +-vmax on equal weights: the bound above is the statement there)
        eff_scale = 1.0 if family in (1, 2) or family >= 6 else scale
        if family < 6 or family == V_OFFSET:
            note("bf16 tensors, bf16 out", float((ob.float() - refb).abs().max()) / vmag, bf16_tol(eff_scale, False, causal, n), desc)
        # the accurate P of FA_KERNEL_AUTO (two bf16 terms, one launch): the fp32 bar with margin on every data family (round 2's one-term
        # fp16 P needed 2^-10 * max|v| on the hostile ones)
        # (against rung 0 on every slab; where rung 0's own fp32 chain is the larger error -- coherent inputs -- fp64 decides: coverage())
        qbn, kbn, vbn = (t.float().cpu().numpy() for t in (qb, kb, vb))
        coverage("bf16 tensors, fp32 out", of, refb, qbn, kbn, vbn, causal, scale, TOL_PB2, TOL_PB2, False, vmag, desc)
        if case % 3 == 0:    # the NB = 2 tiling forced, and the split kernel beside it
            o2, lse2 = fa.forward(qb, kb, vb, causal, scale=scale, out_dtype=torch.float32, kernel="pb2:1", return_lse=True)
            coverage("kernel=pb2:1", o2, refb, qbn, kbn, vbn, causal, scale, TOL_PB2, TOL_PB2, False, vmag, desc)
            coverage("kernel=pb2:1, LSE", lse2, lse_refb, qbn, kbn, vbn, causal, scale, 2e-4, 2e-4, False, vmag, desc, lse=True)
        # the LSE sees what O / l hides (a clamped or saturated P): row sums of 8-bit-rounded P stay within 2e-2, of 11-bit ones 2e-3
        note("bf16 tensors, bf16 out, LSE", float((lse_b - lse_refb).abs().max()), 2e-2, desc)
        note("bf16 tensors, fp32 out, LSE", float((lse_f - lse_refb).abs().max()), 1e-3, desc)
        if verbose and case % 50 == 49:
            print(f"{case + 1} cases ok", flush=True)
    lines.append(f"soak: {cases} cases, seed {seed}" + (f", rows up to {max_n} keys" if max_n else "") + ": all within tolerance")
    for key, (err, tol, desc) in worst.items():
        lines.append(f"  worst {key}: {err:.3e} of {tol:.1e} ({err / tol:.2f})   ({desc})")
    lines.append("  routes (tensor kind, fa_last_forward_route): " + ", ".join(f"{k[0]}/{k[1]}: {v}" for k, v in sorted(routes.items())))
    stats1 = fa.stats()
    lines.append(f"  slow paths counted by the kernels during the run (fa_get_stats): tiles_redone {stats1['tiles_redone'] - stats0['tiles_redone']}, "
                 f"workgroups_fp32 {stats1['workgroups_fp32'] - stats0['workgroups_fp32']}")
    text = "\n".join(lines)
    if verbose:
        print(text)
    if out:
        with open(out, "a") as f:
            f.write(text + "\n")
    return worst, routes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-n", type=int, default=0, help="cap the row length (the pytest slice uses 4500)")
    ap.add_argument("--only", type=int, default=-1, help="re-run one case of the seed (the draws of the others are replayed, their launches skipped) and print what every fp32 path reads on it")
    a = ap.parse_args()
    try:
        run(a.cases, a.seed, a.out, a.only, a.max_n)
    except AssertionError:
        sys.exit(1)


if __name__ == "__main__":
    main()
