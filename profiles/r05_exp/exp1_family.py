"""Round 5, experiment 1 (GPU): the coherent-rounding family of tests/adversarial.py through every fp32 kernel, against the fp64
oracle and against the reference's own arithmetic (fp32 FMA chain, emulated); range cases of the fp16-term split; timings."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import flashattention_c_amd as fa
import adversarial as adv

dev = torch.device("cuda", 0)
def run(q, k, v, kernel, causal=False, scale=1.0):
    o, lse = fa.forward(torch.from_numpy(q).to(dev), torch.from_numpy(k).to(dev), torch.from_numpy(v).to(dev), causal, scale=scale, kernel=kernel, return_lse=True)
    route = fa.last_forward_route() if kernel == "auto" else -1
    return o.cpu().numpy().astype(np.float64), lse.cpu().numpy().astype(np.float64), route

print("== family: max|O| err / max|LSE| err vs fp64; 'chain' = the fp32 FMA chain's own error (emulated)")
rows = []
for d in (32, 64, 128):
    for name in adv.FAMILIES:
        for width in (30.0, 60.0, 85.0, 89.5, 200.0):
            q, k, v = adv.make(name, d, width, n=512, bh=2, seed=d)
            o_ref, l_ref = adv.attention_f64(q, k, v)
            ce = adv.reference_arithmetic_error(q, k, v)
            rec = {"d": d, "family": name, "width": width, "chain": ce}
            for kern in ("exact", "split", "auto"):
                o, l, route = run(q, k, v, kern)
                rec[kern] = (float(np.abs(o - o_ref).max()), float(np.abs(l - l_ref).max()))
                if kern == "auto": rec["route"] = route
            rows.append(rec)
            print(f"d={d:3d} {name:17s} w={width:5.1f} chain {ce[0]:.1e}/{ce[1]:.1e}  exact {rec['exact'][0]:.1e}/{rec['exact'][1]:.1e}  split {rec['split'][0]:.1e}/{rec['split'][1]:.1e}  auto {rec['auto'][0]:.1e}/{rec['auto'][1]:.1e} route {rec['route']}", flush=True)

print("== range cases (d=64, n=512): split unguarded vs auto")
rng = np.random.default_rng(1)
def rc(tag, q, k, v, scale=1.0):
    o_ref, l_ref = adv.attention_f64(q, k, v, scale=scale)
    out = []
    for kern in ("exact", "split", "auto"):
        o, l, route = run(q, k, v, kern, scale=scale)
        out.append(f"{kern} {np.abs(o - o_ref).max():.1e}/{np.abs(l - l_ref).max():.1e}" + (f" route {route}" if kern == "auto" else ""))
    print(f"{tag:34s} " + "  ".join(out), flush=True)
g = lambda *s: rng.standard_normal(s).astype(np.float32)
rc("q*1e3, k*1e-3", g(2, 512, 64) * 1e3, g(2, 512, 64) * 1e-3, g(2, 512, 64))
rc("q*1e-3, k*1e3", g(2, 512, 64) * 1e-3, g(2, 512, 64) * 1e3, g(2, 512, 64))
rc("q*1e4, k*1e-4 coherent", np.full((2, 512, 64), 1e4, np.float32) * (1 + 1e-3 * g(2, 512, 1)), (1e-4 * (1 + 0.3 * g(2, 512, 1))).astype(np.float32) * np.ones((1, 1, 64), np.float32), g(2, 512, 64))
rc("k has one 7e4 element", g(2, 512, 64), np.where(np.arange(512 * 64).reshape(1, 512, 64) == 777, 7e4, g(2, 512, 64)).astype(np.float32), g(2, 512, 64), scale=1e-4)
rc("q*30 (wide logits)", g(2, 512, 64) * 30, g(2, 512, 64), g(2, 512, 64))
rc("k*1e-30", g(2, 512, 64), g(2, 512, 64) * 1e-30, g(2, 512, 64))
rc("all tiny 1e-6", g(2, 512, 64) * 1e-6, g(2, 512, 64) * 1e-6, g(2, 512, 64))

print("== random data at the BASELINE scales (sampled slabs vs fp64)")
for (bh, n, d, scale) in ((4, 1024, 64, 1.0), (2, 8192, 64, 1.0), (2, 4096, 128, 1.0), (2, 4096, 32, 1.0), (2, 4096, 64, 0.125)):
    q, k, v = g(bh, n, d), g(bh, n, d), g(bh, n, d)
    o_ref, l_ref = adv.attention_f64(q, k, v, scale=scale)
    for causal in (False, True):
        if causal: o_r, l_r = adv.attention_f64(q, k, v, True, scale)
        else: o_r, l_r = o_ref, l_ref
        out = []
        for kern in ("exact", "split", "auto"):
            o, l, route = run(q, k, v, kern, causal, scale)
            out.append(f"{kern} {np.abs(o - o_r).max():.1e}/{np.abs(l - l_r).max():.1e}")
        print(f"bh={bh} n={n} d={d} scale={scale} causal={int(causal)}: " + "  ".join(out), flush=True)

print("== timings (ms, fa_time_forward, warm)")
def t(bh, n, d, causal, kernel, iters=20):
    q, k, v = (torch.randn(bh, n, d, device=dev) for _ in range(3))
    fa.time_forward(q, k, v, causal, kernel=kernel, warmup=10, iters=5)
    return min(fa.time_forward(q, k, v, causal, kernel=kernel, warmup=3, iters=iters) for _ in range(3))
for (bh, n, d) in ((16, 8192, 64), (128, 1024, 64), (16, 8192, 128), (16, 8192, 32), (8, 8192, 64), (1, 8192, 64), (4, 8192, 64)):
    for causal in (False, True):
        r = {kern: t(bh, n, d, causal, kern) for kern in ("auto", "split", "exact")}
        fl = (2 if causal else 4) * bh * n * n * d
        print(f"bh={bh:4d} n={n} d={d:3d} causal={int(causal)}: auto {r['auto']:.4f}  split {r['split']:.4f}  exact {r['exact']:.4f} ms  (exact {fl / r['exact'] / 1e9:.1f} TF = {fl / r['exact'] / 1e9 / 157.3:.3f} of fp32 peak)", flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r05_exp1_family.json"), "w"))
