#!/usr/bin/env python3
"""profiles/make_tables.py TAG -- rewrite the measured-results blocks of docs/results.md and README.md from ONE collection of the final binary:
profiles/TAG_config_table.txt (profiles/config_table.sh) and profiles/TAG_bench_line{,_c3,_accurate}.json (profiles/collect.sh).

The blocks sit between `<!-- results:begin -->` / `<!-- results:end -->` markers.  Boxes of the pool differ by +-4 %, so the tables are
regenerated, not edited, whenever the profiles are re-collected (python profiles/make_tables.py r03g)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = [("c4", "== c4 non-causal"), ("bh128", "== bh=128"), ("causal", "== causal c4"), ("dims", "== d=32, d=128"), ("f32s", "== fp32 tensors, split kernel"),
        ("exact", "== fp32 tensors, exact"), ("pb2", "== two-term"), ("p16x2", "== round 3 accurate"), ("split", "== hi + lo"), ("ks", "== key-split"), ("unsplit", "== the same launches without"),
        ("p16", "== one-term"), ("wide", "== wide head dims"), ("llmc", "== llm.c")]


def parse(path):
    sec, cur = {}, None
    for line in open(path):
        line = line.strip()
        if line.startswith("=="):
            cur = next((k for k, pre in KEYS if line.startswith(pre)), None)
            if cur is not None:
                sec[cur] = []
            continue
        if cur is None or not line.startswith("{"):
            continue
        try:
            sec[cur].append(json.loads(line))
        except ValueError:
            pass
    return sec


def ms(j, nd=3):
    return f"{j['ms']:.{nd}f}"


def tf(j):
    return f"{j['tflops']:.0f}"


def fr(j, peak=2500.0):
    return f"{j['tflops'] / peak:.3f}"


def replace_block(path, body, name="results"):
    text = open(path).read()
    pat = re.compile(r"(<!-- %s:begin -->\n).*?(<!-- %s:end -->)" % (name, name), re.S)
    assert pat.search(text), f"{path}: no {name} markers"
    open(path, "w").write(pat.sub(lambda m: m.group(1) + body + m.group(2), text))


def main():
    tag = sys.argv[1]
    s = parse(os.path.join(ROOT, "profiles", f"{tag}_config_table.txt"))
    b4 = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_line.json")))
    b3 = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_line_c3.json")))
    ba = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_line_accurate.json")))
    ex = b4["extra"]
    c4, bh, ca, dm, f3, exa, x2, r3, sp, ks, us, p1 = (s[k] for k in ("c4", "bh128", "causal", "dims", "f32s", "exact", "pb2", "p16x2", "split", "ks", "unsplit", "p16"))
    # pb2 section order: c4 at scale 1, 0.5, 1/sqrt(d); causal; bh=128; d=128; d=128 causal; d=32; c2 shape; c4 NB=2; bh=128 NB=2
    # p16x2 (ablation library) order: c4; causal; bh=128; d=128; d=128 causal; d=32; c2 shape
    llmc = s["llmc"][-1]
    c5 = ex["c5"]
    third = 2500.0 / 3.0
    rows = [
        "| config | kernel | ms | TFLOP/s | frac |",
        "|---|---|---|---|---|",
        f"| **c4** B=2 H=8 N=8192 d=64 bf16 | `fa_fwd_bf16_x4_kernel` (optimistic mix, sampled reference) | {ms(c4[0], 4)} (bench line of the call: {b4['roofline']['kernel_ms']:.4f}) | **{tf(c4[0])}** ({b4['roofline']['achieved']:.0f}) | **{fr(c4[0])}** ({b4['roofline']['frac']:.3f}) |",
        f"| same through the two-wave kernel / lazily rescaled mix only (ablation library) | `fa_fwd_bf16_pp3_kernel` / `x4` variant 42 | {ms(c4[1])} / {ms(c4[2])} | {tf(c4[1])} / {tf(c4[2])} | {fr(c4[1])} / {fr(c4[2])} |",
        f"| **c4, fp32 output = the accurate path (P as bf16 hi + bf16 lo, ONE launch)**; at scale 0.5; at 1/√d; through the NB = 2 tiling | `fa_fwd_bf16_x4_pb2_kernel` (`x2_pb2`) | {ms(x2[0])}; {ms(x2[1])}; {ms(x2[2])}; {ms(x2[9])} (bench `--accurate`: {ba['roofline']['kernel_ms']:.3f}) | **{tf(x2[0])}**; {tf(x2[1])}; {tf(x2[2])}; {tf(x2[9])} | **{fr(x2[0])}**; {fr(x2[1])}; {fr(x2[2])}; {fr(x2[9])} |",
        f"| same through round 3's accurate path (ablation library: V → fp16 copy + two fp16 terms of P + empty fallback launch), same run | `fa_fwd_bf16_x2_p16x2_kernel` | {ms(r3[0])} | {tf(r3[0])} | {fr(r3[0])} |",
        f"| same, ONE fp16 term (ablation library, `kernel=\"p16\"`: ≈ 1e-3, at the bar) | `fa_fwd_bf16_x4_p16_kernel` | {ms(p1[0])} | {tf(p1[0])} | {fr(p1[0])} |",
        f"| same, hi + lo bf16 terms of P and Q′ in the split kernel (`kernel=\"split\"`: slabs beyond 4 GiB) | `fa_fwd_f32_split_kernel<…, IN_BF16>` | {ms(sp[0])} | {tf(sp[0])} | {fr(sp[0], 1250.0)} of peak at 2× FLOP |",
        f"| c5 per-GPU shard BH=128 (bf16 P / two bf16 terms, NB = 4 and NB = 2 / round 3's chain); all 1024 slabs on one GPU (bench `extra.c5`) | x4 / x4_pb2, x2_pb2 / x2_p16x2; x4 | {ms(bh[0])} / {ms(x2[4], 2)}, {ms(x2[10], 2)} / {ms(r3[2], 2)}; {c5['ms_per_step']:.2f} | **{tf(bh[0])}** / {tf(x2[4])}, {tf(x2[10])} / {tf(r3[2])}; {c5['tflops']:.0f} | **{fr(bh[0])}** / {fr(x2[4])}, {fr(x2[10])} / {fr(r3[2])}; {c5['frac_bf16_mfma_peak_per_gpu']:.3f} |",
        f"| c4 shape, causal (bf16 P / two bf16 terms / round 3's chain); BH=128 causal | `fa_fwd_bf16_x2_kernel<64>` / `x2_pb2` / `x2_p16x2` (paired tile order, §4.4); `x2` | {ms(ca[0], 4)} / {ms(x2[3])} / {ms(r3[1])}; {ms(ca[1])} | {tf(ca[0])} / {tf(x2[3])} / {tf(r3[1])}; {tf(ca[1])} | {fr(ca[0])} / {fr(x2[3])} / {fr(r3[1])}; {fr(ca[1])} |",
        f"| **d=128** (BH=16, N=8192), non-causal / causal; two bf16 terms; round 3's chain | `fa_fwd_bf16_x2_kernel<128>`; `x2_pb2`; `x2_p16x2` | {ms(dm[1])} / {ms(dm[2])}; {ms(x2[5])} / {ms(x2[6])}; {ms(r3[3])} / {ms(r3[4])} | **{tf(dm[1])}** / {tf(dm[2])}; {tf(x2[5])} / {tf(x2[6])}; {tf(r3[3])} / {tf(r3[4])} | **{fr(dm[1])}** / {fr(dm[2])}; {fr(x2[5])} / {fr(x2[6])}; {fr(r3[3])} / {fr(r3[4])} |",
        f"| d=32 (bf16 P; two bf16 terms; round 3's chain); c2's shape 128 × 1024 (two bf16 terms; round 3's chain) | `fa_fwd_bf16_x2_kernel<32>`; `x2_pb2`; `x2_p16x2` | {ms(dm[0])}; {ms(x2[7])}; {ms(r3[5])}; {ms(x2[8])}; {ms(r3[6])} | {tf(dm[0])}; {tf(x2[7])}; {tf(r3[5])}; {tf(x2[8])}; {tf(r3[6])} | {fr(dm[0])}; {fr(x2[7])}; {fr(r3[5])}; {fr(x2[8])}; {fr(r3[6])} |",
        f"| one slab, BH=1 N=8192 d=64 bf16, non-causal / causal (key-split launches, §8) | `fa_fwd_bf16_x2_kernel<64>` × 8 key shares + `fa_combine_splits_kernel` | {ms(ks[0])} / {ms(ks[3])} ({ms(us[0])} / {ms(us[3])} unsplit) | {tf(ks[0])} / {tf(ks[3])} | {fr(ks[0])} / {fr(ks[3])} |",
        f"| BH=2, 4 non-causal; BH=2, 4, 8 causal (key-split) | same | {ms(ks[1])}, {ms(ks[2])} ({ms(us[1])}, {ms(us[2])} unsplit); {ms(ks[4])}, {ms(ks[5])}, {ms(ks[6])} ({ms(us[4])}, {ms(us[5])}, {ms(us[6])} unsplit) | {tf(ks[1])}, {tf(ks[2])}; {tf(ks[4])}, {tf(ks[5])}, {tf(ks[6])} | {fr(ks[1])}, {fr(ks[2])}; {fr(ks[4])}, {fr(ks[5])}, {fr(ks[6])} |",
        f"| BH=1, 2 fp32 tensors N=8192, non-causal; BH=1, 2, 4 causal (key-split, every share guarded: round 3) | `fa_fwd_f32_split_kernel` × 8 / 4 / 2 key shares + combine | {ms(ks[7])}, {ms(ks[8])} ({ms(us[7])}, {ms(us[8])} unsplit, unguarded); {ms(ks[9])}, {ms(ks[10])}, {ms(ks[11])} ({ms(us[9])}, {ms(us[10])}, {ms(us[11])}) | {tf(ks[7])}, {tf(ks[8])}; {tf(ks[9])}, {tf(ks[10])}, {tf(ks[11])} | — |",
        f"| **c3** B=2 H=8 N=8192 d=64 fp32 | `fa_fwd_f32_split_kernel` (Q·Kᵀ: 3 fp16 products of fp16 hi/lo terms, hi·hi first; P·V: 3 bf16 products of bf16 hi/lo terms; exponent reference in the accumulator init; AUTO = the same launch with the range guard: {b3['ms_per_step']:.3f} bench) | {ms(f3[0])} | **{tf(f3[0])}** | {fr(f3[0], third)} of the 16-bit MFMA peak at 3× FLOP |",
        f"| same, **fp32 arithmetic** (`FA_KERNEL_MFMA` — the figure to quote for \"c3 fp32\" in the reference's sense: bench `extra.c3.reference_arithmetic`) | `fa_fwd_f32_kernel` | {ms(exa[0])} | {exa[0]['tflops']:.1f} | **{fr(exa[0], 157.3)}** of fp32 peak |",
        f"| c3 shape causal; d=128; d=32 (fp32 tensors, split) | `fa_fwd_f32_split_kernel` | {ms(f3[4])}; {ms(f3[6])}; {ms(f3[7])} | {tf(f3[4])}; {tf(f3[6])}; {tf(f3[7])} | — |",
        f"| **c2** B=8 H=16 N=1024 d=64 fp32 | split (AUTO: {ex['c2']['ms']:.3f} bench) / exact (`extra.c2.reference_arithmetic`) | {ms(f3[5])} / {ms(exa[1])} | {tf(f3[5])} / {tf(exa[1])} | {fr(f3[5], third)} of bf16 peak at 3× FLOP / {fr(exa[1], 157.3)} of fp32 peak |",
        (f"| head dims outside 32 / 64 / 128 (round 6), fp32 tensors through `forward()`: d = 96, 160, 192, 224, 256 at BH=16 N=8192; d = 96, 256 causal; d = 96 one slab (key shares); d = 80 (rung 0, BH=16 N=2048); bf16 tensors d = 96, 256 at BH=16 N=8192 | `fa_fwd_f32_kernel` (exact fp32 MFMA, `fa_fwd_f32_wide{{,_bf16}}.hip`); `fa_naive_f32_kernel` | "
         + ", ".join(ms(w, 2) for w in s["wide"][:5]) + "; " + ", ".join(ms(w, 2) for w in s["wide"][5:7]) + f"; {ms(s['wide'][7])}; {ms(s['wide'][8], 1)}; " + ", ".join(ms(w, 2) for w in s["wide"][9:11]) + " | "
         + ", ".join(f"{w['tflops']:.0f}" for w in s["wide"][:5]) + "; " + ", ".join(f"{w['tflops']:.0f}" for w in s["wide"][5:7]) + f"; {s['wide'][7]['tflops']:.0f}; {s['wide'][8]['tflops']:.1f}; " + ", ".join(f"{w['tflops']:.0f}" for w in s["wide"][9:11]) + " | "
         + ", ".join(fr(w, 157.3) for w in s["wide"][:5]) + " of the fp32 MFMA peak |") if "wide" in s and len(s["wide"]) >= 11 else "| head dims outside 32 / 64 / 128 | not in this collection | | | |",
        f"| llm.c harness size B=6 T=4096 C=768 NH=12 fp32, causal, 1/√d (`fa_driver --mode llmc`) | `fa_forward_packed_qkv` → split kernel | {ms(llmc)} | {tf(llmc)} | max-abs {llmc['max_abs_err_vs_naive']:.1e} vs rung 0 (bar 1e-4) |",
    ]
    design = (f"Generated by `python profiles/make_tables.py {tag}` from `profiles/{tag}_config_table.txt` and `profiles/{tag}_bench_line*.json` "
              f"(library sha256 {b4['roofline']['lib_sha256']}…):\n\n" + "\n".join(rows) + "\n")
    replace_block(os.path.join(ROOT, "docs", "results.md"), design)

    # the c4 sentence of DESIGN.md section 7: profiler passes of the same collection
    pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")))
    k4 = next(v for k, v in pmc.items() if "fa_fwd_bf16_x4_kernel" in k)
    traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert traffic["_tag"] == tag and b4["roofline"]["lib_sha256"] and traffic["lib_sha256"].startswith(b4["roofline"]["lib_sha256"][:16]), "pmc_traffic.json is from another collection"
    summ = open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.txt")).read()
    m = re.search(r"== pass stats: kernel durations.*?\n.*?\n\s*(\d+)\s+(\d+)\s+\d+\s+\d+\s+\d+\s+void fa::fa_fwd_bf16_x4_kernel", summ, re.S)
    calls, avg_ns = int(m.group(1)), int(m.group(2))
    gui = k4["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
    sec7 = (f"c4 from the final passes (`{tag}`): {avg_ns / 1e3:.1f} µs average of {calls} dispatches under the profiler, `SQ_VALU_MFMA_BUSY_CYCLES` ÷ (1024 SIMDs ×\n"
            f"{gui / 1e3:.1f} k cycles) = {k4['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui):.2f} of the matrix pipe, {gui / avg_ns:.2f} GHz, HBM traffic {traffic['c4_hbm_bytes_per_launch'] / 1e6:.2f} MB against 67.11 MB\n"
            f"algorithmic, `SQ_LDS_BANK_CONFLICT` {k4['SQ_LDS_BANK_CONFLICT']:.0f}; the bench line of that call: {b4['roofline']['kernel_ms']:.4f} ms = **{b4['roofline']['frac']:.3f}**.\n")
    # ... and of the accurate path (pass stats_acc: one dispatch per forward; pmc_acc_sq: its counters)
    ka = next((v for k, v in pmc.items() if "fa_fwd_bf16_x4_pb2_kernel" in k), None)
    ma = re.search(r"== pass stats_acc: kernel durations.*?\n.*?\n\s*(\d+)\s+(\d+)\s+\d+\s+\d+\s+\d+\s+void fa::fa_fwd_bf16_x4_pb2_kernel", summ, re.S)
    if ka is not None and ma is not None:
        calls_a, avg_a = int(ma.group(1)), int(ma.group(2))
        gui_a = ka["GRBM_GUI_ACTIVE"] / 8.0
        sec7 += (f"The accurate path in the same collection: {avg_a / 1e3:.1f} µs average of {calls_a} dispatches of `fa_fwd_bf16_x4_pb2_kernel` (one per forward; round 3's\n"
                 f"chain was three), {gui_a / 1e3:.1f} k cycles = {gui_a / 256 / 1e3 * 1.0:.2f} k per four-block step (256 steps per tile; the issue model of section 4.6 gives 2.6 k), `SQ_VALU_MFMA_BUSY_CYCLES` ÷ (1024 × cycles) = "
                 f"{ka['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui_a):.2f} of the pipe,\n{gui_a / avg_a:.2f} GHz, {ka['SQ_INSTS_VALU'] / ka['SQ_INSTS_MFMA']:.2f} VALU instructions per MFMA; "
                 f"the bench line (`--accurate`): {ba['roofline']['kernel_ms']:.4f} ms = **{ba['roofline']['frac']:.3f}** at {ba['roofline']['max_abs_err']:.1e}.\n")
    replace_block(os.path.join(ROOT, "docs", "results.md"), sec7, "profile7")

    acc = ba["roofline"]
    readme = (
        f"Measured on MI355X (ONE collection of the final binary on one box, `profiles/{tag}_config_table.txt`, written by `profiles/make_tables.py`;\n"
        f"identical binaries differ by ±4 % between boxes of the pool, so every headline is quoted with its range):\n\n"
        f"* **The north star's pair — ≥ 60 % of the bf16 MFMA peak AND within 1e-3 of the fp32 reference on B=2 H=8 d=64 N=8192 — is not met.**\n"
        f"  The path INSIDE 1e-3 (bf16 tensors, fp32 output, P as bf16 hi + bf16 lo in one launch: {acc['max_abs_err']:.1e} at the reference's scale 1):\n"
        f"  {ms(x2[0])} ms = **{fr(x2[0])}** of the dense bf16 MFMA peak here; round 6's boxes read 0.325–0.338 with the re-centred reference (A/B on one box: −3.3 % time), the boxes of rounds 4–5 0.30–0.33 (`roofline_at_1e-3` of the bench line).\n"
        f"  The fastest path (bf16 P, bf16 output: {b4['roofline']['max_abs_err']:.1e} there — 15× outside that bar; {ex['c4_scale_rsqrt_d']['max_abs_err']:.1e} at 1/√d): {ms(c4[0])} ms =\n"
        f"  **{fr(c4[0])}** here, 0.483–0.505 on round 6's boxes; the DRIVER's end-of-round runs read 0.466 / 0.485 / 0.488 / 0.468 / 0.477 in rounds 1–5 (0.47–0.49).  `DESIGN.md` §5 / §5.1: the loops\n"
        f"  run at the 1.4 kW package cap; the three sized ideas of VERDICT r05 were taken to their kill tests and measured negative; re-centring the\n"
        f"  optimistic reference on the row sum (more exact zeros in P) bought 2.0 % on this line and 3.3 % on the one above (A/B, `profiles/r06_exp9_recentre.txt`),\n"
        f"  dropping the prologue's sampled reference where the tile re-centres anyway another 1.7 % (`profiles/r06_exp15_sample_vs_recentre.txt`);\n"
        f"  two workgroups sharing a CU taking turns at the issue priority another 2.6 % at d = 32 (`profiles/r06_exp12_take_turns.txt`).\n"
        f"* d=128: {ms(dm[1])} ms ({tf(dm[1])} TFLOP/s, {fr(dm[1])}); causal d=64: {ms(ca[0])} ms ({fr(ca[0])}); d=32: {ms(dm[0])} ms ({fr(dm[0])}); all 1024 slabs of config 5 on one GPU: {c5['ms_per_step']:.2f} ms ({c5['frac_bf16_mfma_peak_per_gpu']:.3f}).\n"
        f"* fp32 tensors (the reference's dtype) {ms(f3[0])} ms = {tf(f3[0])} TFLOP/s (0.667–0.736 ms by box): Q·Kᵀ as three fp16 MFMA products of fp16 hi/lo terms, P·V as three\n"
        f"  bf16 products — within 1e-4 of the fp64 oracle on random data at scale 1; on every input inside `max(1e-3, E_ref) + 3·2⁻¹⁷·max|v − v̄|`, `E_ref` = what the\n"
        f"  reference's own fp32 FMA chain leaves (`DESIGN.md` §4; observed ≤ 3e-4 on the constructed families); workgroups whose operands leave the fp16 range are\n"
        f"  redone in fp32 arithmetic inside the launch.  `kernel=\"exact\"` (fp32 arithmetic, the reference's own rounding): {ms(exa[0], 2)} ms, {100 * exa[0]['tflops'] / 157.3:.0f} % of the fp32 MFMA\n"
        f"  peak, causal {100 * ex['c3_causal']['exact']['frac_f32_mfma_peak']:.0f} %; the same kernel serves head dims 96 … 256 (docs/results.md).\n"
        f"* Grids that leave the chip idle are key-split: one slab of 8192 keys {ms(ks[0])} ms instead of {ms(us[0])}, causal {ms(ks[3])} instead of {ms(us[3])}, fp32 {ms(ks[7])}\n"
        f"  instead of {ms(us[7])}, fp32 causal {ms(ks[9])} instead of {ms(us[9])}.\n")
    replace_block(os.path.join(ROOT, "README.md"), readme)
    print("docs/results.md and README.md results blocks rewritten from", tag)


if __name__ == "__main__":
    main()
