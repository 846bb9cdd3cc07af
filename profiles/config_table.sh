#!/bin/bash
# profiles/config_table.sh -- the per-config timing table of DESIGN.md §3 (C driver, steady clocks) and the README-shape
# harness table.  Run on the GPU box: gpurun -- 'bash profiles/config_table.sh > gpurun_out/config_table.txt 2> gpurun_out/harness.txt'
D=./flashattention.c_amd/fa_driver
DA=./flashattention.c_amd/fa_driver_ablation    # tilings that are not in the product library (round 3: only what the dispatch reaches ships)
W="--warmup 150 --iters 50 --check 0"
echo "== c4 non-causal: dispatch(0)=x4 opt, 7=pp3 opt; ablation library: 42=x4 rescaling mix only, 25=pp3 rescaling mix only"
for v in 0 7; do $D --mode rand --bh 16 --n 8192 --d 64 --dtype bf16 $W --variant $v; done
for v in 42 25; do $DA --mode rand --bh 16 --n 8192 --d 64 --dtype bf16 $W --variant $v; done
echo "== bh=128"
for v in 0 7; do $D --mode rand --bh 128 --n 8192 --d 64 --dtype bf16 --warmup 20 --iters 10 --check 0 --variant $v; done
echo "== causal c4 / bh=128"
$D --mode rand --bh 16 --n 8192 --d 64 --dtype bf16 $W --variant 0 --causal 1
$D --mode rand --bh 128 --n 8192 --d 64 --dtype bf16 --warmup 20 --iters 10 --check 0 --variant 0 --causal 1
echo "== d=32, d=128"
$D --mode rand --bh 16 --n 8192 --d 32 --dtype bf16 $W --variant 0
$D --mode rand --bh 16 --n 8192 --d 128 --dtype bf16 $W --variant 0
$D --mode rand --bh 16 --n 8192 --d 128 --dtype bf16 $W --variant 0 --causal 1
echo "== fp32 tensors, split kernel (f32s; variant 0 = product choice, 1..4 = its tilings): c3, c3 causal, c2, d=128, d=32"
for v in 0 1 3 4; do $D --mode rand --bh 16 --n 8192 --d 64 --dtype f32s --warmup 60 --iters 20 --check 0 --variant $v; done
$D --mode rand --bh 16 --n 8192 --d 64 --dtype f32s --warmup 60 --iters 20 --check 0 --causal 1
$D --mode rand --bh 128 --n 1024 --d 64 --dtype f32s --warmup 200 --iters 50 --check 0
$D --mode rand --bh 16 --n 8192 --d 128 --dtype f32s --warmup 40 --iters 10 --check 0
$D --mode rand --bh 16 --n 8192 --d 32 --dtype f32s --warmup 60 --iters 20 --check 0
echo "== fp32 tensors, exact fp32 arithmetic: c3, c2; round 5: c3 causal (paired tiles), c2 causal, d=128, d=128 causal, d=32, d=32 causal, BH=1 over key shares, BH=1 unsplit (--variant 1), BH=1 causal, BH=4 causal"
E="--dtype f32 --check 0"
$D --mode rand --bh 16 --n 8192 --d 64 $E --warmup 20 --iters 10
$D --mode rand --bh 128 --n 1024 --d 64 $E --warmup 100 --iters 30
$D --mode rand --bh 16 --n 8192 --d 64 $E --warmup 20 --iters 10 --causal 1
$D --mode rand --bh 128 --n 1024 --d 64 $E --warmup 100 --iters 30 --causal 1
$D --mode rand --bh 16 --n 8192 --d 128 $E --warmup 10 --iters 6
$D --mode rand --bh 16 --n 8192 --d 128 $E --warmup 10 --iters 6 --causal 1
$D --mode rand --bh 16 --n 8192 --d 32 $E --warmup 20 --iters 10
$D --mode rand --bh 16 --n 8192 --d 32 $E --warmup 20 --iters 10 --causal 1
$D --mode rand --bh 1 --n 8192 --d 64 $E --warmup 50 --iters 30
$D --mode rand --bh 1 --n 8192 --d 64 $E --warmup 50 --iters 30 --variant 1
$D --mode rand --bh 1 --n 8192 --d 64 $E --warmup 50 --iters 30 --causal 1
$D --mode rand --bh 4 --n 8192 --d 64 $E --warmup 50 --iters 30 --causal 1
echo "== two-term bf16-P kernels (FA_KERNEL_AUTO for an fp32 output, one launch): --kernel pb2 --out_f32 1  (c4 at scale 1, 0.5, 1/sqrt(d); causal; bh=128; d=128; d=128 causal; d=32; c2 shape; c4 through the NB = 2 tiling; bh=128 through NB = 2)"
P="--dtype bf16 --kernel pb2 --out_f32 1 --warmup 100 --iters 30 --check 0"
for sc in 1 0.5 0.125; do $D --mode rand --bh 16 --n 8192 --d 64 $P --scale $sc; done
$D --mode rand --bh 16 --n 8192 --d 64 $P --causal 1
$D --mode rand --bh 128 --n 8192 --d 64 $P --iters 10
$D --mode rand --bh 16 --n 8192 --d 128 $P
$D --mode rand --bh 16 --n 8192 --d 128 $P --causal 1
$D --mode rand --bh 16 --n 8192 --d 32 $P
$D --mode rand --bh 128 --n 1024 --d 64 $P
$D --mode rand --bh 16 --n 8192 --d 64 $P --variant 1
$D --mode rand --bh 128 --n 8192 --d 64 $P --iters 10 --variant 1
echo "== round 3 accurate path, ablation library (V -> fp16 copy, two fp16 terms of P, empty fallback launch): --kernel p16x2 --out_f32 1  (c4; causal; bh=128; d=128; d=128 causal; d=32; c2 shape)"
P="--dtype bf16 --kernel p16x2 --out_f32 1 --warmup 100 --iters 30 --check 0"
$DA --mode rand --bh 16 --n 8192 --d 64 $P
$DA --mode rand --bh 16 --n 8192 --d 64 $P --causal 1
$DA --mode rand --bh 128 --n 8192 --d 64 $P --iters 10
$DA --mode rand --bh 16 --n 8192 --d 128 $P
$DA --mode rand --bh 16 --n 8192 --d 128 $P --causal 1
$DA --mode rand --bh 16 --n 8192 --d 32 $P
$DA --mode rand --bh 128 --n 1024 --d 64 $P
echo "== hi + lo bf16 terms (--kernel split --out_f32 1): c4"
$D --mode rand --bh 16 --n 8192 --d 64 --dtype bf16 --kernel split --out_f32 1 --warmup 100 --iters 30 --check 0
echo "== key-split launches (grids that leave the chip idle): bf16 non-causal / causal BH = 1, 2, 4, 8; fp32 BH = 1, 2; fp32 causal BH = 1, 2, 4"
K="--dtype bf16 --kernel auto --warmup 100 --iters 50 --check 0"
for b in 1 2 4; do $D --mode rand --bh $b --n 8192 --d 64 $K; done
for b in 1 2 4 8; do $D --mode rand --bh $b --n 8192 --d 64 $K --causal 1; done
for b in 1 2; do $D --mode rand --bh $b --n 8192 --d 64 --dtype f32s --kernel auto --warmup 100 --iters 50 --check 0; done
for b in 1 2 4; do $D --mode rand --bh $b --n 8192 --d 64 --dtype f32s --kernel auto --warmup 100 --iters 50 --check 0 --causal 1; done
echo "== the same launches without the key split (bf16: --kernel mfma --variant 50; fp32 tensors: --kernel split): bf16 non-causal BH = 1, 2, 4; causal 1, 2, 4, 8; fp32 1, 2; fp32 causal 1, 2, 4"
U="--dtype bf16 --kernel mfma --variant 50 --warmup 100 --iters 50 --check 0"
for b in 1 2 4; do $D --mode rand --bh $b --n 8192 --d 64 $U; done
for b in 1 2 4 8; do $D --mode rand --bh $b --n 8192 --d 64 $U --causal 1; done
for b in 1 2; do $D --mode rand --bh $b --n 8192 --d 64 --dtype f32s --kernel split --warmup 100 --iters 50 --check 0; done
for b in 1 2 4; do $D --mode rand --bh $b --n 8192 --d 64 --dtype f32s --kernel split --warmup 100 --iters 50 --check 0 --causal 1; done
echo "== one-term fp16-P kernels (ablation library): --kernel p16 --out_f32 1  (c4 at scale 1, 0.5, 1/sqrt(d); causal; bh=128; d=128; d=32; c2 shape)"
P="--dtype bf16 --kernel p16 --out_f32 1 --warmup 100 --iters 30 --check 0"
for sc in 1 0.5 0.125; do $DA --mode rand --bh 16 --n 8192 --d 64 $P --scale $sc; done
$DA --mode rand --bh 16 --n 8192 --d 64 $P --causal 1
$DA --mode rand --bh 128 --n 8192 --d 64 $P --iters 10
$DA --mode rand --bh 16 --n 8192 --d 128 $P
$DA --mode rand --bh 16 --n 8192 --d 128 $P --causal 1
$DA --mode rand --bh 16 --n 8192 --d 32 $P
$DA --mode rand --bh 128 --n 1024 --d 64 $P
echo "== wide head dims (round 6): fp32 tensors, FA_KERNEL_AUTO = the exact fp32 MFMA kernel: d = 96, 160, 192, 224, 256 at BH=16 N=8192; d = 96, 256 causal; d = 96 BH=1 (key shares); then rung 0 at d = 80 (BH=16 N=2048) and bf16 tensors at d = 96, 256 (the same kernel, widened on load)"
E="--dtype f32 --kernel auto --check 0 --warmup 10 --iters 6"
for hd in 96 160 192 224 256; do $D --mode rand --bh 16 --n 8192 --d $hd $E; done
for hd in 96 256; do $D --mode rand --bh 16 --n 8192 --d $hd $E --causal 1; done
$D --mode rand --bh 1 --n 8192 --d 96 $E --warmup 50 --iters 30
$D --mode rand --bh 16 --n 2048 --d 80 $E --warmup 3 --iters 3
for hd in 96 256; do $D --mode rand --bh 16 --n 8192 --d $hd --dtype bf16 --kernel auto --check 0 --warmup 10 --iters 6; done
echo "== llm.c harness size"
$D --mode llmc | tail -1
H="python3 -m flashattention_c_amd.harness.bench_flashattention --iters 50 --warmup 100"
{
for hd in 64 32 128; do for dt in f32 bf16; do
  $H --batch_size 2 --n_head 8 --seq_len 8192 --head_dim $hd --dtype $dt
done; done
for hd in 64 32; do for dt in f32 bf16; do
  $H --batch_size 16 --n_head 8 --seq_len 1024 --head_dim $hd --dtype $dt
done; done
$H --batch_size 2 --n_head 8 --seq_len 8192 --head_dim 64 --dtype bf16 --masking
$H --batch_size 2 --n_head 8 --seq_len 8192 --head_dim 128 --dtype bf16 --masking
$H --batch_size 2 --n_head 8 --seq_len 8192 --head_dim 64 --dtype f32 --masking
} 1>&2
