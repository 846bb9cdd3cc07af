#!/bin/bash
# r06 experiment 14: bf16 tensors at head dims 96 ... 256 on the exact fp32 MFMA kernel (widened on load) instead of the rung-0 kernel
D=flashattention.c_amd/fa_driver
for d in 96 160 192 224 256; do for c in 0 1; do
  echo "d=$d causal=$c bf16: $($D --mode rand --dtype bf16 --kernel auto --bh 16 --n 8192 --d $d --causal $c --check 1 --warmup 3 --iters 5 2>&1 | grep -o '"ms": [0-9.]*\|"max_abs_err_vs_naive": [0-9.e+-]*\|"nan": [0-9]*' | tr '\n' ' ') | f32: $($D --mode rand --dtype f32 --kernel auto --bh 16 --n 8192 --d $d --causal $c --check 0 --warmup 3 --iters 5 2>&1 | grep -o '"ms": [0-9.]*')"
done; done
$D --mode rand --dtype bf16 --out_f32 1 --kernel auto --bh 16 --n 8192 --d 96 --check 1 --iters 3 2>&1 | cut -c1-300
$D --mode rand --dtype bf16 --kernel auto --bh 1 --n 8192 --d 256 --causal 1 --check 1 --iters 3 2>&1 | cut -c1-300
