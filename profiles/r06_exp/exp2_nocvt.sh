#!/bin/bash
# r06 experiment 2: kill test of the one-time K/V split pre-pass.  Ablation library built with
# FA_EXTRA_ABL_FLAGS="-DFA_SPLIT_NOCVT=1": a K/V piece is converted with 1 VALU per element instead of 3.5 (no centring, no lo residual,
# no guard maximum) -- an upper bound of what a pre-pass (0 VALU per element, tiles by LDS-DMA) could take out of the main loop.
D=flashattention.c_amd
out=gpurun_out/r06_exp2.txt
: > $out
run() { drv=$1; shift; $drv --mode rand --check 0 --warmup 30 --iters 50 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
$D/fa_driver_ablation --mode rand --check 1 --iters 5 --dtype f32 --kernel auto --bh 16 --n 2048 --d 64 >> $out 2>&1
for r in 1 2 3 4 5; do
  echo "rep $r" >> $out
  for shape in "--bh 16 --n 8192 --d 64" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 128 --n 1024 --d 64" "--bh 16 --n 8192 --d 128" "--bh 16 --n 8192 --d 32"; do
    echo "[$shape] product $(run $D/fa_driver --dtype f32 --kernel auto $shape) | nocvt $(run $D/fa_driver_ablation --dtype f32 --kernel auto $shape)" >> $out
  done
done
cat $out
