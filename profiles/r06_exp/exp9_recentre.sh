#!/bin/bash
# r06 experiment 9: optimistic mix re-centred on the row sum after 8, 24, 72 stages (more exact zeros in P -> less matrix-core power).
# old = previous commit (ab_old/), new = this tree, interleaved; errors vs rung 0 first.
out=gpurun_out/r06_exp9.txt
: > $out
NEW=flashattention.c_amd/fa_driver; OLD=ab_old/fa_driver
run() { drv=$1; shift; $drv --mode rand --check 0 --warmup 100 --iters 50 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
for shape in "--dtype bf16 --bh 16 --n 8192 --d 64" "--dtype bf16 --out_f32 1 --bh 16 --n 8192 --d 64" "--dtype bf16 --bh 16 --n 8192 --d 64 --causal 1" "--dtype bf16 --bh 16 --n 8192 --d 128" "--dtype bf16 --bh 16 --n 8192 --d 32" "--dtype bf16 --out_f32 1 --bh 4 --n 16384 --d 64 --scale 2"; do
  echo "check [$shape] old $($OLD --mode rand --check 1 --iters 2 --kernel auto $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*') | new $($NEW --mode rand --check 1 --iters 2 --kernel auto $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out
done
for r in 1 2 3 4 5; do
  echo "rep $r" >> $out
  for shape in "--dtype bf16 --bh 16 --n 8192 --d 64" "--dtype bf16 --out_f32 1 --bh 16 --n 8192 --d 64" "--dtype bf16 --bh 16 --n 8192 --d 64 --causal 1" "--dtype bf16 --bh 16 --n 8192 --d 128" "--dtype bf16 --bh 16 --n 8192 --d 32" "--dtype bf16 --bh 128 --n 8192 --d 64 --iters 10" "--dtype bf16 --bh 16 --n 8192 --d 64 --scale 0.125" "--dtype bf16 --bh 32 --n 4096 --d 64" "--dtype bf16 --bh 16 --n 8192 --d 64 --scale 2"; do
    echo "[$shape] old $(run $OLD --kernel auto $shape) | new $(run $NEW --kernel auto $shape)" >> $out
  done
done
cat $out
