#!/bin/bash
# r06 experiment 3: d = 32 bf16-P kernel with Q' = Q*scale*log2e as bf16 hi + lo and the exponent offset in the accumulator init ("Q2"):
# no v_fma in front of the exponentials.  A/B against the previous commit's library (ab_old/), interleaved; errors vs the rung-0 kernel.
out=gpurun_out/r06_exp3.txt
: > $out
NEW=flashattention.c_amd/fa_driver; OLD=ab_old/fa_driver
run() { drv=$1; shift; $drv --mode rand --check 0 --warmup 30 --iters 50 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
for shape in "--bh 16 --n 8192 --d 32" "--bh 16 --n 8192 --d 32 --causal 1" "--bh 128 --n 1024 --d 32" "--bh 3 --n 777 --d 32 --causal 1" "--bh 16 --n 8192 --d 32 --scale 0.17677"; do
  echo "check new [$shape]: $($NEW --mode rand --check 1 --iters 3 --dtype bf16 --kernel auto $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out
  echo "check old [$shape]: $($OLD --mode rand --check 1 --iters 3 --dtype bf16 --kernel auto $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out
done
for r in 1 2 3 4 5; do
  echo "rep $r" >> $out
  for shape in "--bh 16 --n 8192 --d 32" "--bh 16 --n 8192 --d 32 --causal 1" "--bh 128 --n 1024 --d 32" "--bh 128 --n 1024 --d 32 --causal 1" "--bh 64 --n 2048 --d 32" "--bh 128 --n 8192 --d 32" "--bh 8 --n 8192 --d 32 --causal 1"; do
    echo "[$shape] old $(run $OLD --dtype bf16 --kernel auto $shape) | new $(run $NEW --dtype bf16 --kernel auto $shape)" >> $out
  done
done
cat $out
