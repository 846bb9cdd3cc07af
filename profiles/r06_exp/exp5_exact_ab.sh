#!/bin/bash
# r06 experiment 5: exact fp32 kernel -- (a) o *= alpha skipped when no lane's running maximum moved (all head dims), (b) four partial
# K.Q^T accumulators at head dims above 128 (one wave per SIMD there).  old = previous commit (ab_old/), new = this tree; interleaved.
out=gpurun_out/r06_exp5.txt
: > $out
NEW=flashattention.c_amd/fa_driver; OLD=ab_old/fa_driver
run() { drv=$1; shift; $drv --mode rand --check 0 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
echo "check: $($NEW --mode rand --check 1 --iters 2 --dtype f32 --kernel mfma --bh 3 --n 2000 --d 256 --causal 1 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*')  $($NEW --mode rand --check 1 --iters 2 --dtype f32 --kernel mfma --bh 3 --n 2000 --d 64 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*')" >> $out
for r in 1 2 3; do
  echo "rep $r" >> $out
  for shape in "--bh 16 --n 8192 --d 64" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 128 --n 1024 --d 64" "--bh 16 --n 8192 --d 128" "--bh 16 --n 8192 --d 32" "--bh 1 --n 8192 --d 64"; do
    echo "[$shape] old $(run $OLD --dtype f32 --kernel mfma --warmup 10 --iters 10 $shape) | new $(run $NEW --dtype f32 --kernel mfma --warmup 10 --iters 10 $shape)" >> $out
  done
  for shape in "--bh 16 --n 8192 --d 96" "--bh 16 --n 8192 --d 160" "--bh 16 --n 8192 --d 192" "--bh 16 --n 8192 --d 256" "--bh 16 --n 8192 --d 256 --causal 1"; do
    echo "[$shape] old $(run $OLD --dtype f32 --kernel auto --warmup 4 --iters 4 $shape) | new $(run $NEW --dtype f32 --kernel auto --warmup 4 --iters 4 $shape)" >> $out
  done
done
cat $out
