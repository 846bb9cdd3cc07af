#!/bin/bash
# r06 experiment 16: one-wave-per-SIMD bf16 kernels request Q and K(0) first and wait for those only before the first scores (the other tiles of the
# first barrier group land under them).  old = ab_old/ (previous commit: Q last, everything waited for), new = this tree; interleaved reps.
out=gpurun_out/r06_exp16.txt
: > $out
NEW=flashattention.c_amd/fa_driver; OLD=ab_old/fa_driver
run() { drv=$1; shift; $drv --mode rand --check 0 --warmup 60 --iters 60 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
SHAPES=("--dtype bf16 --bh 16 --n 8192 --d 64" "--dtype bf16 --out_f32 1 --kernel pb2 --bh 16 --n 8192 --d 64" "--dtype bf16 --bh 16 --n 8192 --d 64 --causal 1" "--dtype bf16 --bh 16 --n 8192 --d 128" "--dtype bf16 --bh 16 --n 8192 --d 32" "--dtype bf16 --bh 128 --n 8192 --d 64 --iters 10" "--dtype bf16 --bh 32 --n 4096 --d 64" "--dtype bf16 --bh 128 --n 1024 --d 32" "--dtype bf16 --bh 64 --n 2048 --d 64 --causal 1" "--dtype bf16 --bh 1 --n 8192 --d 64" "--dtype bf16 --bh 16 --n 2048 --d 128")
for shape in "${SHAPES[@]}" "--dtype bf16 --bh 3 --n 100 --d 32" "--dtype bf16 --bh 3 --n 64 --d 128 --causal 1" "--dtype bf16 --bh 5 --n 129 --d 64"; do
  echo "check [$shape] old $($OLD --mode rand --check 1 --iters 2 $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*') | new $($NEW --mode rand --check 1 --iters 2 $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out
done
for r in 1 2 3 4 5; do
  echo "rep $r" >> $out
  for shape in "${SHAPES[@]}"; do echo "[$shape] old $(run $OLD $shape) | new $(run $NEW $shape)" >> $out; done
done
grep "^check" $out | cut -c1-220
grep -v "^check\|^rep\|^#" $out | grep old | awk -F'[][]' '{print $2}' | sort -u | while read s; do o=$(grep -F "[$s]" $out | grep -v check | grep old | sed 's/.*old "ms": \([0-9.]*\) | new "ms": \([0-9.]*\)/\1 \2/' | awk '{a+=$1;b+=$2;n++} END {printf "old %.4f new %.4f (%+.1f %%)", a/n, b/n, (b/a-1)*100}'); echo "$s: $o"; done
