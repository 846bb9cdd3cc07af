#!/bin/bash
# r06 experiment 13: fp32 default, pipelined pass -- reference rows, K(0) and Q requested together before anything is waited for;
# K(1) in flight under the scores of tile 0.  old = ab_old/ (previous commit), new = this tree; interleaved reps.
out=gpurun_out/r06_exp13.txt
: > $out
NEW=flashattention.c_amd/fa_driver; OLD=ab_old/fa_driver
run() { drv=$1; shift; $drv --mode rand --check 0 --warmup 60 --iters 100 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
SHAPES=("--bh 128 --n 1024 --d 64" "--bh 128 --n 1024 --d 32" "--bh 16 --n 8192 --d 64" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 1 --n 8192 --d 64" "--bh 1 --n 8192 --d 64 --causal 1" "--bh 72 --n 4096 --d 64 --causal 1 --scale 0.125" "--bh 16 --n 8192 --d 128" "--bh 64 --n 512 --d 64" "--bh 32 --n 2048 --d 128" "--bh 256 --n 512 --d 32 --causal 1" "--bh 16 --n 8192 --d 32")
for shape in "${SHAPES[@]}" "--bh 3 --n 1000 --d 64 --causal 1" "--bh 2 --n 65 --d 128" "--bh 5 --n 130 --d 32"; do
  echo "check [$shape] old $($OLD --mode rand --dtype f32 --kernel auto --check 1 --iters 2 $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*') | new $($NEW --mode rand --dtype f32 --kernel auto --check 1 --iters 2 $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out
done
for r in 1 2 3 4 5; do
  echo "rep $r" >> $out
  for shape in "${SHAPES[@]}"; do
    echo "[$shape] old $(run $OLD --dtype f32 --kernel auto $shape) | new $(run $NEW --dtype f32 --kernel auto $shape)" >> $out
  done
done
grep "^check" $out | cut -c1-220
grep -v "^check\|^rep\|^#" $out | grep old | awk -F'[][]' '{print $2}' | sort -u | while read s; do o=$(grep -F "[$s]" $out | grep -v check | grep old | sed 's/.*old "ms": \([0-9.]*\) | new "ms": \([0-9.]*\)/\1 \2/' | awk '{a+=$1;b+=$2;n++} END {printf "old %.4f new %.4f (%+.1f %%)", a/n, b/n, (b/a-1)*100}'); echo "$s: $o"; done
