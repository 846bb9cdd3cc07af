#!/bin/bash
# r06 experiment 10: fp32 split kernel (FA_KERNEL_AUTO for fp32 tensors), pipelined pass re-centred on the row sum after 16, 48, 144 tiles.
out=gpurun_out/r06_exp10.txt
: > $out
NEW=flashattention.c_amd/fa_driver; OLD=ab_old/fa_driver
run() { drv=$1; shift; $drv --mode rand --check 0 --warmup 40 --iters 30 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1; }
for shape in "--bh 16 --n 8192 --d 64" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 16 --n 8192 --d 128" "--bh 16 --n 8192 --d 32" "--bh 4 --n 16384 --d 64 --scale 2" "--bh 16 --n 8192 --d 64 --scale 0.125" "--bh 3 --n 5000 --d 128 --causal 1"; do
  echo "check [$shape] old $($OLD --mode rand --check 1 --iters 2 --dtype f32 --kernel auto $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*') | new $($NEW --mode rand --check 1 --iters 2 --dtype f32 --kernel auto $shape 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out
done
for r in 1 2 3 4 5; do
  echo "rep $r" >> $out
  for shape in "--bh 16 --n 8192 --d 64" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 16 --n 8192 --d 128" "--bh 16 --n 8192 --d 128 --causal 1" "--bh 16 --n 8192 --d 32" "--bh 16 --n 8192 --d 64 --scale 0.125" "--bh 128 --n 1024 --d 64" "--bh 32 --n 4096 --d 64" "--bh 1 --n 8192 --d 64"; do
    echo "[$shape] old $(run $OLD --dtype f32 --kernel auto $shape) | new $(run $NEW --dtype f32 --kernel auto $shape)" >> $out
  done
done
grep "^check" $out | cut -c1-220
grep -v "^check\|^rep" $out | awk -F'[][]' '{print $2}' | sort -u | while read s; do o=$(grep -F "[$s]" $out | grep -v check | sed 's/.*old "ms": \([0-9.]*\) | new "ms": \([0-9.]*\)/\1 \2/' | awk '{a+=$1;b+=$2;n++} END {printf "old %.4f new %.4f (%+.1f %%)", a/n, b/n, (b/a-1)*100}'); echo "$s: $o"; done
