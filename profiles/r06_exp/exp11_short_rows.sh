#!/bin/bash
# r06 experiment 11: where a short-row launch spends its time, and whether any shipped tiling beats the dispatch there.
out=gpurun_out/r06_exp11.txt
: > $out
D=flashattention.c_amd/fa_driver; A=flashattention.c_amd/fa_driver_ablation
for shape in "--bh 128 --n 1024 --d 32" "--bh 128 --n 1024 --d 64" "--bh 16 --n 8192 --d 32"; do
  echo "## sweep bf16 [$shape]" >> $out
  for v in 0 1 7 24 50; do
    for r in 1 2 3; do $D --mode rand --dtype bf16 --check 0 --warmup 50 --iters 100 --variant $v $shape 2>&1 | grep -o '"variant": [0-9]*\|"ms": [0-9.]*' | tr '\n' ' ' >> $out; echo >> $out; done
  done
done
echo "## prof4 variant 70 (cycle-stamped x2 kernel)" >> $out
for shape in "--bh 128 --n 1024 --d 32" "--bh 16 --n 8192 --d 32" "--bh 128 --n 1024 --d 64"; do
  for r in 1 2; do echo "[$shape] $($A --mode prof4 --variant 70 --dtype bf16 $shape 2>&1 | tail -1)" >> $out; done
done
echo "## fp32 default, c2 (128 x 1024 d = 64) and 128 x 1024 d = 32: tilings" >> $out
for shape in "--bh 128 --n 1024 --d 64" "--bh 128 --n 1024 --d 32"; do
  for v in 0 1 2 3 4; do
    for r in 1 2 3; do echo "[$shape] v$v $($D --mode rand --dtype f32s --check 0 --warmup 50 --iters 100 --variant $v $shape 2>&1 | grep -o '"ms": [0-9.]*')" >> $out; done
  done
done
cat $out
