#!/bin/bash
# r06 experiment 12b: clock-based take-turns, period bit 10 / 12 / 14 of the shader clock, against no turns (ablation libraries built with
# FA_EXTRA_ABL_FLAGS=-DFA_XN_TURN_BIT=..., -DFA_XN_TAKE_TURNS=0)
out=gpurun_out/r06_exp12b.txt
: > $out
for rep in 1 2 3; do
for t in turn0 bit14 bit15 bit16 bit17; do
  for shape in "--bh 16 --n 8192 --d 32" "--bh 128 --n 1024 --d 32" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 16 --n 8192 --d 32 --causal 1" "--bh 24 --n 4096 --d 32" "--bh 32 --n 2048 --d 64 --causal 1"; do
    echo "$t [$shape] $(ab_old/$t/fa_driver_ablation --mode rand --dtype bf16 --variant 50 --check 0 --warmup 60 --iters 60 $shape 2>&1 | grep -o '"ms": [0-9.]*' | head -1)" >> $out
  done
done
done
for t in turn0 bit14 bit15 bit16 bit17; do
  for shape in "--bh 16 --n 8192 --d 32" "--bh 128 --n 1024 --d 32"; do
    echo "$t [$shape]" >> $out; ab_old/$t/fa_driver_ablation --mode prof4 --variant 70 --dtype bf16 $shape 2>&1 | grep "^#" | grep -v "pos " >> $out
  done
done
for shape in "--bh 16 --n 8192 --d 32" "--bh 128 --n 1024 --d 32" "--bh 16 --n 8192 --d 64 --causal 1" "--bh 16 --n 8192 --d 32 --causal 1" "--bh 24 --n 4096 --d 32" "--bh 32 --n 2048 --d 64 --causal 1"; do
  for t in turn0 bit14 bit15 bit16 bit17; do
    echo "$shape $t: $(grep -F "$t [$shape] \"ms\"" $out | sed 's/.*"ms": //' | awk '{a+=$1;n++} END {printf "%.4f", a/n}')"
  done
done
grep -A2 "^turn0 \[\|^bit1[4567] \[" $out | grep -v '"ms"' | tail -40
