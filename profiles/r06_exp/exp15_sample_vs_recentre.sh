#!/bin/bash
# r06 experiment 15: with the reference re-centred on the row sum, is the prologue's sampled reference (five extra K.Q^T passes, ~1 % of a c4 tile)
# still worth its time?  Ablation libraries: base = product; s0 = no sampling beyond sub-tile 0; rN = first re-centring N stages into the tile.
out=gpurun_out/r06_exp15.txt
: > $out
V="base s0 s0r4 s0r2 s1r4 r4"
SHAPES=("--variant 30 --bh 16 --n 8192 --d 64" "--variant 30 --bh 128 --n 8192 --d 64 --iters 10 --warmup 10" "--variant 50 --bh 16 --n 8192 --d 128" "--variant 50 --bh 16 --n 8192 --d 32" "--variant 50 --bh 16 --n 8192 --d 64 --causal 1" "--variant 30 --bh 16 --n 8192 --d 64 --out_f32 1 --kernel pb2 --variant 0" "--variant 30 --bh 32 --n 4096 --d 64")
for v in $V; do echo "check $v $(ab_old/$v/fa_driver_ablation --mode rand --dtype bf16 --variant 30 --bh 16 --n 8192 --d 64 --check 1 --iters 2 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*') | pb2 $(ab_old/$v/fa_driver_ablation --mode rand --dtype bf16 --out_f32 1 --kernel pb2 --bh 16 --n 8192 --d 64 --check 1 --iters 2 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')" >> $out; done
for r in 1 2 3 4; do
  for shape in "${SHAPES[@]}"; do
    for v in $V; do echo "$v [$shape] $(ab_old/$v/fa_driver_ablation --mode rand --dtype bf16 --check 0 --warmup 60 --iters 60 $shape 2>&1 | grep -o '"ms": [0-9.]*' | head -1)" >> $out; done
  done
done
grep "^check" $out
for shape in "${SHAPES[@]}"; do for v in $V; do echo "$shape | $v: $(grep -F "$v [$shape] " $out | sed 's/.*"ms": //' | awk '{a+=$1;n++} END {printf "%.4f", a/n}')"; done; done
