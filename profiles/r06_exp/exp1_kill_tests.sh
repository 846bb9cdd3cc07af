#!/bin/bash
# r06 experiment 1: kill tests for VERDICT r05 items 1 (causal key shares) and 2 (K/V split pre-pass), plus this box's baseline.
# product library vs ablation library built with FA_EXTRA_ABL_FLAGS="-DFA_SPLIT_NOCVT=1" (timing only: K/V tiles stored unconverted).
D=flashattention.c_amd
out=gpurun_out/r06_exp1.txt
: > $out
run() { # lib-driver args...
  drv=$1; shift
  $drv --mode rand --check 0 --warmup 30 --iters 50 "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1
}
for r in 1 2 3; do
  echo "rep $r" >> $out
  echo "c3      product $(run $D/fa_driver --dtype f32 --kernel auto --bh 16 --n 8192 --d 64) | nocvt $(run $D/fa_driver_ablation --dtype f32 --kernel auto --bh 16 --n 8192 --d 64)" >> $out
  echo "c3caus  product $(run $D/fa_driver --dtype f32 --kernel auto --bh 16 --n 8192 --d 64 --causal 1) | nocvt $(run $D/fa_driver_ablation --dtype f32 --kernel auto --bh 16 --n 8192 --d 64 --causal 1)" >> $out
  echo "c2      product $(run $D/fa_driver --dtype f32 --kernel auto --bh 128 --n 1024 --d 64) | nocvt $(run $D/fa_driver_ablation --dtype f32 --kernel auto --bh 128 --n 1024 --d 64)" >> $out
  echo "c4      product $(run $D/fa_driver --dtype bf16 --kernel auto --bh 16 --n 8192 --d 64)" >> $out
  echo "c4caus  product $(run $D/fa_driver --dtype bf16 --kernel auto --bh 16 --n 8192 --d 64 --causal 1) | forceS2 $(FA_EXP_FORCE_S=2 run $D/fa_driver_ablation --dtype bf16 --kernel auto --bh 16 --n 8192 --d 64 --causal 1) | forceS4 $(FA_EXP_FORCE_S=4 run $D/fa_driver_ablation --dtype bf16 --kernel auto --bh 16 --n 8192 --d 64 --causal 1)" >> $out
  echo "c4pb2   product $(run $D/fa_driver --dtype bf16 --out_f32 1 --kernel auto --bh 16 --n 8192 --d 64)" >> $out
done
cat $out
# kernel-level split of the forced key-share launch (main kernel vs combine) and of the product's unsplit launch
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for tag in S2 S1; do
  if [ $tag = S2 ]; then export FA_EXP_FORCE_S=2; else unset FA_EXP_FORCE_S; fi
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- $R/$D/fa_driver_ablation --mode rand --check 0 --warmup 30 --iters 100 --dtype bf16 --kernel auto --bh 16 --n 8192 --d 64 --causal 1 > /tmp/prof_$tag.log 2>&1
  echo "== kernel stats, c4 causal, $tag" >> $R/$out
  find /tmp/prof_$tag -name '*kernel_stats.csv' -exec cat {} \; | cut -c1-220 >> $R/$out
done
cd $R
tail -20 $out
