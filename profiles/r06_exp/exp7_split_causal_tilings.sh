#!/bin/bash
# r06 experiment 7: fp32 split kernel, causal, many slabs of 2048 .. 4096 keys (the llm.c harness size is 72 x 4096): tilings 1 / 3 / 4 against the dispatch
D=flashattention.c_amd/fa_driver
for shape in "72 4096" "128 4096" "64 4096" "32 4096" "128 2048" "256 2048" "48 6144" "96 3072"; do
  set -- $shape
  line="bh=$1 n=$2:"
  for v in 0 1 3 4; do
    line="$line v$v $($D --mode rand --check 0 --warmup 30 --iters 40 --dtype f32s --variant $v --bh $1 --n $2 --d 64 --causal 1 --scale 0.125 2>&1 | grep -o '"ms": [0-9.]*' | head -1 | cut -d' ' -f2)"
  done
  echo "$line"
done
