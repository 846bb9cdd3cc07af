#!/bin/bash
# A/B of two libraries through the C driver, interleaved: $1 = repetitions; remaining args = driver arguments
reps=$1; shift
D=flashattention.c_amd
for r in $(seq $reps); do
  a=$($D/fa_driver "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1)
  b=$($D/fa_driver_ablation "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1)
  echo "rep $r: product $a | ablation $b   [$*]"
done
