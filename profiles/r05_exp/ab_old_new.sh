#!/bin/bash
# A/B of the previous library (ab_old/: built from the previous commit's sources, same driver) against the current one, interleaved:
# $1 = repetitions; remaining args = driver arguments
reps=$1; shift
for r in $(seq $reps); do
  a=$(ab_old/fa_driver "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1)
  b=$(flashattention.c_amd/fa_driver "$@" 2>&1 | grep -o '"ms": [0-9.]*' | head -1)
  echo "rep $r: old $a | new $b   [$*]"
done
