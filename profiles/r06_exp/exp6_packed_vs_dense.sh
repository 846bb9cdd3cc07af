#!/bin/bash
# r06 experiment 6: does the packed (B, T, 3C) layout of fa_forward_packed_qkv cost anything against the dense (BH, N, d) layout of the same shape?
D=flashattention.c_amd/fa_driver
for r in 1 2 3; do
  echo "rep $r: packed $($D --mode llmc --iters 100 2>&1 | tail -1 | grep -o '"ms": [0-9.]*') | dense 72 x 4096 causal scale 0.125 $($D --mode rand --check 0 --warmup 50 --iters 100 --dtype f32 --kernel auto --bh 72 --n 4096 --d 64 --causal 1 --scale 0.125 2>&1 | grep -o '"ms": [0-9.]*' | head -1) | dense at scale 1 $($D --mode rand --check 0 --warmup 50 --iters 100 --dtype f32 --kernel auto --bh 72 --n 4096 --d 64 --causal 1 2>&1 | grep -o '"ms": [0-9.]*' | head -1) | exact $($D --mode rand --check 0 --warmup 20 --iters 30 --dtype f32 --kernel mfma --bh 72 --n 4096 --d 64 --causal 1 --scale 0.125 2>&1 | grep -o '"ms": [0-9.]*' | head -1)"
done
