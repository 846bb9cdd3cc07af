#!/bin/bash
# profiles/power_sample.sh -- socket power and shader clock (rocm-smi, four samples a second apart) while fa_driver launches one kernel back to back:
# evidence for the power-budget reading of DESIGN.md section 5.  Run on the GPU box: gpurun -- "bash profiles/power_sample.sh"
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
out=gpurun_out/$TAG/power.txt
: > $out
D=./flashattention.c_amd/fa_driver
sample() {  # label, driver args
  echo "== $1" >> $out; shift
  $D --mode rand --check 0 --warmup 200 --iters 40000 "$@" > /tmp/drv.json 2>&1 &
  pid=$!
  sleep 2
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction\|Average" >> $out
    echo "--" >> $out
    sleep 1
  done
  wait $pid
  cat /tmp/drv.json | tail -1 | cut -c1-300 >> $out
}
echo "idle:" >> $out; rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|max" >> $out
sample "c4 bf16 P (x4)" --dtype bf16 --bh 16 --n 8192 --d 64
sample "c4 two-term P" --dtype bf16 --out_f32 1 --kernel pb2 --bh 16 --n 8192 --d 64
sample "c4 at scale 4 (sparser P)" --dtype bf16 --bh 16 --n 8192 --d 64 --scale 4
sample "c3 fp32 tensors (split kernel: fp16 terms for Q.K^T since round 5)" --dtype f32s --kernel auto --bh 16 --n 8192 --d 64 --iters 12000
sample "c3 exact fp32" --dtype f32 --bh 16 --n 8192 --d 64 --iters 4000
sample "d128" --dtype bf16 --bh 16 --n 8192 --d 128 --iters 20000
sample "d32 (round 6: is the d = 32 loop at the power cap too?)" --dtype bf16 --bh 16 --n 8192 --d 32 --iters 40000
sample "c4 causal" --dtype bf16 --bh 16 --n 8192 --d 64 --causal 1 --iters 50000
cat $out
