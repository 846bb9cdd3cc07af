#!/bin/bash
# profiles/dispatch_sweep.sh PART -- the sweeps behind profiles/r03_short_rows.txt: every tiling of the product library against the
# dispatch (variant 0) over a grid of shapes, one line per run ("dtype [kernel] d D bh BH n N c CAUSAL [v VARIANT] ms").  Run on the GPU
# box (gpurun -- 'bash profiles/dispatch_sweep.sh short > gpurun_out/sweep_short.txt'), then python3 profiles/dispatch_sweep_report.py FILE.
# PART: f32short (round 5: the fp32 half of short) | short (rows of 128 .. 2048 keys, fp32 modes and bf16 variants) | small (bf16, small grids) | mid (bf16 d = 64 mid-size) |
#       f32mid (fp32 tensors, mid-size) | f32auto (fp32 AUTO against unsplit) | bf16auto (bf16 AUTO against unsplit / phase) | long (bf16 long rows)
PART=${1:-short}
cd "${GRAFT_REPO_ROOT:-.}"
D=./flashattention.c_amd/fa_driver
PV='import json,sys; j=json.loads(sys.stdin.read()); print(j["dtype"], "d", j["d"], "bh", j["bh"], "n", j["n"], "c", j["causal"], "v", j["variant"], j["ms"])'
PK='import json,sys; j=json.loads(sys.stdin.read()); print(j["dtype"], j["kernel"], "d", j["d"], "bh", j["bh"], "n", j["n"], "c", j["causal"], j["ms"])'
runv() { $D --mode rand --check 0 "$@" 2>/dev/null | tail -1 | python3 -c "$PV" 2>/dev/null || true; }
runk() { $D --mode rand --check 0 "$@" 2>/dev/null | tail -1 | python3 -c "$PK" 2>/dev/null || true; }
case $PART in
short)
  for d in 64 32 128; do for c in 0 1; do for n in 128 256 384 512 640 768 896 1024 1280 1536 2048; do
    bh=$((131072 / n)); if [ $d = 128 ]; then vs="1 3 5"; else vs="1 3 4"; fi
    for v in 0 $vs; do runv --bh $bh --n $n --d $d --dtype f32s --variant $v --causal $c --warmup 100 --iters 50; done
  done; done; done
  for d in 64 32 128; do for c in 0 1; do for n in 128 256 512 768; do
    bh=$((131072 / n)); for v in 0 1 7 50; do runv --bh $bh --n $n --d $d --dtype bf16 --variant $v --causal $c --warmup 100 --iters 50; done
  done; done; done ;;
f32short)   # round 5: the fp32 half of `short` (the split kernel's K.Q'^T went to fp16 terms, hi.hi first: is choose_split() still right?)
  for d in 64 32 128; do for c in 0 1; do for n in 128 256 384 512 640 768 896 1024 1280 1536 2048; do
    bh=$((131072 / n)); if [ $d = 128 ]; then vs="1 3 5"; else vs="1 3 4"; fi
    for v in 0 $vs; do runv --bh $bh --n $n --d $d --dtype f32s --variant $v --causal $c --warmup 100 --iters 50; done
  done; done; done ;;
small)
  for c in 0 1; do for n in 256 512 1024 1536 2048 3072; do for bh in 2 4 8 16 32; do
    for v in 0 1 7 50; do runv --bh $bh --n $n --d 64 --dtype bf16 --variant $v --causal $c --warmup 100 --iters 50; done
  done; done; done
  for d in 32 128; do for c in 0 1; do for n in 256 512 1536 3072; do for bh in 4 16 48; do
    for v in 0 1 50; do runv --bh $bh --n $n --d $d --dtype bf16 --variant $v --causal $c --warmup 100 --iters 50; done
  done; done; done; done ;;
mid)
  for c in 0 1; do for n in 1024 1536 2048 3072 4096 6144; do for bh in 4 8 16 24 32 48 64 128; do
    for v in 0 7 30 50; do runv --bh $bh --n $n --d 64 --dtype bf16 --variant $v --causal $c --warmup 60 --iters 30; done
  done; done; done ;;
f32mid)
  for d in 64 32; do for c in 0 1; do for n in 1024 2048 3072 4096 8192; do for bh in 2 4 8 16 32 64; do
    for v in 0 1 3 4; do runv --bh $bh --n $n --d $d --dtype f32s --variant $v --causal $c --warmup 40 --iters 20; done
  done; done; done; done
  for c in 0 1; do for n in 1024 2048 4096 8192; do for bh in 2 8 32; do
    for v in 0 1 3 5; do runv --bh $bh --n $n --d 128 --dtype f32s --variant $v --causal $c --warmup 40 --iters 20; done
  done; done; done ;;
f32auto)
  for d in 64 32 128; do for c in 0 1; do for n in 4096 8192 16384; do for bh in 1 2 4 8 12 16; do
    for k in auto split; do runk --bh $bh --n $n --d $d --dtype f32s --kernel $k --causal $c --warmup 40 --iters 20; done
  done; done; done; done ;;
bf16auto)
  for d in 64 32 128; do for c in 0 1; do for n in 4096 8192 16384; do for bh in 1 2 4 8 12 16; do
    runk --bh $bh --n $n --d $d --dtype bf16 --kernel auto --causal $c --warmup 60 --iters 30
    runk --bh $bh --n $n --d $d --dtype bf16 --kernel mfma --variant 50 --causal $c --warmup 60 --iters 30
    runk --bh $bh --n $n --d $d --dtype bf16 --kernel mfma --variant 1 --causal $c --warmup 60 --iters 30 | sed 's/mfma/phase/'
  done; done; done; done ;;
long)
  for n in 4096 8192 12288 16384 32768; do for bh in 12 16 20 24 32 40 48 64 128; do
    if [ $((bh * n)) -gt 2200000 ]; then continue; fi
    for v in 0 7 30 50; do runv --bh $bh --n $n --d 64 --dtype bf16 --variant $v --causal 0 --warmup 20 --iters 10; done
  done; done ;;
esac
