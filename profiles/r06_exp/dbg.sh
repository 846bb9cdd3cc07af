A=flashattention.c_amd/fa_driver_ablation
for n in 32 64 96 128 160 192 256; do
 for v in 50 52; do
  echo "n=$n variant $v: $($A --mode rand --check 1 --iters 2 --dtype bf16 --kernel mfma --variant $v --bh 2 --n $n --d 32 2>&1 | grep -o '"max_abs_err_vs_naive": [0-9.e+-]*, "nan": [0-9]*')"
 done
done
