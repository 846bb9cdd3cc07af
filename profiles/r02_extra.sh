#!/bin/bash
# profiles/r02_extra.sh -- run on the GPU box: the register-file microbenchmark and the in-kernel cycle stamps of DESIGN.md section 4
# (needs `python flashattention.c_amd/build.py --ablation` and profiles/ubench/ubench_regfile built).  Writes under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
$R/profiles/ubench/ubench_regfile > $OUT/r02_ubench_regfile.txt 2>&1
# per-instruction issue costs alone and beside an MFMA (hipcc --offload-arch=gfx950 -O3 -o ubench_valu_mix ubench_valu_mix.hip)
[ -x $R/profiles/ubench/ubench_valu_mix ] && $R/profiles/ubench/ubench_valu_mix > $OUT/r02_ubench_valu_mix.txt 2>&1
D=$R/flashattention.c_amd/fa_driver_ablation
{
echo "# fa_driver_ablation --mode prof4: s_memtime stamps around the fast loop / prologue / epilogue of every wave of the NB = 4 kernel (c4 unless noted)"
echo "# variants: 60 full, 61 no LDS fragment reads, 62 no LDS-DMA issue, 63 no VALU units, 64 no MFMA, 65 K fragments not re-read,"
echo "#           66 no DMA wait + barrier, 67 no waits for V^T fragments, 68 MFMA + VALU only"
for r in 1 2; do for v in 60 61 62 63 64 65 66 67 68; do $D --mode prof4 --check 0 --variant $v; done; done
echo "# BH = 128 (c5 per-GPU shard)"
$D --mode prof4 --check 0 --variant 60 --bh 128
} > $OUT/r02_prof4_cycles.txt 2>&1
tail -3 $OUT/r02_prof4_cycles.txt
