#!/bin/bash
# full collection of the final binary: rocprofv3 stats + PMC passes, bench lines, config table, README-shape harness, power
TAG=$1
bash profiles/collect.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
bash profiles/config_table.sh > gpurun_out/${TAG}_config_table.txt 2> gpurun_out/${TAG}_readme_shapes_harness.txt
bash profiles/power_sample.sh $TAG > /dev/null 2>&1 || true
cp gpurun_out/$TAG/power.txt gpurun_out/${TAG}_power.txt
ls gpurun_out/prof_$TAG | head -40
tail -5 gpurun_out/collect_$TAG.log
