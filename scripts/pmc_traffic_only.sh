#!/bin/bash
# Re-collect profiles/rNN_pmc_traffic.json only (two PMC passes on the eager step), e.g. after an edit that changes the
# source hash but not the kernels.   usage: scripts/pmc_traffic_only.sh <tag> [outdir]
tag=${1:-r03}
out=${2:-gpurun_out/traffic_$tag}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export DUSTY_GAN_GRAPH=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/$out/fetch -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > /dev/null 2> $root/$out/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/$out/write -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > /dev/null 2> $root/$out/write.log
unset DUSTY_GAN_GRAPH
cd $root
python3 scripts/pmc_traffic.py $out/${tag}_pmc_traffic.json $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1)
rm -rf $out/fetch $out/write
