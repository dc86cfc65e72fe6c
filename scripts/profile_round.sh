#!/bin/bash
# The profiles a round commits (run on the GPU box from the repo root): kernel-trace stats of bench.py, HBM-side traffic
# (two PMC passes on the eager step), SQ counters of the eager step, and the bench lines of the other configurations.
#   usage: scripts/profile_round.sh <tag, e.g. r02a> [outdir]
tag=${1:-r03}
out=${2:-gpurun_out/prof_$tag}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline > $root/$out/bench_under_profiler.json 2> $root/$out/kt.log
export DUSTY_GAN_GRAPH=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/$out/fetch -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > /dev/null 2> $root/$out/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/$out/write -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > /dev/null 2> $root/$out/write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $root/$out/sq -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > /dev/null 2> $root/$out/sq.log
unset DUSTY_GAN_GRAPH
cd $root
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_bench_bf16_b32.csv
python3 scripts/pmc_traffic.py $out/${tag}_pmc_traffic.json $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $out/${tag}_pmc_sq.json $(find $out/sq -name "*counter_collection.csv" | head -1) > $out/${tag}_pmc_sq_summary.txt
# bench.py's `roofline.traffic` reads the committed PMC file (bench.PMC_FILE): put this collection there first, so that the
# bench lines below carry it (same kernel sources -> the hash matches); the caller copies the same file into profiles/
pmc_file=$(python3 -c "import re; print(re.search(r'^PMC_FILE = \"([^\"]+)\"', open('bench.py').read(), re.M).group(1))")
cp $out/${tag}_pmc_traffic.json $pmc_file
# bench lines: the driver's default, then the other configurations of SURVEY.md §8d
python3 bench.py > $out/${tag}_bench_config2.json 2> /dev/null
python3 bench.py --arch dusty1 --no-cpu-baseline --no-other-configs > $out/${tag}_bench_config3_dusty1.json 2> /dev/null
python3 bench.py --arch dusty2 --no-cpu-baseline --no-other-configs > $out/${tag}_bench_config4_share_dusty2.json 2> /dev/null
python3 bench.py --arch dusty2 --shape 128 2048 --batch 64 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --soak-steps 200 > $out/${tag}_bench_config5_share_128x2048_b64.json 2> /dev/null
python3 bench.py --precision fp32 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --soak-steps 100 > $out/${tag}_bench_config2_fp32_parity.json 2> /dev/null
python3 bench.py --precision fp32x3 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --soak-steps 300 > $out/${tag}_bench_config2_fp32x3.json 2> /dev/null
python3 bench.py --gp 0 --no-augment --no-cpu-baseline --no-other-configs > $out/${tag}_bench_config2_nogp_noaug.json 2> /dev/null
# the multi-rank schedule on one GPU through RCCL (a process group of one rank): collectives inside the graph / between segments
DUSTY_GAN_FORCE_SEG=1 DUSTY_BENCH_BACKEND=nccl python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline > $out/${tag}_bench_schedule_rccl_1rank_in_graph.json 2> /dev/null
DUSTY_GAN_FORCE_SEG=1 DUSTY_BENCH_BACKEND=nccl DUSTY_GAN_GRAPH_COMM=0 python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline > $out/${tag}_bench_schedule_rccl_1rank_segments.json 2> /dev/null
python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline > $out/${tag}_bench_schedule_one_graph.json 2> /dev/null
bash scripts/step_sequence.sh $out/seq > $out/${tag}_step_sequence.txt 2>&1
rm -rf $out/kt $out/fetch $out/write $out/sq
ls -la $out
