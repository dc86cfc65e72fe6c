#!/bin/bash
# Same-box pair of step sequences: the sources in .ab_prev/ (scripts/ab_prev.sh <ref>) and the working tree, back to back.
#   usage: scripts/seq_pair.sh <outdir>
out=${1:-gpurun_out/seqpair}
root=$(pwd)
mkdir -p $out
( cd .ab_prev && bash scripts/step_sequence.sh gpurun_out/sp1 > $root/$out/prev.txt 2>&1 )
bash scripts/step_sequence.sh $out/cur > $out/cur.txt 2>&1
( cd .ab_prev && bash scripts/step_sequence.sh gpurun_out/sp2 > $root/$out/prev2.txt 2>&1 )
bash scripts/step_sequence.sh $out/cur2 > $out/cur2.txt 2>&1
tail -1 $out/prev.txt $out/cur.txt $out/prev2.txt $out/cur2.txt
