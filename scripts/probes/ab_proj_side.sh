# Proj.weight's fused optimizer on a side stream beside the generator's weight gradients (DUSTY_GAN_PROJ_SIDE=1) against the
# single chain: ms per step (40 steps after 10), twice each, then the launches of one replayed step with the side stream
F="--soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 40 --warmup 10"
for v in 0 1 0 1; do
  echo -n "side=$v: "; DUSTY_GAN_PROJ_SIDE=$v python bench.py $F 2>gpurun_out/ab_side.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_device']['p50'], d['scalars_last_step'])" || tail -5 gpurun_out/ab_side.err
done
DUSTY_GAN_PROJ_SIDE=1 bash scripts/step_sequence.sh gpurun_out/ab_side > gpurun_out/ab_side_seq.txt 2>&1; tail -12 gpurun_out/ab_side_seq.txt
