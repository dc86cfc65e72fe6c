F="--soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 40 --warmup 10"
for prec in fp32x3 fp32; do for v in 1 0 1 0; do
  echo -n "$prec fuse_fp32=$v: "; DUSTY_GAN_FUSE_PROJ_FP32=$v python bench.py $F --precision $prec 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done
