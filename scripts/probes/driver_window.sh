# the driver's form of the line, three times: ms_per_step, the first five timed steps, the soak beside it
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-roofline --no-clock 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['ms_per_step'], d['step_ms_device']['first_steps'], 'p50', d['step_ms_device']['p50'], 'soak', d['soak']['ms_per_step_wall'], d['soak']['vs_timed_region'])"; done
