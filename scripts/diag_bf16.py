import sys; sys.path.insert(0,'.')
import torch
from tests.test_gpu_step import run_both
from tests.golden_util import rel_l2
tr, _, res = run_both("dusty2", (64, 256), 128, 64, 256, 4, amp=True)
sc_ref, ex, synth, gD, gG, scal = res[0]
print("scal", scal, sc_ref)
for k in ("depth_orig","confidence","depth"): print(k, rel_l2(synth[k], ex["synth"][k]))
print("mask mismatch", (synth["mask"] != ex["synth"]["mask"]).float().mean().item())
for k,v in ex["grad_D"].items(): print("gD", k, rel_l2(gD[k], v))
for k,v in ex["grad_G"].items(): print("gG", k, rel_l2(gG[k], v))
