"""Measure the engine's bf16 mode against the bf16-emulating oracle (and against the plain fp32 oracle) on one whole step:
what the tolerances of tests/test_gpu_configs.py are set from.  usage: python scripts/emu_gap.py [arch] [B] [H W]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.golden_util import rel_l2  # noqa: E402
from tests.test_gpu_step import _cos, run_both  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "none"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
shape = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 1024)
for emulate, sync in ((True, True), (False, True), (False, False)):
    t0 = time.time()
    tr, state, res = run_both(arch, shape, 512, 64, 512, B, amp=True, emulate=emulate, sync=sync)
    sc_ref, ex, synth, gD, gG, scal = res[0]
    print(f"--- arch {arch} B {B} {shape} emulate={emulate} sync={sync} ({time.time() - t0:.1f} s)")
    keys = ["loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial", "loss/D/gradient_penalty", "loss/G/adversarial"]
    for k, v in zip(keys, scal):
        print(f"  {k:28s} engine {v:+.6f} oracle {sc_ref[k]:+.6f} rel {abs(v - sc_ref[k]) / max(1.0, abs(sc_ref[k])):.2e}")
    for k in synth:
        if k == "mask":
            print("  mask mismatch", float((synth[k] != ex["synth"][k]).float().mean()))
        else:
            print(f"  {k:12s} rel_l2 {rel_l2(synth[k], ex['synth'][k]):.2e}")
    for name, got, ref in (("grad_D", gD, ex["grad_D"]), ("grad_G", gG, ex["grad_G"])):
        for k, v in ref.items():
            if v.abs().max() > 0:
                print(f"  {name} {k:40s} rel_l2 {rel_l2(got[k], v):.3e} 1-cos {1 - _cos(got[k], v):.2e}")
    del tr
    torch.cuda.empty_cache()
