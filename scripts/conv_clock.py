"""The clock the chip HOLDS inside the dominant conv kernel (MI355X_MICROARCH.md, DVFS give-back, check 6): the diagnostic
build of the ping-pong conv (make -C dusty_gan_amd/csrc diag DIAGBITS=8) stamps s_memtime (shader cycles) and s_memrealtime
(100 MHz) around workgroup 0's tile loop; after >= 2 s of back-to-back launches on random data the quotient is the in-kernel
clock.  Prints one JSON line.  Never the library the tests or the benchmark use: bench.py runs this file as a CHILD process
(DUSTY_GAN_LIB_DIAG=1) after its timed region and prices `roofline.frac_at_held_clock` against it.
usage: DUSTY_GAN_LIB_DIAG=1 python scripts/conv_clock.py [seconds]"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSTY_GAN_LIB_DIAG", "1")
import torch  # noqa: E402

from dusty_gan_amd import _lib as L  # noqa: E402
from dusty_gan_amd.engine import Ops  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
dev = "cuda"
o = Ops(torch.bfloat16)
o.force = 2
torch.manual_seed(0)
# the step's launches of the family, one of each geometry class (name, mode, adj, Hc, Wc, K, N, samples)
LAYERS = [("down3 fwd 2B", L.MODE_S2, 0, 8, 128, 128, 256, 64), ("up1 fwd B", L.MODE_UP, 0, 4, 64, 512, 256, 32),
          ("down4 bwd 2B", L.MODE_UP, 1, 4, 64, 512, 256, 64)]
out_rec = {}
for name, mode, adj, Hc, Wc, K, N, n in LAYERS:
    if mode == L.MODE_S2:
        hin, win, ho, wo = 2 * Hc, 2 * Wc, Hc, Wc
    else:
        hin, win, ho, wo = Hc, Wc, 2 * Hc, 2 * Wc
    x = torch.randn(n * hin * win * K, device=dev).to(torch.bfloat16)
    w = torch.randn(16 * N * K, device=dev).to(torch.bfloat16)
    out = torch.empty(n * ho * wo * N, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(n * ho * wo * N, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    epi = L.EPI_MASK if adj else L.EPI_LRELU

    def run():
        o.conv(mode, adj, True, n, Hc, Wc, K, N, x, (hin * win * K, K, 1), out, (ho * wo * N, N, 1), w.data_ptr(), 0.01, epi,
               bias=None if adj else bias.data_ptr(), bias_mod=N, aux=aux if adj else None)
    run()
    torch.cuda.synchronize()
    t0 = time.time()
    clocks = []
    while time.time() - t0 < seconds / len(LAYERS):
        for _ in range(200):
            run()
        torch.cuda.synchronize()
        st = out[:256].view(torch.float32).tolist()
        for k in (0, 4):                       # waves 0 and 4 of workgroup 0
            ph = st[64 + 8 * k:64 + 8 * k + 6]
            if ph[5] > 0:
                clocks.append(ph[2] / ph[5] * 0.1)   # shader cycles per 10 ns tick -> GHz
    out_rec[name] = round(statistics.median(clocks), 4) if clocks else None
vals = [v for v in out_rec.values() if v]
print(json.dumps({"clock_ghz": round(statistics.median(vals), 4) if vals else None, "per_layer": out_rec,
                  "method": "s_memtime / s_memrealtime around workgroup 0's tile loop, diagnostic build of pp::conv_pp_kernel, "
                            f"{seconds:.0f} s of back-to-back launches on random bf16 data"}))
