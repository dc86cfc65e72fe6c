"""The N > 1 path on CPU: two gloo ranks, each with half the batch, exchanging through dusty_gan_amd.utils.dist exactly
as the trainer does (parameter broadcast, ONE SUM all-reduce per network on the flat gradient buffer, 1/world folded
into Adam, one packed scalar reduce) must reproduce the single-process full-batch step -- the reference's DDP
semantics (trainers/dcgan_amp.py:68-69, 235, 309, 319-323).  The per-rank compute is the CPU oracle (the HIP kernels
need a GPU; their single-rank parity is the -m gpu suite)."""
import datetime
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

# ranks of one node meet on the loopback interface (see tests/test_gpu_ddp.py); a bounded timeout instead of gloo's 30 min
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
PG_TIMEOUT = datetime.timedelta(seconds=240)

from oracle import dusty_oracle as O
from tests.golden_util import rel_l2

ARCH, SHAPE, NZ, CB, CM, B = "dusty2", (32, 64), 8, 4, 16, 4


def make_inputs():
    gen = torch.Generator().manual_seed(99)
    H, W = SHAPE
    G = O.init_G(f"{ARCH}/dcgan_eqlr", NZ, CB, CM, SHAPE, gen)
    D = O.init_D(1, CB, CM, SHAPE, gen)
    x = torch.rand(B, 1, H, W, generator=gen) * 2 - 1
    rand = {"z": torch.randn(B, NZ, generator=gen),
            "noise": {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=gen), torch.rand(B, 1, H, W, generator=gen)),
                      "image": O.logistic_noise(torch.rand(B, 1, 1, 1, generator=gen), torch.rand(B, 1, 1, 1, generator=gen))},
            "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
    return G, D, x, rand


def shard(x, rand, s):
    return x[s], {"z": rand["z"][s], "noise": {k: v[s] for k, v in rand["noise"].items()},
                  "aug": [{k: v[s] for k, v in rp.items()} for rp in rand["aug"]]}


class FlatNet:
    """the trainer's ParamStore idea on the CPU: one flat fp32 buffer per network + a flat gradient buffer"""

    def __init__(self, params):
        self.keys = [k for k in params if k not in O.PARAM_BUFFERS]
        self.shapes = [params[k].shape for k in self.keys]
        self.flat = torch.cat([params[k].flatten() for k in self.keys]).clone()
        self.grad = torch.zeros_like(self.flat)
        self.buffers = {k: params[k] for k in params if k in O.PARAM_BUFFERS}

    def as_dict(self, buf=None):
        buf = self.flat if buf is None else buf
        out, off = dict(self.buffers), 0
        for k, s in zip(self.keys, self.shapes):
            n = int(torch.tensor(s).prod())
            out[k] = buf[off:off + n].view(s)
            off += n
        return out


def grads_of(G, D, x, rand, cfg):
    """one oracle step WITHOUT parameter updates -> (grad_D, grad_G, scalars)"""
    Gc = {k: v.clone() for k, v in G.items()}
    Dc = {k: v.clone() for k, v in D.items()}
    cfg0 = O.StepConfig(arch=cfg.arch, lr_g=0.0, lr_d=0.0, ema_decay=1.0)
    sc, ex = O.train_step(Gc, Dc, {k: v.clone() for k, v in Gc.items()}, O.new_optim_state(Gc), O.new_optim_state(Dc), 1,
                          cfg0, x, rand, return_grads=True)
    return ex["grad_D"], ex["grad_G"], sc


def worker(rank, world, init_file, out_dir):
    from dusty_gan_amd.utils import dist as DD
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    torch.set_num_threads(2)
    G, D, x, rand = make_inputs()
    netG, netD = FlatNet(G), FlatNet(D)
    if rank != 0:  # ranks start from different weights; the broadcast must make them rank 0's
        netG.flat.add_(1.0)
        netD.flat.mul_(0.5)
    DD.broadcast_params([netG.flat, netD.flat], src=0)
    lb = DD.local_batch(B, world, 1)
    s = slice(rank * lb, (rank + 1) * lb)
    xs, rs = shard(x, rand, s)
    cfg = O.StepConfig(arch=ARCH)
    gD, gG, sc = grads_of(netG.as_dict(), netD.as_dict(), xs, rs, cfg)
    netD.grad.copy_(torch.cat([gD[k].flatten() for k in netD.keys]))
    netG.grad.copy_(torch.cat([gG[k].flatten() for k in netG.keys]))
    _, gsD = DD.allreduce_grads(netD.grad)
    _, gsG = DD.allreduce_grads(netG.grad)
    scal, finish = DD.mean_scalars(torch.tensor([sc[k] for k in sorted(sc)]))   # (asynchronous: `finish` completes it)
    scal = finish() if finish is not None else scal
    torch.save({"gD": netD.grad * gsD, "gG": netG.grad * gsG, "scal": scal, "G0": netG.flat, "keys": sorted(sc)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_step_equals_full_batch():
    world = 2
    with tempfile.TemporaryDirectory() as td:
        init_file = os.path.join(td, "init")
        mp.spawn(worker, args=(world, init_file, td), nprocs=world, join=True)
        outs = [torch.load(os.path.join(td, f"rank{r}.pt")) for r in range(world)]
    G, D, x, rand = make_inputs()
    netG, netD = FlatNet(G), FlatNet(D)
    gD, gG, sc = grads_of(G, D, x, rand, O.StepConfig(arch=ARCH))
    refD = torch.cat([gD[k].flatten() for k in netD.keys])
    refG = torch.cat([gG[k].flatten() for k in netG.keys])
    for o in outs:
        assert torch.equal(o["G0"], netG.flat)  # broadcast made every rank start from rank 0's weights
        assert rel_l2(o["gD"], refD) < 1e-4
        assert rel_l2(o["gG"], refG) < 1e-4
        for k, v in zip(o["keys"], o["scal"].tolist()):
            assert abs(v - sc[k]) < 1e-4 * max(1.0, abs(sc[k])), k
    assert torch.equal(outs[0]["gD"], outs[1]["gD"]) and torch.equal(outs[0]["gG"], outs[1]["gG"])


def test_local_batch_and_accumulation_schedule():
    from dusty_gan_amd.utils import dist as DD
    from dusty_gan_amd.utils.context_manager import gradient_accumulation
    assert DD.local_batch(256, 8, 1) == 32 and DD.local_batch(64, 2, 4) == 8
    with pytest.raises(AssertionError):
        DD.local_batch(30, 4, 1)
    assert list(gradient_accumulation(3, True, ())) == [0, 1, 2]  # the reference's yield (utils/context_manager.py:35)
    from dusty_gan_amd.utils.context_manager import sync_round
    assert [sync_round(i, 3) for i in range(3)] == [False, False, True]
    assert DD.world_size() == 1 and DD.rank() == 0
    g = torch.ones(4)
    assert DD.allreduce_grads(g) == (None, 1.0)


def agree_worker(rank, world, init_file, out_dir):
    from dusty_gan_amd.utils import dist as DD
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    # round 0: every rank captured; round 1: rank 1 could not; round 2: nobody could
    seen = [DD.all_agree(True), DD.all_agree(rank != 1), DD.all_agree(False)]
    what = [DD.capture_decision(True, ok, ev, world) for ok, ev in zip((True, rank != 1, False), seen)]
    torch.save({"seen": seen, "what": what}, os.path.join(out_dir, f"agree{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_agree_on_the_capture_outcome():
    """One rank that cannot capture the collectives inside the step's hipGraph takes every rank to the segmented replay
    (trainers/dcgan_amp.py `_step_graph`): the flag is MIN-reduced, and the decision table is the same on every rank."""
    from dusty_gan_amd.utils import dist as DD
    world = 2
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(agree_worker, args=(world, os.path.join(td, "init"), td), nprocs=world, join=True)
        outs = [torch.load(os.path.join(td, f"agree{r}.pt")) for r in range(world)]
    for o in outs:
        assert o["seen"] == [True, False, False]
        assert o["what"] == ["keep", "segments", "segments"]   # the rank that DID capture discards its graph too
    # the table itself (no process group: the flag is the rank's own)
    assert DD.all_agree(True) is True and DD.all_agree(False) is False
    assert DD.capture_decision(True, True, True, 8) == "keep"
    assert DD.capture_decision(True, True, False, 8) == "segments"
    assert DD.capture_decision(True, False, False, 1) == "segments"
    assert DD.capture_decision(False, True, True, 8) == "keep"
    assert DD.capture_decision(False, False, False, 8) == "eager"     # a multi-rank job does not die of a refused capture
    assert DD.capture_decision(False, False, False, 1) == "raise"
