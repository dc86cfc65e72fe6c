"""The data-parallel path of the Trainer on REAL kernels: two ranks share the one GPU of the test box (gloo moves the
CUDA buffers through the host, RCCL refuses two ranks on one device) and must reproduce the single-process
full-batch step: identical parameter broadcast, gradient averaging (one all-reduce per network), packed scalar
reduce.  The 8-GPU RCCL run uses exactly this code with backend "nccl"."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dusty_oracle as O
from tests.golden_util import rel_l2

pytestmark = pytest.mark.gpu

ARCH, SHAPE, NZ, CB, CM, B = "dusty2", (32, 64), 8, 4, 16, 4


def make_rand():
    gen = torch.Generator().manual_seed(77)
    H, W = SHAPE
    x = torch.rand(B, 1, H, W, generator=gen) * 2 - 1
    rand = {"z": torch.randn(B, NZ, generator=gen),
            "noise": {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=gen), torch.rand(B, 1, H, W, generator=gen)),
                      "image": O.logistic_noise(torch.rand(B, 1, 1, 1, generator=gen), torch.rand(B, 1, 1, 1, generator=gen))},
            "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
    return x, rand


def shard(x, rand, s):
    return x[s].contiguous(), {"z": rand["z"][s], "noise": {k: v[s] for k, v in rand["noise"].items()},
                               "aug": [{k: v[s] for k, v in rp.items()} for rp in rand["aug"]]}


def run_steps(tr, x, rand, steps=2):
    out = []
    for i in range(steps):
        xd = x.to("cuda")
        s = tr.step(i, reals=[(xd, torch.ones_like(xd))], rands=[rand])
        out.append(dict(s.items()))
    return out


def worker(rank, world, init_file, out_dir, sdG, sdD):
    from tests.test_gpu_step import make_trainer
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # ranks build DIFFERENT nets; the constructor's broadcast must fix that
    tr = make_trainer(ARCH, True, SHAPE, NZ, CB, CM, B // world)
    if rank == 0:
        tr.G.load_state_dict(sdG)
        tr.D.load_state_dict(sdD)
    from dusty_gan_amd.utils import dist as DD
    DD.broadcast_params([tr.G.store.flat, tr.D.store.flat], src=0)
    tr.G.refresh() if hasattr(tr.G, "refresh") else tr.G.backbone.refresh()
    tr.D.refresh()
    tr.G_ema.store.flat.copy_(tr.G.store.flat)
    x, rand = make_rand()
    lb = B // world
    xs, rs = shard(x, rand, slice(rank * lb, (rank + 1) * lb))
    scal = run_steps(tr, xs, rs)
    torch.save({"G": tr.G.store.flat.cpu(), "D": tr.D.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "scal": scal},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process():
    from tests.test_gpu_step import make_trainer
    torch.manual_seed(5)
    ref = make_trainer(ARCH, True, SHAPE, NZ, CB, CM, B)
    sdG = {k: v.detach().cpu().clone() for k, v in ref.G.state_dict().items()}
    sdD = {k: v.detach().cpu().clone() for k, v in ref.D.state_dict().items()}
    x, rand = make_rand()
    scal_ref = run_steps(ref, x, rand)
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(worker, args=(2, os.path.join(td, "init"), td, sdG, sdD), nprocs=2, join=True)
        outs = [torch.load(os.path.join(td, f"r{r}.pt")) for r in range(2)]
    for o in outs:
        assert rel_l2(o["D"], ref.D.store.flat.cpu()) < 1e-4
        assert rel_l2(o["G"], ref.G.store.flat.cpu()) < 1e-4
        assert rel_l2(o["E"], ref.G_ema.store.flat.cpu()) < 1e-4
        for s_got, s_ref in zip(o["scal"], scal_ref):
            for k, v in s_ref.items():
                assert abs(s_got[k] - v) < 1e-4 * max(1.0, abs(v)), k
    assert torch.equal(outs[0]["G"], outs[1]["G"]) and torch.equal(outs[0]["D"], outs[1]["D"])


def graph_worker(rank, world, init_file, out_dir, use_graph):
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_GRAPH"] = "1" if use_graph else "0"
    os.environ["DUSTY_GAN_GRAPH_DDP"] = "1"  # the segmented replay is opt-in for world > 1
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.manual_seed(300)  # same seed on both ranks: same initial nets, same device RNG streams
    tr = make_trainer(ARCH, True, SHAPE, NZ, CB, CM, B // world)
    scal = [dict(tr.step(i).items()) for i in range(5)]
    segs = 0 if tr._graph is None else sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph)
    torch.save({"G": tr.G.store.flat.cpu(), "D": tr.D.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "scal": scal,
                "segs": segs}, os.path.join(out_dir, f"g{int(use_graph)}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_segmented_graph_matches_eager():
    """world > 1 (opt-in, DUSTY_GAN_GRAPH_DDP=1): the step is replayed as hipGraph segments with the collectives (D gradient all-reduce, Proj operand
    gather, G tail all-reduce) called between them; 5 iterations must train exactly like the eager launch sequence."""
    res = {}
    for use_graph in (True, False):
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(graph_worker, args=(2, os.path.join(td, "init"), td, use_graph), nprocs=2, join=True)
            res[use_graph] = [torch.load(os.path.join(td, f"g{int(use_graph)}_r{r}.pt")) for r in range(2)]
    assert res[True][0]["segs"] == 4 and res[False][0]["segs"] == 0  # 3 collective points -> 4 graph segments
    for r in range(2):
        a, b = res[True][r], res[False][r]
        for k in ("G", "D", "E"):
            assert rel_l2(a[k], b[k]) < 1e-5, k
        for x, y in zip(a["scal"], b["scal"]):
            for k in x:
                assert abs(x[k] - y[k]) < 1e-4 * max(1.0, abs(y[k])), k
    assert torch.equal(res[True][0]["G"], res[True][1]["G"])


def fused_worker(rank, world, init_file, out_dir, fuse, pl=0.0):
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_GRAPH"] = "0"
    os.environ["DUSTY_GAN_FUSE_PROJ"] = "1" if fuse else "0"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.manual_seed(400)
    # nz = 128 and Np = 2 * 4 * 16 = 128: the smallest Proj the MFMA epilogue takes; global batch 2 x 40 = 80 > 64
    tr = make_trainer("dusty1", True, (32, 64), 128, 4, 16, 40, amp=True, pl=pl)
    scal = [dict(tr.step(i).items()) for i in range(3)]
    used = tr.optim_G.regen_grad is not None  # set only while Proj.weight's gradient lives inside the optimizer kernel
    torch.save({"G": tr.G.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "V": tr.G.store.v.cpu(), "scal": scal,
                "used": used}, os.path.join(out_dir, f"f{int(fuse)}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pl", [0.0, 2.0])
def test_two_ranks_fused_proj_optimizer_matches_unfused(pl):
    """world > 1, bf16: Proj.weight's global-batch gradient is formed from the all-gathered operands inside the optimizer
    (epilogue of the MFMA gradient GEMM) instead of being written and re-read; 3 iterations must train like the unfused
    sequence (same bf16 operands, same fp32 accumulation -> agreement far below bf16 resolution)."""
    res = {}
    for fuse in (True, False):
        with tempfile.TemporaryDirectory() as td:
            # pl > 0: the path-length block appends its two operand pairs to what is all-gathered (3 x 40 rows per rank)
            mp.spawn(fused_worker, args=(2, os.path.join(td, "init"), td, fuse, pl), nprocs=2, join=True)
            res[fuse] = [torch.load(os.path.join(td, f"f{int(fuse)}_r{r}.pt")) for r in range(2)]
    assert all(o["used"] for o in res[True]) and not any(o["used"] for o in res[False])
    for r in range(2):
        a, b = res[True][r], res[False][r]
        # not bit-equal: the split-K atomics of the other weight gradients reorder fp32 sums from run to run, and a
        # last-bit difference can flip a bf16 rounding of the next step's shadows; a wrong kernel would be O(1) off
        assert rel_l2(a["G"], b["G"]) < 1e-4, rel_l2(a["G"], b["G"])
        assert rel_l2(a["E"], b["E"]) < 1e-4, rel_l2(a["E"], b["E"])
        assert rel_l2(a["V"], b["V"]) < 5e-3, rel_l2(a["V"], b["V"])
        for x, y in zip(a["scal"], b["scal"]):
            for k in x:
                assert abs(x[k] - y[k]) < 5e-3 * max(1.0, abs(y[k])), (k, x[k], y[k])
    assert torch.equal(res[True][0]["G"], res[True][1]["G"])
