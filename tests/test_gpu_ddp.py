"""The data-parallel path of the Trainer on REAL kernels: two ranks share the one GPU of the test box (gloo moves the
CUDA buffers through the host, RCCL refuses two ranks on one device) and must reproduce the single-process
full-batch step: identical parameter broadcast, gradient averaging (one all-reduce per network), packed scalar
reduce.  The 8-GPU RCCL run uses exactly this code with backend "nccl"."""
import datetime
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dusty_oracle as O
from tests.golden_util import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

# Ranks of one node meet on the loopback interface: without this gloo binds to whatever the box's hostname resolves to,
# and on a box where that address is not reachable the ranks wait for each other until the process-group timeout (one
# MI355X box of the pool hung this file for 18 minutes, twice).  A bounded timeout turns any such wait into a failure.
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
PG_TIMEOUT = datetime.timedelta(seconds=240)

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
        xd = x.to(tr.device)
        s = tr.step(i, reals=[(xd, torch.ones_like(xd))], rands=[rand])
        out.append(dict(s.items()))
    return out


def worker(rank, world, init_file, out_dir, sdG, sdD, backend="gloo"):
    from tests.test_gpu_step import make_trainer
    if backend == "nccl":   # one device per rank (RCCL refuses two ranks on one device)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", init_method=f"file://{init_file}", rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank), timeout=PG_TIMEOUT)
    else:
        dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    torch.manual_seed(100 + rank)  # ranks build DIFFERENT nets; the constructor's broadcast must fix that
    tr = make_trainer(ARCH, True, SHAPE, NZ, CB, CM, B // world, gpu=rank if backend == "nccl" else 0)
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
    # (measured ~1e-8; 5e-4 leaves room for one flipped pixel of the hard Gumbel threshold - the 1.4e-4 cluster of
    #  scripts/resume_noise.py - where a wrong exchange is O(1e-2))
    for o in outs:
        assert rel_l2(o["D"], ref.D.store.flat.cpu()) < 5e-4
        assert rel_l2(o["G"], ref.G.store.flat.cpu()) < 5e-4
        assert rel_l2(o["E"], ref.G_ema.store.flat.cpu()) < 5e-4
        for s_got, s_ref in zip(o["scal"], scal_ref):
            for k, v in s_ref.items():
                assert abs(s_got[k] - v) < 1e-3 * max(1.0, abs(v)), k
    assert torch.equal(outs[0]["G"], outs[1]["G"]) and torch.equal(outs[0]["D"], outs[1]["D"])


def graph_worker(rank, world, init_file, out_dir, use_graph, full=False):
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_GRAPH"] = "1" if use_graph else "0"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    torch.manual_seed(300)  # same seed on both ranks: same initial nets, same device RNG streams
    if full:  # the benchmark's nets and image size, bf16, 8 images per rank
        tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8, amp=True)
    else:
        tr = make_trainer(ARCH, True, SHAPE, NZ, CB, CM, B // world)
    scal = []
    for i in range(6 if full else 5):
        s = tr.step(i)
        if full and i % 2 == 1:
            torch.cuda.synchronize()  # host-side synchronisation between replays (bench.py's sync / barrier pattern)
            dist.barrier()
        scal.append(dict(s.items()))
    segs = 0 if tr._graph is None else sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph)
    torch.save({"G": tr.G.store.flat.cpu(), "D": tr.D.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "scal": scal,
                "segs": segs}, os.path.join(out_dir, f"g{int(use_graph)}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: this box has one GPU")
def test_two_ranks_over_rccl_match_single_process():
    """The same check as above on the backend the multi-GPU runs use: two ranks on two devices over RCCL ("nccl") must
    reproduce the single-process full-batch steps.  Skipped on the one-GPU boxes of the builder's pool; it runs wherever
    the test suite sees two or more devices (the driver's multi-GPU node)."""
    from tests.test_gpu_step import make_trainer
    torch.manual_seed(5)
    ref = make_trainer(ARCH, True, SHAPE, NZ, CB, CM, B)
    sdG = {k: v.detach().cpu().clone() for k, v in ref.G.state_dict().items()}
    sdD = {k: v.detach().cpu().clone() for k, v in ref.D.state_dict().items()}
    x, rand = make_rand()
    scal_ref = run_steps(ref, x, rand)
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(worker, args=(2, os.path.join(td, "init"), td, sdG, sdD, "nccl"), nprocs=2, join=True)
        outs = [torch.load(os.path.join(td, f"r{r}.pt")) for r in range(2)]
    for o in outs:
        assert rel_l2(o["D"], ref.D.store.flat.cpu()) < 5e-4
        assert rel_l2(o["G"], ref.G.store.flat.cpu()) < 5e-4
        assert rel_l2(o["E"], ref.G_ema.store.flat.cpu()) < 5e-4
        for s_got, s_ref in zip(o["scal"], scal_ref):
            for k, v in s_ref.items():
                assert abs(s_got[k] - v) < 1e-3 * max(1.0, abs(v)), k
    assert torch.equal(outs[0]["G"], outs[1]["G"]) and torch.equal(outs[0]["D"], outs[1]["D"])


def _det():
    """deterministic sums (engine.DETERMINISTIC): runs that execute the same additions agree on the logged scalars like they
    agree on the parameters - bit for bit, 1e-6 for the one float rounding a different launch form may add (round 6; the 3e-2 of
    rounds 3-5 was the noise allowance of float atomics)"""
    from dusty_gan_amd import engine as E
    return E.DETERMINISTIC


def rccl_graph_worker(rank, world, init_file, out_dir, mode):
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_GRAPH_COMM"] = "1" if mode == "in_graph" else "0"
    os.environ["DUSTY_GAN_GRAPH_DDP"] = "0" if mode == "eager" else "1"
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method=f"file://{init_file}", rank=rank, world_size=world,
                            device_id=torch.device("cuda", rank), timeout=PG_TIMEOUT)
    torch.manual_seed(300)   # one job seed: same initial nets; each rank's device RNG streams derive from it by rank
    tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 16, amp=True, gpu=rank)
    scal = []
    for i in range(6):
        s = tr.step(i)
        if i % 2 == 1:
            torch.cuda.synchronize()
            dist.barrier()
        scal.append(dict(s.items()))
    segs = 0 if tr._graph is None else sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph)
    torch.save({"G": tr.G.store.flat.cpu(), "D": tr.D.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "scal": scal,
                "segs": segs, "captured": tr._comm_captured, "mode": tr.launch_mode()}, os.path.join(out_dir, f"{mode}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: this box has one GPU")
@pytest.mark.parametrize("mode", ["in_graph", "segments"])
def test_two_ranks_over_rccl_graph_forms_match_eager(mode):
    """The two replayed forms of the multi-rank step on the backend and at the size the multi-GPU bench runs (RCCL, two
    devices, 64x1024, bf16, device RNG): collectives captured INSIDE the one hipGraph (the default) and hipGraph segments
    with the collectives between them must each train like the eager launch sequence, and leave both ranks with identical
    parameters.  Skipped on one-GPU boxes (every box this suite has run on so far): bench.py's supervisor is what keeps
    the first real run from being lost if the in-graph form misbehaves there."""
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for m in (mode, "eager"):
            mp.spawn(rccl_graph_worker, args=(2, os.path.join(td, "init_" + m), td, m), nprocs=2, join=True)
            res[m] = [torch.load(os.path.join(td, f"{m}_r{r}.pt")) for r in range(2)]
    a, e = res[mode], res["eager"]
    if mode == "in_graph":
        assert a[0]["captured"] == a[1]["captured"], "the ranks disagree about the capture"
        assert a[0]["segs"] == (1 if a[0]["captured"] else a[0]["segs"]), a[0]["mode"]
    else:
        assert a[0]["segs"] >= 4 and not a[0]["captured"]
    assert e[0]["segs"] == 0
    for k in ("G", "D", "E"):
        assert torch.equal(a[0][k], a[1][k]), k                    # the ranks stay replicas of each other
        assert rel_l2(a[0][k], e[0][k]) < 2e-3, (k, rel_l2(a[0][k], e[0][k]))
    for x, y in zip(a[0]["scal"], e[0]["scal"]):
        for k in x:
            assert abs(x[k] - y[k]) < 3e-2 * max(1.0, abs(y[k])), (k, x[k], y[k])


@pytest.mark.parametrize("full", [False, True], ids=["tiny-fp32", "64x1024-bf16-B8"])
def test_two_ranks_segmented_graph_matches_eager(full):
    """world > 1 (the default there): the step is replayed as hipGraph segments with the collectives (two D gradient
    buckets, Proj operand gather, two G gradient buckets; issued asynchronously, waited where their result is needed)
    called between them; the iterations must train like the eager launch sequence.  `full`: at the benchmark's network
    and image size with host-side synchronisation between replays (the configuration round 1 saw garbage in)."""
    res = {}
    for use_graph in (True, False):
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(graph_worker, args=(2, os.path.join(td, "init"), td, use_graph, full), nprocs=2, join=True)
            res[use_graph] = [torch.load(os.path.join(td, f"g{int(use_graph)}_r{r}.pt")) for r in range(2)]
    assert res[True][0]["segs"] >= 4 and res[False][0]["segs"] == 0  # collective points split the graph
    for r in range(2):
        a, b = res[True][r], res[False][r]
        for k in ("G", "D", "E"):
            # bf16 / full size: atomics reorder sums and Adam (beta1 = 0) turns a sign change of a rounding-noise gradient
            # into a 2 lr step: bounded per element, tiny over a tensor; garbage would be O(1)
            assert rel_l2(a[k], b[k]) < (2e-3 if full else 1e-5), (k, rel_l2(a[k], b[k]))
            if full:  # (step k of Adam can move an element by up to ~sqrt(k) lr while v-hat is young: 6 steps < 0.05)
                assert float((a[k] - b[k]).abs().max()) <= 0.05, k
        for x, y in zip(a["scal"], b["scal"]):
            for k in x:
                assert abs(x[k] - y[k]) < (3e-2 if full else 1e-4) * max(1.0, abs(y[k])), (k, x[k], y[k])
    assert torch.equal(res[True][0]["G"], res[True][1]["G"])


@pytest.mark.parametrize("x3", [False, True], ids=["bf16", "fp32x3"])
def test_forced_segments_single_process_match_one_graph(monkeypatch, x3):
    """DUSTY_GAN_FORCE_SEG=1: one process runs the multi-rank schedule (bucket order, chain-first G backward, operand
    gather, graph segments) with degenerate collectives - it must train like the single-graph replay at the benchmark's
    size, bf16; and in the fp32x3 mode with split-bf16 storage (B = 8), where the gathered Proj operand is a DG_BF16X2 buffer."""
    from tests.test_gpu_step import make_trainer
    monkeypatch.setenv("DUSTY_GAN_FP32_SPLIT", "1" if x3 else "0")

    def run(seg):
        monkeypatch.setenv("DUSTY_GAN_FORCE_SEG", "1" if seg else "0")
        torch.manual_seed(77)
        tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8 if x3 else 32, amp=not x3)
        assert tr.fp32_pairs == x3
        sc = [dict(tr.step(i).items()) for i in range(5)]
        segs = sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph)
        return tr, sc, segs
    a, sa, na = run(True)
    b, sb, nb = run(False)
    assert na >= 4 and nb == 1
    from dusty_gan_amd import engine as E
    for net in ("G", "D", "G_ema"):
        fa, fb = getattr(a, net).store.flat.cpu(), getattr(b, net).store.flat.cpu()
        if E.DETERMINISTIC:   # round 5: the multi-rank schedule issues the same sums in the same order - equal to the bit
            assert torch.equal(fa, fb), (net, rel_l2(fa, fb))
        else:
            assert rel_l2(fa, fb) < 2e-3, (net, rel_l2(fa, fb))
    for x, y in zip(sa, sb):
        for k in x:
            assert abs(x[k] - y[k]) <= (1e-6 if _det() else 3e-2) * max(1.0, abs(y[k])), (k, x[k], y[k])


def test_aborted_capture_leaves_no_pending_partials(monkeypatch):
    """A capture that fails while split-K partials are still pending (round-4 advice): the handler must forget them on the
    CAPTURE stream's workspace - the re-capture as segments runs on that stream, and a stale pending list would be reduced
    from never-written workspace memory into the gradients on every replay.  One process on the multi-rank schedule
    (DUSTY_GAN_FORCE_SEG=1) with the in-graph form forced on and the wait for the Proj-operand gather raising once during
    the capture; the run must then train like one whose capture never failed."""
    import warnings
    from dusty_gan_amd import engine as E
    from tests.test_gpu_step import make_trainer

    def run(fail):
        monkeypatch.setenv("DUSTY_GAN_FORCE_SEG", "1")
        torch.manual_seed(91)
        # (B = 32: every fat layer on the ping-pong conv, whose sums are order-independent - the two runs can be compared
        #  bit for bit; at B = 8 the small layers fall to kernels that still add bias gradients with float atomics)
        tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 32, amp=True)
        sc = [dict(tr.step(i).items()) for i in range(2)]       # the two eager warm-up steps
        seen = {"n": 0, "pending": 0}
        if fail:
            tr._comm_in_graph = True
            orig = tr._comm_wait

            def flaky(*keys):
                if tr._cap is not None and tr._comm_in_graph and "G.gather" in keys and seen["n"] == 0:
                    seen["n"] += 1
                    seen["pending"] = len(E.WGRAD_WS.items)
                    raise RuntimeError("injected capture failure")
                return orig(*keys)
            tr._comm_wait = flaky
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            sc += [dict(tr.step(i).items()) for i in range(2, 6)]
        if fail:
            assert seen["n"] == 1 and any("falling back" in str(w.message) for w in caught)
            assert not tr._comm_in_graph and tr._graph is not None
        torch.cuda.synchronize()
        for ws in E.WGRAD_WS._by_stream.values():
            assert not ws.items, "partials of the aborted capture are still pending"
        return tr, sc, seen
    a, sa, seen = run(True)
    b, sb, _ = run(False)
    for net in ("G", "D", "G_ema"):
        fa, fb = getattr(a, net).store.flat.cpu(), getattr(b, net).store.flat.cpu()
        assert torch.isfinite(fa).all()
        if E.DETERMINISTIC:
            assert torch.equal(fa, fb), (net, rel_l2(fa, fb), seen)
        else:
            assert rel_l2(fa, fb) < 2e-3, (net, rel_l2(fa, fb), seen)
    for x, y in zip(sa, sb):
        for k in x:
            assert abs(x[k] - y[k]) <= (1e-6 if _det() else 3e-2) * max(1.0, abs(y[k])), (k, x[k], y[k])


def rccl_worker(rank, world, init_file, out_dir, in_graph=True):
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_FORCE_SEG"] = "1"
    os.environ["DUSTY_GAN_GRAPH_COMM"] = "1" if in_graph else "0"
    dist.init_process_group("nccl", init_method=f"file://{init_file}", rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0), timeout=PG_TIMEOUT)
    from dusty_gan_amd.utils import dist as DD
    assert DD.through_backend()
    # the coalesced Proj-operand gather against two plain gathers (advisor, round 3): same bytes, and a handle whose wait
    # really orders the consumer behind RCCL's stream
    a = torch.randn(1 << 20, device="cuda").bfloat16()
    b = torch.randn(3 << 18, device="cuda")
    oa, ob = torch.empty_like(a), torch.empty_like(b)
    w = DD.all_gather_pair((oa, ob), (a, b))
    assert len(getattr(w, "works", [1])) > 0, "no work handle behind the coalesced gather"
    w.wait()
    pa, pb = torch.empty_like(a), torch.empty_like(b)
    DD.all_gather_into(pa, a)
    DD.all_gather_into(pb, b)
    torch.cuda.synchronize()
    assert torch.equal(oa, pa) and torch.equal(ob, pb)
    torch.manual_seed(77)
    tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 32, amp=True)
    assert tr._multi
    sc = []
    for i in range(5):
        sc.append(dict(tr.step(i).items()))
        if i % 2 == 1:
            torch.cuda.synchronize()  # host-side synchronisation between replays (bench.py's sync / barrier pattern)
            dist.barrier()
    segs = sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph)
    res = {"G": tr.G.store.flat.cpu(), "D": tr.D.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "scal": sc, "segs": segs,
           "mode": tr.launch_mode(), "captured": tr._comm_captured}
    # steady-state step time of this schedule (device events over 20 replays)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        tr.step(i)
    torch.cuda.synchronize()
    e0.record()
    for i in range(20):
        tr.step(i)
    e1.record()
    torch.cuda.synchronize()
    res["ms"] = e0.elapsed_time(e1) / 20
    res["prof"] = tr.comm_profile(2)
    torch.save(res, os.path.join(out_dir, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("in_graph", [True, False])
def test_single_rank_rccl_runs_the_multi_rank_schedule(monkeypatch, in_graph):
    """The one thing the gloo tests above cannot show on a one-GPU box: the multi-rank schedule on the "nccl" backend
    (RCCL).  A process group of ONE rank with DUSTY_GAN_FORCE_SEG=1 sends every exchange of the step through RCCL -
    asynchronous bucketed all-reduces on RCCL's stream, the Proj operand all-gathers (`all_gather_into_tensor` on byte
    views, coalesced), work handles, thread-local capture beside the RCCL watchdog - at the benchmark's size (bf16,
    B = 32).  in_graph: the collectives and their waits are captured INSIDE the step's one hipGraph (round 4; the default
    on nccl); otherwise they are host calls between hipGraph segments (round 3).  With one rank every collective is an
    identity, so either run must train like the single-graph replay."""
    from tests.test_gpu_step import make_trainer
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(rccl_worker, args=(1, os.path.join(td, "init"), td, in_graph), nprocs=1, join=True)
        a = torch.load(os.path.join(td, "rccl.pt"))
    monkeypatch.setenv("DUSTY_GAN_FORCE_SEG", "0")
    torch.manual_seed(77)
    b = make_trainer("none", True, (64, 1024), 512, 64, 512, 32, amp=True)
    sb = [dict(b.step(i).items()) for i in range(5)]
    if in_graph and a["captured"]:
        assert a["segs"] == 1, a["mode"]
    else:
        if in_graph:   # the runtime refused collectives inside the capture: the fallback must still be the segmented replay
            import warnings
            warnings.warn("collectives were not captured inside the graph on this box: " + a["mode"])
        assert a["segs"] >= 4
        assert any("wait" in k for k in a["prof"]), a["prof"]
    print(f"single-rank RCCL schedule, in_graph={in_graph}: {a['mode']}, {a['ms']:.3f} ms per step")
    for x, y in zip(a["scal"], sb):
        for k in x:
            assert abs(x[k] - y[k]) <= (1e-6 if _det() else 3e-2) * max(1.0, abs(y[k])), (k, x[k], y[k])
    for net, key in (("G", "G"), ("D", "D"), ("G_ema", "E")):
        fb = getattr(b, net).store.flat.cpu()
        from dusty_gan_amd import engine as E
        if E.DETERMINISTIC:   # (measured round 5: bit-identical in both forms)
            assert torch.equal(a[key], fb), (net, rel_l2(a[key], fb))
        else:
            assert rel_l2(a[key], fb) < 2e-3, (net, rel_l2(a[key], fb))


def fused_worker(rank, world, init_file, out_dir, fuse, pl=0.0):
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_GRAPH"] = "0"
    os.environ["DUSTY_GAN_FUSE_PROJ"] = "1" if fuse else "0"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=PG_TIMEOUT)
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
        # not bit-equal: atomics reorder fp32 sums from run to run, and a last-bit difference can flip a bf16 rounding of the
        # next step's shadows or one pixel of the hard Gumbel threshold.  Measured over repeated runs of this test: no
        # flip -> G / E ~1e-5, V ~1e-3; one flipped pixel -> G / E 5.8e-4 (the same value every time), V 0.6-1.9e-2.  A
        # wrong kernel would be O(1) off on Proj.weight, which is 96 % of these vectors.
        assert rel_l2(a["G"], b["G"]) < 2e-3, rel_l2(a["G"], b["G"])
        assert rel_l2(a["E"], b["E"]) < 2e-3, rel_l2(a["E"], b["E"])
        assert rel_l2(a["V"], b["V"]) < 5e-2, rel_l2(a["V"], b["V"])
        for x, y in zip(a["scal"], b["scal"]):
            for k in x:
                assert abs(x[k] - y[k]) < 2e-2 * max(1.0, abs(y[k])), (k, x[k], y[k])
    assert torch.equal(res[True][0]["G"], res[True][1]["G"])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (the reference's train.py:185-186
    mp.spawn) and prints ONE JSON line with n_gpus = 2; gloo lets both ranks share this box's one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DUSTY_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3",
                          "--batch", "4", "--shape", "64", "256", "--no-cpu-baseline"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distributed"]["world_size"] == 2 and out["distributed"]["backend"] == "gloo"
    assert out["config"]["global_batch"] == 8 and out["scaling"] == "weak"
    assert "segments" in out["launch_mode"]
    assert set(out["distributed"]["exposed_ms_per_step"]) >= {"wait D.hi+D.lo", "wait G.gather", "wait G.tail"}
    assert out["distributed"]["exchanges_per_step"] == 4 and out["distributed"]["bytes_per_step"]["total"] > 0
    assert out["step_ms_device"]["p50"] > 0
    assert out["roofline"]["model_self_check"] == "ok" and 0 < out["roofline"]["step"]["frac"] < 1


def x3_worker(rank, world, init_file, out_dir, pairs):
    """the fp32x3 mode on two gloo ranks sharing the GPU (same seed on both: identical nets), device RNG, 3 steps"""
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    from tests.test_gpu_step import make_trainer
    os.environ["DUSTY_GAN_FP32_SPLIT"] = "1"
    Trainer.fp32_pairs_default = pairs
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    torch.manual_seed(300)
    tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
    assert tr.fp32_pairs == pairs and tr.D.engine().x2 == pairs and tr._g_engines()[0].x2 == pairs
    scal = [dict(tr.step(i).items()) for i in range(3)]
    torch.cuda.synchronize()
    torch.save({"G": tr.G.store.flat.cpu(), "D": tr.D.store.flat.cpu(), "E": tr.G_ema.store.flat.cpu(), "scal": scal,
                "mode": tr.launch_mode()}, os.path.join(out_dir, f"x3_p{int(pairs)}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_fp32x3_split_storage_keep_identical_replicas():
    """The fp32x3 mode with split-bf16 storage under the data-parallel schedule on two real ranks (gloo, sharing the GPU): Proj's
    gradient operand is all-gathered as raw bytes of DG_BF16X2 - two ranks' whole 64-element groups side by side are again that
    form (checked on its own below) - and the gradient GEMM runs on the unpacked global batch.  The replicas must stay identical
    to the bit, and the run must train like the same two ranks with fp32 storage (the register-split kernels): the first step's
    scalars to 1e-4, three steps' parameters within 3e-3 (Adam at beta1 = 0 moves every element by ~lr per step whichever way its
    gradient's sign falls: a wrong gradient operand would move half of Proj.weight the other way in the first step already)."""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(9)
    parts = [torch.randn(3 * 64 * k, generator=g).cuda() for k in (5, 2)]
    raw = torch.cat([E.x2_pack(t) for t in parts])
    assert torch.equal(E.x2_unpack(E.tag_x2(raw)), torch.cat([E.x2_unpack(E.x2_pack(t)) for t in parts]))
    with tempfile.TemporaryDirectory() as td:
        for pairs in (True, False):
            mp.spawn(x3_worker, args=(2, os.path.join(td, f"init{int(pairs)}"), td, pairs), nprocs=2, join=True)
        a, b = (torch.load(os.path.join(td, f"x3_p1_r{r}.pt")) for r in range(2))
        ref = torch.load(os.path.join(td, "x3_p0_r0.pt"))
    for k in ("G", "D", "E"):
        assert torch.equal(a[k], b[k]), k
        assert bool(torch.isfinite(a[k]).all())
        assert rel_l2(a[k], ref[k]) < 3e-3, (k, rel_l2(a[k], ref[k]))
    for k in a["scal"][0]:
        assert abs(a["scal"][0][k] - ref["scal"][0][k]) < 1e-4 * max(1.0, abs(ref["scal"][0][k])), k
    for x, y in zip(a["scal"], ref["scal"]):
        for k in x:
            assert abs(x[k] - y[k]) < 2e-2 * max(1.0, abs(y[k])), (k, x[k], y[k])
