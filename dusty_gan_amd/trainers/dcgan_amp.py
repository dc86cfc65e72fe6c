"""Trainer for the DCGAN-eqlr / DUSty training hot path on MI355X -- reference: trainers/dcgan_amp.py.

Drop-in surface: Trainer(cfg, local_cfg) with .device, .start_iteration, .loader, .fetch_reals,
.A, .step(i) -> dict[str,float], .generate, .save_models; plus optimize_D()/optimize_G() which `step` is made of.

What differs underneath:
  * no autograd / DDP / GradScaler: one explicit schedule of HIP kernels per phase, fp32 master parameters in flat
    engine-layout buffers, bf16 (enable_amp) or fp32 activations, one RCCL all-reduce per network per step;
  * the discriminator is piecewise linear, so the R1 backward-data chain d(sum y_real)/dx is reused (scaled per
    sample by dLoss/dy_real) as the real batch's ordinary backward pass, and the R1 double backward is a
    forward-mode tangent pass through the saved leaky-relu masks (SURVEY.md §7 "hard parts");
  * the G-phase D(real) forward of the reference (:259) is dead code for non-relativistic losses and is not run;
  * the path-length regulariser (solver.loss.pl > 0) is a forward-over-reverse pass (Trainer._path_length);
  * scalars are gathered with ONE packed all-reduce and read back lazily.
"""
import math
import os
import os.path as osp
from collections import OrderedDict
from collections.abc import Mapping

import torch
import torch.distributed as dist

from .. import _lib as L
from .. import engine as E
from ..models import define_D, define_G
from ..models.loss import GANLoss
from ..utils import cycle, sigmoid_to_tanh, tanh_to_sigmoid  # noqa: F401  (re-exported like the reference)
from ..utils.context_manager import gradient_accumulation
from ..utils import dist as D_
from ..utils.diff_augment import DiffAugment
from ..utils.lidar import LiDAR
from ..utils.rng import Philox
from ..utils.synthetic import SyntheticLiDAR


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class LazyScalars(Mapping):
    """Read-only mapping str -> float whose values are read back from the device on first access (one D2H copy per
    step instead of the reference's 5-7 blocking `.item()` calls, trainers/dcgan_amp.py:319-323).  Every accessor
    (`[]`, get, items, keys, values, iteration, ==, dict(x)) resolves first; it is deliberately NOT a dict subclass:
    dict's C fast paths (json.dumps, dict.update, copy) would see an empty table before the read-back."""

    def __init__(self, keys, dev_tensor, finish=None):
        self._keys, self._dev, self._vals, self._finish = list(keys), dev_tensor, None, finish

    def _resolve(self):
        if self._vals is None:
            if self._finish is not None:    # the asynchronous rank average of a multi-GPU run completes here
                self._dev = self._finish()
                self._finish = None
            if isinstance(self._dev, RingSlot):
                self._vals = dict(zip(self._keys, self._dev.read()))
            else:
                self._vals = dict(zip(self._keys, self._dev.tolist()))
            self._dev = None
        return self._vals

    def __getitem__(self, k):
        return self._resolve()[k]

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __contains__(self, k):
        return k in self._keys

    def copy(self):
        return dict(self._resolve())

    as_dict = copy

    def __repr__(self):
        return repr(self._resolve())


class RingSlot:
    """One step's logged scalars in the trainer's ring of mapped pinned host memory: the step's last launch
    (dg_counter_add_multi_snap) stored them there, `event` was recorded behind it.  `read()` waits for that event - not
    for the launch stream, on which later steps may already be queued - and copies the slot out; the ring has
    Trainer.SCALAR_RING slots, so a slot is valid until that many later steps have run."""

    def __init__(self, ring, slot, event, idx, serial, trainer):
        self.ring, self.slot, self.event, self.idx, self.serial, self.trainer = ring, slot, event, idx, serial, trainer

    def read(self):
        self.event.synchronize()
        if self.trainer._snap_pos - self.serial > self.ring.shape[0]:
            raise RuntimeError(f"the scalars of step serial {self.serial} were overwritten: read a step's values within "
                               f"{self.ring.shape[0]} steps")
        row = self.ring[self.slot].tolist()
        return [row[i] for i in self.idx]


class FlatAdam:
    """Adam state on a flat ParamStore, exported/imported in torch.optim.Adam's state_dict format so checkpoints
    interchange with the reference (trainers/dcgan_amp.py:116-125, 138-141, 403-404)."""

    def __init__(self, net, lr, betas, eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, float(lr), (float(betas[0]), float(betas[1])), eps
        self.step_count = 0      # host mirror (checkpoints); the kernels read the device counter
        self._step_dev = None

    @property
    def store(self):
        return self.net.store

    def _param_views(self, buf):
        """views of `buf` (grad/m/v) shaped like the module's parameters, in named_parameters() order"""
        st = self.store
        pairs = self.net.backbone._bind_pairs() if hasattr(self.net, "backbone") else self.net._bind_pairs()
        by_param = {id(p): mk for p, mk in pairs}
        views = []
        saved = st.flat
        for _, p in self.net.named_parameters():
            st.flat = buf
            try:
                views.append(by_param[id(p)](st))
            finally:
                st.flat = saved
        return views

    def state_dict(self):
        st = self.store
        state = {}
        if self.step_count > 0:
            if self.betas[0] == 0.0 and getattr(self, "_last_gscale", None) is not None:
                if getattr(self, "regen_grad", None) is not None:
                    self.regen_grad()  # a gradient the fused optimizer kernel never materialised (Proj.weight)
                st.m.copy_(st.grad * self._last_gscale)  # exp_avg of torch.optim.Adam at beta1 = 0
            ms, vs = self._param_views(st.m), self._param_views(st.v)
            for i, (m, v) in enumerate(zip(ms, vs)):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().cpu().contiguous(),
                            "exp_avg_sq": v.detach().cpu().contiguous()}
        n = len(list(self.net.parameters()))
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(n))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        st = self.store
        st.ensure_train_state()
        ms, vs = self._param_views(st.m), self._param_views(st.v)
        steps = []
        with torch.no_grad():
            for i, s in sd["state"].items():
                ms[int(i)].copy_(s["exp_avg"])
                vs[int(i)].copy_(s["exp_avg_sq"])
                steps.append(int(float(s["step"])))
        self.step_count = max(steps) if steps else 0
        self._step_dev = None  # rebuilt from step_count on the next step
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])

    def zero_grad(self, set_to_none=True, skip=()):
        """zero the flat gradient buffer; `skip` names segments the backward pass OVERWRITES (Proj's 268 MB weight
        gradient is written with plain stores when there is a single micro-batch), which need no memset"""
        st = self.store
        if not skip:
            L.zero_(st.grad)
            return
        cuts = sorted((st.seg[k].off, st.seg[k].off + st.seg[k].numel) for k in skip)
        pos = 0
        for a, b in cuts:
            if a > pos:
                L.zero_(st.grad[pos:a])
            pos = b
        if pos < st.n:
            L.zero_(st.grad[pos:])

    # round 6: one launch sums the pending split-K partials / bias-gradient rows, applies Adam + EMA and writes every shadow
    # (dg_adam_fused) - DUSTY_GAN_FUSED_OPT=0 keeps the four-launch form (A/B, tests)
    fused_enabled = os.environ.get("DUSTY_GAN_FUSED_OPT", "1") != "0"

    def will_fuse(self, shadow_dtype):
        """whether `step` will take the one-launch optimizer (the caller may then leave gradient terms to it: `extra`)"""
        return (FlatAdam.fused_enabled and shadow_dtype == torch.bfloat16 and self.betas[0] == 0.0 and E.PROFILE is None
                and self.store.shadow is not None and self.store.shadow is not self.store.flat)

    def _fused_plan(self, st, off, extra, sole=False):
        """the DgOptSeg list of dg_adam_fused for flat[off:] from the store's segments and the split-K workspace's pending items
        (each must be THE one gradient term of a whole segment), or None (the caller then reduces first and runs the plain
        optimizer).  extra: {segment: (src_ptr, is_bf16, coef_ptr, n, stride, scale)} - a dg_batch_wsum term folded in."""
        ws = E.WGRAD_WS._cur()
        by_dw = {}
        for it in ws.items:
            if it[1] in by_dw:
                return None            # two terms for one destination (micro-batches, path-length terms): the reduce orders them
            by_dw[it[1]] = it
        segs, run = [], None

        def flat(a, b):
            q = L.DgOptSeg()
            q.off, q.numel, q.accumulate = a, b - a, 1
            return q
        for name, sg in st.seg.items():
            if sg.off < off:
                continue
            it = by_dw.pop(st.fptr(name, st.grad), None)
            fat = st.is_fat_conv(name) and name in st.coci
            ex = extra.get(name) if extra else None
            if it is None and not fat and ex is None:
                run = sg.off if run is None else run
                continue
            if run is not None:
                segs.append(flat(run, sg.off))
                run = None
            if sg.numel % 4 or (it is not None and it[2] != sg.numel):
                return None
            q = flat(sg.off, sg.off + sg.numel)
            if it is not None:
                q.part, q.splits, q.accumulate = it[0], it[3], it[4]
                if sole and fat:
                    q.accumulate = 0   # (the launch's partial tiles are the segment's ONLY gradient term: the zero-filled
                                       #  gradient buffer need not be read back - x + 0 is x, the bits do not change)
            if fat:
                q.kind, q.ci, q.co, q.shadow_t = 1, sg.shape[2], sg.shape[3], L.ptr(st.coci[name])
            if ex is not None:
                if fat:
                    return None
                q.ws_src, q.ws_bf16, q.ws_coef, q.ws_n, q.ws_stride, q.ws_scale = ex[0], int(ex[1]), ex[2], ex[3], ex[4], ex[5]
            segs.append(q)
        if run is not None:
            segs.append(flat(run, st.n))
        if by_dw or not segs or len(segs) > L.OPT_MAX_SEG:
            return None
        # the longest sums first (a thread of a 64-row piece makes eight dependent batches of loads, one of a 4-row piece one:
        # with the deep pieces' workgroups at the END of the grid the launch would finish on a handful of CUs - dg_wgrad_reduce's rule)
        segs.sort(key=lambda q: -q.splits)
        return (L.DgOptSeg * len(segs))(*segs), len(segs), ws

    def step(self, gscale=1.0, ema_store=None, ema_decay=0.0, shadow_dtype=torch.float32, fused_proj=None, between=None,
             extra=None, sole=False):
        """fused_proj = (dp0, zT, op_dtype, nb, Np, K, wscale): the first segment of the store is Proj.weight [Np][K] and
        its gradient is NOT in st.grad - the kernel forms wscale * dp0^T zT itself (dg_adam_proj_fused).  Returns
        False (and does nothing) if that kernel refuses the shape, so the caller can fall back.
        extra: gradient terms the caller left to the optimizer launch (`will_fuse`): {segment: dg_batch_wsum operands}.
        sole: the caller vouches that a fat conv segment's pending partial tiles are its only gradient term this step (one
        micro-batch, no path-length terms) and that the gradient buffer was zero-filled at the step's start."""
        st = self.store
        off = 0
        assert not (fused_proj is not None and extra), "extra terms are applied before the call can still fall back"
        if fused_proj is not None:
            if self.betas[0] != 0.0:
                return False
            off = fused_proj[4] * fused_proj[5]
        plan = self._fused_plan(st, off, extra, sole) if self.will_fuse(shadow_dtype) else None
        if plan is None:
            if extra:   # (the terms left to the fused launch, as launches of their own)
                for name, (src, is_bf16, coef, n, stride, scale) in extra.items():
                    assert stride == st.seg[name].numel
                    L.check(L.lib().dg_batch_wsum(src, L.DG_BF16 if is_bf16 else L.DG_F32, coef, scale, n, st.seg[name].numel,
                                                  st.fptr(name, st.grad), L.stream_ptr()), "dg_batch_wsum")
            E.WGRAD_WS.flush()  # (split-K partials still waiting to be summed into st.grad)
        if self._step_dev is None or self._step_dev.device != st.flat.device:
            self._step_dev = torch.full((1,), self.step_count, dtype=torch.int64, device=st.flat.device)
        # beta1 == 0 (the reference's solver): exp_avg == scaled gradient, so the kernel neither reads nor writes it
        m_ptr = None if self.betas[0] == 0.0 else L.ptr(st.m)
        lib = L.lib()
        L.Counters.flush_if(self._step_dev)  # (a queued advance from an optimizer step that no step end has flushed)
        sdt = L.dtype_code(shadow_dtype)
        ses = 2 if shadow_dtype == torch.bfloat16 else 4
        ema_ptr = L.ptr(ema_store.flat) if ema_store is not None else None
        if fused_proj is not None:
            dp0, zT, op_dt, nb, Np, K, wscale = fused_proj
            rc = lib.dg_adam_proj_fused(L.ptr(st.flat), L.ptr(st.v), ema_ptr,
                                        None if st.shadow is st.flat else L.ptr(st.shadow), sdt, L.ptr(dp0), L.ptr(zT),
                                        op_dt, nb, Np, K, wscale, gscale, self.lr, self.betas[1], self.eps,
                                        L.ptr(self._step_dev), ema_decay, L.stream_ptr())
            if rc == L.DG_EUNSUPPORTED:
                return False   # (nothing consumed: the caller's fall-back call plans again)
            L.check(rc, "dg_adam_proj_fused")
            if between is not None:
                between()   # (multi-GPU: the tail bucket's exchange has been travelling beside the kernel above)
        self.step_count += 1
        self._last_gscale = gscale
        if plan is not None:
            arr, nseg, ws = plan
            L.check(lib.dg_adam_fused(L.ptr(st.flat), L.ptr(st.grad), L.ptr(st.v), ema_ptr, L.ptr(st.shadow), sdt, arr, nseg,
                                      gscale, self.lr, self.betas[1], self.eps, L.ptr(self._step_dev), ema_decay,
                                      L.stream_ptr()), "dg_adam_fused")
            ws.items, ws.pos, ws.demand = [], 0, 0      # (the pending partials are summed: what `flush` would have left)
        else:
            # step count in device memory (bias corrections computed in the kernel) so the launch is graph-replayable
            L.check(lib.dg_adam_ema_step_dev(L.ptr(st.flat) + 4 * off, L.ptr(st.grad) + 4 * off,
                                             None if m_ptr is None else m_ptr + 4 * off, L.ptr(st.v) + 4 * off,
                                             None if ema_ptr is None else ema_ptr + 4 * off,
                                             None if st.shadow is st.flat else L.ptr(st.shadow) + ses * off,   # (fp32: the master IS the shadow)
                                             sdt, st.n - off, gscale, self.lr, self.betas[0],
                                             self.betas[1], self.eps, L.ptr(self._step_dev), ema_decay, L.stream_ptr()),
                    "dg_adam_ema_step_dev")
        L.Counters.add(self._step_dev, 1)  # (queued: one launch advances all of the step's counters)
        st.refresh_transposed(tail=True, small_only=plan is not None)
        if ema_store is not None:
            ema_store._seen_version = -1  # its shadows are rebuilt lazily when G_ema is used
        return True


def _store(net):
    return net.store


def _backbone(G):
    return G.backbone if hasattr(G, "backbone") else G


def _own_counters(fn):
    """run a Trainer entry point with the trainer's own queue of counter advances current (`_lib.Counters.bind`)"""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with L.Counters.bind(self.counters):
            return fn(self, *a, **kw)
    return wrapped


class Trainer:
    # fp32x3 mode: split-bf16 storage of the fat feature maps (set False for the register-splitting kernels of round 4: A/B
    # measurements, tests of that path)
    fp32_pairs_default = True

    def __init__(self, cfg, local_cfg, loader=None):
        self.counters = L.CounterQueue()   # this trainer's queued Philox / Adam / pool counter advances (never another trainer's)
        with L.Counters.bind(self.counters):
            self._init(cfg, local_cfg, loader)

    def _init(self, cfg, local_cfg, loader=None):
        self.cfg = cfg
        self.local_cfg = local_cfg
        gpu = local_cfg["gpu"] if isinstance(local_cfg, dict) else local_cfg.gpu
        self.device = torch.device("cuda", int(gpu)) if not isinstance(gpu, torch.device) else gpu
        if not torch.cuda.is_available():
            raise RuntimeError("dusty_gan_amd.Trainer needs an MI355X (no CPU path; see oracle/ for the CPU checker)")
        torch.cuda.set_device(self.device)
        L.lib()  # fail loudly now if the HIP library is missing
        self.local_batch = int(local_cfg["batch_size"] if isinstance(local_cfg, dict) else local_cfg.batch_size)

        # setup models (reference :45-51)
        self.cfg.model.gen.shape = self.cfg.dataset.shape
        self.cfg.model.dis.shape = self.cfg.dataset.shape
        self.G = define_G(self.cfg)
        self.D = define_D(self.cfg)
        self.G_ema = define_G(self.cfg)
        self.G_ema.eval()
        self.dtype = _backbone(self.G).compute_dtype
        # one job-level seed (rank 0's torch seed) from which every rank derives its Philox streams: rank r's position
        # in a run is then a function of rank 0's, which is what the checkpoint - written by rank 0 alone, like the
        # reference's (train.py:159-166) - records
        self.world = _world()
        self._job_seed = D_.broadcast_int(torch.initial_seed() & (2**62 - 1), self.device, src=0)
        self.A = DiffAugment(policy=list(self.cfg.solver.augment) if self.cfg.solver.augment is not None else None,
                             seed=self._job_seed + 101 + 7919 * _rank())   # (fixed at construction: the lazy default reads
                                                                            #  torch's GLOBAL seed at the first draw)
        H, W = self.cfg.dataset.shape
        self.H, self.W = int(H), int(W)
        self.lidar = LiDAR(num_ring=H, num_points=W, min_depth=cfg.dataset.min_depth, max_depth=cfg.dataset.max_depth,
                           angle_file=osp.join(cfg.dataset.root, "angles.pt") if cfg.dataset.get("root") else None)
        if str(cfg.dataset.name) == "synthetic":
            self.lidar.use_nominal_angles()
        self.lidar.to(self.device)

        self.G.to(self.device)
        self.D.to(self.device)
        self.G_ema.to(self.device)
        for net in (self.G, self.D):
            net.store.ensure_train_state()

        # DDP construction broadcasts rank 0's parameters (reference :68-69)
        if self.world > 1:
            D_.broadcast_params([self.G.store.flat, self.D.store.flat], src=0)
            self.G.store._seen_version = -1
            self.D.store._seen_version = -1
        self.G_ema.store.flat.copy_(self.G.store.flat)  # ema_inplace(G_ema, G, 0.0) (:51)
        self.G_ema.store._seen_version = -1

        self.ema_decay = 0.5 ** (self.cfg.solver.batch_size / (self.cfg.solver.smoothing_kimg * 1000))

        # data (reference :80-90): a device-resident synthetic pool for `dataset.name == synthetic`, the file datasets
        # through the device-side scan pipeline (datasets/scans.py), or any iterator of {"depth","mask"} batches
        if loader is not None:
            self.loader = loader
        elif str(self.cfg.dataset.name) == "synthetic":
            self.dataset = SyntheticLiDAR(self.local_batch, self.H, self.W, self.device, seed=1234 + _rank(),
                                          pool=int(self.cfg.dataset.get("pool", 4)),
                                          min_depth=cfg.dataset.min_depth, max_depth=cfg.dataset.max_depth)
            self.loader = cycle(self.dataset)
        else:
            from ..datasets import ScanLoader, define_dataset
            self.dataset = define_dataset(self.cfg.dataset, phase="train")  # NotImplementedError for unknown names
            self._scan_loader = ScanLoader(self.dataset, self.local_batch, self.device, world=_world(), rank=_rank(),
                                           num_workers=int(local_cfg["num_workers"] if isinstance(local_cfg, dict)
                                                           else local_cfg.num_workers))
            self.loader = cycle(self._scan_loader)

        # losses (reference :104-113)
        self.loss_weight = dict(self.cfg.solver.loss)
        self.criterion = {"gan": GANLoss(self.cfg.solver.gan_mode)}
        self.gan_code = self.criterion["gan"].code  # NotImplementedError for an unknown metric (models/loss.py:63)
        if self.loss_weight.get("gp", 0) > 0.0:
            self.criterion["gp"] = True
        if self.loss_weight.get("pl", 0) > 0.0:
            self.criterion["pl"] = True  # path-length regularisation (reference :109-111, :268-306)
        self.pl_ema = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._geng_pl = None

        # optimizers (reference :116-125)
        betas = (float(self.cfg.solver.lr.beta1), float(self.cfg.solver.lr.beta2))
        self.optim_G = FlatAdam(self.G, self.cfg.solver.lr.alpha.gen, betas)
        self.optim_D = FlatAdam(self.D, self.cfg.solver.lr.alpha.dis, betas)
        self.enable_amp = bool(cfg.enable_amp)
        # fp32x3 (DUSTY_GAN_FP32_SPLIT=1 with enable_amp: false): fp32 storage everywhere, the fat layers' contractions on the
        # bf16 matrix instructions with each operand split into bf16 hi + lo.  A property of THIS trainer's networks: their
        # engines pass DG_FORCE_FP32X3 with every launch (nothing process-wide; a second trainer of another precision in the
        # same process keeps its own kernels)
        self.fp32_split = (not self.enable_amp) and os.environ.get("DUSTY_GAN_FP32_SPLIT", "0") == "1"
        # ... and keep the fat layers' feature maps as split-bf16 PAIRS (DG_BF16X2: hi | lo halves per 64 channels, 4 bytes per
        # element like fp32), so that those layers run on the bf16 kernels of the timed path - the ping-pong conv and the LDS-DMA
        # weight gradient, three K steps per real one - instead of the register-splitting one-tile kernels.  (The data-parallel
        # schedule gathers Proj's gradient operand as raw bytes: whole 64-element groups per rank, so the gathered buffer is
        # the same form - `_like`.)
        self.fp32_pairs = self.fp32_split and Trainer.fp32_pairs_default
        for net in (_backbone(self.G), self.D, _backbone(self.G_ema)):
            net.fp32_split = self.fp32_split
            net.fp32_pairs = self.fp32_pairs

        # resume (reference :134-144)
        self.start_iteration = 0
        self.batches_drawn = 0
        resume_extra = None
        if self.cfg.resume is not None:
            sd = torch.load(self.cfg.resume, map_location="cpu", weights_only=True)  # tensors, ints, strs only
            resume_extra = sd.get("resume_state")
            self.start_iteration = sd["step"] // self.cfg.solver.batch_size
            self.G.load_state_dict(sd["G"])
            self.D.load_state_dict(sd["D"])
            self.G_ema.load_state_dict(sd["G_ema"])
            self.optim_G.load_state_dict(sd["optim_G"])
            self.optim_D.load_state_dict(sd["optim_D"])
            if "pl" in self.criterion and sd.get("pl_ema") is not None:
                self.pl_ema.copy_(torch.as_tensor(sd["pl_ema"]).reshape(1))

        self.n_acc = int(self.cfg.solver.num_accumulation)
        self.rng = Philox(self._job_seed + 7919 * _rank(), self.device, stream_id=1)
        self.fixed_noise = self.sample_latents(self.local_batch)
        if resume_extra is not None:
            self._restore_position(resume_extra)
        self._geng = None
        self._pending = None
        self._dev_scal = None
        self._pool_ctr = None   # device-resident loader position of the synthetic pool (dg_fetch_reals_pool_sum)
        self._fetch_in_prologue = os.environ.get("DUSTY_GAN_FETCH_IN_PROLOGUE", "1") != "0"   # (A/B switch, tests)
        self._pool_host = 0     # its host mirror: checked against batches_drawn before every graph step
        # the logged scalars leave the device with the step's last launch, into a ring of mapped pinned host memory
        # (RingSlot); DUSTY_GAN_SCALAR_RING=0: a device copy + a blocking read-back per step instead
        self._snap_ring, self._snap_ctr, self._snap_pos = None, None, 0
        # hipGraph replay of the step (single GPU, synthetic device-resident data); DUSTY_GAN_GRAPH=0 disables it
        self.use_graph = os.environ.get("DUSTY_GAN_GRAPH", "1") != "0"
        self._graph, self._eager_steps = None, 0
        self._cap, self._cap_cur, self._cap_pool, self._gather = None, None, None, None
        self._force_seg = os.environ.get("DUSTY_GAN_FORCE_SEG", "0") == "1"  # one process runs the multi-rank schedule
        self._multi = D_.through_backend()  # exchanges go through torch.distributed (utils/dist.py)
        if self._multi:
            D_.side_group()   # (collective: every rank's constructor) the gloo group the capture decision is taken over
        self._works, self._comm_events = {}, None
        self._fuse_proj_ok = os.environ.get("DUSTY_GAN_FUSE_PROJ", "1") != "0"
        self._fuse_proj_fp32 = os.environ.get("DUSTY_GAN_FUSE_PROJ_FP32", "1") != "0"   # (A/B of the round-6 fp32 forms)
        # Collectives INSIDE the captured step (nccl = RCCL only): the process group enqueues a captured collective on its
        # own stream behind an event of the capture stream, and work.wait() joins it back - fork / join edges of ONE hipGraph,
        # no host call and no stream hand-off between graph segments at replay.  "1": try it first and fall back to the
        # segmented replay if the runtime refuses the capture; "0": segments (round-3 behaviour; gloo always)
        self._comm_in_graph = (self._multi and dist.is_initialized() and dist.get_backend() == "nccl"
                               and os.environ.get("DUSTY_GAN_GRAPH_COMM", "1") != "0")
        self._comm_captured = False   # True once a step with captured collectives is being replayed

    # ------------------------------------------------------------------ helpers
    def sample_latents(self, B):
        """reference :151-152"""
        return self.rng.normal(B * self.cfg.model.gen.in_ch).view(B, self.cfg.model.gen.in_ch)

    def _next_batch(self):
        self.batches_drawn += 1
        return next(self.loader)

    def fetch_reals(self, raw_batch):
        """reference :154-160"""
        pol = raw_batch["depth"].to(self.device, non_blocking=True)
        mask = raw_batch["mask"].to(self.device, non_blocking=True).float()
        return self.lidar.fetch_reals(pol, mask, float(self.cfg.model.gen.drop_const))

    def _begin_step(self, draw_B=None, fetch=None):
        """Open the step's accumulator arena and zero both networks' gradient buffers (optim.zero_grad, reference :177 and
        :246) with ONE launch; `optimize_D` / `optimize_G` then skip their own fills (a single micro-batch overwrites
        Proj.weight's 268 MB gradient, or never forms it, so that segment is not filled).  draw_B: the same launch also
        makes the first micro-batch's draws - latents (+ their bfloat16 copy in the generator workspace), Gumbel noise,
        DiffAugment parameters: the numbers `_draw_rand` draws, from the same counters - kept for `_prep_rand`."""
        Gst, Dst = _backbone(self.G).store, self.D.store
        g = Gst.grad
        if self.n_acc == 1 and next(iter(Gst.seg)) == "proj_w":
            g = Gst.grad[Gst.seg["proj_b"].off:]
        ok = g.numel() % 4 == 0 and Dst.grad.numel() % 4 == 0 and g.data_ptr() % 16 == 0
        draws, adv = (), None
        if draw_B is not None:
            draws, adv = self._draw_jobs(int(draw_B))
        L.AccArena.begin(self.device, also=(Dst.grad, g) if ok else (), draws=draws, fetch=fetch)
        if adv is not None:
            adv()
        self._arena_ready = True
        self._grads_zeroed = {"D", "G"} if ok else set()

    def _draw_jobs(self, B):
        """the DgDraw jobs of `_draw_rand(B)` (same generators, same counter ranges, same order) and the function that
        advances the counters behind the launch; the results wait in self._predrawn"""
        dev, f32 = self.device, dict(dtype=torch.float32, device=self.device)
        nz = int(self.cfg.model.gen.in_ch)
        geng = self._g_engines()[0]
        geng.alloc(B, dev)
        r, ra = self.rng, self.A.rng(dev)
        r.sync()
        ra.sync()
        z = torch.empty(B, nz, **f32)
        z_ready = geng.dtype == torch.bfloat16
        base = 0
        jobs = [r.job(0, base, fill_kind=1, n=B * nz, out=L.ptr(z), out_bf16=L.ptr(geng.zT) if z_ready else None)]
        base += (B * nz + 3) // 4
        noise, arch = None, _backbone(self.G).masker
        if arch != "none":
            shapes = [("pixel", (B, 1, self.H, self.W))] + ([("image", (B, 1, 1, 1))] if arch == "dusty2" else [])
            noise = {}
            for k, shp in shapes:
                n = shp[0] * shp[1] * shp[2] * shp[3]
                noise[k] = torch.empty(shp, **f32)
                jobs.append(r.job(1, base, eps=1e-10, n=n, out=L.ptr(noise[k])))
                base += 2 * ((n + 3) // 4)
        n_sets = 4
        uf = torch.empty(3, n_sets * B, **f32)
        qi = torch.empty(4, n_sets * B, dtype=torch.int32, device=dev)
        jobs.append(ra.job(2, 0, B=n_sets * B, H=self.H, W=self.W, uf=L.ptr(uf), qi=L.ptr(qi)))
        self._predrawn = {"z": z, "noise": noise, "aug": self.A.sets_of(uf, qi, n_sets, B), "z_ready": z_ready, "B": B}

        def advance(total=base):
            r.advance(total)
            ra.advance(2 * n_sets * B)
        return jobs, advance

    def _pooled(self):
        """the loader is the device-resident synthetic pool and its batches can be picked by a device-side index"""
        ds = getattr(self, "dataset", None)
        return (isinstance(ds, SyntheticLiDAR) and getattr(ds, "pool_depth", None) is not None
                and (self.H * self.W) % 256 == 0)

    def _fetch_reals_in_step(self, raw_batch, pooled=False, begin=True):
        """fetch_reals as the first launch of a step: the step's accumulator arena is opened first, so the per-sample sums
        the kernel produces beside x_real survive until DiffAugment reads them.  pooled: `raw_batch` is the batch the
        synthetic loader just yielded (number batches_drawn - 1); the kernel picks that same batch ON THE DEVICE from the
        pool by a counter the step advances (dg_fetch_reals_pool_sum), so a hipGraph replay needs no copy of it."""
        if pooled:
            ds = self.dataset
            if self._pool_ctr is None:
                # (the batch this call fetches: the last one drawn, or - an accumulated step draws its num_accumulation batches
                #  up front - the first of those)
                first = self.batches_drawn - (getattr(self, "_drawn_ahead", 1) or 1)
                self._pool_ctr = torch.full((1,), first, dtype=torch.int64, device=self.device)
                self._pool_host = first
            L.Counters.flush_if(self._pool_ctr)
            x = None
            job = (self.lidar.fetch_job(ds.pool_depth, ds.pool_mask, self._pool_ctr, float(self.cfg.model.gen.drop_const))
                   if self._fetch_in_prologue else None)
            if job is not None:
                # round 6: the fetch rides on the step's first launch (arena + gradient zero-fill + the draws); its per-sample
                # sums leave as XSUM_PARTS partials per sample (nothing that launch would have to zero first).  A further
                # micro-batch of an accumulated step (begin=False): the same kernel as a launch of its own.
                if begin:
                    self._begin_step(draw_B=self.local_batch, fetch=job[0])
                else:
                    L.step_prologue([], [], fetch=job[0])
                x = L.tag_sums(job[1], job[2], parts=L.XSUM_PARTS)
            else:
                if begin:
                    self._begin_step(draw_B=self.local_batch)
                x = self.lidar.fetch_reals_pool(ds.pool_depth, ds.pool_mask, self._pool_ctr,
                                                float(self.cfg.model.gen.drop_const))
            L.Counters.add(self._pool_ctr, 1)   # (queued: applied with the step's other counters, inside the graph)
            if not torch.cuda.is_current_stream_capturing():
                self._pool_host += 1            # (host mirror of the device index; a capture executes nothing)
            # no mask on this path: the loader's host batch is NOT what a replay reads (the device picks the batch), and
            # nothing downstream uses the reals' mask (reference :154-160 returns it, :162-325 never reads it)
            return x, None
        if begin:   # (begin=False: a further micro-batch of an accumulated step - the arena is open, the draws follow per micro-batch)
            self._begin_step(draw_B=self.local_batch)   # (a step that fetches its own batch also draws its own parameters)
        return self.fetch_reals(raw_batch)

    def _g_engines(self):
        """one generator workspace per micro-batch: the D phase's G activations are reused by the G phase (:196,256)"""
        if self._geng is None:
            from ..engine import GEngine
            bb = _backbone(self.G)
            first = bb.engine()
            self._geng = [first] + [GEngine(first.cfg, self.dtype, x3=self.fp32_split, x2=self.fp32_pairs) for _ in range(self.n_acc - 1)]
        return self._geng

    def _sample_noise(self, B):
        arch = _backbone(self.G).masker
        if arch == "none":
            return None
        noise = {"pixel": self.rng.logistic_noise((B, 1, self.H, self.W))}
        if arch == "dusty2":
            noise["image"] = self.rng.logistic_noise((B, 1, 1, 1))
        return noise

    def _draw_rand(self, B):
        return {"z": self.sample_latents(B), "noise": self._sample_noise(B),
                "aug": self.A.draw_sets(4, B, self.H, self.W, self.device)}

    def _prep_rand(self, rand, B):
        if rand is None:
            pre, self._predrawn = getattr(self, "_predrawn", None), None
            if pre is not None and pre["B"] == B:
                return pre                # drawn by the step's first launch (`_begin_step`)
            assert pre is None, "pre-drawn parameters for another batch size"
            return self._draw_rand(B)
        self._predrawn = None             # (injected draws: whatever the step's first launch drew is not this step's)
        out = {"z": torch.as_tensor(rand["z"]).to(self.device, torch.float32)}
        nz = rand.get("noise")
        out["noise"] = None if not nz else {k: torch.as_tensor(v).to(self.device, torch.float32) for k, v in nz.items()}
        out["aug"] = [DiffAugment.params_to_device(rp, self.device) for rp in rand["aug"]]
        if "pl" in rand:  # injected draws of the path-length block (parity tests)
            pl = rand["pl"]
            f = lambda t: torch.as_tensor(t).to(self.device, torch.float32)
            out["pl"] = {"z": f(pl["z"]), "y": f(pl["y"]), "pl_ema": f(pl["pl_ema"]).reshape(1),
                         "noise": None if not pl.get("noise") else {k: f(v) for k, v in pl["noise"].items()}}
        return out

    def _coll(self, fn, name=None):
        """Run a collective (or any host-side call that cannot be captured) at this point of the launch sequence.
        Eager: call it.  While the step is being captured for replay: close the current hipGraph segment, remember
        `fn`, open the next segment - the replay alternates graph launches and these calls in the same order.
        `fn` must only touch tensors whose storage is the same on every step (flat buffers, preallocated outputs).
        With `self._comm_events` set (Trainer.comm_profile) HIP events on the launch stream bracket the call: their
        distance is the time the compute stream spent blocked on it, i.e. the EXPOSED communication time."""
        if name is not None:
            inner = fn

            def fn():
                if self._comm_events is None:
                    return inner()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                inner()
                e1.record()
                self._comm_events.append((name, e0, e1))
        if self._cap is None:
            fn()
            return
        if self._comm_in_graph:      # the collective / the wait become nodes of the graph being captured
            fn()
            return
        self._cap_close()
        self._cap.append(fn)
        self._cap_open()

    def _comm_issue(self, key, fn):
        """Start a collective WITHOUT making the compute stream wait for it (async_op=True: RCCL runs it on its own
        stream behind everything issued so far); `_comm_wait(key)` is where its result is first needed."""
        def issue():
            self._works[key] = fn()
        self._coll(issue, name=None)

    def _comm_wait(self, *keys):
        def wait():
            for k in keys:
                w = self._works.pop(k, None)
                if w is not None:
                    w.wait()  # nccl: the current stream waits (no host block); gloo: host wait
        self._coll(wait, name="wait " + "+".join(keys))

    def _cap_open(self):
        import gc
        # No cyclic garbage collection while a capture is open: a collection that starts in the middle of it may finalize objects
        # of EARLIER trainers - their hipGraphs, tensors of their private pools - and destroying a graph or freeing pool memory
        # from inside a capturing thread aborts the process (round 5: torch 2.10's graph context no longer collects on entry,
        # and the suite died in whichever test the allocation counters happened to trigger it).  Collect now, outside.
        # A caller that already runs with the collector off (bench.py, from in front of its warm-up) has taken it out of the
        # picture: nothing can start a collection inside the capture, and the ~70 ms of a full collection here would idle the
        # GPU long enough to send its clock down right in front of the first replays (round 6: the capture step took 76 ms of
        # host time, 66 of them in gc.collect; scripts/probes/capture_idle.py).
        if self._cap_cur is None and getattr(self, "_gc_was_on", None) is None:
            self._gc_was_on = gc.isenabled()
            if self._gc_was_on:
                gc.collect()
                gc.disable()
        # DUSTY_GAN_KEEP_GRAPH=1 (tests): keep the hipGraph_t behind the executable graph so that `graph_kernel_nodes()` can count
        # its nodes - the launch count of the replayed step is a tested property (tests/test_gpu_timed_path.py)
        g = torch.cuda.CUDAGraph(keep_graph=True) if os.environ.get("DUSTY_GAN_KEEP_GRAPH", "0") == "1" else torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of a multi-rank run may poll events while this thread captures
        ctx = torch.cuda.graph(g, pool=self._cap_pool, capture_error_mode="thread_local" if self._multi else "global")
        ctx.__enter__()
        self._cap_cur = (g, ctx)

    def _cap_close(self):
        import warnings
        g, ctx = self._cap_cur
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            ctx.__exit__(None, None, None)
        # two host-side calls in a row (issue of one bucket, wait for both) leave an empty segment between them: drop it
        if not any("is empty" in str(w.message) for w in caught):
            self._cap.append(g)
        self._cap_cur = None
        self._gc_restore()

    def _gc_restore(self):
        import gc
        if getattr(self, "_gc_was_on", None):
            gc.enable()
        self._gc_was_on = None

    def _bucketed(self):
        """multi-rank schedule (also with DUSTY_GAN_FORCE_SEG=1 in one process): bucketed, overlapped exchanges"""
        on = (self.world > 1 or self._force_seg) and self.n_acc == 1
        if on and not getattr(self, "_buckets_checked", False):
            # the buckets are cut by OFFSET in the flat gradient buffers and travel while later kernels still write other
            # offsets: that is only sound for this segment order (engine.g_segments / d_segments)
            dn, gn = list(self.D.store.seg), list(_backbone(self.G).store.seg)
            assert dn[dn.index("d4_w"):] == ["d4_w", "d4_b", "final_w", "final_b"], dn
            assert gn == ["proj_w", "proj_b", "up1_w", "up1_b", "up2_w", "up2_b", "up3_w", "up3_b", "head_w", "head_b"], gn
            self._buckets_checked = True
        return on

    def _allreduce(self, store):
        """SUM all-reduce of a network's flat gradient; returns the factor Adam applies (1/world = DDP's average)."""
        if self._multi:
            E.WGRAD_WS.flush()  # the split-K partials of the weight gradients become the gradient here
            self._coll(lambda: D_.allreduce_grads(store.grad), name="all-reduce grads")
        elif self._force_seg:
            E.WGRAD_WS.flush()
            self._coll(lambda: None)
        # (one rank: the optimizer's launch sums the pending partials itself - FlatAdam.step / dg_adam_fused, round 6 - or
        #  reduces them first where it cannot)
        return 1.0 / self.world

    def _allreduce_async(self, key, buf):
        """issue half of a bucketed gradient exchange (`_comm_wait(key)` completes it)"""
        E.WGRAD_WS.flush()
        if self._multi:
            self._comm_issue(key, lambda: D_.allreduce_grads(buf, async_op=True)[0])
        elif self._force_seg:
            self._coll(lambda: None)

    # ------------------------------------------------------------------ D phase (reference :171-238)
    @_own_counters
    def optimize_D(self, reals=None, rands=None):
        """One discriminator update over `num_accumulation` micro-batches.
        reals: optional list of (x_real, m_real) already through fetch_reals; rands: optional list of randomness
        bundles {"z", "noise", "aug":[4 parameter sets]} (parity tests); defaults draw from the loader / Philox."""
        lib, sp = L.lib(), L.stream_ptr()
        B = self.local_batch
        Gb, D = _backbone(self.G), self.D
        self.G.train()
        gengs = self._g_engines()
        deng = D.engine()
        deng.alloc(3 * B, self.device)
        Dst, Gst = D.store, Gb.store
        Dst.refresh_shadows(self.dtype)
        Gst.refresh_shadows(self.dtype)
        gp = float(self.loss_weight.get("gp", 0.0)) if "gp" in self.criterion else 0.0
        w_gan = float(self.loss_weight["gan"]) / self.n_acc
        self._mb = []
        extra_D = None
        dev = self.device
        # real, fake, adv, gp, G adv, path-length baseline, path-length penalty (sums over micro-batches)
        # every small accumulator of the step (these scalars, per-sample sums, logits) comes zeroed out of ONE arena
        if not getattr(self, "_arena_ready", False):   # (the graph path opens it before its fetch_reals)
            self._begin_step(draw_B=B if rands is None else None)
        self._arena_ready = False
        if "D" in self._grads_zeroed:
            self._grads_zeroed.discard("D")
        else:
            self.optim_D.zero_grad()
        scal = L.AccArena.take(16, dev)[:7]
        f32 = dict(dtype=torch.float32, device=dev)
        for j in gradient_accumulation(self.n_acc, True, (self.G, self.D)):
            if reals is not None:
                x_real, m_real = reals[j]
            else:
                x_real, m_real = self.fetch_reals(self._next_batch())
            rand = self._prep_rand(rands[j] if rands is not None else None, B)
            synth = gengs[j].forward(Gst, rand["z"], rand["noise"], training=True,   # :195 (graph kept = workspaces)
                                     z_ready=bool(rand.get("z_ready")) and j == 0)
            # :199-204  A(real) | A(fake) -> D, one pass; DiffAugment fused into BlurVH's pass where the sums exist
            y = deng.forward_aug(Dst, self.A, [(x_real, rand["aug"][0]), (synth["depth"], rand["aug"][1])], 0)
            # loss + dLoss/dy + the R1 schedule's per-sample vectors + scalar sums + final-bias gradient: one launch
            # chain upstream `up`: 1 for the real half (that IS d sum(y_real)/dx, :218-223), dLoss/dy for the fake half;
            # `rs`: the real half's ordinary backward is the same chain weighted per sample by dLoss/dy_real
            dy = torch.empty(2 * B, **f32)
            up = torch.empty(2 * B, **f32) if gp > 0 else None
            rs = None
            if gp > 0:
                # per-sample weights of the weight-gradient sums over the 3B input slots real | fake | tangent: the kernel
                # below writes [dLoss/dy_real | 1]; the tangent rows' 1s are written once, here
                if getattr(self, "_rs3", None) is None or self._rs3.numel() != 3 * B or self._rs3.device != self.device:
                    self._rs3 = torch.ones(3 * B, **f32)
                rs = self._rs3
            # the loss step, the final conv's backward-data and its weight gradient as one launch; three where the
            # kernel does not take the shape
            fused = deng.final_gan_bwd(Dst, 0, B, self.gan_code, False, float(self.criterion["gan"].smoothing), w_gan,
                                       L.ptr(y), L.ptr(y) + 4 * B, gp > 0, dy, up, rs, L.ptr(scal), True, True)
            if not fused:
                L.check(lib.dg_gan_d_step(self.gan_code, float(self.criterion["gan"].smoothing), L.ptr(y),
                                          L.ptr(y) + 4 * B, B, w_gan, L.ptr(dy), L.ptr(up), L.ptr(rs), L.ptr(scal),
                                          Dst.fptr("final_b", Dst.grad), sp), "dg_gan_d_step")
            # data-parallel runs (one micro-batch): the gradient exchange is cut into two buckets at d4_w (73 % of D's
            # bytes live in [d4_w, end)) and the weight gradients are formed last layer first, so the large bucket
            # travels while the three smaller layers' gradients are still being computed
            bucketed = self._bucketed()
            cut = Dst.seg["d4_w"].off
            if gp > 0:
                deng.backward_data(Dst, 0, 2 * B, up, rs, want_dbias=True, skip_final=fused)
                # R1 (:218-235): g = d sum(y_real) / dx_real, penalty = gp / 2 * mean_b |g_b|^2, and its double backward's
                # tangent v = d penalty / dg = (gp / B) g, pushed forward through D below
                vscale = gp / self.n_acc / B
                ssq = L.AccArena.take(B, dev)
                # (:229: the penalty's logged mean rides on the same launch)
                if ssq is None or not deng.r1_turnaround(Dst, 0, B, 2 * B, vscale, ssq, L.ptr(scal) + 12):
                    g = torch.empty(B, 1, self.H, self.W, **f32)
                    vg = torch.empty_like(g)
                    if ssq is None or not deng.backward_input(Dst, 0, B, vg, r1=(vscale, ssq)):
                        # (shapes the fused adjoint does not take - refused before anything is launched - or no arena:
                        # three passes)
                        deng.backward_input(Dst, 0, B, g)
                        if ssq is None:
                            ssq = torch.empty(B, **f32)
                            L.check(lib.dg_sample_sum(L.ptr(g), B, self.H * self.W, 1, L.ptr(ssq), sp), "dg_sample_sum")
                        else:
                            L.check(lib.dg_sample_sum_acc(L.ptr(g), B, self.H * self.W, 1, L.ptr(ssq), sp), "dg_sample_sum_acc")
                        L.check(lib.dg_scale(L.ptr(g), vscale, g.numel(), L.ptr(vg), sp), "dg_scale")
                    # (:229: the penalty's logged mean rides on the tangent's BlurVH launch)
                    deng.forward(Dst, vg, 2 * B, tangent_of=0, mean=(ssq, B, L.ptr(scal) + 12))
                # weight gradients: real (weighted by dLoss/dy_real) + fake halves and tangent (x) real chain, one launch
                # per fat layer (engine.DEngine.wgrad_r1)
                for layers in (((4,), (3, 2, 1)) if bucketed else ((4, 3, 2, 1),)):
                    deng.wgrad_r1(Dst, B, rs, layers=layers)
                    if 4 in layers:
                        if not fused:
                            deng.final_wgrad(Dst, 0, 2 * B, dy)
                        if (not bucketed and self.n_acc == 1 and not deng.x2 and self.optim_D.will_fuse(self.dtype)):
                            # (round 6: the R1 term of the final conv's weight gradient is summed by the optimizer's launch)
                            extra_D = {"final_w": deng.final_wgrad_term(2 * B, B)}
                        else:
                            deng.final_wgrad(Dst, 2 * B, B, None)
                        if bucketed:
                            self._allreduce_async("D.hi", Dst.grad[cut:])
            else:
                deng.backward_data(Dst, 0, 2 * B, dy, None, want_dbias=True, skip_final=fused)
                if not bucketed:
                    deng.wgrad(Dst, 0, 0, 2 * B, None)
                    if not fused:
                        deng.final_wgrad(Dst, 0, 2 * B, dy)
                else:
                    deng.wgrad(Dst, 0, 0, 2 * B, None, layers=(4,))
                    if not fused:
                        deng.final_wgrad(Dst, 0, 2 * B, dy)
                    self._allreduce_async("D.hi", Dst.grad[cut:])
                    deng.wgrad(Dst, 0, 0, 2 * B, None, layers=(3, 2, 1))
            self._mb.append({"x_real": x_real, "m_real": m_real, "rand": rand, "synth": synth, "geng": gengs[j]})
        if self._bucketed():
            self._allreduce_async("D.lo", Dst.grad[:Dst.seg["d4_w"].off])
            self._comm_wait("D.hi", "D.lo")
            gscale = 1.0 / self.world
        else:
            gscale = self._allreduce(Dst)
        self.optim_D.step(gscale=gscale, shadow_dtype=self.dtype, extra=extra_D, sole=self.n_acc == 1)  # :238
        self._dev_scal = scal
        return scal

    # ------------------------------------------------------------------ G phase (reference :240-316)
    @_own_counters
    def optimize_G(self):
        lib, sp = L.lib(), L.stream_ptr()
        B = self.local_batch
        Gb, D = _backbone(self.G), self.D
        Gst, Dst = Gb.store, D.store
        pl_on = "pl" in self.criterion
        if "G" in getattr(self, "_grads_zeroed", ()):
            self._grads_zeroed.discard("G")     # (filled with the step's arena: _begin_step)
        else:
            self.optim_G.zero_grad(skip=("proj_w",) if self.n_acc == 1 else ())
        gather_proj = (self.world > 1 or self._force_seg) and self.n_acc == 1 and next(iter(Gst.seg)) == "proj_w"
        fuse_proj = (not gather_proj and self.world == 1 and self.n_acc == 1 and self.dtype in (torch.bfloat16, torch.float32)
                     and (self.dtype == torch.bfloat16 or self._fuse_proj_fp32)
                     and next(iter(Gst.seg)) == "proj_w" and Gst.seg["proj_w"].off == 0 and self._fuse_proj_ok
                     and self.optim_G.betas[0] == 0.0 and E.PROFILE is None)
        fuse_gathered = (gather_proj and self.dtype == torch.bfloat16 and Gst.seg["proj_w"].off == 0
                         and self._fuse_proj_ok and self.optim_G.betas[0] == 0.0 and E.PROFILE is None)
        if not (fuse_proj or fuse_gathered):
            self.optim_G.regen_grad = None
        deng = D.engine()
        w_gan = float(self.loss_weight["gan"]) / self.n_acc
        f32 = dict(dtype=torch.float32, device=self.device)
        scal = self._dev_scal
        for j in gradient_accumulation(self.n_acc, True, (self.G, self.D)):
            mb = self._mb[j]
            rand = mb["rand"]
            # :255/:259 A(real) and D(real) feed only the relativistic losses (loss.py:76-85); the other metrics'
            # loss_G never reads pred_real (loss.py:66-75), so that pass is not run for them
            dy = torch.empty(B, **f32)
            if self.criterion["gan"].relativistic:
                # fake first: its slots 0..B carry the backward (:256, :255, :259-260, updated D)
                y = deng.forward_aug(Dst, self.A, [(mb["synth"]["depth"], rand["aug"][3]), (mb["x_real"], rand["aug"][2])], 0)
                y_real = L.ptr(y) + 4 * B
            else:
                y = deng.forward_aug(Dst, self.A, [(mb["synth"]["depth"], rand["aug"][3])], 0)  # :256, :260, updated D
                y_real = None
            fused = deng.final_gan_bwd(Dst, 0, B, self.gan_code, True, 1.0, w_gan, y_real, L.ptr(y), False, dy, None, None,
                                       L.ptr(scal) + 16, False, False)
            if not fused:
                L.check(lib.dg_gan_g_step(self.gan_code, y_real, L.ptr(y), B, w_gan, L.ptr(dy), L.ptr(scal) + 16, sp),
                        "dg_gan_g_step")
            deng.backward_data(Dst, 0, B, dy, None, want_dbias=False, skip_final=fused)
            ddepth = deng.backward_input_aug(Dst, 0, B, self.A, rand["aug"][3], lazy=True)
            overlap = gather_proj and not pl_on  # (the path-length block adds to every gradient after this pass)
            if overlap:
                geng = mb["geng"]
                zg, dg = self._gather_bufs(geng, B)
                # two exchanges: the operand gather starts behind the backward-data chain and travels beside every weight
                # gradient; the tail bucket (all of G's gradient but Proj.weight, 11 MB) starts behind the last weight
                # gradient and travels beside the fused Proj optimizer (0.3 ms), which only needs the gather - the rest of
                # the optimizer waits for it (FlatAdam.step `between`)
                geng.backward(
                    Gst, ddepth, skip_proj=True, chain_first=True,
                    after_chain=lambda: self._comm_issue("G.gather", lambda: D_.all_gather_pair((zg, dg), (geng.zT, geng.dp[0]))))
                self._allreduce_async("G.tail", Gst.grad[Gst.seg["proj_b"].off:])
            else:
                mb["geng"].backward(Gst, ddepth, accumulate_proj=(j > 0), skip_proj=gather_proj or fuse_proj)
            if pl_on:
                # (Proj.weight's two path-length terms follow its adversarial term: materialised here, or - when that
                # gradient is formed inside the optimizer / from gathered operands - appended to its operand list below)
                self._path_length(Gst, rand.get("pl"), B // 2, scal, proj_terms=not (gather_proj or fuse_proj))
        fused = None
        if gather_proj:
            # Proj.weight is 96 % of G's gradient bytes (268 MB fp32) and the last tensor backward produces.  It is a
            # plain linear layer, so instead of all-reducing its gradient every rank gathers the tiny operands
            # (z [B,nz] and dL/da0 [B,h0*w0*C], 8.4 MB per rank in bf16) and forms the GLOBAL-batch gradient locally:
            # same sum, 3.5x less xGMI traffic, and the 268 MB never cross a link.
            geng = self._mb[0]["geng"]
            if not pl_on:
                # started inside the backward pass (right after the data chain / the largest weight gradient)
                zg, dg = self._gather_bufs(geng, B)
                nloc = B
                self._comm_wait("G.gather")
            else:
                dp0, zT, nloc = self._proj_operands(geng, B)
                if self._gather is None or self._gather[0].numel() != self.world * zT.numel():
                    self._gather = (zT.new_empty(self.world * zT.numel()), self._like(dp0, self.world * dp0.numel()))
                zg, dg = self._gather
                self._coll(lambda: (D_.all_gather_into(zg, zT), D_.all_gather_into(dg, dp0)), name="all-gather Proj operands")
            nbg = self.world * nloc
            if fuse_gathered:
                # ... and the global-batch gradient is not even written: the optimizer forms it tile by tile in the
                # epilogue of the gradient GEMM (dg_adam_proj_fused -> MFMA path for nb > 64)
                c = geng.cfg
                Np = c.h0 * c.w0 * c.ch[3]
                fused = (dg, zg, L.dtype_code(self.dtype), nbg, Np, c.nz, 1.0 / math.sqrt(Np))
                self.optim_G.regen_grad = lambda: geng.proj_wgrad(Gst, dg, zg, nbg, False)
            else:
                geng.proj_wgrad(Gst, dg, zg, nbg)
            if not pl_on:
                if fused is None:
                    self._comm_wait("G.tail")   # (no fused Proj optimizer to hide it behind)
            else:
                tail = Gst.grad[Gst.seg["proj_b"].off:]
                E.WGRAD_WS.flush()
                self._coll(lambda: D_.allreduce_grads(tail), name="all-reduce G tail")
            gscale = 1.0 / self.world
        else:
            gscale = self._allreduce(Gst)
            if fuse_proj:
                geng = self._mb[0]["geng"]
                c = geng.cfg
                Np = c.h0 * c.w0 * c.ch[3]
                dp0, zT, nloc = self._proj_operands(geng, B)
                op_dt = L.dtype_code(self.dtype)
                if self.dtype == torch.float32:
                    # the parity-class modes (round 6): fp32 rows - split-bf16 pairs (fp32x3 with split storage) through an fp32
                    # copy, as the unfused GEMM takes them - on the fp32 matrix instructions or, fp32x3, as bf16 pairs in registers
                    if E.is_x2(dp0):
                        dp0 = E.x2_unpack(dp0, geng.ops._f32_copy("wa", dp0))
                    if E.is_x2(zT):
                        zT = E.x2_unpack(zT, geng.ops._f32_copy("wg", zT))
                    if geng.ops.x3:
                        op_dt |= L.DG_FORCE_FP32X3
                fused = (dp0, zT, op_dt, nloc, Np, c.nz, 1.0 / math.sqrt(Np))
                self.optim_G.regen_grad = lambda: geng.proj_wgrad(Gst, dp0, zT, nloc, False)
        # Adam + EMA fused (:312, :316); single-GPU bf16 runs also fold Proj.weight's gradient GEMM into the kernel
        tail_wait = (lambda: self._comm_wait("G.tail")) if (gather_proj and not pl_on and fused is not None) else None
        ok = self.optim_G.step(gscale=gscale, ema_store=_backbone(self.G_ema).store, ema_decay=self.ema_decay,
                               shadow_dtype=self.dtype, fused_proj=fused, between=tail_wait, sole=self.n_acc == 1 and not pl_on)
        if not ok:  # shape the fused kernel does not take: materialise the gradient and run the plain optimizer
            if tail_wait is not None:
                tail_wait()
            self.optim_G.regen_grad()
            self.optim_G.regen_grad = None
            self._fuse_proj_ok = False
            self.optim_G.step(gscale=gscale, ema_store=_backbone(self.G_ema).store, ema_decay=self.ema_decay,
                              shadow_dtype=self.dtype)
        self._mb = []
        return scal

    @staticmethod
    def _like(src, n):
        """an uninitialised buffer of n elements for rows of `src`, in src's storage form (split-bf16 pairs keep their tag)"""
        t = src.new_empty(n)
        return E.tag_x2(t) if E.is_x2(src) else t

    def _gather_bufs(self, geng, B):
        """static all-gather destinations for Proj's gradient operands (z rows, dL/da0 rows) of the global batch"""
        n = self.world * geng.zT.numel()
        if self._gather is None or self._gather[0].numel() != n:
            self._gather = (geng.zT.new_empty(n), self._like(geng.dp[0], self.world * geng.dp[0].numel()))
        return self._gather

    def launch_mode(self):
        """how `step` is being launched (bench.py reports it)"""
        if self._graph is None:
            return "eager launches"
        n = sum(isinstance(g, torch.cuda.CUDAGraph) for g in self._graph)
        if n == 1:
            return "one hipGraph per step" + (", collectives captured inside it" if self._comm_captured else "")
        return f"{n} hipGraph segments per step, collectives between them"

    def comm_bytes(self):
        """bytes each rank contributes to the step's exchanges (the multi-rank schedule at one micro-batch): D's gradient
        in two buckets, Proj's gradient operands (gathered instead of reducing the 268 MB gradient), the rest of G's
        gradient in one bucket, the packed scalars"""
        Gst, Dst = _backbone(self.G).store, self.D.store
        geng = self._g_engines()[0]
        es = 2 if self.dtype == torch.bfloat16 else 4
        cut = Dst.seg["d4_w"].off
        out = {"all-reduce D.hi": 4 * (Dst.n - cut), "all-reduce D.lo": 4 * cut,
               "all-gather G.gather (Proj operands, per rank)": es * (geng.zT.numel() + geng.dp[0].numel()) if geng.ws_B else 0,
               "all-reduce G.tail": 4 * (Gst.n - Gst.seg["proj_b"].off), "all-reduce scalars": 4 * 7}
        out["total"] = sum(out.values())
        out["not exchanged: Proj.weight gradient"] = 4 * Gst.seg["proj_w"].numel
        return out

    def comm_profile(self, steps=3):
        """Exposed communication time per step [ms] by collective: HIP events on the launch stream around every
        host-side collective / wait of `steps` further steps (after the timed region of bench.py)."""
        self._comm_events = []
        for i in range(steps):
            self.step(i)
        torch.cuda.synchronize()
        ev, self._comm_events = self._comm_events, None
        out = {}
        for name, e0, e1 in ev:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) / steps
        return {k: round(v, 4) for k, v in out.items()}

    def _proj_operands(self, geng, B):
        """(gradient rows, input rows, count) whose product is Proj.weight's gradient: the adversarial pair and, with
        the path-length term on, its two pairs (z_pl, tangent chain) and (v, first-order chain) - the three GEMMs share
        the output, so they are ONE GEMM over the concatenated rows.  Static buffers (graph replay, all-gather)."""
        if "pl" not in self.criterion:
            return geng.dp[0], geng.zT, B
        gp = self._geng_pl
        Bp = gp.ws_B
        n = B + 2 * Bp
        if getattr(self, "_cat", None) is None or self._cat[0].numel() != n * geng.dp[0].numel() // B:
            self._cat = (self._like(geng.dp[0], n * geng.dp[0].numel() // B), geng.zT.new_empty(n * geng.zT.numel() // B))
        d, z = self._cat
        nd, nzv = geng.dp[0].numel(), geng.zT.numel()
        d[:nd].copy_(geng.dp[0]); z[:nzv].copy_(geng.zT)
        d[nd:nd + gp.dp2[0].numel()].copy_(gp.dp2[0]); z[nzv:nzv + gp.zT.numel()].copy_(gp.zT)
        d[nd + gp.dp2[0].numel():].copy_(gp.dp[0]); z[nzv + gp.zT.numel():].copy_(gp.vT)
        return d, z, n

    def _path_length(self, Gst, inj, B_pl, scal, proj_terms=True):
        """Path-length regularisation (reference :268-306) for one micro-batch: accumulates
        d[w_pl * mean_b (|J_b^T y_b| - a)^2]/d theta into Gst.grad without autograd.  With v = d penalty / d(J^T y) held
        fixed, <v, J^T y> = <J v, y'(h) y>, so the parameter gradient is a forward-over-reverse pass: the reverse
        (data-only) chain gives J^T y, a forward tangent pass along v gives J v through the saved leaky-relu masks,
        the head's Hessian couples the two, and one more walk down the generator accumulates
        activations (x) tangent-chain + tangent-activations (x) first-order-chain (engine.GEngine._backward_chain)."""
        from ..engine import GEngine
        lib, sp = L.lib(), L.stream_ptr()
        if self._geng_pl is None:
            self._geng_pl = GEngine(self._g_engines()[0].cfg, self.dtype, x3=self.fp32_split, x2=self.fp32_pairs)
        geng = self._geng_pl
        nz = int(self.cfg.model.gen.in_ch)
        if inj is not None:
            z, noise, y = inj["z"], inj["noise"], inj["y"]
            self.pl_ema.copy_(inj["pl_ema"])
        else:
            z = self.sample_latents(B_pl)                       # :272-273
            noise = self._sample_noise(B_pl)                    # the Gumbel draws of G(z_pl), :277
            y = self.rng.normal(B_pl * self.H * self.W).view(B_pl, 1, self.H, self.W)   # randn_like(x_pl), :279
        geng.forward(Gst, z, noise, training=True)              # :277
        yv = torch.empty_like(y)
        L.check(lib.dg_scale(L.ptr(y), 1.0 / math.sqrt(self.H * self.W), y.numel(), L.ptr(yv), sp), "dg_scale")  # :280
        geng.backward(Gst, yv, data_only=True)                  # :282-287, down to Proj's pre-activation ...
        dz = geng.grad_z(Gst)                                   # ... and through Proj to z
        self._pl_dz = dz  # (kept for the parity tests: the reference's `grads`, :282-290)
        v = torch.empty_like(dz)
        w = float(self.loss_weight["pl"]) / self.n_acc
        L.check(lib.dg_pl_penalty(L.ptr(dz), B_pl, nz, w, L.ptr(self.pl_ema), L.ptr(v), L.ptr(scal) + 20, sp),
                "dg_pl_penalty")                                # :294-303
        geng.tangent_forward(Gst, v)
        geng.backward_second(Gst, yv, proj_terms)               # the `loss_G.backward()` share of the penalty, :309

    def _scalar_keys(self):
        keys = ["loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial"]
        idx = [0, 1, 2]
        if "gp" in self.criterion:
            keys.append("loss/D/gradient_penalty")
            idx.append(3)
        keys.append("loss/G/adversarial")
        idx.append(4)
        if "pl" in self.criterion:
            keys += ["loss/G/path_length/baseline", "loss/G/path_length"]
            idx += [5, 6]
        return keys, idx

    SCALAR_RING = 64

    def _use_ring(self):
        import os
        return self.world == 1 and self.n_acc == 1 and os.environ.get("DUSTY_GAN_SCALAR_RING", "1") != "0"

    def _ring_slot(self):
        """the slot the step that has just been launched files its scalars in (host mirror of the device-side serial)"""
        ev = torch.cuda.Event()
        ev.record()
        slot = RingSlot(self._snap_ring, self._snap_pos % self.SCALAR_RING, ev, self._scalar_keys()[1], self._snap_pos, self)
        self._snap_pos += 1
        return slot

    def _step_eager(self, reals=None, rands=None):
        """the launch sequence of one iteration; returns the (locally averaged) scalars: a device tensor, or - single
        process, one micro-batch - the RingSlot they are filed in by the step's last launch (None while capturing)"""
        scal = self.optimize_D(reals, rands)
        if self._use_ring():
            # queued now, applied by the step's LAST launch: the shadow refresh behind the generator's optimizer carries
            # every pending counter advance and the scalar snapshot (the scalars live in `scal`, the step's arena slice,
            # for both phases); whatever is still pending afterwards goes out with the flush below
            if self._snap_ring is None:
                self._snap_ring = torch.zeros(self.SCALAR_RING, 8, dtype=torch.float32).pin_memory()
                self._snap_ctr = torch.full((1,), self._snap_pos, dtype=torch.int64, device=self.device)
            L.Counters.add(self._snap_ctr, 1)
            L.Counters.snapshot(self._snap_ctr, scal.data_ptr(), 8, self._snap_ring.data_ptr(), self.SCALAR_RING)
            L.Counters.ride = True
            try:
                self.optimize_G()
            except BaseException:
                # (advisor, round 3) a generator phase that raises must not leave the ride armed: the NEXT step's first
                # shadow refresh would file a snapshot of half-finished scalars and advance the device serial past the host's
                L.Counters.ride, L.Counters.snap = False, None
                L.Counters.pending.pop(self._snap_ctr.data_ptr(), None)
                raise
            L.Counters.flush()  # (nothing left when the generator's refresh took them)
            return None if self._cap is not None else self._ring_slot()
        scal = self.optimize_G()
        L.Counters.flush()  # Philox offsets and Adam step counts of this step: one launch
        if self.n_acc > 1:
            scal = scal / self.n_acc
        elif self._cap is None:
            scal = scal.clone()  # (a copy: `scal` lives in the step's arena, re-zeroed when the next step begins)
        # (while capturing: the replay loop copies the arena slice after every replay - no copy node inside the graph)
        # (slices, not a Python index list: that would be a host-to-device copy, illegal during graph capture)
        end = 7 if "pl" in self.criterion else 5
        return scal[:end] if "gp" in self.criterion else torch.cat((scal[:3], scal[4:end]))

    def _graph_eligible(self, reals, rands):
        import os
        from .. import engine as E
        # world > 1: the step is replayed as hipGraph segments with the collectives issued between them (eager launches
        # cost more host time than the kernels take); DUSTY_GAN_GRAPH_DDP=0 falls back to eager launches
        if self.world > 1 and os.environ.get("DUSTY_GAN_GRAPH_DDP", "1") == "0":
            return False
        # the replay copies each batch into static device buffers, so the loader must yield fixed-shape device batches
        return (reals is None and rands is None and self.use_graph and E.PROFILE is None
                and getattr(getattr(self, "dataset", None), "graph_safe", False))

    def _step_graph(self):
        """hipGraph replay of the whole iteration (device-resident data): the ~180 kernel launches of a
        step cost ~5 ms of Python/ctypes time when issued one by one, more than the kernels themselves; captured once
        they replay from one host call.  Everything that changes between steps lives in device memory (Philox
        counters, Adam step counts, the input batch in a static buffer), so replays draw fresh randomness."""
        # an eager draw between two steps (validation() / generate() call sample_latents on the trainer's generator)
        # leaves its counter advance queued: apply it now, outside the graph - a replay reads the device counters as
        # they are, and an advance still pending when the capture starts would be baked into the graph and re-added
        # on every replay
        L.Counters.flush()
        pooled = self._pooled()
        batch = getattr(self, "_retry_batch", None)
        if batch is None:
            # num_accumulation micro-batches per step (utils/context_manager.py:21-35): the host loader moves on by as many
            # batches as the device-side pool index does inside the replay
            batches = [self._next_batch() for _ in range(self.n_acc)]
            self._drawn_ahead = self.n_acc
            batch = batches[0] if self.n_acc == 1 else batches
            if pooled and self._pool_ctr is not None and self._pool_host != self.batches_drawn - self.n_acc:
                # another consumer of the loader (an eager step with injected draws, a script calling _next_batch) has moved
                # the host position past the device-side pool index: the step fetches what the loader just yielded
                L.Counters.flush_if(self._pool_ctr)
                self._pool_host = self.batches_drawn - self.n_acc
                self._pool_ctr.fill_(self._pool_host)
        mbs = batch if isinstance(batch, list) else [batch]

        def fetch_all(srcs):   # every micro-batch's fetch_reals at the head of the step (the first one opens the arena)
            return [self._fetch_reals_in_step(b, pooled, begin=(j == 0)) for j, b in enumerate(srcs)]
        if self._graph is None:
            if self._eager_steps < 2:  # warm-up: workspaces, shadows and counters must exist before the capture
                self._eager_steps += 1
                return self._step_eager(reals=fetch_all(mbs))
            if not pooled:  # any other fixed-shape device loader: the replay reads static copies of the batch
                self._g_pol = batch["depth"].to(self.device).clone()
                self._g_mask = batch["mask"].to(self.device).clone()
            torch.cuda.synchronize()
            # One hipGraph per stretch between collectives (world > 1: gradient all-reduce of D, operand gather and
            # tail all-reduce of G); a single graph when world == 1.  The segments share one memory pool, so a
            # workspace allocated while capturing one segment stays valid in the next.
            self._cap, self._cap_pool = [], torch.cuda.graph_pool_handle()
            counts = (self.optim_D.step_count, self.optim_G.step_count)
            in_graph, err = self._comm_in_graph, None
            try:
                self._cap_open()
                src = mbs if pooled else [{"depth": self._g_pol, "mask": self._g_mask}]
                self._g_out = self._step_eager(reals=fetch_all(src))
                self._cap_close()
            except BaseException as exc:
                err = exc
                if self._cap_cur is not None:
                    # the split-K partials queued by the aborted capture were never written: forget them while the CAPTURE
                    # stream is still current (E.WGRAD_WS is per stream - after __exit__ the default stream's workspace would be
                    # the one reset, and the retry on this capture stream would reduce never-written memory into the gradients
                    # on every replay).  The workspace itself goes too: its buffer was allocated inside the aborted capture's
                    # memory pool, which dies with that capture - the retry allocates a fresh one in ITS pool.
                    E.WGRAD_WS.discard()
                    E.WGRAD_WS.drop_stream()
                    try:
                        self._cap_cur[1].__exit__(None, None, None)
                    except BaseException:
                        pass
                else:
                    E.WGRAD_WS.discard()
                self._cap_cur = None
                self._gc_restore()
                L.Counters.pending.clear()  # (advances queued by the aborted capture were never going to run)
                L.Counters.snap, L.Counters.ride = None, False
                self._works.clear()
                if not isinstance(exc, Exception):   # KeyboardInterrupt / SystemExit: not a refused capture
                    self._cap = None
                    raise
            # The ranks take ONE decision about collectives inside the graph: a rank that fell back to segments beside peers
            # replaying captured collectives would be the only one making host-side calls (and a mixed job has never run
            # anywhere).  Nothing has executed yet - the capture only recorded - so every rank reaches this exchange.
            everyone = D_.all_agree(err is None, self.device) if (in_graph and self.world > 1) else err is None
            what = D_.capture_decision(in_graph, err is None, everyone, self.world)
            if what != "keep":
                import warnings
                self._cap = None
                self._works.clear()
                self.optim_D.step_count, self.optim_G.step_count = counts   # nothing was executed: restore the host mirrors
                self._mb = []
                if what == "segments":
                    why = f"{type(err).__name__}: {err}" if err is not None else "a peer rank could not capture them"
                    warnings.warn(f"capturing the step with its collectives failed ({why}); falling back to hipGraph segments")
                    self._comm_in_graph = False
                    torch.cuda.synchronize()
                    return self._step_graph_retry(batch, pooled)
                if what == "raise":
                    raise err
                # multi-rank: a runtime that refuses the capture must not take the job down - keep launching eagerly
                warnings.warn("hipGraph capture of the training step failed; continuing with eager launches")
                self.use_graph = False
                return self._step_eager(reals=fetch_all(mbs))
            self._comm_captured = in_graph
            self._graph, self._cap = self._cap, None
            # the capture did not execute anything, but the host mirrors of the Adam step counts advanced once
            self.optim_D.step_count -= 1
            self.optim_G.step_count -= 1
        if not pooled:
            self._g_pol.copy_(batch["depth"], non_blocking=True)
            self._g_mask.copy_(batch["mask"], non_blocking=True)
        for item in self._graph:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            else:
                item()
        self.optim_D.step_count += 1
        self.optim_G.step_count += 1
        if pooled:
            self._pool_host += self.n_acc   # (the replay advanced the device-side pool index by its micro-batches)
        # the replayed Adam+EMA kernel rewrote G_ema's master through raw pointers: its low-precision / transposed
        # shadows (built lazily, only when G_ema is evaluated) are stale now
        _backbone(self.G_ema).store._seen_version = -1
        return self._ring_slot() if self._g_out is None else self._g_out.clone()

    def graph_kernel_nodes(self):
        """kernel nodes of the captured step, summed over its hipGraph segments - i.e. the launches one replayed step makes;
        None unless the step has been captured with DUSTY_GAN_KEEP_GRAPH=1 (hipGraphGetNodes / hipGraphNodeGetType on the
        kept hipGraph_t; HIP runtime through ctypes: host-side introspection, nothing on the compute path)"""
        if self._graph is None:
            return None
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        total = 0
        for item in self._graph:
            if not isinstance(item, torch.cuda.CUDAGraph):
                continue
            try:
                raw = item.raw_cuda_graph()
            except Exception:  # noqa: BLE001  (captured without keep_graph)
                return None
            n = C.c_size_t(0)
            if hip.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n)) != 0:
                return None
            nodes = (C.c_void_p * n.value)()
            if hip.hipGraphGetNodes(C.c_void_p(raw), nodes, C.byref(n)) != 0:
                return None
            for nd in nodes:
                t = C.c_int(-1)
                if hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t)) == 0 and t.value == 0:   # hipGraphNodeTypeKernel
                    total += 1
        return total

    def _step_graph_retry(self, batch, pooled):
        """second capture attempt of `_step_graph` (segments) on the batch the first one drew"""
        self._retry_batch = batch
        try:
            return self._step_graph()
        finally:
            self._retry_batch = None

    @_own_counters
    def step(self, i=0, reals=None, rands=None):
        """One training iteration (reference :162-325).  Returns dict[str,float] of globally averaged scalars."""
        if self._graph_eligible(reals, rands) and (self.n_acc == 1 or self._pooled()):
            out = self._step_graph()
        else:
            out = self._step_eager(reals, rands)
        if isinstance(out, RingSlot):
            return LazyScalars(self._scalar_keys()[0], out)
        out, finish = D_.mean_scalars(out)  # one packed, asynchronous collective instead of 5-7 blocking ones (:319-323)
        return LazyScalars(self._scalar_keys()[0], out, finish)

    # ------------------------------------------------------------------ inference / checkpoints
    def postprocess(self, synth):
        """reference :327-329 -> utils.postprocess (utils/__init__.py:163-178): [0,1] depth maps, sigmoid confidence and
        the point map on the sensor's angle grid (surface normals, a rendering aid, are not produced)"""
        from ..utils.lidar import postprocess
        return postprocess(synth, self.lidar)

    @_own_counters
    @torch.no_grad()
    def generate(self, ema=False):
        """reference :331-340"""
        net = self.G_ema if ema else self.G
        net.eval()
        synth = net(self.fixed_noise)
        return self.postprocess({k: v.clone() for k, v in synth.items()})

    def _val_batches(self):
        """the validation set as device batches {"depth","mask"} (reference :92-101: the 'val' split, drop_last=False)"""
        if isinstance(getattr(self, "dataset", None), SyntheticLiDAR):
            return list(self.dataset)  # the resident pool doubles as the validation set
        if getattr(self, "val_dataset", None) is None:
            from ..datasets import ScanLoader, define_dataset
            self.val_dataset = define_dataset(self.cfg.dataset, phase="val")
            nw = int(self.local_cfg["num_workers"] if isinstance(self.local_cfg, dict) else self.local_cfg.num_workers)
            self.val_loader = ScanLoader(self.val_dataset, self.local_batch, self.device, num_workers=nw,
                                         shuffle=True, drop_last=False)
        return self.val_loader

    @_own_counters
    @torch.no_grad()
    def validation(self, swd_rand=None, return_data=False):
        """reference :342-393: real (validation split) vs synthetic (G_ema) sets of equal size N ->
        SWD on the 2-D inverse-depth maps, JSD on the halved point clouds, COV / MMD / 1-NNA with the Chamfer distance
        on clouds downsampled to `solver.validation.num_points` by furthest point sampling.  Every stage runs on the
        GPU (csrc/lidar_io.hip, csrc/metrics.hip); returns dict[str,float] with the reference's keys."""
        from ..utils.metrics import compute_cov_mmd_1nna, compute_jsd, compute_swd
        from ..utils.sampling import downsample_point_clouds
        num_points = int(self.cfg.solver.validation.num_points)

        def inv_to_xyz(inv):
            xyz = self.lidar.inv_to_xyz(inv, 1e-8, from_tanh=True)  # tanh_to_sigmoid(.).clamp_(0,1) fused (:345)
            xyz = xyz.flatten(2).transpose(1, 2).contiguous()        # (B,N,3)
            return downsample_point_clouds(xyz, num_points)

        self.G_ema.eval()
        data = {"real-2d": [], "real-3d": [], "fake-2d": [], "fake-3d": []}
        for item in self._val_batches():
            x_real, _ = self.fetch_reals(item)
            data["real-2d"].append(x_real)
            data["real-3d"].append(inv_to_xyz(x_real))
        N = sum(t.shape[0] for t in data["real-2d"])
        for _ in range(0, N, self.local_batch):
            # (the generator returns views of its persistent workspace: keep a copy, not the view)
            x_fake = self.G_ema(latent=self.sample_latents(self.local_batch))["depth"].clone()
            data["fake-2d"].append(x_fake)
            data["fake-3d"].append(inv_to_xyz(x_fake))
        for key in data:
            data[key] = torch.cat(data[key], dim=0)[:N]
        scores = {}
        scores.update(compute_swd(data["fake-2d"], data["real-2d"], rand=swd_rand))
        scores["jsd"] = compute_jsd(data["fake-3d"] / 2.0, data["real-3d"] / 2.0)
        scores.update(compute_cov_mmd_1nna(data["fake-3d"], data["real-3d"], 512, ("cd",)))
        return (scores, data) if return_data else scores

    @_own_counters
    def state(self, step):
        def sd(m):
            return OrderedDict((k, v.detach().cpu().contiguous()) for k, v in m.state_dict().items())
        return {"step": step, "G": sd(self.G), "D": sd(self.D), "G_ema": sd(self.G_ema),
                "optim_G": self.optim_G.state_dict(), "optim_D": self.optim_D.state_dict(),
                "pl_ema": self.pl_ema.detach().cpu().reshape(()) if "pl" in self.criterion else None,
                # what the reference's checkpoint omits: without it a resumed run draws other latents,
                # augmentations and batches than the uninterrupted one.  An extra key: the reference's loaders ignore it.
                "resume_state": self._position()}

    def _position(self):
        """Philox streams (seed, stream id, counter) of the trainer and of DiffAugment, the fixed evaluation latents and
        the number of batches drawn from the loader, per rank"""
        def ph(r):
            return None if r is None else {"seed": r.seed, "stream_id": r.stream_id, "offset": r.offset}
        return {"rank": _rank(), "world": self.world, "rng": ph(self.rng), "augment_rng": ph(self.A._rng),
                "fixed_noise": self.fixed_noise.detach().cpu(), "batches_drawn": self.batches_drawn}

    def _restore_position(self, st):
        """Continue from a checkpoint's position record.  The record is the WRITING rank's (rank 0 saves, as in the
        reference); every rank's streams are `job seed + 7919 rank` and every rank makes the same number of draws per
        step, so rank r's position is the writer's with its own seed: same counter offsets, same loader position, and
        its fixed evaluation latents are the first draw of its own stream (what __init__ just made)."""
        from ..utils.rng import Philox
        if int(st.get("world", 1)) != self.world:
            import warnings
            warnings.warn("checkpoint position belongs to another world size; random streams and the loader restart")
            return
        shift = 7919 * (_rank() - int(st.get("rank", 0)))

        def mk(d):
            r = Philox(int(d["seed"]) + shift, self.device, stream_id=d["stream_id"])
            L.Counters.flush_if(r.ctr)
            r.ctr.fill_(int(d["offset"]))
            return r
        if shift == 0:
            self.fixed_noise = st["fixed_noise"].to(self.device)
        else:
            self.fixed_noise = Philox(int(st["rng"]["seed"]) + shift, self.device,
                                      stream_id=st["rng"]["stream_id"]).normal(self.fixed_noise.numel()).view_as(self.fixed_noise)
        self.rng = mk(st["rng"])
        if st.get("augment_rng") is not None:
            self.A._rng = mk(st["augment_rng"])
        # the loader: same epoch, same position inside it
        n = int(st["batches_drawn"])
        self.batches_drawn = n
        self._pool_ctr = None   # the device-side pool index is re-derived from batches_drawn at the next pooled fetch
        src = getattr(self, "dataset", None)
        from ..datasets.scans import ScanLoader
        inner = getattr(self, "_scan_loader", None)
        if isinstance(inner, ScanLoader):
            per = len(inner)
            inner.epoch, inner.skip = n // per, n % per
        elif isinstance(src, SyntheticLiDAR):
            for _ in range(n % len(src)):
                next(self.loader)

    @_own_counters
    def save_models(self, suffix, step, directory="models"):
        """reference :395-409 (same keys, reference-shaped tensors)"""
        import os
        os.makedirs(directory, exist_ok=True)
        path = osp.join(directory, "checkpoint_{}.pth".format(str(suffix)))
        torch.save(self.state(step), path)
        return path
