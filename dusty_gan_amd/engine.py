"""MI355X engine for the dusty-gan training hot path: parameter storage, workspaces and the explicit
forward / backward-data / weight-gradient schedule of one `Trainer.step` (reference: trainers/dcgan_amp.py:162-325).

No autograd: every pass is a kernel of libdustygan_hip.so launched on torch's current HIP stream.  torch owns the
device memory (flat parameter stores, workspaces) and nothing else.  See DESIGN.md for the schedule and layouts.
"""
import ctypes as C
import math
import os
from collections import OrderedDict

import torch

from . import _lib as L

ARCH_ID = {"none": 0, "dusty1": 1, "dusty2": 2}

# bench.py sets this to a list to time every conv / wgrad launch with HIP events (kernel family, algorithmic FLOPs,
# algorithmic bytes, start event, end event, shape tag).  None = no instrumentation (the normal path).
PROFILE = None
# tests / bench set this to a list to record what every conv / wgrad call launches: ("conv", DgConvPlan fields ..., tag)
# or ("wgrad", variant, tag).  None = off.
TRACE = None


def _align(n, a=64):
    return (n + a - 1) // a * a


class Segment:
    __slots__ = ("name", "off", "shape", "numel", "kind")

    def __init__(self, name, off, shape, kind):
        self.name, self.off, self.shape, self.kind = name, off, tuple(shape), kind
        n = 1
        for s in shape:
            n *= s
        self.numel = n


class ParamStore:
    """Flat fp32 parameter storage in ENGINE layout (+ grad / Adam state / low-precision shadow, same offsets).

    Conv weights are stored [ky][kx][ci][co]; the reference-shaped nn.Parameters are strided views of `flat`
    (see models/gans/dcgan_eqlr.py in this package), so state_dict() keeps the reference's keys and shapes.
    """

    def __init__(self, segments):
        self.seg = OrderedDict()
        off = 0
        for name, shape, kind in segments:
            s = Segment(name, off, shape, kind)
            self.seg[name] = s
            off = _align(off + s.numel)
        self.n = off
        self.flat = torch.zeros(self.n, dtype=torch.float32)
        self.grad = None
        self.m = None
        self.v = None
        self.shadow = None  # T copy of flat (same offsets)
        self.coci = {}  # name -> transposed conv shadow [16][co][ci] in T
        self._tdesc = None  # device descriptor table of refresh_transposed
        self.shadow_dtype = None
        self._seen_version = -1
        self.up_frags = {}  # (name, Hc, adj) -> [buffer, DgUpFrag]: weight fragments of the thin matrix-core MODE_UP kernel
        # DG_BF16X2 twins of the fat conv layers' fp32 shadows (fp32x3 mode with split storage: `enable_x2`), both forms:
        # name -> tensor for [tap][ci][co] (x2_cico) and [tap][co][ci] (x2_coci); rebuilt behind every shadow refresh
        self.x2 = False
        self.x2_cico, self.x2_coci = {}, {}

    # -- views
    def view(self, name, buf=None):
        s = self.seg[name]
        buf = self.flat if buf is None else buf
        return buf[s.off:s.off + s.numel].view(s.shape)

    def apply(self, fn):
        aliased = self.shadow is self.flat          # (fp32: the shadow is the master itself)
        self.flat = fn(self.flat)
        for k in ("grad", "m", "v", "shadow"):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, self.flat if (k == "shadow" and aliased) else fn(t))
        self.coci = {k: fn(v) for k, v in self.coci.items()}
        self._drop_x2()
        self._tdesc = None  # the descriptor table holds the old pointers
        self._seen_version = -1
        self.up_frags = {}  # (re-registered by the engines, on the new device)

    @property
    def device(self):
        return self.flat.device

    def ensure_train_state(self):
        if self.grad is None:
            self.grad = torch.zeros_like(self.flat)
            self.m = torch.zeros_like(self.flat)
            self.v = torch.zeros_like(self.flat)

    def up_frag(self, name, master_strides, N, Hc, adj):
        """Device pointer of the thin MODE_UP kernel's weight fragments for segment `name` (DgConv.up_frag), kept current
        by `refresh_transposed` - i.e. rebuilt by the launch that follows every optimizer step, instead of a preparation
        launch in front of each of the three convolutions of a step that use them.  master_strides = (tap, n, k) element
        strides of the weight in the fp32 master.  bf16 shadows only (None otherwise: the kernel prepares its own)."""
        if self.shadow_dtype != torch.bfloat16:
            return None
        key = (name, int(Hc), int(adj))
        e = self.up_frags.get(key)
        if e is None:
            if len(self.up_frags) >= 4:
                return None
            buf = torch.empty(L.UP_FRAG_BYTES, dtype=torch.uint8, device=self.flat.device)
            d = L.DgUpFrag()
            d.off, d.frag = self.seg[name].off, L.ptr(buf)
            d.m_st, d.m_sn, d.m_sk = master_strides
            d.N, d.Hc, d.adj = int(N), int(Hc), int(adj)
            e = self.up_frags[key] = [buf, d]
            self.refresh_transposed()  # (first use: build it now; from here on it follows the shadows)
        return L.ptr(e[0])

    def refresh_shadows(self, dtype, force=False):
        """(Re)build the T-typed copies the conv kernels read.  Cheap no-op when nothing changed."""
        ver = self.flat._version
        if not force and self.shadow is not None and self.shadow_dtype == dtype and ver == self._seen_version:
            return
        lib, st = L.lib(), L.stream_ptr()
        if self.shadow is None or self.shadow_dtype != dtype or self.shadow.device != self.flat.device:
            # fp32: the "[tap][ci][co] shadow in T" IS the master - no second 280 MB buffer, no copy here, and the optimizer's
            # shadow store lands on the line it has just written (round 5; rounds 1-4 kept and rewrote a copy every step)
            self.shadow = self.flat if dtype == torch.float32 else torch.empty(self.n, dtype=dtype, device=self.flat.device)
            self.coci = {}
            self._tdesc = None
            self.up_frags = {}
            self.shadow_dtype = dtype
        if self.shadow is not self.flat:
            L.check(lib.dg_cast(L.ptr(self.flat), L.ptr(self.shadow), L.dtype_code(dtype), self.n, st), "dg_cast")
        self.refresh_transposed()
        self._seen_version = ver

    # -- split-bf16 twins (fp32x3 mode with split storage)
    def enable_x2(self):
        """keep DG_BF16X2 copies of the fat conv layers' fp32 shadows (both layouts) and register them in X2_TWIN, so that
        Ops.conv finds the split weights behind the fp32 shadow pointer the engines pass"""
        if not self.x2:
            self.x2 = True
            self._seen_version = -1   # (the next refresh_shadows builds them)

    def __del__(self):   # (the twin registry is keyed by device pointers: a dead store must not leave its entries behind)
        try:
            self._drop_x2()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _drop_x2(self):
        for t in list(self.x2_cico.values()) + list(self.x2_coci.values()):
            for k in [k for k, v in X2_TWIN.items() if v == t.data_ptr()]:
                del X2_TWIN[k]
        self.x2_cico, self.x2_coci = {}, {}
        self._x2_key = None

    def _refresh_x2(self):
        if not self.x2 or self.shadow is None or self.shadow_dtype != torch.float32:
            return
        fat = [(name, s) for name, s in self.seg.items()
               if s.kind == "conv" and s.shape[2] % 64 == 0 and s.shape[3] % 64 == 0 and name in self.coci]
        if not fat:
            return
        key = (self.shadow.data_ptr(),) + tuple(self.coci[name].data_ptr() for name, _ in fat)
        if getattr(self, "_x2_key", None) != key:   # (first use, or the fp32 shadows moved: new twins, new registrations)
            self._drop_x2()
            self._x2_key = key
            for name, s in fat:
                self.x2_cico[name] = tag_x2(torch.empty(s.numel, dtype=torch.float32, device=self.flat.device))
                self.x2_coci[name] = tag_x2(torch.empty(s.numel, dtype=torch.float32, device=self.flat.device))
                X2_TWIN[self.sptr(name)] = self.x2_cico[name].data_ptr()
                X2_TWIN[self.coci[name].data_ptr()] = self.x2_coci[name].data_ptr()
            srcs, dsts, ns = [], [], []
            for name, s in fat:
                srcs += [self.sptr(name), self.coci[name].data_ptr()]
                dsts += [self.x2_cico[name].data_ptr(), self.x2_coci[name].data_ptr()]
                ns += [s.numel, s.numel]
            k = len(srcs)
            self._x2_args = ((C.c_void_p * k)(*srcs), (C.c_void_p * k)(*dsts), (C.c_long * k)(*ns), k)
        a = self._x2_args
        for i in range(0, a[3], 16):   # (16 buffers per launch; the benchmark's networks have 6 each)
            n = min(16, a[3] - i)
            L.check(L.lib().dg_cast_x2_multi(C.byref(a[0], i * C.sizeof(C.c_void_p)), C.byref(a[1], i * C.sizeof(C.c_void_p)),
                                             C.byref(a[2], i * C.sizeof(C.c_long)), n, L.stream_ptr()), "dg_cast_x2_multi")

    def is_fat_conv(self, name):
        """conv segments whose [tap][co][ci] shadow the fused optimizer writes itself (dg_adam_fused kind 1: 1024-element tiles)"""
        s = self.seg[name]
        return s.kind == "conv" and s.shape[2] % 16 == 0 and (s.shape[3] == 64 or s.shape[3] % 128 == 0)

    def refresh_transposed(self, tail=False, small_only=False):
        """[tap][ci][co] fp32 master -> [tap][co][ci] T shadow of every conv segment, one launch per network.  tail: the call
        behind an optimizer step - when the trainer has flagged it (`_lib.Counters.ride`) this is the step's last launch
        and carries the pending counter advances and the scalar snapshot.  (+ the split-bf16 twins, when enabled)
        small_only (behind dg_adam_fused, which has written the fat layers' transposed shadows itself): only the conv
        segments that kernel does not tile (Down1 / Head: a few KB), the thin kernels' weight fragments and the ride."""
        self._refresh_transposed(tail, small_only)
        self._refresh_x2()

    def _refresh_transposed(self, tail=False, small_only=False):
        lib, st = L.lib(), L.stream_ptr()
        convs = [(name, s) for name, s in self.seg.items() if s.kind == "conv"]
        if not convs:
            return
        if self._tdesc is None or self._tdesc_dtype != self.shadow_dtype or self._tdesc.device != self.flat.device:
            desc, tiles = [], 0
            small, stiles = [], 0
            for name, s in convs:
                _, _, ci, co = s.shape
                self.coci[name] = torch.empty(16 * ci * co, dtype=self.shadow_dtype, device=self.flat.device)
                desc += [s.off, L.ptr(self.coci[name]), ci, co, tiles]
                nt = 16 * ((ci + 31) // 32) * ((co + 31) // 32)
                tiles += nt
                if not self.is_fat_conv(name):
                    small += [s.off, L.ptr(self.coci[name]), ci, co, stiles]
                    stiles += nt
            self._tdesc = torch.tensor(desc, dtype=torch.int64).to(self.flat.device)
            self._tdesc_tiles, self._tdesc_dtype = tiles, self.shadow_dtype
            self._tdesc_small = torch.tensor(small, dtype=torch.int64).to(self.flat.device) if small else None
            self._tdesc_small_tiles, self._tdesc_small_n = stiles, len(small) // 5
        tdesc, ntiles, nconv = self._tdesc, self._tdesc_tiles, len(convs)
        if small_only:
            tdesc, ntiles, nconv = self._tdesc_small, self._tdesc_small_tiles, self._tdesc_small_n
        nf = len(self.up_frags) if self.shadow_dtype == torch.bfloat16 else 0
        arr = (L.DgUpFrag * max(nf, 1))(*[d for _, d in self.up_frags.values()][:nf])
        ride = L.Counters.take_for_ride() if tail else None
        if ride is not None:
            L.check(lib.dg_transpose_shadow_multi_tail(L.ptr(self.flat), L.ptr(tdesc), nconv, ntiles,
                                                       L.dtype_code(self.shadow_dtype), arr, nf, *ride, st),
                    "dg_transpose_shadow_multi_tail")
            return
        if nf:
            L.check(lib.dg_transpose_shadow_multi_frags(L.ptr(self.flat), L.ptr(tdesc), nconv, ntiles,
                                                        L.dtype_code(self.shadow_dtype), arr, nf, st),
                    "dg_transpose_shadow_multi_frags")
            return
        if nconv:
            L.check(lib.dg_transpose_shadow_multi(L.ptr(self.flat), L.ptr(tdesc), nconv, ntiles,
                                                  L.dtype_code(self.shadow_dtype), st), "dg_transpose_shadow_multi")

    def sptr(self, name):
        """device pointer of the T shadow of a segment"""
        return L.ptr(self.shadow) + self.shadow.element_size() * self.seg[name].off

    def fptr(self, name, buf=None):
        buf = self.flat if buf is None else buf
        return L.ptr(buf) + 4 * self.seg[name].off


def g_segments(nz, ch, shape, nheads):
    h0, w0 = shape[0] >> 4, shape[1] >> 4
    segs = [("proj_w", (h0, w0, ch[3], nz), "gemm"), ("proj_b", (ch[3],), "bias")]
    for i, (ci, co) in enumerate(((ch[3], ch[2]), (ch[2], ch[1]), (ch[1], ch[0])), start=1):
        segs += [(f"up{i}_w", (4, 4, ci, co), "conv"), (f"up{i}_b", (co,), "bias")]
    segs += [("head_w", (4, 4, ch[0], nheads), "conv"), ("head_b", (nheads,), "bias")]
    return segs


def d_segments(in_ch, ch, shape):
    h0, w0 = shape[0] >> 4, shape[1] >> 4
    segs = []
    ci = 2 * in_ch
    for i in range(4):
        segs += [(f"d{i + 1}_w", (4, 4, ci, ch[i]), "conv"), (f"d{i + 1}_b", (ch[i],), "bias")]
        ci = ch[i]
    segs += [("final_w", (h0, w0, ch[3]), "vec"), ("final_b", (1,), "bias")]
    return segs


class NetCfg:
    def __init__(self, shape, nz, ch, arch="none", ring=True, tau=1.0, drop_const=-1.0, dis_in_ch=1):
        self.H, self.W = int(shape[0]), int(shape[1])
        if self.H % 16 or self.W % 16 or self.H < 32:
            raise ValueError(f"shape {shape}: H and W must be multiples of 16 and H >= 32 (SURVEY.md §0.2)")
        self.nz, self.ch = int(nz), [int(c) for c in ch]
        self.arch, self.ring, self.tau, self.drop_const = arch, bool(ring), float(tau), float(drop_const)
        self.nheads = 1 + ARCH_ID[arch]
        self.h0, self.w0 = self.H >> 4, self.W >> 4
        if dis_in_ch != 1:
            raise NotImplementedError("discriminator in_ch != 1 (the range-image path is single channel)")


def conv_algorithmic(mode, B, Hc, Wc, K, N, ies, oes, wes, reads_aux, mask_bits=0):
    """(FLOPs, bytes) one conv-like launch needs at minimum: 2 MAC per tap, every operand moved once.
    MODE_S2: output on the coarse grid (Hc x Wc, N channels), input on the fine grid (2Hc x 2Wc, K channels), 16 taps per
    output pixel; MODE_UP: output on the fine grid, input on the coarse grid, 4 taps per output pixel (16 weight taps over
    the four sub-pixel parities); MODE_GEMM: B rows.  `reads_aux`: EPI_MASK also reads the saved activation at every
    output element - or, with `mask_bits` & 2 (DgConvPlan.mask_bits), one saved BIT per output element; `mask_bits` & 1:
    an EPI_LRELU launch also writes one bit per output element."""
    if mode == L.MODE_GEMM:
        pin, pout, taps_px, wtaps = B, B, 1, 1
    elif mode == L.MODE_S2:
        pin, pout, taps_px, wtaps = B * 4 * Hc * Wc, B * Hc * Wc, 16, 16
    else:
        pin, pout, taps_px, wtaps = B * Hc * Wc, B * 4 * Hc * Wc, 4, 16
    flops = 2.0 * pout * N * K * taps_px
    nbytes = pin * K * ies + pout * N * oes + wtaps * N * K * wes
    if reads_aux:
        nbytes += pout * N // 8 if (mask_bits & 2) else pout * N * oes
    elif mask_bits & 1:
        nbytes += pout * N // 8
    return flops, nbytes


def wgrad_algorithmic(wmode, B, Hc, Wc, Ci, Co, aes, ges):
    """(FLOPs, bytes) of one weight-gradient launch.  wmode 0 (Down): the layer input `a` lives on the fine grid, the
    gradient `g` on the coarse grid; wmode 1 (Up): `a` coarse, `g` fine; wmode 2: a plain [rows, Ci]^T [rows, Co] GEMM.
    16 taps per coarse pixel either way; dW (fp32) written once."""
    rows = B * Hc * Wc
    if wmode == 2:
        return 2.0 * rows * Ci * Co, rows * (Ci * aes + Co * ges) + Ci * Co * 4
    fa, fg = (4, 1) if wmode == 0 else (1, 4)
    return 2.0 * rows * Ci * Co * 16, rows * (fa * Ci * aes + fg * Co * ges) + 16 * Ci * Co * 4


class WgradWorkspace:
    """Split-K partial tiles of the step's weight-gradient launches (DgWgrad.ws): the MFMA LDS-DMA kernel stores every
    split's [16][Ci][Co] partial here with plain stores, and `flush` sums them into the gradients with ONE
    dg_wgrad_reduce launch per <= 16 layers - instead of fp32 atomics onto dW, which the memory side executes at ~1.3 TB/s
    (8x dW's bytes per launch: VERDICT r02).  A bump allocator over one caller-owned buffer; the reduce is deferred to
    the point where the gradient is first needed (before an exchange / the optimizer), so one launch serves a whole
    network.  Fixed summation order: the gradients are bit-reproducible."""
    FLOATS = 96 << 20   # cap, 384 MB: D's three fat layers at 3B rows + G's three (80 MB each) with room for the path-length terms
    START = 4 << 20     # first allocation (16 MB): tiny nets never need more; the buffer doubles up to the cap on demand

    def __init__(self, start=None):
        self.buf, self.pos, self.items = None, 0, []
        # what the tests of the large-batch plans read: the most floats ever pending, reduces forced by a full buffer,
        # requests larger than the cap (those launches fall back to fp32 atomics onto dW)
        self.hwm, self.early_flushes, self.refused = 0, 0, 0
        # first allocation: another stream's workspace on this device has already found out what a step needs (the step's graph
        # is captured on a stream of its own: starting small there baked the growth's extra reduce launches into every replay)
        self.start = max(self.START, min(int(start or 0), self.FLOATS))
        self.demand = 0     # floats requested since the caller's last flush, growth's own reduces not counted: what to grow to
        self.grows = 0      # allocations (1 = the first one sufficed)
        self.pre_reduce = None   # set by an open Ops.grouped() block: issues its queued launches before any reduce

    def take(self, nfloats, device):
        if nfloats > self.FLOATS:
            self.refused += 1
            return None
        if self.buf is not None and (self.buf.device != device or self.buf.numel() > self.FLOATS):
            self._reduce_pending()
            self.buf = None
        need = self.pos + nfloats
        self.demand += (nfloats + 63) // 64 * 64
        if self.buf is None or need > self.buf.numel():
            size = self.start if self.buf is None else self.buf.numel()
            while size < min(max(need, self.demand), self.FLOATS):
                size *= 2
            size = min(size, self.FLOATS)
            if self.buf is None or size > self.buf.numel():
                # grow: the pending partials are summed first (stream order keeps the old buffer alive until that reduce has
                # run: the caching allocator hands its memory out again only to later work of this stream)
                self._reduce_pending()
                self.buf = torch.empty(size, dtype=torch.float32, device=device)
                self.grows += 1
            if self.pos + nfloats > self.buf.numel():   # at the cap and still full: reduce early, start over
                self.early_flushes += 1
                self._reduce_pending()    # (stream order: the reduce has read the partials before the next launch overwrites them)
        off = self.pos
        self.pos = (off + nfloats + 63) // 64 * 64
        self.hwm = max(self.hwm, self.pos)
        return L.ptr(self.buf) + 4 * off

    def add(self, ws_ptr, dw_ptr, numel, splits, accumulate):
        self.items.append((ws_ptr, dw_ptr, numel, splits, accumulate))

    def flush(self):
        """sum every pending layer's partials into its gradient (stream order after the launches that wrote them)"""
        self.demand = 0
        self._reduce_pending()

    def discard(self):
        """forget the pending partials WITHOUT summing them: the launches that were to write them never ran (an aborted
        stream capture) - a later flush would add never-written workspace memory onto the gradients"""
        self.items, self.pos, self.demand = [], 0, 0

    def _reduce_pending(self):
        if self.items and self.pre_reduce is not None:
            # launches whose partials are about to be summed may still be waiting in an open `Ops.grouped()` block (they were
            # queued, not issued): issue them first - the reduce follows them in stream order
            self.pre_reduce()
        items, self.items, self.pos = self.items, [], 0
        # one launch per <= 16 layers, and never two items with the same destination in one launch (micro-batches, the
        # path-length terms: their blocks would read-modify-write the same dW concurrently) - those follow in stream order
        chunks, cur, seen = [], [], set()
        for it in items:
            if len(cur) == 16 or it[1] in seen:
                chunks.append(cur)
                cur, seen = [], set()
            cur.append(it)
            seen.add(it[1])
        if cur:
            chunks.append(cur)
        for chunk in chunks:
            arr = (L.DgWgradReduce * len(chunk))()
            for a, (ws, dw, numel, splits, acc) in zip(arr, chunk):
                a.ws, a.dw, a.numel, a.splits, a.accumulate = ws, dw, numel, splits, acc
            if PROFILE is None:
                L.check(L.lib().dg_wgrad_reduce(arr, len(chunk), L.stream_ptr()), "dg_wgrad_reduce")
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(L.lib().dg_wgrad_reduce(arr, len(chunk), L.stream_ptr()), "dg_wgrad_reduce")
            e1.record()
            nbytes = sum(4 * numel * (splits + 1 + acc) for _, _, numel, splits, acc in chunk)
            PROFILE.append(("wgrad_reduce_kernel", 0.0, nbytes, e0, e1, f"{len(chunk)} layers"))


class _PerStreamWorkspace:
    """`engine.WGRAD_WS`: one WgradWorkspace per (device, HIP stream).  Launches of one stream are ordered, so engines that
    share a stream may share the bump allocator (a reduce always precedes the reuse of its partials); engines on DIFFERENT
    streams - a second trainer, a validation model on a side stream - get buffers of their own instead of interleaving
    allocations in one (round-3 review: a process-global singleton)."""

    def __init__(self):
        self._by_stream = {}

    def _cur(self):
        key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream) if torch.cuda.is_available() else (-1, 0)
        ws = self._by_stream.get(key)
        if ws is None:
            known = [w.buf.numel() for (dev, _), w in self._by_stream.items() if dev == key[0] and w.buf is not None]
            ws = self._by_stream[key] = WgradWorkspace(start=max(known, default=0))
        return ws

    def __getattr__(self, name):          # take / add / flush / items / pos / hwm ... of the current stream's workspace
        return getattr(self._cur(), name)

    def pending_elsewhere(self):
        """layers whose partials wait in ANOTHER stream's workspace of the current device: `grads_ready()` on this stream
        does not sum them (a reader of `st.grad` on the wrong stream would see incomplete gradients)"""
        if not torch.cuda.is_available():
            return 0
        key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
        return sum(len(w.items) for k, w in self._by_stream.items() if k[0] == key[0] and k != key)

    def drop_stream(self, stream=None):
        """release the workspace of `stream` (default: the current one) - called when the graph / trainer that owned the
        stream goes away, so that a recycled stream handle does not inherit its buffer or its pending list"""
        if not torch.cuda.is_available():
            return
        h = (stream or torch.cuda.current_stream()).cuda_stream
        self._by_stream.pop((torch.cuda.current_device(), h), None)

    def __setattr__(self, name, value):
        if name == "_by_stream":
            object.__setattr__(self, name, value)
        else:
            setattr(self._cur(), name, value)


WGRAD_WS = _PerStreamWorkspace()


def grads_ready():
    """The engines' backward passes leave the split-K partial tiles of the fat layers' weight gradients in `WGRAD_WS`
    (`defer=True`): `st.grad` is complete only after this call (what `FlatAdam.step` and the trainer's all-reduce helpers
    do first).  Anything else that reads a ParamStore's gradient after GEngine.backward / DEngine.wgrad calls it.
    Only the CURRENT stream's workspace is summed (a reduce must follow its launches in stream order): partials still pending
    on another stream of this device mean that stream's gradients are incomplete - said out loud (round-4 advice)."""
    other = WGRAD_WS.pending_elsewhere()
    if other:
        import warnings
        warnings.warn(f"engine.grads_ready(): {other} layer(s) have split-K partials pending on ANOTHER stream of this device; "
                      "call grads_ready() on the stream that ran their backward pass before reading those gradients")
    WGRAD_WS.flush()


class MaskBits:
    """Saved leaky-relu masks at 1 bit per element (DgConv.mask_out / mask_in; the reference's autograd keeps the sign of
    every FusedLeakyReLU pre-activation, models/ops/common.py:99-106).  An engine registers one uint8 buffer of numel / 8
    bytes per activation buffer; `Ops.conv` then hands its slice to every EPI_LRELU launch that WRITES the activation
    (mask_out) and to every EPI_MASK launch whose `aux` it is (mask_in), so the backward / tangent passes read 1/16 of the
    bytes.  bf16 feature maps with >= 16 channels only; everything else keeps reading `aux`."""
    enabled = os.environ.get("DUSTY_GAN_MASK_BITS", "1") != "0"   # (A/B switch, exercised by tests/test_gpu_ops.py)

    @classmethod
    def register(cls, act):
        """give the activation buffer `act` a bit buffer (kept as an attribute of the tensor object the engine passes around)"""
        if not cls.enabled or act.dtype != torch.bfloat16 or act.numel() % 128:
            return None
        act._dg_bits = torch.zeros(act.numel() // 8, dtype=torch.uint8, device=act.device)
        return act._dg_bits

    @staticmethod
    def slice_ptr(act, off, N, strides):
        """device pointer of the bits of `act` from element `off`, or None (no bits for this tensor / geometry)"""
        bits = getattr(act, "_dg_bits", None)
        if bits is None or N % 16 or strides[0] % 16 or strides[1] % 16 or strides[2] != 1 or off % 16:
            return None
        return L.ptr(bits) + off // 8


# Bit-reproducible gradients (round 5; DUSTY_GAN_DETERMINISTIC=0 restores the float atomics): bias-gradient sums leave the
# matrix-core conv kernels and the fused final-conv backward as per-workgroup partial ROWS in the split-K workspace and are
# summed by the reduce launch that already runs (fixed order), and the cross-block sums into the step's accumulator arena are
# fixed-point integer adds (_lib.AccArena registers the arena's shadow: dg_det_arena).  Covers the kernels of the bf16 timed
# path; the direct / VALU fall-back kernels of tiny or odd shapes keep their float atomics.
DETERMINISTIC = os.environ.get("DUSTY_GAN_DETERMINISTIC", "1") != "0"


# ---- split-bf16 pairs (DG_BF16X2, round 5): the storage form of the fp32x3 mode's fat feature maps and weight shadows.  A
# tensor keeps torch dtype float32 (4 bytes per element, same element strides) and carries the attribute `_dg_x2 = True`: its
# bytes are (hi | lo) bf16 halves per 64 channels (include/dusty_gan_hip.h).  Ops.conv / Ops.wgrad pass the dtype code by the
# tag; kernels that do not take the form get an fp32 copy (Proj's GEMMs, the final conv's pointwise kernels: small tensors).
X2_TWIN = {}   # device pointer of an fp32 weight shadow -> device pointer of its DG_BF16X2 copy (ParamStore.refresh_x2)


def is_x2(t):
    return t is not None and getattr(t, "_dg_x2", False)


def tag_x2(t):
    """mark a float32 tensor as holding DG_BF16X2 data (256-byte aligned, whole 64-channel groups: the allocator's job)"""
    assert t.dtype == torch.float32 and t.data_ptr() % 256 == 0 and t.numel() % 64 == 0
    t._dg_x2 = True
    return t


def x2_pack(src, dst=None, off=0, n=None):
    """fp32 `src` (n elements from `off`) -> DG_BF16X2 at the same offsets of `dst` (a tagged tensor; default: a new one)"""
    n = src.numel() - off if n is None else n
    if dst is None:
        dst = tag_x2(torch.empty(src.numel(), dtype=torch.float32, device=src.device))
    L.check(L.lib().dg_cast(L.ptr(src) + 4 * off, L.ptr(dst) + 4 * off, L.DG_BF16X2, n, L.stream_ptr()), "dg_cast")
    return dst


def x2_unpack(src, dst=None, off=0, n=None):
    """DG_BF16X2 `src` -> fp32 `dst` (same offsets)"""
    n = src.numel() - off if n is None else n
    if dst is None:
        dst = torch.empty(src.numel(), dtype=torch.float32, device=src.device)
    L.check(L.lib().dg_uncast(L.ptr(src) + 4 * off, L.DG_BF16X2, L.ptr(dst) + 4 * off, n, L.stream_ptr()), "dg_uncast")
    return dst


class _WgradGroup:
    def __init__(self, ops):
        self.ops = ops

    def __enter__(self):
        self.outer = self.ops._group is not None    # (a block inside a block joins the outer one's launch)
        if not self.outer:
            self.ops._group = []
            self.ws = WGRAD_WS._cur()
            self.ws.pre_reduce = self._issue_now
        return self

    def _issue_now(self):
        """the workspace is about to sum pending partials (it grows, or it is full): the queued launches must run first"""
        items, self.ops._group = self.ops._group, []
        self.ops._launch_group(items)

    def __exit__(self, et, ev, tb):
        if self.outer:
            return False
        items, self.ops._group = self.ops._group, None
        self.ws.pre_reduce = None
        if et is None:
            self.ops._launch_group(items)
        return False


class Ops:
    """Thin typed wrappers over the C ABI (struct filling); all launches go to torch's current stream."""
    group_enabled = os.environ.get("DUSTY_GAN_WGRAD_GROUP", "1") != "0"
    # workgroups a group launch aims at, in residency rounds of 512 (0: every layer keeps the 512 of a launch of its own;
    # 1 / 2 / 3 rounds measured 3.7 / 1.5 / 0.9 % slower on the step, so this is a test / experiment knob, not a switch)
    group_rounds = 0
    default_wg_cap = 0  # parity tests lower it so that small problems walk several tiles per persistent workgroup
    _dbias_ws = {}      # device -> DgConv.dbias_ws scratch (launches of one stream share it: each leaves it zero)

    def __init__(self, dtype, x3=False):
        self.lib = L.lib()
        self.dtype = dtype
        # fp32x3: DG_F32 operands on the bf16 matrix instructions, split into bf16 hi + lo (DG_FORCE_FP32X3 on every call of
        # THIS Ops: per engine, nothing process-wide)
        self.x3 = bool(x3) and dtype == torch.float32
        self.dt = L.dtype_code(dtype)
        self.es = 2 if dtype == torch.bfloat16 else 4
        # dg_conv / dg_wgrad `force` (0 auto; the parity tests set 1 direct, 2 MFMA, 3 thin, 4 / 5 persistent kernels)
        self.force = 0
        self.wg_cap = Ops.default_wg_cap  # dg_conv_ex: cap on the persistent conv's workgroup count (0 = one residency wave)
        self.use_ws = True  # split-K partials through WGRAD_WS + dg_wgrad_reduce (False: fp32 atomics onto dW)
        self._group = None  # the open `grouped()` block's launches
        self._x2_scratch = {}  # (role, numel) -> fp32 copy of a DG_BF16X2 operand for kernels that do not take the form

    def _f32_copy(self, role, t):
        """fp32 scratch of the same size as the tagged tensor `t` (one per role and size, reused launch after launch)"""
        key = (role, t.numel(), str(t.device))
        b = self._x2_scratch.get(key)
        if b is None:
            b = self._x2_scratch[key] = torch.empty(t.numel(), dtype=torch.float32, device=t.device)
        return b

    @property
    def _f(self):
        """the `force` argument of the C ABI: kernel-family code | flag bits"""
        return self.force | (L.DG_FORCE_FP32X3 if self.x3 else 0)

    def conv(self, mode, adj, ring, B, Hc, Wc, K, N, x, x_strides, out, out_strides, w_ptr, scale, epi,
             bias=None, bias_mod=0, aux=None, dbias=None, rowscale=None, in_dt=None, out_dt=None, nscale=None,
             x_off=0, out_off=0, aux_off=0, w_strides=None, w_dt=None, up_frag=None, defer_db=False, tanh_sums=None):
        """tanh_sums (a float32 buffer of >= B * 256 elements): ask the launch for DgConv.tanh_sum_parts - tanh on the output and
        the per-workgroup sums of it; returns DgConvPlan.sum_parts (> 0: done, the partial sums per sample; 0: the kernel that
        takes this launch does not do it, nothing of it happened).
        defer_db: the bias-gradient rows of a deterministic launch (DETERMINISTIC, kernels with DgConvPlan.dbias_rows) wait in
        WGRAD_WS for the caller's flush like a deferred weight gradient's partials; otherwise they are summed at once."""
        x2_out = None
        if mode == L.MODE_GEMM and (is_x2(x) or is_x2(out) or is_x2(aux)):
            # Proj's GEMMs run on the fp32 kernels: split-bf16 operands through fp32 copies (4 M elements at the benchmark)
            if is_x2(x):
                x = x2_unpack(x, self._f32_copy("x", x), x_off, B * K)
            if is_x2(aux):
                aux = x2_unpack(aux, self._f32_copy("aux", aux), aux_off, B * N)
            if is_x2(out):
                x2_out, out = out, self._f32_copy("out", out)
        if is_x2(x):
            in_dt = L.DG_BF16X2
        if is_x2(out):
            out_dt = L.DG_BF16X2
            assert aux is None or is_x2(aux), "DG_EPI_MASK: aux has out's layout and dtype"
        if in_dt == L.DG_BF16X2 and out_dt == L.DG_BF16X2 and w_strides is None:
            # both sides split-bf16: the ping-pong conv with the weight shadow's split twin
            tw = X2_TWIN.get(int(w_ptr))
            if tw is None:
                raise L.DgError("no DG_BF16X2 twin registered for this weight shadow (ParamStore.enable_x2)")
            w_ptr, w_dt = tw, L.DG_BF16X2
        p = L.DgConv()
        p.mode, p.adj, p.ring = mode, adj, int(ring)
        p.B, p.Hc, p.Wc, p.K, p.N = B, Hc, Wc, K, N
        in_dt = self.dt if in_dt is None else in_dt
        out_dt = self.dt if out_dt is None else out_dt
        ies = 2 if in_dt == L.DG_BF16 else 4
        oes = 2 if out_dt == L.DG_BF16 else 4
        p.in_ = L.ptr(x) + ies * x_off
        p.in_sb, p.in_sp, p.in_sk = x_strides
        p.out = L.ptr(out) + oes * out_off
        p.out_sb, p.out_sp, p.out_sn = out_strides
        p.w = w_ptr
        if w_strides is None:
            p.w_st, p.w_sn, p.w_sk = N * K, K, 1  # shadow in T, [tap][n][k]
        else:
            p.w_st, p.w_sk, p.w_sn = w_strides  # e.g. the fp32 master [tap][ci][co] read in place
        p.scale, p.epi = scale, epi
        p.bias, p.bias_mod = bias, bias_mod
        p.aux = None if aux is None else L.ptr(aux) + oes * aux_off
        p.dbias, p.rowscale = dbias, L.ptr(rowscale)
        p.in_dtype, p.out_dtype, p.w_dtype = in_dt, out_dt, (self.dt if w_dt is None else w_dt)
        p.nscale = L.ptr(nscale)
        p.up_frag = up_frag
        sum_parts = 0
        if tanh_sums is not None:
            p.tanh_sum_parts = L.ptr(tanh_sums)
            pl = L.DgConvPlan()
            L.check(self.lib.dg_conv_plan(C.byref(p), self._f, self.wg_cap, C.byref(pl)), "dg_conv_plan")
            sum_parts = pl.sum_parts if B * pl.sum_parts <= tanh_sums.numel() else 0
            if not sum_parts:
                p.tanh_sum_parts = None
        if epi == L.EPI_LRELU and out_dt == L.DG_BF16:
            p.mask_out = MaskBits.slice_ptr(out, out_off, N, out_strides)
        elif epi == L.EPI_MASK and aux is not None and out_dt == L.DG_BF16:
            p.mask_in = MaskBits.slice_ptr(aux, aux_off, N, out_strides)
        if dbias is not None:  # staging scratch of the bias-gradient rows (zero between launches; one per process and device)
            ws = Ops._dbias_ws.get(str(x.device))
            if ws is None:
                ws = Ops._dbias_ws[str(x.device)] = torch.zeros(L.DBIAS_WS_FLOATS, dtype=torch.float32, device=x.device)
            p.dbias_ws = L.ptr(ws)
        db_rows = 0
        if dbias is not None and DETERMINISTIC:
            pl = L.DgConvPlan()
            L.check(self.lib.dg_conv_plan(C.byref(p), self._f, self.wg_cap, C.byref(pl)), "dg_conv_plan")
            if pl.dbias_rows > 0:
                part = WGRAD_WS.take(pl.dbias_rows * N, x.device)
                if part is not None:
                    db_rows, p.dbias_part = pl.dbias_rows, part
        if TRACE is not None:
            pl = L.DgConvPlan()
            L.check(self.lib.dg_conv_plan(C.byref(p), self._f, self.wg_cap, C.byref(pl)), "dg_conv_plan")
            TRACE.append(("conv", pl.family, pl.bm, pl.bn, pl.tiles, pl.workgroups, pl.tiles_per_wg,
                          f"mode{mode}adj{adj} B{B} {Hc}x{Wc} K{K} N{N}", pl.thin_mfma,
                          (1 if p.mask_out else 0) | (pl.mask_bits & 2 if p.mask_in else 0)))
        if PROFILE is None:
            L.check(self.lib.dg_conv_ex(C.byref(p), self._f, self.wg_cap, L.stream_ptr()), "dg_conv_ex")
            self._db_rows_done(p, db_rows, dbias, N, defer_db)
            if x2_out is not None:
                x2_pack(out, x2_out, out_off, B * N)
            return sum_parts
        # bench.py's instrumented pass: HIP events on the launch stream around this one kernel
        choice = self.lib.dg_conv_kernel_choice(C.byref(p)) if self.force == 0 else self.force
        wes = 2 if p.w_dtype == L.DG_BF16 else 4
        mb = 0
        if p.mask_out or p.mask_in:
            pl = L.DgConvPlan()
            L.check(self.lib.dg_conv_plan(C.byref(p), self._f, self.wg_cap, C.byref(pl)), "dg_conv_plan")
            mb = pl.mask_bits if p.mask_in else 1  # (mask_out behind a kernel without it: the packing launch writes the same bytes)
        flops, nbytes = conv_algorithmic(mode, B, Hc, Wc, K, N, ies, oes, wes, epi == L.EPI_MASK, mb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(self.lib.dg_conv_ex(C.byref(p), self._f, self.wg_cap, L.stream_ptr()), "dg_conv_ex")
        e1.record()
        PROFILE.append(({2: "conv_mfma_kernel", 3: "conv_thin_kernel"}.get(choice, "conv_direct_kernel"), flops, nbytes, e0, e1,
                        f"mode{mode}adj{adj} B{B} {Hc}x{Wc} K{K} N{N}"))
        self._db_rows_done(p, db_rows, dbias, N, defer_db)
        if x2_out is not None:
            x2_pack(out, x2_out, out_off, B * N)
        return sum_parts

    def _db_rows_done(self, p, rows, dbias, N, defer_db):
        """the launch left `rows` partial bias-gradient rows in the workspace: queue their sum onto dbias"""
        if rows:
            WGRAD_WS.add(p.dbias_part, dbias, N, rows, 1)
            if not defer_db:
                WGRAD_WS.flush()

    def wgrad_plan(self, p, accumulate=1):
        pl = L.DgWgradPlan()
        L.check(self.lib.dg_wgrad_plan(C.byref(p), accumulate, self._f, C.byref(pl)), "dg_wgrad_plan")
        return pl

    def wgrad(self, wmode, ring, B, Hc, Wc, Ci, Co, a, a_strides, g, g_strides, dw_ptr, scale, rowscale=None,
              accumulate=1, a_dt=None, g_dt=None, a_off=0, g_off=0, g_mod=0, defer=False):
        """dW (+)= scale * sum_b rowscale[b] a_b (x) g_(b % g_mod).  Kernels with a split-K workspace form (the MFMA LDS-DMA
        kernel) write their partial tiles to WGRAD_WS; `defer=True` leaves the sum to the caller's WGRAD_WS.flush() (one
        launch for a whole network), otherwise it follows at once.  g_mod > 0 needs that kernel: check `wgrad_takes_map`."""
        p = self._wgrad_params(wmode, ring, B, Hc, Wc, Ci, Co, a, a_strides, g, g_strides, dw_ptr, scale, rowscale, a_dt,
                               g_dt, a_off, g_off, g_mod)
        pl = self.wgrad_plan(p, accumulate) if self.use_ws else None
        if (self._group is not None and defer and pl is not None and pl.ws_floats > 0 and pl.variant == 5
                and len(self._group) < self.GROUP_MAX and pl.ws_floats <= WgradWorkspace.FLOATS):
            # inside `with ops.grouped():` - the launch joins the group's ONE launch (dg_wgrad_group) at the end of the block:
            # its geometry (K split, workspace) is decided there, for the group as a whole
            self._group.append((p, (wmode, B, Hc, Wc, Ci, Co), (a, g, rowscale), dw_ptr, int(accumulate)))
            return
        if pl is not None and pl.ws_floats > 0:
            p.ws = WGRAD_WS.take(pl.ws_floats, a.device)
        if TRACE is not None:
            TRACE.append(("wgrad", self.lib.dg_wgrad_kernel_variant(C.byref(p), self._f),
                          f"wmode{wmode} B{B} {Hc}x{Wc} Ci{Ci} Co{Co}", 0 if pl is None else pl.splits,
                          0 if pl is None else pl.tap_pairs, bool(p.ws), g_mod))
        if PROFILE is None:
            L.check(self.lib.dg_wgrad(C.byref(p), accumulate, self._f, L.stream_ptr()), "dg_wgrad")
        else:
            choice = self.lib.dg_wgrad_kernel_choice(C.byref(p)) if self.force in (0, 7, 8) else self.force
            flops, nbytes = wgrad_algorithmic(wmode, B, Hc, Wc, Ci, Co, 2 if p.a_dtype == L.DG_BF16 else 4,
                                              2 if p.g_dtype == L.DG_BF16 else 4)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(self.lib.dg_wgrad(C.byref(p), accumulate, self._f, L.stream_ptr()), "dg_wgrad")
            e1.record()
            PROFILE.append(({2: "wgrad_mfma_kernel", 3: "wgrad_thin_kernel"}.get(choice, "wgrad_direct_kernel"), flops, nbytes,
                            e0, e1, f"wmode{wmode} B{B} {Hc}x{Wc} Ci{Ci} Co{Co}"))
        if p.ws:
            WGRAD_WS.add(p.ws, dw_ptr, (1 if wmode == 2 else 16) * Ci * Co, pl.splits, int(accumulate))
            if not defer:
                WGRAD_WS.flush()

    GROUP_MAX = 4   # (GROUP_MAX of csrc/wgrad_mfma_dma.hip)

    def grouped(self):
        """`with ops.grouped():` - the deferred weight-gradient launches of the block that run on the MFMA LDS-DMA kernel with a
        split-K workspace become ONE launch (dg_wgrad_group: the layers of a network are independent of each other, and one
        grid lets the ring fill / partial-tile stores of one layer overlap the matrix work of the next); anything else in the
        block launches at once as before.  Nothing reads the partials before the caller's WGRAD_WS.flush(), which comes after
        the block.  DUSTY_GAN_WGRAD_GROUP=0: single launches (A/B, tests)."""
        return _WgradGroup(self)

    def _launch_group(self, items):
        """issue the queued launches of a `grouped()` block: ONE dg_wgrad_group launch when there are several (geometry from
        dg_wgrad_group_plan - with Ops.group_rounds > 0 every layer gets its FLOP share of rounds x 512 workgroups instead of
        512 of its own: fewer partial tiles), single launches otherwise.  The workspace of the whole group is taken in ONE
        piece, so a reduce that the allocation triggers only ever sums partials of launches already issued."""
        if not items:
            return
        if len(items) > 1:
            # a group whose partial tiles do not fit the workspace cap together goes out in pieces that do (each piece takes
            # its workspace in one allocation, which may reduce the pieces before it early: they have been issued by then)
            need = []
            for it in items:
                pl = self.wgrad_plan(it[0], 1)
                need.append((pl.ws_floats + 63) // 64 * 64)
            if sum(need) > WgradWorkspace.FLOATS:
                piece, tot = [], 0
                for it, nf in zip(items, need):
                    if piece and tot + nf > WgradWorkspace.FLOATS:
                        self._launch_group(piece)
                        piece, tot = [], 0
                    piece.append(it)
                    tot += nf
                if len(piece) < len(items):
                    self._launch_group(piece)
                    return
        n = len(items)
        arr = (L.DgWgrad * n)()
        for i, it in enumerate(items):
            C.memmove(C.byref(arr, i * C.sizeof(L.DgWgrad)), C.byref(it[0]), C.sizeof(L.DgWgrad))
        plans = (L.DgWgradPlan * n)()
        grouped = n > 1 and Ops.group_enabled
        if grouped:
            L.check(self.lib.dg_wgrad_group_plan(arr, n, self._f, Ops.group_rounds, plans), "dg_wgrad_group_plan")
        else:
            for i in range(n):
                L.check(self.lib.dg_wgrad_plan(C.byref(arr[i]), 1, self._f, C.byref(plans[i])), "dg_wgrad_plan")
        sizes = [(plans[i].ws_floats + 63) // 64 * 64 for i in range(n)]
        base = WGRAD_WS.take(sum(sizes), items[0][2][0].device)
        if base is None:     # (larger than the workspace cap: atomics onto dW, one launch each)
            for it in items:
                L.check(self.lib.dg_wgrad(C.byref(it[0]), it[4], self._f, L.stream_ptr()), "dg_wgrad")
            return
        off = 0
        for i in range(n):
            arr[i].ws = base + 4 * off
            off += sizes[i]
        descs = [f"wmode{d[0]} B{d[1]} {d[2]}x{d[3]} Ci{d[4]} Co{d[5]}" for _, d, _, _, _ in items]
        if TRACE is not None:
            for i, it in enumerate(items):
                TRACE.append(("wgrad", 5, descs[i], plans[i].splits, plans[i].tap_pairs, True, it[0].g_mod))
            if grouped:
                TRACE.append(("wgrad_group", n, descs))

        def algo(i):
            p, (wmode, B, Hc, Wc, Ci, Co) = items[i][0], items[i][1]
            return wgrad_algorithmic(wmode, B, Hc, Wc, Ci, Co, 2 if p.a_dtype == L.DG_BF16 else 4, 2 if p.g_dtype == L.DG_BF16 else 4)
        if grouped:
            if PROFILE is None:
                L.check(self.lib.dg_wgrad_group(arr, n, self._f, Ops.group_rounds, L.stream_ptr()), "dg_wgrad_group")
            else:
                fb = [algo(i) for i in range(n)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(self.lib.dg_wgrad_group(arr, n, self._f, Ops.group_rounds, L.stream_ptr()), "dg_wgrad_group")
                e1.record()
                PROFILE.append(("wgrad_mfma_kernel", sum(f for f, _ in fb), sum(b for _, b in fb), e0, e1, f"group of {n} layers"))
        else:
            for i in range(n):
                if PROFILE is None:
                    L.check(self.lib.dg_wgrad(C.byref(arr[i]), 1, self._f, L.stream_ptr()), "dg_wgrad")
                    continue
                f, b = algo(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(self.lib.dg_wgrad(C.byref(arr[i]), 1, self._f, L.stream_ptr()), "dg_wgrad")
                e1.record()
                PROFILE.append(("wgrad_mfma_kernel", f, b, e0, e1, descs[i]))
        for i, it in enumerate(items):
            (wmode, B, Hc, Wc, Ci, Co) = it[1]
            WGRAD_WS.add(arr[i].ws, it[3], 16 * Ci * Co, plans[i].splits, it[4])

    def wgrad_takes_map(self, wmode, ring, B, Hc, Wc, Ci, Co, a, a_strides, g, g_strides, dw_ptr):
        """whether the kernel that would run this launch has the g-sample index map (DgWgrad.g_mod): the LDS-DMA kernel
        and Down1's thin matrix-core kernel"""
        p = self._wgrad_params(wmode, ring, B, Hc, Wc, Ci, Co, a, a_strides, g, g_strides, dw_ptr, 1.0, None, None, None,
                               0, 0, 0)
        return self.lib.dg_wgrad_has_sample_map(C.byref(p), self._f) == 1

    def _wgrad_params(self, wmode, ring, B, Hc, Wc, Ci, Co, a, a_strides, g, g_strides, dw_ptr, scale, rowscale, a_dt, g_dt,
                      a_off, g_off, g_mod):
        if wmode == 2:   # Proj's GEMM shape runs on the fp32 kernels: split-bf16 operands through fp32 copies
            if is_x2(a):
                a = x2_unpack(a, self._f32_copy("wa", a))
            if is_x2(g):
                g = x2_unpack(g, self._f32_copy("wg", g))
        if is_x2(a):
            a_dt = L.DG_BF16X2
        if is_x2(g):
            g_dt = L.DG_BF16X2
        p = L.DgWgrad()
        p.wmode, p.ring = wmode, int(ring)
        p.B, p.Hc, p.Wc, p.Ci, p.Co = B, Hc, Wc, Ci, Co
        a_dt = self.dt if a_dt is None else a_dt
        g_dt = self.dt if g_dt is None else g_dt
        p.a = L.ptr(a) + (2 if a_dt == L.DG_BF16 else 4) * a_off
        p.a_sb, p.a_sp, p.a_sc = a_strides
        p.g = L.ptr(g) + (2 if g_dt == L.DG_BF16 else 4) * g_off
        p.g_sb, p.g_sp, p.g_sc = g_strides
        p.dw, p.scale, p.rowscale = dw_ptr, scale, L.ptr(rowscale)
        p.a_dtype, p.g_dtype = a_dt, g_dt
        p.ws, p.g_mod = None, int(g_mod)
        return p


def x2_eligible(cfg):
    """networks whose fat layers the split-bf16 kernels tile (the ping-pong conv: 64-channel groups, circular power-of-two rows
    of at least 64 columns at every level); anything else keeps fp32 storage and the register-split kernels"""
    return bool(cfg.ring and all(c % 64 == 0 for c in cfg.ch) and (cfg.W & (cfg.W - 1)) == 0 and cfg.w0 >= 64)


class GEngine:
    """Generator forward / backward on one ParamStore (models/gans/dcgan_eqlr.py:49-72 + models/dusty.py)."""
    head_tanh_fused = os.environ.get("DUSTY_GAN_HEAD_TANH", "1") != "0"   # (A/B switch, tests: the separate head_post launch)

    def __init__(self, cfg: NetCfg, dtype, x3=False, x2=False):
        self.cfg, self.dtype = cfg, dtype
        self.ops = Ops(dtype, x3=x3)
        # fp32x3 with split storage: the feature maps a0..a3 and their gradient chains are DG_BF16X2 (tagged float32 tensors)
        self.x2_asked = bool(x2)
        self.x2 = self.x2_asked and self.ops.x3 and x2_eligible(cfg)
        self.ws_B = 0

    def alloc(self, B, device):
        if self.ws_B == B and self.a[0].device == device:
            return
        c = self.cfg
        T = self.dtype
        hw = [(c.h0 << i, c.w0 << i) for i in range(5)]  # grids of a0..a3 and the image
        chs = [c.ch[3], c.ch[2], c.ch[1], c.ch[0]]
        self.grid = hw
        self.a = [torch.empty(B * hw[i][0] * hw[i][1] * chs[i], dtype=T, device=device) for i in range(4)]
        self.dp = [torch.empty_like(t) for t in self.a]  # gradients w.r.t. the pre-activations of a0..a3
        if self.x2:
            for t in self.a + self.dp:
                tag_x2(t)
        # 1-bit leaky-relu mask of a3 only: its consumer (the Head's backward-data) is HBM-bound and a3 is 57 % of the
        # generator's activation bytes.  Measured per layer (scripts/bench_conv.py, batch 32): writing the bits costs the Up
        # forward passes 3.6-5.5 us each (the MODE_UP tiles write every other pixel: 2-byte pieces 64 bytes apart) and
        # saves Up2 / Up3 backward-data 0-1 us - a1 and a2 keep reading the activation itself (a0 = Proj's output: 2 % of the bytes)
        self.abits = [MaskBits.register(self.a[3])]
        HW = c.H * c.W
        self.gout = torch.empty(B, c.nheads, c.H, c.W, dtype=torch.float32, device=device)
        self.draw = torch.empty_like(self.gout)
        self.cp = 2 if c.nheads <= 2 else 4  # channel padding of the pixel-major head gradient
        self.draw_pm = (torch.empty(B, c.H, c.W, self.cp, dtype=torch.bfloat16, device=device)
                        if (T == torch.bfloat16 and c.nheads <= 4 and c.ring) else None)
        self.hp_ws = torch.zeros(B * 1024, dtype=torch.float32, device=device)  # dg_head_post_bwd's per-sample bias staging
        self.mask = torch.empty(B, max(c.nheads - 1, 1), c.H, c.W, dtype=torch.float32, device=device)
        self.depth = torch.empty(B, 1, c.H, c.W, dtype=torch.float32, device=device)
        self.dsum_parts = torch.empty(B * 256, dtype=torch.float32, device=device)   # DgConv.tanh_sum_parts (baseline generator)
        self.zT = torch.empty(B * c.nz, dtype=T, device=device)
        self.noise_pixel = None
        self.noise_image = None
        k = c.nheads - 1
        s_depth = 1.0 / math.sqrt(1 * 16)
        s_conf = 1.0 / math.sqrt(k * 16) if k else 0.0
        self.head_scales = (s_depth, s_conf)
        self.nscale = torch.tensor([s_depth] + [s_conf] * k, dtype=torch.float32, device=device)
        self.HW = HW
        self.ws_B = B

    def forward(self, st: ParamStore, z, noise=None, training=True, z_ready=False):
        """z [B,nz] fp32; noise: dict(pixel [B,1,H,W], image [B,1,1,1]) logistic noise (dusty archs).
        z_ready: self.zT already holds z in the compute dtype (written by the launch that drew z: dg_step_prologue).
        Returns the reference's output dict (views of engine workspaces)."""
        c, o, lib = self.cfg, self.ops, L.lib()
        B = z.shape[0]
        self.alloc(B, z.device)
        if self.x2:
            st.enable_x2()
        st.refresh_shadows(self.dtype)
        sp = L.stream_ptr()
        chs = [c.ch[3], c.ch[2], c.ch[1], c.ch[0]]
        if not z_ready:
            z = z.contiguous().float()
            L.check(lib.dg_cast(L.ptr(z), L.ptr(self.zT), o.dt, B * c.nz, sp), "dg_cast")
        # Proj (dcgan_eqlr.py:6-16): GEMM [B,nz] x [N',nz]^T, N' = h0*w0*C3 in (y,x,c) order
        Np = c.h0 * c.w0 * chs[0]
        o.conv(L.MODE_GEMM, 0, 1, B, 1, 1, c.nz, Np, self.zT, (c.nz, 0, 1), self.a[0], (Np, 0, 1), st.sptr("proj_w"),
               1.0 / math.sqrt(Np), L.EPI_LRELU, bias=st.fptr("proj_b"), bias_mod=chs[0])
        # Up x3 (dcgan_eqlr.py:19-26)
        for i in (1, 2, 3):
            hc, wc = self.grid[i - 1]
            ci, co = chs[i - 1], chs[i]
            o.conv(L.MODE_UP, 0, c.ring, B, hc, wc, ci, co, self.a[i - 1], (hc * wc * ci, ci, 1), self.a[i],
                   (4 * hc * wc * co, co, 1), L.ptr(st.coci[f"up{i}_w"]), 1.0 / math.sqrt(co * 16), L.EPI_LRELU,
                   bias=st.fptr(f"up{i}_b"), bias_mod=co)
        # Head (dcgan_eqlr.py:29-46), all heads in one pass, planar fp32 output
        hc, wc = self.grid[3]
        # (weight (tap, n = head, k = ci) in the fp32 master [tap][ci][co]: strides (ci co, 1, co))
        frag = (st.up_frag("head_w", (chs[3] * c.nheads, 1, c.nheads), c.nheads, hc, 0)
                if (c.ring and chs[3] == 64 and c.nheads <= 3) else None)
        arch = ARCH_ID[c.arch]
        # the baseline generator's head (one channel, then torch.tanh, dcgan_eqlr.py:69-72): the thin matrix-core kernel applies
        # the tanh itself and stores the image's per-sample sums as partials (DgConv.tanh_sum_parts, round 6) - no head
        # post-processing launch; `depth` then IS gout
        parts = o.conv(L.MODE_UP, 0, c.ring, B, hc, wc, chs[3], c.nheads, self.a[3], (hc * wc * chs[3], chs[3], 1), self.gout,
                       (c.nheads * self.HW, 1, self.HW), L.ptr(st.coci["head_w"]), 1.0, L.EPI_LINEAR,
                       bias=st.fptr("head_b"), bias_mod=c.nheads, out_dt=L.DG_F32, nscale=self.nscale, up_frag=frag,
                       tanh_sums=self.dsum_parts if (arch == 0 and GEngine.head_tanh_fused) else None)
        if parts:
            out = OrderedDict()
            out["depth"] = L.tag_sums(self.gout.view(B, 1, c.H, c.W), self.dsum_parts, parts=parts)
            return out
        if arch:
            if noise is None or "pixel" not in noise:
                raise ValueError("dusty generator needs logistic noise (engine.sample_noise or injected)")
            self.noise_pixel = noise["pixel"].contiguous().float()
            self.noise_image = noise["image"].contiguous().float() if (arch == 2 and training) else None
            if arch == 2 and training and self.noise_image is None:
                raise ValueError("dusty2 in training mode needs image-level noise")
        sums = L.AccArena.take(B, self.depth.device) if self.HW % 256 == 0 else None
        hp_args = (L.ptr(self.gout), L.ptr(self.noise_pixel) if arch else None,
                   L.ptr(self.noise_image) if arch == 2 and training else None, arch, int(training), c.tau, c.drop_const, B,
                   self.HW, L.ptr(self.mask), L.ptr(self.depth))
        if sums is not None:  # per-sample sums of the depth image in the same pass (DiffAugment's contrast reads them)
            L.check(lib.dg_head_post_fwd_sum(*hp_args, L.ptr(sums), sp), "dg_head_post_fwd_sum")
            L.tag_sums(self.depth, sums)
        else:
            L.check(lib.dg_head_post_fwd(*hp_args, sp), "dg_head_post_fwd")
        out = OrderedDict()
        out["depth"] = self.depth
        if arch:
            out["confidence"] = self.gout[:, 1:]
            out["depth_orig"] = self.gout[:, 0:1]
            out["mask"] = self.mask
        return out

    def proj_wgrad(self, st: ParamStore, dp0, zT, nb, accumulate=False):
        """Proj weight gradient dW[n'][k] = s * sum_b dp0[b][n'] z[b][k] over `nb` samples (the local batch, or the
        all-gathered global batch in data-parallel runs: utils/dist.py).  Plain stores unless accumulating."""
        c = self.cfg
        Np = c.h0 * c.w0 * c.ch[3]
        self.ops.wgrad(2, 1, 1, 1, nb, Np, c.nz, dp0, (0, Np, 1), zT, (0, c.nz, 1), st.fptr("proj_w", st.grad),
                       1.0 / math.sqrt(Np), accumulate=int(accumulate))

    def backward(self, st: ParamStore, ddepth, accumulate_proj=False, skip_proj=False, data_only=False,
                 chain_first=False, after_chain=None, after_up1=None):
        """ddepth [B,1,H,W] fp32 = dLoss/d(output depth).  Accumulates every G parameter gradient into st.grad
        (the autograd work of loss_G.backward(), trainers/dcgan_amp.py:309).  data_only: just the backward-data chain
        (self.draw, self.dp[3..0] = gradients w.r.t. the pre-activations), no parameter gradient is touched - the
        first half of the path-length regulariser's d(sum x y)/dz.
        chain_first (data-parallel runs): the whole backward-data chain first - after it dp[0], Proj's gradient operand,
        and every bias gradient are final (`after_chain()`: the trainer starts the operand all-gather) - then the weight
        gradients largest first (`after_up1()`: the bucket that holds 75 % of the remaining gradient bytes is final), so
        both exchanges run beside the remaining weight-gradient kernels.  Same kernels, same results."""
        if data_only:
            return self._backward_chain(st, ddepth, self.draw, self.draw_pm, self.dp, acts=None, chain=None,
                                        second_of=None)
        c, o, lib = self.cfg, self.ops, L.lib()
        B = ddepth.shape[0]
        sp = L.stream_ptr()
        chs = [c.ch[3], c.ch[2], c.ch[1], c.ch[0]]
        arch = ARCH_ID[c.arch]
        s_depth, s_conf = self.head_scales
        head = (L.ptr(self.gout), L.ptr(self.noise_pixel) if arch else None, L.ptr(self.noise_image) if arch == 2 else None,
                L.ptr(self.mask) if arch else None)
        tail = (s_depth, s_conf, None if self.draw_pm is not None else L.ptr(self.draw), st.fptr("head_b", st.grad),
                L.ptr(self.draw_pm), self.cp, L.ptr(self.hp_ws), sp)
        if isinstance(ddepth, AugGrad):  # DiffAugment's adjoint gather inside this launch: the upstream gradient is never written
            g = ddepth
            rc = lib.dg_head_post_bwd_aug(*head, L.ptr(g.gy), *g.args, g.A.mask, L.ptr(g.gsum), arch, c.tau, c.drop_const, B,
                                          c.H, c.W, *tail)
            if rc == L.DG_EUNSUPPORTED:
                ddepth = g.materialize()
            else:
                L.check(rc, "dg_head_post_bwd_aug")
                ddepth = None
        if ddepth is not None:
            L.check(lib.dg_head_post_bwd(*head, L.ptr(ddepth), arch, c.tau, c.drop_const, B, self.HW, *tail), "dg_head_post_bwd")
        pl = (c.nheads * self.HW, 1, self.HW)
        pm = self.draw_pm is not None  # bf16: the pixel-major copy of the head gradient feeds the two thin MFMA kernels
        cp = self.cp
        hsrc, hstr, hkw_w, hkw_c = ((self.draw_pm, (self.HW * cp, cp, 1), {}, {}) if pm else
                                    (self.draw, pl, {"g_dt": L.DG_F32}, {"in_dt": L.DG_F32}))

        def head_wgrad():
            hc, wc = self.grid[3]
            o.wgrad(1, c.ring, B, hc, wc, chs[3], c.nheads, self.a[3], (hc * wc * chs[3], chs[3], 1), hsrc, hstr,
                    st.fptr("head_w", st.grad), 1.0, defer=True, **hkw_w)

        def head_bwd_data():  # gradient w.r.t. Up3's pre-activation, fused lrelu' mask + bias grad
            hc, wc = self.grid[3]
            o.conv(L.MODE_S2, 1, c.ring, B, hc, wc, c.nheads, chs[3], hsrc, hstr, self.dp[3],
                   (hc * wc * chs[3], chs[3], 1), st.sptr("head_w"), 1.0, L.EPI_MASK, aux=self.a[3],
                   dbias=st.fptr("up3_b", st.grad), bias_mod=chs[3], defer_db=True, **hkw_c)

        def up_wgrad(i):
            hc, wc = self.grid[i - 1]
            ci, co = chs[i - 1], chs[i]
            o.wgrad(1, c.ring, B, hc, wc, ci, co, self.a[i - 1], (hc * wc * ci, ci, 1), self.dp[i],
                    (4 * hc * wc * co, co, 1), st.fptr(f"up{i}_w", st.grad), 1.0 / math.sqrt(co * 16), defer=True)

        def up_bwd_data(i):
            hc, wc = self.grid[i - 1]
            ci, co = chs[i - 1], chs[i]
            prev_b = f"up{i - 1}_b" if i > 1 else "proj_b"
            o.conv(L.MODE_S2, 1, c.ring, B, hc, wc, co, ci, self.dp[i], (4 * hc * wc * co, co, 1), self.dp[i - 1],
                   (hc * wc * ci, ci, 1), st.sptr(f"up{i}_w"), 1.0 / math.sqrt(co * 16), L.EPI_MASK, aux=self.a[i - 1],
                   dbias=st.fptr(prev_b, st.grad), bias_mod=ci, defer_db=True)

        # The weight gradients of Up1-3 only read finished buffers (a[i-1], dp[i]: nothing below overwrites them), so they are
        # collected while the backward-data chain is issued and leave as ONE launch behind it (dg_wgrad_group, round 5: the ring
        # fill and the partial-tile stores of one layer under the matrix work of the next - the three B-sized launches ran at
        # 0.31 of the matrix peak, a quarter of each being ramp).
        if chain_first:
            head_bwd_data()
            for i in (3, 2, 1):
                up_bwd_data(i)
            if after_chain is not None:
                after_chain()
            with o.grouped():
                up_wgrad(1)
                up_wgrad(2)
                up_wgrad(3)
            if after_up1 is not None:
                after_up1()
            head_wgrad()
        else:
            with o.grouped():
                head_wgrad()
                head_bwd_data()
                for i in (3, 2, 1):
                    up_wgrad(i)
                    up_bwd_data(i)
        if not skip_proj:
            self.proj_wgrad(st, self.dp[0], self.zT, B, accumulate_proj)

    # ------------------------------------------------------------------ path-length regulariser (trainers/dcgan_amp.py:268-306)
    def _backward_chain(self, st, ddepth, draw, draw_pm, dp, acts, chain, second_of, thead=None):
        """One walk down the generator from the head to Proj's pre-activation.
        acts is None  : data only (first-order chain of d(sum x y)/dz).
        acts given    : the forward-over-reverse walk.  `draw`/`dp` receive the TANGENT gradient chain (upstream =
                        Hessian of the head post-processing applied to `thead`); weight gradients accumulate
                        a (x) tangent-chain + tangent-activations (x) first-order chain, bias gradients the sums of
                        the tangent chain.  acts = tangent activations, chain = (draw1, draw_pm1, dp1) of the
                        first-order walk."""
        c, o, lib = self.cfg, self.ops, L.lib()
        B = ddepth.shape[0]
        sp = L.stream_ptr()
        chs = [c.ch[3], c.ch[2], c.ch[1], c.ch[0]]
        arch = ARCH_ID[c.arch]
        s_depth, s_conf = self.head_scales
        full = acts is not None
        g = st.grad
        common = (L.ptr(self.gout), L.ptr(self.noise_pixel) if arch else None,
                  L.ptr(self.noise_image) if arch == 2 else None, L.ptr(self.mask) if arch else None, L.ptr(ddepth))
        if full:
            L.check(lib.dg_head_post_bwd2(*common, L.ptr(thead), arch, c.tau, c.drop_const, B, self.HW, s_depth, s_conf,
                                          L.ptr(draw), st.fptr("head_b", g), L.ptr(draw_pm), self.cp, sp),
                    "dg_head_post_bwd2")
        else:
            L.check(lib.dg_head_post_bwd(*common, arch, c.tau, c.drop_const, B, self.HW, s_depth, s_conf,
                                         None if draw_pm is not None else L.ptr(draw),   # (pixel-major copy only)
                                         None, L.ptr(draw_pm), self.cp, None, sp), "dg_head_post_bwd")
        hc, wc = self.grid[3]
        pl = (c.nheads * self.HW, 1, self.HW)
        pm = draw_pm is not None
        cp = self.cp
        gsrc, gstr, gdt = (draw_pm, (self.HW * cp, cp, 1), None) if pm else (draw, pl, L.DG_F32)
        if full:
            draw1, draw_pm1, dp1 = chain
            g1 = (draw_pm1, (self.HW * cp, cp, 1), None) if pm else (draw1, pl, L.DG_F32)
            for a_src, (gs, gst, gd) in ((self.a[3], (gsrc, gstr, gdt)), (acts[3], g1)):
                kw = {} if gd is None else {"g_dt": gd}
                o.wgrad(1, c.ring, B, hc, wc, chs[3], c.nheads, a_src, (hc * wc * chs[3], chs[3], 1), gs, gst,
                        st.fptr("head_w", g), 1.0, defer=True, **kw)
        kw = {} if gdt is None else {"in_dt": gdt}
        o.conv(L.MODE_S2, 1, c.ring, B, hc, wc, c.nheads, chs[3], gsrc, gstr, dp[3], (hc * wc * chs[3], chs[3], 1),
               st.sptr("head_w"), 1.0, L.EPI_MASK, aux=self.a[3], dbias=st.fptr("up3_b", g) if full else None,
               bias_mod=chs[3], defer_db=True, **kw)
        for i in (3, 2, 1):
            hc, wc = self.grid[i - 1]
            ci, co = chs[i - 1], chs[i]
            s = 1.0 / math.sqrt(co * 16)
            if full:
                for a_src, g_src in ((self.a[i - 1], dp[i]), (acts[i - 1], chain[2][i])):
                    o.wgrad(1, c.ring, B, hc, wc, ci, co, a_src, (hc * wc * ci, ci, 1), g_src, (4 * hc * wc * co, co, 1),
                            st.fptr(f"up{i}_w", g), s, defer=True)
            prev_b = f"up{i - 1}_b" if i > 1 else "proj_b"
            o.conv(L.MODE_S2, 1, c.ring, B, hc, wc, co, ci, dp[i], (4 * hc * wc * co, co, 1), dp[i - 1],
                   (hc * wc * ci, ci, 1), st.sptr(f"up{i}_w"), s, L.EPI_MASK, aux=self.a[i - 1],
                   dbias=st.fptr(prev_b, g) if full else None, bias_mod=ci, defer_db=True)
        if full and second_of:
            self.proj_wgrad(st, dp[0], self.zT, B, True)        # z (x) tangent chain
            self.proj_wgrad(st, chain[2][0], self.vT, B, True)  # v (x) first-order chain

    def grad_z(self, st: ParamStore):
        """d(sum x y)/dz [B,nz] fp32 from the data-only backward's dp[0] (Proj: a0 = lrelu(s z W^T + b)):
        dz^T [nz][B] = s * sum_n' W[n'][:] (x) dp0[:, n'] - a reduction over Proj's 131 072 output rows, i.e. the
        weight-gradient GEMM shape (wmode 2: rows = "pixels") with W as the layer input and dp0^T, zero-padded to 64
        columns, as the gradient; split-K on the MFMA kernel instead of a one-off VALU kernel."""
        c = self.cfg
        B = self.ws_B
        Np = c.h0 * c.w0 * c.ch[3]
        Bp = (B + 63) // 64 * 64
        if getattr(self, "_dzT", None) is None or self._dzT.shape != (Np, Bp):
            self._dzT = torch.zeros(Np, Bp, dtype=self.dtype, device=self.dp[0].device)
            self._dzw = torch.empty(c.nz, Bp, dtype=torch.float32, device=self.dp[0].device)
        dp0 = x2_unpack(self.dp[0]) if is_x2(self.dp[0]) else self.dp[0]
        self._dzT[:, :B].copy_(dp0.view(B, Np).t())
        L.zero_(self._dzw)
        shadow = st.shadow[st.seg["proj_w"].off:st.seg["proj_w"].off + Np * c.nz]
        # the kernel splits its reduction over (sample, row) units: present the Np rows as R "samples" of Np / R rows
        R = 512 if Np % (512 * 64) == 0 else 1
        self.ops.wgrad(2, 1, R, 1, Np // R, c.nz, Bp, shadow, ((Np // R) * c.nz, c.nz, 1), self._dzT,
                       ((Np // R) * Bp, Bp, 1), L.ptr(self._dzw), 1.0 / math.sqrt(Np), accumulate=1)
        return self._dzw[:, :B].t().contiguous()

    def tangent_forward(self, st: ParamStore, v):
        """Forward-mode derivative of the head pre-activations along the latent direction v [B,nz]: the generator with
        the saved leaky-relu masks, no biases.  Fills self.ta[0..3] (tangent activations) and self.tout [B,heads,H,W]."""
        c, o, lib = self.cfg, self.ops, L.lib()
        B = v.shape[0]
        chs = [c.ch[3], c.ch[2], c.ch[1], c.ch[0]]
        if getattr(self, "ta", None) is None or self.ta[0].numel() != self.a[0].numel():
            self.ta = [torch.empty_like(t) for t in self.a]
            self.dp2 = [torch.empty_like(t) for t in self.a]
            if self.x2:
                for t in self.ta + self.dp2:
                    tag_x2(t)
            self.vT = torch.empty_like(self.zT)
            self.tout = torch.empty_like(self.gout)
            self.draw2 = torch.empty_like(self.gout)
            self.draw_pm2 = None if self.draw_pm is None else torch.empty_like(self.draw_pm)
        L.check(lib.dg_cast(L.ptr(v.contiguous().float()), L.ptr(self.vT), o.dt, B * c.nz, L.stream_ptr()), "dg_cast")
        Np = c.h0 * c.w0 * chs[0]
        o.conv(L.MODE_GEMM, 0, 1, B, 1, 1, c.nz, Np, self.vT, (c.nz, 0, 1), self.ta[0], (Np, 0, 1), st.sptr("proj_w"),
               1.0 / math.sqrt(Np), L.EPI_MASK, aux=self.a[0])
        for i in (1, 2, 3):
            hc, wc = self.grid[i - 1]
            ci, co = chs[i - 1], chs[i]
            o.conv(L.MODE_UP, 0, c.ring, B, hc, wc, ci, co, self.ta[i - 1], (hc * wc * ci, ci, 1), self.ta[i],
                   (4 * hc * wc * co, co, 1), L.ptr(st.coci[f"up{i}_w"]), 1.0 / math.sqrt(co * 16), L.EPI_MASK,
                   aux=self.a[i])
        hc, wc = self.grid[3]
        o.conv(L.MODE_UP, 0, c.ring, B, hc, wc, chs[3], c.nheads, self.ta[3], (hc * wc * chs[3], chs[3], 1), self.tout,
               (c.nheads * self.HW, 1, self.HW), L.ptr(st.coci["head_w"]), 1.0, L.EPI_LINEAR, out_dt=L.DG_F32,
               nscale=self.nscale)

    def backward_second(self, st: ParamStore, y, proj_terms=True):
        """the parameter gradient of <v, d(sum x y)/dz> (v folded into the tangents): accumulates into st.grad.
        proj_terms=False leaves Proj.weight's two terms, dp2[0]^T z and dp[0]^T v, to the caller (who appends them to
        the operand list of the fused optimizer / the all-gather)."""
        self._backward_chain(st, y, self.draw2, self.draw_pm2, self.dp2, acts=self.ta,
                             chain=(self.draw, self.draw_pm, self.dp), second_of=proj_terms, thead=self.tout)


class AugGrad:
    """d loss / d (generator output) not yet formed: DiffAugment's adjoint gather of `gy` (the BlurVH adjoint's output, with
    the window sums `gsum`), which GEngine.backward evaluates inside the head post-processing's backward
    (dg_head_post_bwd_aug) - or `materialize()`s where that form does not apply."""

    def __init__(self, A, gy, rp, gsum, args, keep):
        self.A, self.gy, self.rp, self.gsum, self.args, self.keep = A, gy, rp, gsum, args, keep
        self.shape = gy.shape

    def materialize(self):
        return self.A.backward_pre(self.gy, self.rp, self.gsum)


class DEngine:
    """Discriminator passes (models/gans/dcgan_eqlr.py:85-96): forward, the shared backward-data chain, the R1
    tangent pass and the weight gradients."""

    def __init__(self, cfg: NetCfg, dtype, x3=False, x2=False):
        self.cfg, self.dtype = cfg, dtype
        self.ops = Ops(dtype, x3=x3)
        # fp32x3 with split storage: h1..h4 and e1..e4 are DG_BF16X2; the final conv's pointwise kernels (logits, loss step,
        # its weight gradient) see fp32 copies of h4 / write an fp32 e4 that is packed behind them (131 072 elements a sample)
        self.x2_asked = bool(x2)
        self.x2 = self.x2_asked and self.ops.x3 and x2_eligible(cfg)
        self.ws_B = 0

    def alloc(self, nb, device):
        """Workspaces for `nb` images (the D phase uses 3B slots: real | fake | R1 tangent)."""
        if self.ws_B >= nb and self.h[0].device == device:
            return
        c, T = self.cfg, self.dtype
        self.grid = [(c.H >> i, c.W >> i) for i in range(5)]  # h0 (image) .. d4
        self.chs = [2, c.ch[0], c.ch[1], c.ch[2], c.ch[3]]
        self.per = [self.grid[i][0] * self.grid[i][1] * self.chs[i] for i in range(5)]
        self.h = [torch.empty(nb * self.per[i], dtype=T, device=device) for i in range(5)]
        self.e = [torch.empty(nb * self.per[i], dtype=T, device=device) for i in range(5)]
        # 1-bit leaky-relu masks of h1..h3 (h4's only reader of bits is the R1 tangent's last layer: measured in the step,
        # writing them cost the two Down4 forward launches +6 us and saved that pass 0.8 us)
        self.hbits = [MaskBits.register(t) for t in self.h[1:4]]
        self.h4f = self.e4f = None
        if self.x2:
            for t in self.h[1:] + self.e[1:]:
                tag_x2(t)
            self.h4f, self.e4f = torch.empty_like(self.h[4]), torch.empty_like(self.e[4])
        self.y = torch.empty(nb, dtype=torch.float32, device=device)
        self.ws_B = nb

    def forward(self, st, x, slot, tangent_of=None, mean=None):
        """x [n,1,H,W] fp32 written to batch slots [slot, slot+n).  tangent_of = slot of the saved activations whose
        lrelu masks gate the R1 tangent pass (then no bias, no activation: the Jacobian-vector product).
        mean = (src tensor, n, accumulator pointer): acc[0] += mean(src[0..n)) rides on the BlurVH launch (dg_mean_acc
        otherwise: the R1 penalty's logged value)."""
        c, o, lib = self.cfg, self.ops, L.lib()
        n = x.shape[0]
        dst = L.ptr(self.h[0]) + o.es * slot * self.per[0]
        if mean is not None:
            rc = lib.dg_blur_fwd_mean(L.ptr(x), dst, o.dt, n, c.H, c.W, int(c.ring), L.ptr(mean[0]), int(mean[1]), mean[2],
                                      L.stream_ptr())
            if rc == L.DG_EUNSUPPORTED:
                L.check(lib.dg_mean_acc(L.ptr(mean[0]), int(mean[1]), mean[2], L.stream_ptr()), "dg_mean_acc")
                mean = None
            else:
                L.check(rc, "dg_blur_fwd_mean")
        if mean is None:
            L.check(lib.dg_blur_fwd(L.ptr(x), dst, o.dt, n, c.H, c.W, int(c.ring), L.stream_ptr()), "dg_blur_fwd")
        return self._layers(st, n, slot, tangent_of)

    def forward_aug(self, st, A, sources, slot):
        """D(A(x_0) | A(x_1)): `sources` = one or two (x [n,1,H,W] fp32, DiffAugment draws) pairs filling consecutive
        slots (trainers/dcgan_amp.py:199-204: real | fake; :256-260: fake).  The augmented images are never needed again
        (D is piecewise linear and BlurVH linear: the backward passes need the draws, not the pixels), so DiffAugment and
        BlurVH run as ONE pass that writes h[0] directly (dg_diffaug_blur_fwd) when every source carries the per-sample
        sums its producer made; otherwise DiffAugment, then BlurVH."""
        c, o, lib = self.cfg, self.ops, L.lib()
        n = sources[0][0].shape[0]
        sets = [A.aug_set(x, rp) for x, rp in sources] if (c.W % 4 == 0 and len(sources) <= 2) else [None]
        if all(q is not None for q in sets):
            arr = (L.DgAugSet * len(sets))(*[q[0] for q in sets])
            rc = lib.dg_diffaug_blur_fwd(arr, len(sets), A.mask, n, c.H, c.W, int(c.ring),
                                         L.ptr(self.h[0]) + o.es * slot * self.per[0], o.dt, L.stream_ptr())
            if rc != L.DG_EUNSUPPORTED:
                L.check(rc, "dg_diffaug_blur_fwd")
                return self._layers(st, n * len(sets), slot, None)
        xcat = torch.empty(n * len(sources), 1, c.H, c.W, dtype=torch.float32, device=sources[0][0].device)
        for k, (x, rp) in enumerate(sources):
            A.apply(x, rp, out=xcat[k * n:(k + 1) * n])
        return self.forward(st, xcat, slot)

    def _layers(self, st, n, slot, tangent_of):
        """Down x4 + the final conv on h[0][slot : slot + n]"""
        c, o, lib = self.cfg, self.ops, L.lib()
        sp = L.stream_ptr()
        if self.x2:
            st.enable_x2()
        st.refresh_shadows(self.dtype)
        es = o.es
        for i in range(1, 5):
            hc, wc = self.grid[i]
            ci, co = self.chs[i - 1], self.chs[i]
            tang = tangent_of is not None
            o.conv(L.MODE_S2, 0, c.ring, n, hc, wc, ci, co, self.h[i - 1], (self.per[i - 1], ci, 1), self.h[i],
                   (self.per[i], co, 1), L.ptr(st.coci[f"d{i}_w"]), 1.0 / math.sqrt(ci * 16),
                   L.EPI_MASK if tang else L.EPI_LRELU, bias=None if tang else st.fptr(f"d{i}_b"), bias_mod=co,
                   aux=self.h[i] if tang else None, x_off=slot * self.per[i - 1], out_off=slot * self.per[i],
                   aux_off=(tangent_of or 0) * self.per[i])
        if tangent_of is None:
            nf = self.per[4]
            h4 = self._h4(slot, n)
            y = L.AccArena.take(n, self.y.device)  # logits: a pre-zeroed slice of the step's arena while a step runs
            if y is not None:
                L.check(lib.dg_final_fwd_acc(L.ptr(h4) + es * slot * nf, o.dt, st.fptr("final_w"), st.fptr("final_b"),
                                             1.0 / math.sqrt(nf), n, nf, L.ptr(y), sp), "dg_final_fwd_acc")
                return y
            L.check(lib.dg_final_fwd(L.ptr(h4) + es * slot * nf, o.dt, st.fptr("final_w"), st.fptr("final_b"),
                                     1.0 / math.sqrt(nf), n, nf, L.ptr(self.y) + 4 * slot, sp), "dg_final_fwd")
        return self.y[slot:slot + n]

    def _h4(self, slot, n):
        """h4 as the pointwise kernels of the final conv read it: the tensor itself, or (split storage) the fp32 copy with the
        slots [slot, slot + n) brought up to date"""
        if not self.x2:
            return self.h[4]
        return x2_unpack(self.h[4], self.h4f, slot * self.per[4], n * self.per[4])

    def _e4(self):
        """where the final conv's backward-data writes e4 (split storage: fp32, packed into e[4] by `_e4_done`)"""
        return self.e4f if self.x2 else self.e[4]

    def _e4_done(self, slot, n):
        if self.x2:
            x2_pack(self.e4f, self.e[4], slot * self.per[4], n * self.per[4])

    def _bwd_layer(self, st, i, slot, n, rowscale, want_dbias):
        """e[i-1] = lrelu'(h[i-1]) * sqrt2 * s_i * conv_i^T(e[i])  (no mask for i == 1: BlurVH has no activation)"""
        c, o = self.cfg, self.ops
        hc, wc = self.grid[i]
        ci, co = self.chs[i - 1], self.chs[i]
        first = i == 1
        # Down1 (64 -> <= 3 image channels): the thin matrix-core kernel's weight fragments follow the shadows
        # (weight (tap, n = ci, k = co) in the master [tap][ci][co]: strides (ci co, co, 1))
        frag = st.up_frag("d1_w", (ci * co, co, 1), ci, hc, 1) if (first and c.ring and co == 64 and ci <= 3) else None
        o.conv(L.MODE_UP, 1, c.ring, n, hc, wc, co, ci, self.e[i], (self.per[i], co, 1), self.e[i - 1],
               (self.per[i - 1], ci, 1), st.sptr(f"d{i}_w"),
               1.0 / math.sqrt(ci * 16), L.EPI_LINEAR if first else L.EPI_MASK,
               aux=None if first else self.h[i - 1],
               dbias=(st.fptr(f"d{i - 1}_b", st.grad) if (want_dbias and not first) else None), bias_mod=ci,
               rowscale=rowscale, x_off=slot * self.per[i], out_off=slot * self.per[i - 1],
               aux_off=slot * self.per[i - 1], up_frag=frag, defer_db=True)

    def backward_data(self, st, slot, n, up, rowscale, want_dbias, skip_final=False):
        """Backward-data chain over batch slots [slot, slot+n): e4 = up*s_f*wf*mask4, then e3, e2, e1 (each the
        gradient w.r.t. a layer's pre-activation).  up: per-sample upstream gradient or None (= 1, the R1 chain);
        rowscale: per-sample weight of the bias-gradient sums (dLoss/dy_real for the shared real chain).
        skip_final: e4 is there already (`final_gan_bwd`)."""
        o, lib = self.ops, L.lib()
        nf = self.per[4]
        if not skip_final:
            L.check(lib.dg_final_bwd_data(L.ptr(self._h4(slot, n)) + o.es * slot * nf, o.dt, st.fptr("final_w"), L.ptr(up),
                                          L.ptr(rowscale), 1.0 / math.sqrt(nf), n, nf, self.chs[4],
                                          L.ptr(self._e4()) + o.es * slot * nf,
                                          st.fptr("d4_b", st.grad) if want_dbias else None, L.stream_ptr()),
                    "dg_final_bwd_data")
            self._e4_done(slot, n)
        for i in (4, 3, 2):
            self._bwd_layer(st, i, slot, n, rowscale, want_dbias)

    def final_gan_bwd(self, st, slot, B, metric, mode_g, smoothing, w_gan, y_real, y_fake, r1, dy, up, rs, acc_ptr,
                      want_dbias, want_wgrad):
        """The loss step + the final conv's backward-data (+ its weight gradient with coefficients dy) over the 2B (D
        phase, [real | fake] from `slot`) or B (G phase) samples as ONE launch (dg_final_gan_bwd).  y_real / y_fake:
        device pointers of the logits.  False - nothing launched - when the kernel does not take the shape: the caller
        then issues dg_gan_*_step, backward_data and final_wgrad."""
        if PROFILE is not None:
            return False
        o, lib = self.ops, L.lib()
        nf = self.per[4]
        C4 = self.chs[4]
        # Down4's bias gradient: one partial per element of the final map in the split-K workspace, summed per channel by the
        # reduce launch that runs anyway (DETERMINISTIC) - instead of 131 072 float atomics onto 512 addresses
        part = WGRAD_WS.take(nf, self.h[4].device) if (want_dbias and DETERMINISTIC and nf % C4 == 0 and C4 % 4 == 0) else None
        nsl = B if mode_g else 2 * B             # samples the launch covers from `slot`
        h4 = self._h4(slot, nsl)
        rc = lib.dg_final_gan_bwd(metric, int(mode_g), float(smoothing), y_real, y_fake, B, w_gan, int(r1), L.ptr(dy),
                                  L.ptr(up), L.ptr(rs), acc_ptr,
                                  st.fptr("final_b", st.grad) if not mode_g else None,
                                  L.ptr(h4) + o.es * slot * nf, o.dt, st.fptr("final_w"), 1.0 / math.sqrt(nf), nf,
                                  C4, L.ptr(self._e4()) + o.es * slot * nf,
                                  st.fptr("d4_b", st.grad) if want_dbias else None,
                                  st.fptr("final_w", st.grad) if want_wgrad else None, part, L.stream_ptr())
        if rc == L.DG_EUNSUPPORTED:
            return False   # (nothing was launched; the bump allocation is simply re-used by the next take after the flush)
        L.check(rc, "dg_final_gan_bwd")
        self._e4_done(slot, nsl)
        if part is not None:
            WGRAD_WS.add(part, st.fptr("d4_b", st.grad), C4, nf // C4, 1)
        return True

    def r1_fused_ok(self):
        """whether dg_blur_bwd_r1 (BlurVH adjoint + R1 tangent + |g|^2 in one pass) takes this image shape
        (pointwise.hip: four pixels per thread, 1024-pixel blocks)"""
        c = self.cfg
        return c.W % 4 == 0 and (c.H * c.W) % 1024 == 0

    def backward_input(self, st, slot, n, dx, r1=None):
        """Continue a chain from e1 to the image: Down1 backward-data + BlurVH adjoint -> dx [n,1,H,W] fp32.
        r1 = (oscale, ssq): the R1 form - dx = oscale * g and ssq[b] += |g_b|^2 in the same pass (ssq pre-zeroed).
        Returns True when dx was written; False - with NOTHING launched - when r1 was asked for a shape the fused
        adjoint does not take (the caller then runs the plain form and its own sum / scale passes)."""
        c, o, lib = self.cfg, self.ops, L.lib()
        if r1 is not None and not self.r1_fused_ok():
            return False
        self._bwd_layer(st, 1, slot, n, None, False)
        if r1 is not None:
            L.check(lib.dg_blur_bwd_r1(L.ptr(self.e[0]) + o.es * slot * self.per[0], o.dt, L.ptr(dx), float(r1[0]),
                                       L.ptr(r1[1]), n, c.H, c.W, int(c.ring), L.stream_ptr()), "dg_blur_bwd_r1")
            return True
        L.check(lib.dg_blur_bwd(L.ptr(self.e[0]) + o.es * slot * self.per[0], o.dt, L.ptr(dx), n, c.H, c.W,
                                int(c.ring), L.stream_ptr()), "dg_blur_bwd")
        return True

    def r1_turnaround(self, st, slot, n, tslot, oscale, ssq, mean_ptr):
        """R1 at the image, round 6 (trainers/dcgan_amp.py:218-235): Down1 backward-data on the chain of slots [slot, slot+n),
        then ONE launch (dg_blur_r1_tangent) that forms g = BlurVH^T(e0), adds |g_b|^2 to ssq[b] (and their mean over the n
        samples to *mean_ptr) and writes BlurVH(oscale g) - the tangent's first feature map - into h[0] at slots
        [tslot, tslot+n); then the tangent's Down layers through the saved masks of `slot`.  g itself is never written.
        False - NOTHING launched - where that kernel does not take the image shape: the caller runs backward_input +
        forward(tangent_of=...)."""
        c, o, lib = self.cfg, self.ops, L.lib()
        if c.W % 4 or c.H % 4 or 6 * c.W * 4 > 60 * 1024:
            return False
        self._bwd_layer(st, 1, slot, n, None, False)
        L.check(lib.dg_blur_r1_tangent(L.ptr(self.e[0]) + o.es * slot * self.per[0], o.dt,
                                       L.ptr(self.h[0]) + o.es * tslot * self.per[0], float(oscale), L.ptr(ssq), mean_ptr, int(n),
                                       n, c.H, c.W, int(c.ring), L.stream_ptr()), "dg_blur_r1_tangent")
        self._layers(st, n, tslot, slot)
        return True

    def backward_input_aug(self, st, slot, n, A, rp, lazy=False):
        """d loss / d x for D(A(x)): Down1 backward-data, then BlurVH's adjoint and DiffAugment's adjoint - two launches:
        the BlurVH adjoint also accumulates the masked per-sample sums DiffAugment's contrast term needs
        (dg_blur_bwd_augsum + dg_diffaug_bwd_pre); three (adjoint, sum, gather) where that form does not apply."""
        c, o, lib = self.cfg, self.ops, L.lib()
        self._bwd_layer(st, 1, slot, n, None, False)
        dx = torch.empty(n, 1, c.H, c.W, dtype=torch.float32, device=self.e[0].device)
        gsum = L.AccArena.take(n, dx.device) if c.W % 4 == 0 else None
        if gsum is not None:
            args, keep = A._args(rp, n, dx.device)
            rc = lib.dg_blur_bwd_augsum(L.ptr(self.e[0]) + o.es * slot * self.per[0], o.dt, L.ptr(dx), args[2], args[4],
                                        args[5], A.mask, L.ptr(gsum), n, c.H, c.W, int(c.ring), L.stream_ptr())
            if rc != L.DG_EUNSUPPORTED:
                L.check(rc, "dg_blur_bwd_augsum")
                if lazy:  # the gather runs inside the consumer (GEngine.backward: dg_head_post_bwd_aug)
                    return AugGrad(A, dx, rp, gsum, args, keep)
                return A.backward_pre(dx, rp, gsum)
        L.check(lib.dg_blur_bwd(L.ptr(self.e[0]) + o.es * slot * self.per[0], o.dt, L.ptr(dx), n, c.H, c.W,
                                int(c.ring), L.stream_ptr()), "dg_blur_bwd")
        return A.backward(dx, rp)

    def wgrad(self, st, a_slot, g_slot, n, rowscale, layers=(1, 2, 3, 4), g_mod=0):
        """dW_i += s_i * sum_b rowscale[b] * (h_{i-1}[a_slot+b] (x) e_i[g_slot + b % g_mod]) for the Down layers in `layers`.
        The split-K partials of the fat layers wait in engine.WGRAD_WS: the caller flushes before the gradient is read."""
        c, o = self.cfg, self.ops
        with o.grouped():   # the fat layers' launches as ONE (dg_wgrad_group)
            for i in layers:
                hc, wc = self.grid[i]
                ci, co = self.chs[i - 1], self.chs[i]
                o.wgrad(0, c.ring, n, hc, wc, ci, co, self.h[i - 1], (self.per[i - 1], ci, 1), self.e[i],
                        (self.per[i], co, 1), st.fptr(f"d{i}_w", st.grad), 1.0 / math.sqrt(ci * 16), rowscale=rowscale,
                        a_off=a_slot * self.per[i - 1], g_off=g_slot * self.per[i], g_mod=g_mod, defer=True)

    def wgrad_r1(self, st, B, rs3, layers=(1, 2, 3, 4)):
        """The D phase's weight gradients with R1 on (trainers/dcgan_amp.py:235 through :218-232): per layer
        wgrad(h[real | fake], e[real | fake], rs) + wgrad(t, e[real]) - the ordinary term over 2B samples and the R1 term
        (tangent activations t in slots [2B, 3B) against the real batch's chain).  Both sums have the same output, so on the
        kernel that has the gradient-sample index map they are ONE launch over the 3B input slots with g sample = b % 2B
        and per-sample weights rs3 = [dLoss/dy_real | 1 | 1]: half the launches and half the split-K partial tiles."""
        c, o = self.cfg, self.ops
        with o.grouped():   # Down2-4 at 3B samples: one launch for the three layers (round 5)
            for i in layers:
                hc, wc = self.grid[i]
                ci, co = self.chs[i - 1], self.chs[i]
                if o.wgrad_takes_map(0, c.ring, 3 * B, hc, wc, ci, co, self.h[i - 1], (self.per[i - 1], ci, 1), self.e[i],
                                     (self.per[i], co, 1), st.fptr(f"d{i}_w", st.grad)):
                    self.wgrad(st, 0, 0, 3 * B, rs3, layers=(i,), g_mod=2 * B)
                else:
                    self.wgrad(st, 0, 0, 2 * B, rs3, layers=(i,))
                    self.wgrad(st, 2 * B, 0, B, None, layers=(i,))

    def final_wgrad_term(self, slot, n):
        """the operands of final_wgrad(st, slot, n, None) for the optimizer's launch to sum (FlatAdam.step `extra`, dg_adam_fused
        ws_*): (source pointer, is_bf16, coefficient pointer, samples, sample stride, scale)"""
        nf = self.per[4]
        return (L.ptr(self.h[4]) + self.ops.es * slot * nf, self.dtype == torch.bfloat16, None, int(n), nf, 1.0 / math.sqrt(nf))

    def final_wgrad(self, st, slot, n, coef):
        """dwf += s_f * sum_b coef[b] * h4[slot+b]"""
        o, lib = self.ops, L.lib()
        nf = self.per[4]
        L.check(lib.dg_batch_wsum(L.ptr(self._h4(slot, n)) + o.es * slot * nf, o.dt, L.ptr(coef), 1.0 / math.sqrt(nf), n,
                                  nf, st.fptr("final_w", st.grad), L.stream_ptr()), "dg_batch_wsum")
