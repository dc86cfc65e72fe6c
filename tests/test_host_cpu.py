"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/dusty_gan_hip.h declares,
the ctypes structs match the header, the config tree loads, the model mirrors keep the reference's state_dict
layout, and the host-side helpers behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_util import load, sub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from dusty_gan_amd import _lib
    _lib.build()  # hipcc cross-compiles gfx950 without a GPU; a no-op when up to date
    return _lib


def test_library_exports_every_declared_symbol(built):
    h = open(os.path.join(ROOT, "include", "dusty_gan_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(dg_\w+)\s*\(", h, flags=re.M))
    assert len(declared) >= 25
    lib = built.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported by libdustygan_hip.so"
    # every binding the Python side uses is declared in the header
    assert set(built.PROTOTYPES) <= declared
    assert lib.dg_version().decode().startswith("dusty_gan_hip")


def _header_structs():
    """{struct name: [field names in order]} of every `typedef struct X {...} X;` in the header"""
    h = open(os.path.join(ROOT, "include", "dusty_gan_hip.h")).read()
    out = {}
    for cname, body in re.findall(r"typedef struct (\w+) \{(.*?)\} \1;", h, flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = [n.strip().lstrip("*").strip() for n in decl.split(",")]
            names[0] = names[0].split()[-1].lstrip("*")
            fields += [re.sub(r"\[.*\]", "", n) for n in names]
        out[cname] = fields
    return out


def _c_layout(tmp_path):
    """sizeof / offsetof of every struct of the header as the C compiler lays them out (gcc, the ABI the library was built to)"""
    import subprocess
    structs = _header_structs()
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "dusty_gan_hip.h"', 'int main(void) {']
    for cname, fields in structs.items():
        src.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for f in fields:
            src.append(f'  printf("{cname} {f} %zu\\n", offsetof({cname}, {f}));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    lay = {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, f, v = line.split()
        lay.setdefault(cname, {})[f] = int(v)
    return structs, lay


def _check_against_c(cname, cls, structs, lay, where):
    mine = [f[0].rstrip("_") for f in cls._fields_]
    assert mine == structs[cname], (where, cname, mine, structs[cname])
    assert C.sizeof(cls) == lay[cname]["sizeof"], (where, cname, C.sizeof(cls), lay[cname]["sizeof"])
    for (fname, _t), hname in zip(cls._fields_, structs[cname]):
        assert getattr(cls, fname).offset == lay[cname][hname], (where, cname, fname)


def test_ctypes_structs_match_header(built, tmp_path):
    """every struct of include/dusty_gan_hip.h has a ctypes mirror in _lib.py with the same field names, order, offsets
    and size as the C compiler's layout"""
    structs, lay = _c_layout(tmp_path)
    assert set(structs) >= {"DgConv", "DgWgrad", "DgConvPlan", "DgWgradPlan", "DgWgradReduce", "DgAugSet", "DgUpFrag", "DgDraw"}
    for cname in structs:
        assert hasattr(built, cname), f"{cname} is declared in the header but _lib.py has no ctypes mirror"
        _check_against_c(cname, getattr(built, cname), structs, lay, "_lib.py")


def test_integration_md_stubs_match_header(built, tmp_path):
    """INTEGRATION.md is the binding a maintainer copies: every ctypes Structure it shows must have the header's field
    names, order, offsets and sizeof (round 3 shipped a DgConv stub two fields short), and every `lib.dg_*` call it shows
    must be an exported symbol."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    classes = {}
    for b in blocks:
        for m in re.finditer(r"^class (\w+)\(C\.Structure\):.*?\n(?=\S|\Z)", b, flags=re.S | re.M):
            ns = {"C": C}
            exec(m.group(0), ns)  # a class statement with a literal _fields_ list
            classes[m.group(1)] = ns[m.group(1)]
    assert "DgConv" in classes and "DgWgrad" in classes, sorted(classes)
    structs, lay = _c_layout(tmp_path)
    for cname, cls in classes.items():
        assert cname in structs, f"INTEGRATION.md shows a struct the header does not declare: {cname}"
        _check_against_c(cname, cls, structs, lay, "INTEGRATION.md")
        ref = getattr(built, cname)
        assert [(n.rstrip("_"), t) for n, t in cls._fields_] == [(n.rstrip("_"), t) for n, t in ref._fields_], cname
    lib = built.lib()
    for name in set(re.findall(r"\blib\.(dg_\w+)", md)):
        assert hasattr(lib, name), f"INTEGRATION.md calls {name}, which the library does not export"


def test_error_codes_without_gpu(built):
    """argument validation happens before any launch, so it can be exercised on the CPU"""
    lib = built.lib()
    p = built.DgConv()
    assert lib.dg_conv(C.byref(p), 0, None) == built.DG_EINVAL
    w = built.DgWgrad()
    assert lib.dg_wgrad(C.byref(w), 1, 0, None) == built.DG_EINVAL
    with pytest.raises(built.DgError):
        built.check(built.DG_EUNSUPPORTED, "x")
    assert built.policy_mask(["brightness", "cutout"]) == 17
    # GANLoss dispatch: the reference's seven metric names (models/loss.py:39-61), anything else is NotImplementedError
    from dusty_gan_amd.models.loss import GANLoss, METRICS
    assert [GANLoss(m).code for m in METRICS] == list(range(7))
    assert [m for m in METRICS if GANLoss(m).relativistic] == ["ragan", "rahinge", "ralsgan"]
    with pytest.raises(NotImplementedError):
        GANLoss("nope").code
    one = (C.c_float * 1)()
    assert lib.dg_gan_d_step(0, 1.0, None, one, 1, 1.0, one, None, None, one, None, None) == built.DG_EINVAL
    assert lib.dg_gan_g_step(0, None, None, 1, 1.0, one, one, None) == built.DG_EINVAL
    with pytest.raises(KeyError):
        built.policy_mask(["nope"])


def test_config_tree_and_overrides():
    from dusty_gan_amd.utils.config import load_config
    cfg = load_config([])
    assert cfg.model.gen.arch == "dusty1/dcgan_eqlr" and cfg.dataset.shape == [64, 256]
    assert cfg.solver.batch_size == 32 and cfg.solver.loss.gp == 1 and cfg.solver.lr.beta1 == 0.0
    assert cfg.solver.augment == ["brightness", "saturation", "contrast", "translation", "cutout"]
    cfg = load_config(["model=dusty2_dcgan_eqlr", "dataset.shape=[64,1024]", "solver.loss.gp=0", "enable_amp=false"])
    assert cfg.model.gen.out_ch == {"depth": 1, "confidence": 2} and cfg.dataset.shape == [64, 1024]
    assert cfg.solver.loss.gp == 0 and cfg.enable_amp is False
    with pytest.raises(FileNotFoundError):
        load_config(["solver=nsgan"])  # the reference's broken default


@pytest.mark.parametrize("case", ["none_ring", "dusty1_ring", "dusty2_noring"])
def test_state_dict_layout_matches_reference(case):
    """keys, shapes and values round-trip through the strided views of the flat engine-layout store"""
    from dusty_gan_amd.models import define_D, define_G
    from dusty_gan_amd.utils.config import load_config
    g = load("step_" + case)
    arch = str(g["meta/arch"])
    model = {"none": "dcgan_eqlr", "dusty1": "dusty1_dcgan_eqlr", "dusty2": "dusty2_dcgan_eqlr"}[arch]
    cfg = load_config([f"model={model}", "dataset.shape=[32,64]", "model.gen.in_ch=8", "model.gen.ch_base=4",
                       "model.gen.ch_max=16", "model.dis.ch_base=4", "model.dis.ch_max=16",
                       f"model.ring={str(bool(g['meta/ring'])).lower()}", "enable_amp=false"])
    cfg.model.gen.shape = cfg.dataset.shape
    cfg.model.dis.shape = cfg.dataset.shape
    G, D = define_G(cfg), define_D(cfg)
    refG, refD = sub(g, "init/G"), sub(g, "init/D")
    assert list(G.state_dict().keys()) == list(refG.keys())
    assert list(D.state_dict().keys()) == list(refD.keys())
    G.load_state_dict(refG)
    D.load_state_dict(refD)
    for k, v in G.state_dict().items():
        assert v.shape == refG[k].shape and torch.equal(v, refG[k]), k
    for k, v in D.state_dict().items():
        assert v.shape == refD[k].shape and torch.allclose(v, refD[k]), k
    # engine layout underneath: conv weights are [ky][kx][ci][co]
    st = D.store
    w = refD["2.1.module.weight"]  # Conv2d (Co,Ci,4,4)
    assert torch.equal(st.view("d2_w"), w.permute(2, 3, 1, 0))
    # parameter count equals the reference's (probe numbers of SURVEY.md §8a at full size are checked on the GPU box)
    assert sum(p.numel() for p in G.parameters()) == sum(v.numel() for k, v in refG.items() if k != "drop_const")
    # no CPU execution path
    with pytest.raises(RuntimeError):
        D(torch.zeros(1, 1, 32, 64))
    with pytest.raises(NotImplementedError):
        cfg.model.gen.arch = "dusty3/dcgan_eqlr"
        define_G(cfg)
    with pytest.raises(NotImplementedError):
        cfg.model.dis.arch = "stylegan"
        define_D(cfg)


def test_full_size_parameter_counts():
    """SURVEY.md §8a probe: G 69 863 361 / 69 864 386 / 69 865 411 (none/dusty1/dusty2), D 2 886 593 at 64x1024"""
    from dusty_gan_amd.models.gans.dcgan_eqlr import Discriminator, Generator
    for heads, n in (({"depth": 1}, 69863361), ({"depth": 1, "confidence": 1}, 69864386),
                     ({"depth": 1, "confidence": 2}, 69865411)):
        G = Generator(512, heads, 64, 512, (64, 1024), True)
        assert sum(p.numel() for p in G.parameters()) == n
    D = Discriminator(1, 64, 512, (64, 1024), True)
    assert sum(p.numel() for p in D.parameters()) == 2886593


def test_lazy_scalars_and_flat_adam_format():
    from dusty_gan_amd.models.gans.dcgan_eqlr import Discriminator
    from dusty_gan_amd.trainers.dcgan_amp import FlatAdam, LazyScalars
    s = LazyScalars(["a", "b"], torch.tensor([1.5, 2.5]))
    assert len(s) == 2 and "a" in s and s["b"] == 2.5 and dict(s.items()) == {"a": 1.5, "b": 2.5}
    D = Discriminator(1, 4, 16, (32, 64), True)
    D.store.ensure_train_state()
    opt = FlatAdam(D, 0.002, (0.0, 0.99))
    D.store.v.uniform_()
    opt.step_count = 3
    sd = opt.state_dict()
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in D.parameters()], lr=0.002, betas=(0.0, 0.99))
    assert sd["param_groups"][0]["params"] == ref.state_dict()["param_groups"][0]["params"]
    assert sd["state"][0]["exp_avg_sq"].shape == tuple(next(iter(D.parameters())).shape)
    opt2 = FlatAdam(Discriminator(1, 4, 16, (32, 64), True), 0.1, (0.5, 0.5))
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and opt2.lr == 0.002
    for a, b in zip(opt2._param_views(opt2.store.v), opt._param_views(D.store.v)):  # (alignment gaps are not state)
        assert torch.equal(a, b)


def test_philox_oracle_known_answers():
    """Random123 kat_vectors for philox4x32-10 pin oracle/philox.py (which in turn pins the HIP generator on the GPU)"""
    from oracle.philox import bits, philox4x32_10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    assert list(bits(0, 0, 0, 1)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]


def test_scan_dataset_file_lists_and_sampler(tmp_path):
    """datasets/kitti.py:46-52 and datasets/mpo.py:45-52 file lists, split tables and the loader's sample order
    (host logic only: no scan is decoded here)"""
    import numpy as np
    from dusty_gan_amd.datasets import KITTIOdometry, SparseMPO, define_dataset
    from dusty_gan_amd.datasets.scans import KITTI_CONFIG, MPO_CONFIG, sampler_indices
    from oracle import lidar_oracle as LO
    for seq, n in ((0, 3), (8, 2), (16, 1)):
        d = tmp_path / "kitti" / "sequences" / f"{seq:02d}" / "velodyne"
        d.mkdir(parents=True)
        for i in range(n):
            np.save(d / f"{i:06d}.npy", np.zeros((2, 4, 4), np.float32))
    k = KITTIOdometry(str(tmp_path / "kitti"), "train", shape=(2, 4))
    assert len(k) == 3 and all("/00/velodyne/" in p for p in k.datalist) and k.datalist == sorted(k.datalist)
    assert len(KITTIOdometry(str(tmp_path / "kitti"), "val")) == 2 and len(KITTIOdometry(str(tmp_path / "kitti"), "custom")) == 1
    assert KITTI_CONFIG["split"]["train"] == [0, 1, 2, 3, 4, 5, 6, 7, 9, 10] and KITTI_CONFIG["split"]["val"] == [8]
    assert "Number of datapoints: 3" in repr(k)
    (tmp_path / "mpo" / "Data").mkdir(parents=True)
    for name in ("class0_set000_scan00001", "class1_set000_scan00002", "class0_set007_scan00001", "class0_set010_scan00001"):
        np.save(tmp_path / "mpo" / "Data" / f"{name}.npy", np.zeros((2, 4, 4), np.float32))
    m = SparseMPO(str(tmp_path / "mpo"), "train", shape=(2, 4))
    assert len(m) == 2 and len(SparseMPO(str(tmp_path / "mpo"), "val")) == 1 and len(SparseMPO(str(tmp_path / "mpo"), "test")) == 1
    assert MPO_CONFIG["split"]["test"] == [8, 9, 10]
    with pytest.raises(AssertionError):
        KITTIOdometry(str(tmp_path / "kitti"), "train", modality=("reflectance",))  # '"depth" is required'
    with pytest.raises(NotImplementedError):
        KITTIOdometry(str(tmp_path / "kitti"), "train", modality=("depth", "reflectance"))
    from dusty_gan_amd.utils.config import load_config
    cfg = load_config(["dataset=sparse_mpo", f"dataset.root={tmp_path / 'mpo'}"]).dataset
    ds = define_dataset(cfg, "train")
    assert isinstance(ds, SparseMPO) and ds.flip is True and define_dataset(cfg, "val").flip is False
    for n, world in ((10, 1), (10, 4), (3, 8)):
        for rank in range(world):
            assert sampler_indices(n, world, rank) == LO.sampler_indices(n, world, rank)


def test_registry_is_a_table_and_refuses_unknown_archs():
    """models.define_G / define_D (reference models/__init__.py:5-50): "<masker>/<backbone>" strings from tables, the
    reference's NotImplementedError for anything that is not in them"""
    from dusty_gan_amd import models
    from dusty_gan_amd.utils.config import load_config
    assert set(models.GENERATORS) == {"dcgan_eqlr"} == set(models.DISCRIMINATORS) and set(models.MASKERS) == {"none", "dusty1", "dusty2"}
    cfg = load_config(["model=dusty1_dcgan_eqlr", "dataset=synthetic", "dataset.shape=[32,64]"])
    cfg.model.gen.shape = cfg.model.dis.shape = cfg.dataset.shape
    G, D = models.define_G(cfg), models.define_D(cfg)
    assert type(G).__name__ == "DUSty1" and type(G.backbone).__name__ == "Generator" and type(D).__name__ == "Discriminator"
    for bad in ("dusty3/dcgan_eqlr", "none/stylegan"):
        cfg.model.gen.arch = bad
        with pytest.raises(NotImplementedError):
            models.define_G(cfg)
    cfg.model.dis.arch = "patchgan"
    with pytest.raises(NotImplementedError):
        models.define_D(cfg)


def test_algorithmic_work_model_matches_the_survey():
    """engine.conv_algorithmic / wgrad_algorithmic (what bench.py's roofline divides by): FLOPs of SURVEY.md section 8a/8d per
    sample at 64x1024 - 536 870 912 MAC for each fat layer pass - and byte counts from the operand shapes: MODE_S2 reads the
    FINE grid and writes the coarse one, MODE_UP the reverse (round 2 counted MODE_UP's input at the fine pixel count and one
    weight-gradient family ran "above" the HBM peak), EPI_MASK also reads the saved activation, 16 weight taps either way."""
    from dusty_gan_amd import _lib as L
    from dusty_gan_amd import engine as E
    # Down2 forward: 64 -> 128 channels, coarse 16 x 256
    fl, nb = E.conv_algorithmic(L.MODE_S2, 1, 16, 256, 64, 128, 2, 2, 2, False)
    assert fl == 2 * 536870912 and nb == 4 * 4096 * 64 * 2 + 4096 * 128 * 2 + 16 * 128 * 64 * 2
    # Up3 forward: 128 -> 64 channels, coarse 16 x 256 -> fine 32 x 512
    fl, nb = E.conv_algorithmic(L.MODE_UP, 1, 16, 256, 128, 64, 2, 2, 2, False)
    assert fl == 2 * 536870912 and nb == 4096 * 128 * 2 + 4 * 4096 * 64 * 2 + 16 * 64 * 128 * 2
    # Down2 backward-data (MODE_UP, mask epilogue): the mask source doubles the output-side bytes
    _, nb_m = E.conv_algorithmic(L.MODE_UP, 1, 16, 256, 128, 64, 2, 2, 2, True)
    assert nb_m - nb == 4 * 4096 * 64 * 2
    # Proj: B rows
    fl, nb = E.conv_algorithmic(L.MODE_GEMM, 32, 1, 1, 512, 131072, 2, 2, 2, False)
    assert fl == 2 * 32 * 67108864 and nb == 32 * 512 * 2 + 32 * 131072 * 2 + 131072 * 512 * 2
    # weight gradients: Down (input fine, gradient coarse) and Up (input coarse, gradient fine), dW in fp32 once
    fl, nb = E.wgrad_algorithmic(0, 1, 16, 256, 64, 128, 2, 2)
    assert fl == 2 * 536870912 and nb == 4096 * (4 * 64 + 128) * 2 + 16 * 64 * 128 * 4
    fl, nb = E.wgrad_algorithmic(1, 1, 16, 256, 128, 64, 2, 2)
    assert fl == 2 * 536870912 and nb == 4096 * (128 + 4 * 64) * 2 + 16 * 128 * 64 * 4
    # bench.py's whole-step model at config 2: executed FLOPs 3 F_G + 10 F_D per sample
    import bench
    sf, sb = bench.step_model([64, 1024], "none", 1.0, 32, 2, 69863424, 2886656)
    f_g = 2 * (67108864 + 3 * 536870912 + 16777216)
    f_d = 2 * (393216 + 33554432 + 3 * 536870912 + 131072)
    assert sf == 32 * (3 * f_g + 10 * f_d) and 5.5e9 < sb < 6.2e9



def test_wgrad_workspace_never_reduces_two_items_into_one_destination_per_launch(monkeypatch):
    """engine.WgradWorkspace.flush: at most 16 layers per dg_wgrad_reduce launch and never two items with the same dW in one
    launch (micro-batches and the path-length terms queue several partial sets for one gradient: their blocks would
    read-modify-write it concurrently) - the repeats follow in later launches, in queue order."""
    from dusty_gan_amd import _lib as L
    from dusty_gan_amd import engine as E
    launches = []

    class FakeLib:
        def dg_wgrad_reduce(self, arr, n, stream):
            launches.append([(arr[i].ws, arr[i].dw, arr[i].numel, arr[i].splits, arr[i].accumulate) for i in range(n)])
            return 0
    monkeypatch.setattr(L, "lib", lambda: FakeLib())
    monkeypatch.setattr(L, "stream_ptr", lambda: None)
    ws = E.WgradWorkspace()
    dsts = [10, 20, 30, 10, 40, 20, 10] + list(range(50, 250, 10))          # the last run of 20 distinct ones: 16 + 4
    items = [(1000 + i, dw, 64, 4, 1) for i, dw in enumerate(dsts)]
    for it in items:
        ws.add(*it)
    ws.flush()
    assert ws.items == [] and ws.pos == 0
    assert [it for chunk in launches for it in chunk] == items                 # everything, in order
    for chunk in launches:
        assert len(chunk) <= 16 and len({it[1] for it in chunk}) == len(chunk)  # no destination twice in a launch
    assert [len(c) for c in launches] == [3, 3, 16, 5]                          # [10 20 30] [10 40 20] [10 50 ... 190] [200 ... 240]


def test_counters_ride_and_mid_step_flush_without_gpu(monkeypatch):
    """_lib.Counters: a riding snapshot stays queued through a mid-step flush (a consumer syncing its counter) and leaves
    with `take_for_ride`; a final flush takes everything; a snapshot without its counter's advance is refused."""
    from dusty_gan_amd import _lib as L
    calls = []

    class T:                       # stands in for a device counter tensor
        def __init__(self, p):
            self.p = p

        def data_ptr(self):
            return self.p

    class FakeLib:
        def dg_counter_add_multi(self, ptrs, dels, k, stream):
            calls.append(("add", [ptrs[i] for i in range(k)], [dels[i] for i in range(k)]))
            return 0

        def dg_counter_add_multi_snap(self, ptrs, dels, k, idx, src, n, ring, slots, stream):
            calls.append(("snap", [ptrs[i] for i in range(k)], [dels[i] for i in range(k)], idx, src, n, ring, slots))
            return 0
    monkeypatch.setattr(L, "lib", lambda: FakeLib())
    monkeypatch.setattr(L, "stream_ptr", lambda: None)
    C = L.Counters
    C.pending.clear(); C.snap = None; C.ride = False
    a, b, s = T(0x100), T(0x200), T(0x300)
    C.add(a, 3); C.add(b, 1); C.add(s, 1)
    C.snapshot(s, 0xAAA0, 8, 0xBBB0, 64)
    C.ride = True
    C.flush_if(a)                                    # mid-step: a and b go out, the snapshot and its counter stay
    assert calls == [("add", [0x100, 0x200], [3, 1])] and C.ride and C.snap is not None and list(C.pending) == [0x300]
    C.add(a, 2)
    ride = C.take_for_ride()
    assert ride is not None and not C.ride and C.snap is None and not C.pending
    ptrs, dels, k, idx, src, n, ring, slots = ride
    assert k == 2 and sorted(ptrs[i] for i in range(k)) == [0x100, 0x300] and ptrs[idx] == 0x300
    assert (src, n, ring, slots) == (0xAAA0, 8, 0xBBB0, 64)
    assert C.take_for_ride() is None                 # nothing flagged any more
    C.add(s, 1); C.snapshot(s, 0xAAA0, 8, 0xBBB0, 64)
    C.flush()                                        # a final flush files the snapshot itself
    assert calls[-1][0] == "snap" and calls[-1][3] == 0 and not C.pending and C.snap is None
    C.snapshot(s, 0xAAA0, 8, 0xBBB0, 64)             # no advance queued for its counter
    C.add(a, 1)
    with pytest.raises(RuntimeError):
        C.flush()
    C.pending.clear(); C.snap = None; C.ride = False


def test_counter_queues_are_per_owner():
    """`_lib.Counters` (queued Philox / Adam / pool counter advances) reads and writes the CURRENT owner's queue: what one
    trainer queues is invisible to the flush / the riding launch of another trainer's step (round-3 review, item 10)"""
    import torch
    from dusty_gan_amd import _lib as L
    a, b = L.CounterQueue(), L.CounterQueue()
    t = torch.zeros(1, dtype=torch.int64)
    with L.Counters.bind(a):
        L.Counters.add(t, 3)
        L.Counters.ride = True
        with L.Counters.bind(b):
            assert not L.Counters.pending and not L.Counters.ride and L.Counters.snap is None
            L.Counters.add(t, 1)
        assert L.Counters.pending[t.data_ptr()][1] == 3 and L.Counters.ride
    assert t.data_ptr() not in L.Counters.pending and not L.Counters.ride      # the process-wide default queue: untouched
    assert b.pending[t.data_ptr()][1] == 1 and a.ride and not b.ride
    a.pending.clear(); b.pending.clear()


def test_two_word_fixed_point_split_is_exact():
    """csrc/common.h dg_fix2 (round 6: order-independent bias-gradient sums of the kernels off the timed path), restated in
    numpy: v = hi 2^-20 + lo 2^-60 with hi = rint(v 2^20) and lo = rint((v - hi 2^-20) 2^60).  Claims checked on a wide random
    sample of floats (magnitudes 2^-40 ... 2^41 (all below the 4e12 cut-off), both signs, plus edge values): (1) the residual v - hi 2^-20 that the kernel
    forms with ONE float fma is exactly representable in float32 (so the float op loses nothing); (2) for |v| >= 2^-36 the
    pair reproduces v exactly, below that to 2^-60; (3) integer sums of the pairs do not depend on the order of the terms and
    equal the exact sum of the values to 2^-60 per term (checked against Python integers / Fraction)."""
    from fractions import Fraction
    rng = np.random.default_rng(7)
    mant = rng.uniform(1.0, 2.0, 20000).astype(np.float32)
    expo = rng.integers(-40, 41, 20000)
    v = (np.ldexp(mant, expo) * rng.choice([-1.0, 1.0], 20000)).astype(np.float32)
    v = np.concatenate([v, np.float32([0.0, 1.0, -1.0, 0.5 / 1048576, 0.50001 / 1048576, 1.5 / 1048576, 15.999999, 16.0,
                                       2.0 ** -36, 2.0 ** -37, 3.9e12, 1e-30])])
    assert np.all(np.abs(v) < 4.0e12)
    t = np.rint(v * np.float32(1048576.0)).astype(np.float32)          # exact scaling by a power of two, integer-valued
    hi = t.astype(np.int64)
    r64 = v.astype(np.float64) - t.astype(np.float64) / 1048576.0       # what fma(-t, 2^-20, v) computes before rounding
    assert np.array_equal(r64.astype(np.float32).astype(np.float64), r64), "the residual is not a float: the fma would round"
    assert np.all(np.abs(r64) <= 2.0 ** -21)
    lo = np.rint(r64 * 2.0 ** 60).astype(np.int64)
    back = [Fraction(int(h), 2 ** 20) + Fraction(int(l), 2 ** 60) for h, l in zip(hi, lo)]
    exact = [Fraction(float(x)) for x in v]
    for x, b, e in zip(v, back, exact):
        if abs(float(x)) >= 2.0 ** -36:
            assert b == e, float(x)
        else:
            assert abs(b - e) <= Fraction(1, 2 ** 61), float(x)
    # order independence + value of the sum (what the last workgroup converts: one rounding of the exact integer pair)
    idx = rng.permutation(len(v))
    assert int(hi.sum()) == int(hi[idx].sum()) and sum(int(x) for x in lo) == sum(int(x) for x in lo[idx])
    big = np.abs(v) >= 2.0 ** -36
    tot = Fraction(sum(int(x) for x in hi[big]), 2 ** 20) + Fraction(sum(int(x) for x in lo[big]), 2 ** 60)
    assert tot == sum((Fraction(float(x)) for x in v[big]), Fraction(0))
