"""ctypes binding of libdustygan_hip.so (the C ABI declared in include/dusty_gan_hip.h).

There is deliberately NO fallback: if the HIP library is missing, importing the product path raises.
PyTorch is used only as the owner of device memory and of the HIP stream.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libdustygan_hip.so")
if os.environ.get("DUSTY_GAN_LIB_DIAG"):  # kernel-development aid: `make -C csrc diag` (ablation switches, stamps);
    _d = os.environ["DUSTY_GAN_LIB_DIAG"]  # "1" or the suffix of a renamed copy (libdustygan_hip_diag<suffix>.so)
    LIB_PATH = os.path.join(CSRC, "libdustygan_hip_diag%s.so" % ("" if _d == "1" else _d))

DG_OK, DG_EINVAL, DG_EUNSUPPORTED, DG_EHIP = 0, 1, 2, 3
DG_F32, DG_BF16 = 0, 1
DG_BF16X2 = 2   # split-bf16 pairs (hi | lo per 64 channels), 4 bytes per element: include/dusty_gan_hip.h
MODE_S2, MODE_UP, MODE_GEMM = 0, 1, 2
EPI_LINEAR, EPI_LRELU, EPI_MASK = 0, 1, 2
DG_FORCE_FP32X3 = 0x100   # flag bit of the `force` arguments (include/dusty_gan_hip.h)
POLICY_BITS = {"brightness": 1, "saturation": 2, "contrast": 4, "translation": 8, "cutout": 16}

_ERR = {1: "DG_EINVAL (bad argument)", 2: "DG_EUNSUPPORTED (shape not supported by the requested kernel)",
        3: "DG_EHIP (HIP runtime error)"}


class DgError(RuntimeError):
    pass


class DgConv(C.Structure):
    _fields_ = [
        ("mode", C.c_int), ("adj", C.c_int), ("ring", C.c_int),
        ("B", C.c_int), ("Hc", C.c_int), ("Wc", C.c_int),
        ("K", C.c_int), ("N", C.c_int),
        ("in_", C.c_void_p), ("in_sb", C.c_long), ("in_sp", C.c_long), ("in_sk", C.c_long),
        ("out", C.c_void_p), ("out_sb", C.c_long), ("out_sp", C.c_long), ("out_sn", C.c_long),
        ("w", C.c_void_p), ("w_st", C.c_long), ("w_sk", C.c_long), ("w_sn", C.c_long),
        ("scale", C.c_float), ("epi", C.c_int),
        ("bias", C.c_void_p), ("bias_mod", C.c_int),
        ("aux", C.c_void_p), ("dbias", C.c_void_p), ("rowscale", C.c_void_p),
        ("in_dtype", C.c_int), ("out_dtype", C.c_int), ("w_dtype", C.c_int),
        ("nscale", C.c_void_p),
        ("up_frag", C.c_void_p),
        ("dbias_ws", C.c_void_p),
        ("dbias_part", C.c_void_p),
        ("mask_out", C.c_void_p),
        ("mask_in", C.c_void_p),
        ("tanh_sum_parts", C.c_void_p),
    ]


UP_FRAG_BYTES = 3 * 18 * 1024
DBIAS_WS_FLOATS = 32 * 1024


class DgUpFrag(C.Structure):
    _fields_ = [("off", C.c_longlong), ("frag", C.c_void_p), ("m_st", C.c_longlong), ("m_sn", C.c_longlong),
                ("m_sk", C.c_longlong), ("N", C.c_int), ("Hc", C.c_int), ("adj", C.c_int)]


class DgDraw(C.Structure):
    _fields_ = [("kind", C.c_int), ("fill_kind", C.c_int), ("seed", C.c_uint64), ("stream_id", C.c_uint64),
                ("offset_dev", C.c_void_p), ("base", C.c_ulonglong), ("lo", C.c_float), ("hi", C.c_float),
                ("eps", C.c_float), ("ilo", C.c_int), ("ihi", C.c_int), ("n", C.c_long), ("out", C.c_void_p),
                ("out_bf16", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("uf", C.c_void_p),
                ("qi", C.c_void_p)]


class DgWgrad(C.Structure):
    _fields_ = [
        ("wmode", C.c_int), ("ring", C.c_int),
        ("B", C.c_int), ("Hc", C.c_int), ("Wc", C.c_int),
        ("Ci", C.c_int), ("Co", C.c_int),
        ("a", C.c_void_p), ("a_sb", C.c_long), ("a_sp", C.c_long), ("a_sc", C.c_long),
        ("g", C.c_void_p), ("g_sb", C.c_long), ("g_sp", C.c_long), ("g_sc", C.c_long),
        ("dw", C.c_void_p), ("scale", C.c_float), ("rowscale", C.c_void_p),
        ("a_dtype", C.c_int), ("g_dtype", C.c_int),
        ("ws", C.c_void_p), ("g_mod", C.c_int),
    ]


XSUM_PARTS = 8   # DG_XSUM_PARTS of include/dusty_gan_hip.h


class DgFetch(C.Structure):
    _fields_ = [("pol", C.c_void_p), ("mask", C.c_void_p), ("pool_ctr", C.c_void_p), ("npool", C.c_int),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("drop_const", C.c_float), ("B", C.c_int), ("HW", C.c_long),
                ("out", C.c_void_p), ("parts", C.c_void_p)]


class DgAugSet(C.Structure):
    _fields_ = [("x", C.c_void_p), ("xsum", C.c_void_p), ("xsum_parts", C.c_int), ("u_b", C.c_void_p), ("u_c", C.c_void_p), ("t_h", C.c_void_p),
                ("t_w", C.c_void_p), ("o_x", C.c_void_p), ("o_y", C.c_void_p)]


class DgOptSeg(C.Structure):
    _fields_ = [("off", C.c_longlong), ("numel", C.c_longlong), ("part", C.c_void_p), ("splits", C.c_int), ("accumulate", C.c_int),
                ("kind", C.c_int), ("ci", C.c_int), ("co", C.c_int), ("shadow_t", C.c_void_p), ("ws_src", C.c_void_p),
                ("ws_coef", C.c_void_p), ("ws_stride", C.c_longlong), ("ws_n", C.c_int), ("ws_bf16", C.c_int),
                ("ws_scale", C.c_float), ("first_block", C.c_int)]


OPT_MAX_SEG = 20   # DG_OPT_MAX_SEG


class DgWgradPlan(C.Structure):
    _fields_ = [("variant", C.c_int), ("splits", C.c_int), ("ws_floats", C.c_long), ("tap_pairs", C.c_int)]


class DgWgradReduce(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dw", C.c_void_p), ("numel", C.c_long), ("splits", C.c_int), ("accumulate", C.c_int)]


class DgConvPlan(C.Structure):
    _fields_ = [("family", C.c_int), ("bm", C.c_int), ("bn", C.c_int), ("tiles", C.c_int), ("workgroups", C.c_int),
                ("tiles_per_wg", C.c_int), ("thin_mfma", C.c_int), ("mask_bits", C.c_int), ("dbias_rows", C.c_int),
                ("sum_parts", C.c_int)]


_P, _I, _L, _F, _D, _U64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_uint64

# name -> argtypes (all return int except dg_version); mirrors include/dusty_gan_hip.h one to one
PROTOTYPES = {
    "dg_conv": [C.POINTER(DgConv), _I, _P],
    "dg_conv_ex": [C.POINTER(DgConv), _I, _I, _P],
    "dg_conv_plan": [C.POINTER(DgConv), _I, _I, C.POINTER(DgConvPlan)],
    "dg_conv_mfma_supported": [C.POINTER(DgConv)],
    "dg_conv_kernel_choice": [C.POINTER(DgConv)],
    "dg_wgrad": [C.POINTER(DgWgrad), _I, _I, _P],
    "dg_wgrad_plan": [C.POINTER(DgWgrad), _I, _I, C.POINTER(DgWgradPlan)],
    "dg_wgrad_group": [C.POINTER(DgWgrad), _I, _I, _I, _P],
    "dg_wgrad_group_plan": [C.POINTER(DgWgrad), _I, _I, _I, C.POINTER(DgWgradPlan)],
    "dg_wgrad_reduce": [C.POINTER(DgWgradReduce), _I, _P],
    "dg_wgrad_mfma_supported": [C.POINTER(DgWgrad)],
    "dg_wgrad_kernel_choice": [C.POINTER(DgWgrad)],
    "dg_wgrad_kernel_variant": [C.POINTER(DgWgrad), _I],
    "dg_wgrad_has_sample_map": [C.POINTER(DgWgrad), _I],
    "dg_blur_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dg_blur_fwd_mean": [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    "dg_blur_bwd": [_P, _I, _P, _I, _I, _I, _I, _P],
    "dg_blur_bwd_r1": [_P, _I, _P, _F, _P, _I, _I, _I, _I, _P],
    "dg_blur_r1_tangent": [_P, _I, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "dg_final_fwd": [_P, _I, _P, _P, _F, _I, _L, _P, _P],
    "dg_final_fwd_acc": [_P, _I, _P, _P, _F, _I, _L, _P, _P],
    "dg_final_bwd_data": [_P, _I, _P, _P, _P, _F, _I, _L, _I, _P, _P, _P],
    "dg_batch_wsum": [_P, _I, _P, _F, _I, _L, _P, _P],
    "dg_head_post_fwd": [_P, _P, _P, _I, _I, _F, _F, _I, _L, _P, _P, _P],
    "dg_head_post_fwd_sum": [_P, _P, _P, _I, _I, _F, _F, _I, _L, _P, _P, _P, _P],
    "dg_head_post_bwd": [_P, _P, _P, _P, _P, _I, _F, _F, _I, _L, _F, _F, _P, _P, _P, _I, _P, _P],
    "dg_logistic_noise": [_P, _P, _F, _L, _P, _P],
    "dg_diffaug_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "dg_diffaug_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "dg_diffaug_fwd_acc": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "dg_diffaug_fwd_pre": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "dg_diffaug_bwd_acc": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "dg_diffaug_blur_fwd": [C.POINTER(DgAugSet), _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "dg_blur_bwd_augsum": [_P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "dg_diffaug_bwd_pre": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "dg_head_post_bwd_aug": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _F, _F, _I, _I, _I, _F, _F, _P, _P, _P,
                             _I, _P, _P],
    "dg_fetch_reals_pool_sum": [_P, _P, _P, _I, _F, _F, _F, _I, _L, _P, _P, _P],
    "dg_nsgan_d": [_P, _P, _I, _F, _P, _P, _P, _P],
    "dg_nsgan_g": [_P, _I, _F, _P, _P, _P],
    "dg_nsgan_d_step": [_P, _P, _I, _F, _P, _P, _P, _P, _P, _P],
    "dg_nsgan_g_step": [_P, _I, _F, _P, _P, _P],
    "dg_gan_d_step": [_I, _F, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P],
    "dg_gan_g_step": [_I, _P, _P, _I, _F, _P, _P, _P],
    "dg_final_gan_bwd": [_I, _I, _F, _P, _P, _I, _F, _I, _P, _P, _P, _P, _P, _P, _I, _P, _F, _L, _I, _P, _P, _P, _P, _P],
    "dg_det_arena": [_P, _L, _P],
    "dg_mean_acc": [_P, _I, _P, _P],
    "dg_fetch_reals": [_P, _P, _F, _F, _F, _L, _P, _P],
    "dg_fetch_reals_sum": [_P, _P, _F, _F, _F, _I, _L, _P, _P, _P],
    "dg_pl_penalty": [_P, _I, _I, _F, _P, _P, _P, _P],
    "dg_head_post_bwd2": [_P, _P, _P, _P, _P, _P, _I, _F, _F, _I, _L, _F, _F, _P, _P, _P, _I, _P],
    "dg_scan_to_polar": [_P, _I, _I, _I, _I, _I, _I, _P, _D, _D, _F, _P, _P, _P, _P, _P],
    "dg_inv_to_xyz": [_P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P],
    "dg_unit_map": [_P, _L, _I, _P, _P],
    "dg_normals": [_P, _I, _I, _I, _I, _P, _P],
    "dg_fps": [_P, _I, _I, _I, _P, _P, _P, _P],
    "dg_chamfer_dir": [_P, _I, _I, _P, _I, _I, _P, _P],
    "dg_emd": [_P, _I, _I, _P, _I, _I, _I, _P, _P],
    "dg_grid_vote": [_P, _L, _P, _I, _P, _P],
    "dg_jsd": [_P, _P, _I, _P, _P],
    "dg_pyr_down": [_P, _L, _I, _I, _P, _P],
    "dg_pyr_up_sub": [_P, _P, _L, _I, _I, _P],
    "dg_extract_patches": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    "dg_sample_sum": [_P, _I, _L, _I, _P, _P],
    "dg_sample_sum_acc": [_P, _I, _L, _I, _P, _P],
    "dg_scale": [_P, _F, _L, _P, _P],
    "dg_zero": [_P, _L, _P],
    "dg_zero_multi": [_P, _P, _I, _P],
    "dg_adam_ema_step": [_P, _P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    "dg_cast": [_P, _P, _I, _L, _P],
    "dg_uncast": [_P, _I, _P, _L, _P],
    "dg_cast_x2_multi": [_P, _P, _P, _I, _P],
    "dg_transpose_shadow": [_P, _P, _I, _I, _I, _P],
    "dg_transpose_shadow_multi": [_P, _P, _I, _I, _I, _P],
    "dg_transpose_shadow_multi_frags": [_P, _P, _I, _I, _I, C.POINTER(DgUpFrag), _I, _P],
    "dg_transpose_shadow_multi_tail": [_P, _P, _I, _I, _I, C.POINTER(DgUpFrag), _I, _P, _P, _I, _I, _P, _I, _P, _I, _P],
    "dg_philox_bits": [_U64, _U64, _U64, _L, _P, _P],
    "dg_philox_fill": [_U64, _U64, _U64, _I, _F, _F, _I, _I, _L, _P, _P],
    "dg_aug_draw": [_U64, _U64, _U64, _I, _I, _I, _P, _P, _P],
    "dg_counter_add": [_P, _U64, _P],
    "dg_step_prologue": [_P, _P, _I, C.POINTER(DgDraw), _I, _P],
    "dg_step_prologue_fetch": [_P, _P, _I, C.POINTER(DgDraw), _I, C.POINTER(DgFetch), _P],
    "dg_counter_add_multi": [_P, _P, _I, _P],
    "dg_counter_add_multi_snap": [_P, _P, _I, _I, _P, _I, _P, _I, _P],
    "dg_philox_fill_dev": [_U64, _U64, _P, _I, _F, _F, _I, _I, _L, _P, _P],
    "dg_aug_draw_dev": [_U64, _U64, _P, _I, _I, _I, _P, _P, _P],
    "dg_philox_logistic_dev": [_U64, _U64, _P, _F, _L, _P, _P],
    "dg_adam_ema_step_dev": [_P, _P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _F, _P, _F, _P],
    "dg_adam_fused": [_P, _P, _P, _P, _P, _I, C.POINTER(DgOptSeg), _I, _F, _F, _F, _F, _P, _F, _P],
    "dg_adam_proj_fused": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _L, _I, _F, _F, _F, _F, _F, _P, _F, _P],
}

_lib = None


def build(verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise DgError("building libdustygan_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout)
    return LIB_PATH


def build_diag(bits=8):
    """The diagnostic twin of the library (libdustygan_hip_diag.so: the ping-pong conv with cycle stamps; `make diag`).  Never
    loaded by the product path, the tests or the timed region: scripts/conv_clock.py reads the clock the chip holds inside
    the conv kernel from it (bench.py `roofline.clock_ghz`).  Returns its path, or None when the build fails (a missing
    diagnostic library costs one key of the bench line, never the line)."""
    res = subprocess.run(["make", "-C", CSRC, "diag", f"DIAGBITS={int(bits)}"], capture_output=True, text=True)
    path = os.path.join(CSRC, "libdustygan_hip_diag.so")
    return path if res.returncode == 0 and os.path.exists(path) else None


def lib():
    """Load (once) and return the ctypes handle.  Raises DgError when the library is absent: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DgError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C dusty_gan_amd/csrc`). "
            "There is no CPU or PyTorch fallback for the hot path.")
    h = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(h, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    h.dg_version.restype = C.c_char_p
    h.dg_version.argtypes = []
    _lib = h
    return h


def check(rc, what=""):
    if rc != DG_OK:
        raise DgError(f"{what or 'libdustygan_hip call'} failed: {_ERR.get(rc, rc)}")


def dtype_code(t):
    import torch
    if t == torch.float32:
        return DG_F32
    if t == torch.bfloat16:
        return DG_BF16
    raise DgError(f"unsupported dtype {t}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def zero_(t):
    """t.zero_() as a kernel of this library (fp32 tensors; see dg_zero in include/dusty_gan_hip.h)"""
    import torch
    assert t.dtype == torch.float32 and t.is_contiguous()
    check(lib().dg_zero(t.data_ptr(), t.numel(), stream_ptr()), "dg_zero")
    return t


def zero_multi(tensors):
    """zero-fill up to 4 contiguous fp32 tensors (numel % 4 == 0) in one launch"""
    import ctypes as C
    import torch
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    for i in range(0, len(ts), 4):
        chunk = ts[i:i + 4]
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0 for t in chunk)
        ptrs = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
        cnts = (C.c_long * len(chunk))(*[t.numel() for t in chunk])
        check(lib().dg_zero_multi(ptrs, cnts, len(chunk), stream_ptr()), "dg_zero_multi")


def step_prologue(tensors, draws, fetch=None):
    """zero-fill up to 4 contiguous fp32 tensors and run up to 6 DgDraw jobs in ONE launch (dg_step_prologue); fetch: a DgFetch -
    fetch_reals of the step's batch as more workgroups of the same launch (dg_step_prologue_fetch)"""
    import ctypes as C
    import torch
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    assert len(ts) <= 4 and len(draws) <= 6
    assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0 for t in ts)
    ptrs = (C.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
    cnts = (C.c_long * max(len(ts), 1))(*[t.numel() for t in ts])
    arr = (DgDraw * max(len(draws), 1))(*draws)
    if fetch is not None:
        check(lib().dg_step_prologue_fetch(ptrs, cnts, len(ts), arr, len(draws), C.byref(fetch), stream_ptr()),
              "dg_step_prologue_fetch")
        return
    check(lib().dg_step_prologue(ptrs, cnts, len(ts), arr, len(draws), stream_ptr()), "dg_step_prologue")


class AccArena:
    """Small fp32 accumulators that must start at zero (per-sample sums, logits, the logged scalars), carved from ONE
    buffer per DEVICE that one kernel zero-fills at the start of a training step - instead of one ~5 us zero-fill node per
    accumulator (11 per step).  `take(n)` hands out a slice that has been zero since the last `begin()` and is handed
    out once; None when the arena is not in use or exhausted (the caller then uses the self-zeroing entry point).
    Round 6: one arena (+ fixed-point shadow) per device, allocated once and kept for the life of the process: a captured
    hipGraph holds raw pointers into it, and the library's per-device registration (dg_det_arena) points at it - a second
    trainer on another device gets that device's own arena instead of re-allocating (and silently un-registering) the first
    one's (round-5 review, item 7).  Trainers that share a device share its arena: their steps are ordered on the device, and
    every step opens its epoch with a zero-fill."""
    SIZE = 8192
    buf, pos, epoch = None, 0, 0               # the arena of the device `begin` was last called for; epoch counts begin() calls
    shadow = None                              # fixed-point shadow words of the slots (deterministic cross-block sums)
    _by_dev = {}                               # device index -> (buf, shadow)

    @staticmethod
    def _same(dev, want):
        import torch
        want = torch.device(want)
        return dev.type == want.type and (want.index is None or dev.index == want.index)

    @classmethod
    def _for_device(cls, device):
        import torch
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ent = cls._by_dev.get(idx)
        if ent is None:
            dev = torch.device("cuda", idx)
            buf = torch.empty(cls.SIZE, dtype=torch.float32, device=dev)
            shadow = None
            # bit-reproducible sums into the arena's slots (csrc/common.h dg_acc_add): a shadow of 128 bytes per float, zero at
            # rest, registered with the library for this device; DUSTY_GAN_DETERMINISTIC=0 keeps the float atomics
            if os.environ.get("DUSTY_GAN_DETERMINISTIC", "1") != "0":
                shadow = torch.zeros(cls.SIZE * 32, dtype=torch.int32, device=dev)   # DG_DET_STRIDE = 128 bytes per slot
                torch.cuda.synchronize(dev)
                with torch.cuda.device(dev):
                    check(lib().dg_det_arena(ptr(buf), cls.SIZE, ptr(shadow)), "dg_det_arena")
            ent = cls._by_dev[idx] = (buf, shadow)
        return ent

    @classmethod
    def begin(cls, device, also=(), draws=(), fetch=None):
        """open a new epoch: the arena - and the fp32 buffers in `also` (the step's gradient buffers) - zero-filled by
        one launch, which also runs the DgDraw jobs in `draws` (the step's parameter draws) and, with `fetch` (a DgFetch),
        fetch_reals of the step's batch"""
        cls.buf, cls.shadow = cls._for_device(device)
        if draws or fetch is not None:
            step_prologue([cls.buf] + list(also), list(draws), fetch)
        else:
            zero_multi([cls.buf] + list(also))
        cls.pos = 0
        cls.epoch += 1

    @classmethod
    def take(cls, n, device=None):
        import torch
        if cls.buf is None or (device is not None and not cls._same(cls.buf.device, device)):
            return None
        a = (cls.pos + 15) // 16 * 16          # 64-byte slices
        if a + n > cls.SIZE:
            return None
        cls.pos = a + n
        return cls.buf[a:a + n]


class CounterQueue:
    """The queued counter advances of one owner (a Trainer; the process-wide default for everything else):
    pending  {counter pointer: [tensor, delta]}
    snap     (counter tensor, src pointer, n, ring pointer, ring slots): filed by the next flush (dg_counter_add_multi_snap)
    ride     the next ParamStore.refresh_transposed carries the pending advances (and the snapshot) in its launch"""
    _all = None   # weak set of every queue (flush_if looks for a counter's queued advance in all of them)

    def __init__(self):
        import weakref
        self.pending, self.snap, self.ride = {}, None, False
        if CounterQueue._all is None:
            CounterQueue._all = weakref.WeakSet()
        CounterQueue._all.add(self)


class _CountersMeta(type):
    """`Counters.pending / .snap / .ride` read and write the CURRENT queue (`Counters.bind`), so the classmethods below and
    their callers are written as for one queue"""
    _STATE = ("pending", "snap", "ride")

    def __getattr__(cls, name):
        if name in _CountersMeta._STATE:
            return getattr(cls._stack[-1], name)
        raise AttributeError(name)

    def __setattr__(cls, name, value):
        if name in _CountersMeta._STATE:
            setattr(cls._stack[-1], name, value)
        else:
            type.__setattr__(cls, name, value)


class Counters(metaclass=_CountersMeta):
    """Device counters (Philox offsets, Adam step counts) are advanced behind their consumers.  The advances of a step are
    queued and applied by ONE kernel at the end of the step (`flush`); anything that reads a counter with a queued
    advance - the next draw from the same generator, `Philox.offset`, the next optimizer step - flushes first.
    The queue is per owner (round-3 review: it was one process-global): a Trainer binds its own `CounterQueue` around its
    public entry points, so the flush / the riding launch at the end of ITS step - possibly being captured into ITS graph -
    never carries advances another trainer of the process has queued."""
    _stack = [CounterQueue()]   # [-1] = the current queue; [0] = the process-wide default

    @classmethod
    def current(cls):
        return cls._stack[-1]

    @classmethod
    def bind(cls, queue):
        """context manager: `queue` is the current queue inside the block"""
        import contextlib

        @contextlib.contextmanager
        def _bound():
            cls._stack.append(queue)
            try:
                yield queue
            finally:
                cls._stack.pop()
        return _bound()

    @classmethod
    def snapshot(cls, ctr, src_ptr, n, ring_ptr, ring):
        """with the next flush - which must advance `ctr` - src[0..n) goes to slot (old ctr) % ring of the ring buffer"""
        cls.snap = (ctr, int(src_ptr), int(n), int(ring_ptr), int(ring))

    @classmethod
    def add(cls, t, n):
        e = cls.pending.setdefault(t.data_ptr(), [t, 0])
        e[1] += int(n)

    @classmethod
    def flush_if(cls, t):
        if t is None:
            return
        if t.data_ptr() in cls.pending:
            cls.flush(mid_step=True)
            return
        for q in list(CounterQueue._all or ()):   # (a counter read outside its owner's entry points)
            if q is not cls._stack[-1] and t.data_ptr() in q.pending:
                with cls.bind(q):
                    cls.flush(mid_step=True)

    @classmethod
    def take_for_ride(cls):
        """(counter pointers, deltas, k, snap_idx, src, n, ring_ptr, ring) of everything pending, for a launch that applies
        them itself (dg_transpose_shadow_multi_tail); None when nothing rides (no flag, nothing pending, more than 8)"""
        import ctypes as C
        if not cls.ride or not cls.pending or len(cls.pending) > 8:
            return None
        cls.ride = False
        items = list(cls.pending.values())
        snap, cls.snap = cls.snap, None
        at = [j for j, (t, _) in enumerate(items) if snap is not None and t.data_ptr() == snap[0].data_ptr()]
        if snap is not None and not at:
            raise RuntimeError("Counters.snapshot: the snapshot's counter has no queued advance")
        cls.pending.clear()
        ptrs = (C.c_void_p * len(items))(*[t.data_ptr() for t, _ in items])
        dels = (C.c_uint64 * len(items))(*[d for _, d in items])
        if snap is None:
            return ptrs, dels, len(items), -1, None, 0, None, 1
        return ptrs, dels, len(items), at[0], snap[1], snap[2], snap[3], snap[4]

    @classmethod
    def flush(cls, mid_step=False):
        """apply everything pending.  mid_step (a consumer needs its counter current while a step with a riding snapshot is
        still running): the snapshot and its counter stay queued - the scalars are not complete yet."""
        import ctypes as C
        held = None
        if mid_step and cls.ride and cls.snap is not None:
            held = cls.pending.pop(cls.snap[0].data_ptr(), None)
        else:
            cls.ride = False
        items = list(cls.pending.values())
        cls.pending.clear()
        if held is not None:
            cls.pending[cls.snap[0].data_ptr()] = held
            snap = None
        else:
            snap, cls.snap = cls.snap, None
        if snap is not None and snap[0].data_ptr() not in [t.data_ptr() for t, _ in items]:
            raise RuntimeError("Counters.snapshot: the snapshot's counter has no queued advance")
        for i in range(0, len(items), 8):
            chunk = items[i:i + 8]
            ptrs = (C.c_void_p * len(chunk))(*[t.data_ptr() for t, _ in chunk])
            dels = (C.c_uint64 * len(chunk))(*[d for _, d in chunk])
            at = [j for j, (t, _) in enumerate(chunk) if snap is not None and t.data_ptr() == snap[0].data_ptr()]
            if at:
                check(lib().dg_counter_add_multi_snap(ptrs, dels, len(chunk), at[0], snap[1], snap[2], snap[3], snap[4],
                                                      stream_ptr()), "dg_counter_add_multi_snap")
            else:
                check(lib().dg_counter_add_multi(ptrs, dels, len(chunk), stream_ptr()), "dg_counter_add_multi")


def tag_sums(t, sums, parts=1):
    """remember on an image tensor the per-sample sums its producer kernel accumulated (DiffAugment's contrast reads them
    instead of making its own pass) - valid while the arena epoch lasts and the tensor is not rewritten.  parts > 1: `sums` is
    [B][parts] partial sums, to be added in index order (dg_step_prologue_fetch)"""
    t._dg_sums = (sums, AccArena.epoch, t._version, int(parts))
    return t


def tagged_sums(t, with_parts=False):
    """the per-sample sums tagged on `t`, or None; with_parts: (sums, parts) - readers that only take one float per sample
    (with_parts=False) see None for a tensor that carries partial sums"""
    tag = getattr(t, "_dg_sums", None)
    if tag is None or tag[1] != AccArena.epoch or tag[2] != t._version:
        return None
    if with_parts:
        return tag[0], tag[3]
    return tag[0] if tag[3] == 1 else None


def policy_mask(policy):
    m = 0
    for p in policy or ():
        if p not in POLICY_BITS:
            raise KeyError(p)
        m |= POLICY_BITS[p]
    return m
