"""Host-side helpers mirrored from the reference's utils/__init__.py (only the four on the hot path,
SURVEY.md §2) plus the engine's Philox stream."""


def set_requires_grad(net, requires_grad: bool = True):
    """reference: utils/__init__.py:59-61.  The engine has no autograd; kept so callers need not change."""
    for param in net.parameters():
        param.requires_grad = requires_grad


def sigmoid_to_tanh(x):
    """[0,1] -> [-1,+1]  (reference: utils/__init__.py:70-73)"""
    return x * 2.0 - 1.0


def tanh_to_sigmoid(x):
    """[-1,+1] -> [0,1]  (reference: utils/__init__.py:76-79)"""
    return (x + 1.0) / 2.0


def cycle(iterable):
    """reference: utils/__init__.py:110-113"""
    while True:
        for i in iterable:
            yield i


def setup(model_path, config_path, ema=True, fix_noise=True, cuda=True):
    """reference: utils/__init__.py:116-160 -- what evaluate_*.py / demo.py call to get a generator back from a
    checkpoint: (cfg, G, lidar, device).  `cuda=False` is refused: this engine has no CPU path."""
    import os.path as osp

    import torch

    from ..models import define_G
    from ..models.dusty import GumbelSigmoid
    from .config import load_config_file
    from .lidar import LiDAR
    if not (cuda and torch.cuda.is_available()):
        raise RuntimeError("dusty_gan_amd.utils.setup needs an MI355X (no CPU path)")
    device = torch.device("cuda")
    cfg = load_config_file(config_path)
    cfg.model.gen.shape = cfg.dataset.shape
    cfg.model.dis.shape = cfg.dataset.shape
    assert ".pth" in model_path
    checkpoint = torch.load(model_path, map_location="cpu")
    G_state_dict = checkpoint["G_ema"] if ema else checkpoint["G"]
    print("#iterations:", checkpoint["step"])
    G = define_G(cfg)
    G.eval()
    G.load_state_dict(G_state_dict)
    G.to(device)
    if fix_noise:
        for m in G.modules():
            if isinstance(m, GumbelSigmoid):
                m.fix_on_first_use = True
    root = cfg.dataset.get("root")
    lidar = LiDAR(num_ring=cfg.dataset.shape[0], num_points=cfg.dataset.shape[1], min_depth=cfg.dataset.min_depth,
                  max_depth=cfg.dataset.max_depth, angle_file=osp.join(root, "angles.pt") if root else None)
    if str(cfg.dataset.name) == "synthetic":
        lidar.use_nominal_angles()
    lidar.to(device)
    return cfg, G, lidar, device
