"""Coverage, minimum matching distance and 1-NN accuracy -- reference: utils/metrics/cov_mmd_1nna.py:55-148.
The three pairwise matrices come from the all-pairs Chamfer kernel; what remains are reductions over [N,N] matrices
(device tensors, torch reductions).  Both metrics of the reference: "cd" (what the trainer asks for,
trainers/dcgan_amp.py:389-391) and "emd" (approximate matching; 10 annealing levels x 3 sweeps per pair, so an
all-pairs matrix of it is minutes of GPU time at validation size - as in the reference)."""
import torch

from .distance import chamfer_distance_matrix, emd_distance_matrix


def _compute_cov_mmd(M_rg):
    N_ref, N_gen = M_rg.shape
    mmd_gen, min_idx_gen = M_rg.min(dim=0)
    mmd_ref, _ = M_rg.min(dim=1)
    return {"mmd": mmd_ref.mean().item(), "mmd-sample": mmd_gen.mean().item(),
            "cov": float(len(torch.unique(min_idx_gen))) / float(N_ref)}


def _compute_nna(M_rr, M_rg, M_gg, k, sqrt=False):
    N_ref, N_gen = M_rg.shape
    device = M_rg.device
    label = torch.cat([torch.ones(N_ref, device=device), torch.zeros(N_gen, device=device)], dim=0)
    M = torch.cat([torch.cat((M_rr, M_rg), dim=1), torch.cat((M_rg.t(), M_gg), dim=1)], dim=0)
    M = M.abs().sqrt() if sqrt else M
    M = M + torch.diag(float("inf") * torch.ones_like(label))  # leave-one-out
    _, idx = M.topk(k=k, dim=0, largest=False)
    count = torch.zeros_like(label)
    for i in range(0, k):
        count = count + label.index_select(0, idx[i])
    pred = (count / k >= 0.5).float()
    s = {"tp": (pred * label).sum().item(), "fp": (pred * (1 - label)).sum().item(),
         "fn": ((1 - pred) * label).sum().item(), "tn": ((1 - pred) * (1 - label)).sum().item()}
    s.update({"precision": s["tp"] / (s["tp"] + s["fp"] + 1e-10), "recall": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
              "accuracy_t": s["tp"] / (s["tp"] + s["fn"] + 1e-10), "accuracy_f": s["tn"] / (s["tn"] + s["fp"] + 1e-10),
              "accuracy": torch.eq(label, pred).float().mean().item()})
    return s


@torch.no_grad()
def compute_cov_mmd_1nna(pcs_gen, pcs_ref, batch_size=512, metrics=("cd",), verbose=True):
    """same signature and result keys as the reference (:113-148); `batch_size` / `verbose` only shaped its Python loop"""
    assert isinstance(metrics, tuple)
    results = {}
    for metric in metrics:
        if metric not in ("cd", "emd"):
            raise NotImplementedError(f"metric '{metric}'")
        pairwise = chamfer_distance_matrix if metric == "cd" else emd_distance_matrix
        M_rr = pairwise(pcs_ref, pcs_ref)
        M_rg = pairwise(pcs_ref, pcs_gen)
        M_gg = pairwise(pcs_gen, pcs_gen)
        for k, v in _compute_cov_mmd(M_rg).items():
            results["{}-{}".format(k, metric)] = v
        for k, v in _compute_nna(M_rr, M_rg, M_gg, k=1, sqrt=False).items():
            results["1-nn-{}-{}".format(k, metric)] = v
    return results
