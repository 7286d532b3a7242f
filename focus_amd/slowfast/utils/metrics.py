"""Evaluation metrics of the hot-path models (mirror of slowfast/utils/metrics.py:10-83, 107-200; SURVEY.md section 8(f)
rank 4).  FG-ARI is computed ON THE DEVICE for the whole batch: the reference moves every clip's masks to the host and
runs scipy's comb over a numpy contingency table per clip (metrics.py:39-83); here the tables are one batched matmul and
comb(x, 2) = x (x - 1) / 2 in float64 (exact for counts below 2^26)."""
import torch


def _comb2(x):
    return x * (x - 1.0) * 0.5


def ari_from_tables(table):
    """compute_ari (metrics.py:10-36) for a batch of contingency tables [B, r, s] (any float/int dtype) -> [B] float64."""
    t = table.double()
    a, b = t.sum(dim=2), t.sum(dim=1)                         # row / column totals
    n = a.sum(dim=1)
    comb_a, comb_b, comb_n, comb_t = _comb2(a).sum(1), _comb2(b).sum(1), _comb2(n), _comb2(t).sum((1, 2))
    perfect = (comb_b == comb_a) & (comb_a == comb_n) & (comb_n == comb_t)
    expected = comb_a * comb_b / comb_n
    ari = (comb_t - expected) / (0.5 * (comb_a + comb_b) - expected)
    return torch.where(perfect, torch.ones_like(ari), ari)


def evaluate_ari(true_mask, pred_mask):
    """metrics.py:56-83.  true_mask [B, N0, D] (0/1), pred_mask [B, N1, D] (scores: arg-max over N1 binarises them).
    Returns the average ARI over the batch as a Python float, like the reference; everything up to that scalar stays on
    the masks' device."""
    B, K, D = pred_mask.shape
    onehot = torch.zeros_like(pred_mask, dtype=torch.float64)
    onehot.scatter_(1, torch.argmax(pred_mask, dim=1, keepdim=True), 1.0)
    truth = (true_mask.to(pred_mask.device).to(torch.uint8) != 0).double()    # .byte() of the reference: non-zero = member
    table = truth @ onehot.transpose(1, 2)                                      # [B, N0, N1]
    return float(ari_from_tables(table).sum() / B)


def topks_correct(preds, labels, ks):
    """metrics.py:107-138: number of samples whose label is among the k highest scores, for every k in ks."""
    assert preds.size(0) == labels.size(0), "Batch dim of predictions and labels must match"
    top = torch.topk(preds, max(ks), dim=1, largest=True, sorted=True)[1]        # [N, max_k]
    hit = top.eq(labels.view(-1, 1))
    return [hit[:, :k].float().sum() for k in ks]


def topk_errors(preds, labels, ks):
    """metrics.py:141-151."""
    return [(1.0 - x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]


def topk_accuracies(preds, labels, ks):
    """metrics.py:154-163."""
    return [(x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]


def multitask_topks_correct(preds, labels, ks=(1,)):
    """metrics.py:167-196: a sample counts for k when EVERY task (EPIC-Kitchens verb and noun) has its label in its top k."""
    max_k = max(int(k) for k in ks)
    hits = 0
    for output, label in zip(preds, labels):
        idx = output.topk(max_k, dim=1, largest=True, sorted=True)[1]
        hits = hits + idx.eq(label.view(-1, 1)).to(torch.int32)                   # [N, max_k]
    tasks = len(preds)
    return [torch.ge(hits[:, :k].float().sum(1), tasks).float().sum(0) for k in ks]


def multitask_topk_accuracies(preds, labels, ks):
    """metrics.py:199-209."""
    n = preds[0].size(0)
    return [(x / n) * 100.0 for x in multitask_topks_correct(preds, labels, ks)]
