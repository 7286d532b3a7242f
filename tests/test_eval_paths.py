"""Evaluation paths: FG-ARI on the device, top-k metrics, multi-view ensembling and the two
eval loops, against a fixture produced by the reference's own slowfast/utils/metrics.py (oracle/make_golden.py
main_metrics) and against the per-clip semantics of meters.py:300-410."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from focus_amd.slowfast.utils import meters, metrics


def fixture():
    return np.load(os.path.join(GOLDEN, "metrics.npz"), allow_pickle=False)


def test_fg_ari_matches_the_reference_metric():
    z = fixture()
    true, pred = torch.from_numpy(z["true"]), torch.from_numpy(z["pred"])
    for b in range(true.shape[0]):
        assert abs(metrics.evaluate_ari(true[b:b + 1, 1:], pred[b:b + 1]) - float(z["ari_each"][b])) < 1e-12, b
    assert abs(metrics.evaluate_ari(true[:, 1:], pred) - float(z["ari_fg"])) < 1e-12
    assert abs(metrics.evaluate_ari(true, pred) - float(z["ari_all"])) < 1e-12
    assert float(z["ari_each"][0]) == 1.0                                   # the perfect case of metrics.py:27-29
    # contingency tables directly (compute_ari): identical partitions -> 1, independent ones -> about 0
    t = torch.tensor([[[5, 0], [0, 7]], [[3, 3], [3, 3]]])
    a = metrics.ari_from_tables(t)
    assert float(a[0]) == 1.0 and abs(float(a[1])) < 0.11          # 12 points: exactly -0.1


def test_topk_and_multitask_counts_match_the_reference():
    z = fixture()
    s, l = torch.from_numpy(z["scores"]), torch.from_numpy(z["labels"])
    s2, l2 = torch.from_numpy(z["scores2"]), torch.from_numpy(z["labels2"])
    assert [float(x) for x in metrics.topks_correct(s, l, (1, 5))] == list(z["topk"])
    assert np.allclose([float(x) for x in metrics.topk_accuracies(s, l, (1, 5))], z["topk_acc"])
    assert np.allclose([float(x) for x in metrics.topk_errors(s, l, (1, 5))], 100.0 - z["topk_acc"])
    assert [float(x) for x in metrics.multitask_topks_correct((s, s2), (l, l2), (1, 5))] == list(z["multitask"])
    assert np.allclose([float(x) for x in metrics.multitask_topk_accuracies((s, s2), (l, l2), (1, 5))], z["multitask"] / 64 * 100)


@pytest.mark.parametrize("method", ["sum", "max"])
def test_multi_view_ensembling(method):
    """meters.py:300-332: clip i belongs to video i // num_clips; the views of a video are summed (or max-ed); batches may
    arrive in any order and split a video's clips."""
    g = torch.Generator().manual_seed(3)
    V, NC, C = 7, 6, 13
    preds = torch.rand(V * NC, C, generator=g)
    labels_v = torch.randint(1, C, (V,), generator=g)
    clip_ids = torch.randperm(V * NC, generator=g)
    m = meters.TestMeter(V, NC, C, overall_iters=4, ensemble_method=method)
    for chunk in clip_ids.split(11):
        m.update_stats(preds[chunk], labels_v[chunk // NC], chunk)
    want = torch.zeros(V, C)
    for i in range(V * NC):                                                  # the reference's per-clip loop
        want[i // NC] = want[i // NC] + preds[i] if method == "sum" else torch.maximum(want[i // NC], preds[i])
    assert torch.allclose(m.video_preds, want, atol=1e-6) and torch.equal(m.video_labels, labels_v)
    st = m.finalize_metrics(ks=(1, 5))
    assert st["complete"] and st["split"] == "test_final"
    top1 = float((want.argmax(1) == labels_v).float().mean()) * 100
    assert st["top1_acc"] == "%.2f" % top1
    with pytest.raises(AssertionError):
        m.update_stats(preds[:1], labels_v[:1] + 1, torch.zeros(1, dtype=torch.long))   # a clip contradicting its video's label


def test_perform_test_sums_the_views_of_every_video():
    from focus_amd.train import perform_test

    class Scores(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, inputs, meta):
            return inputs[0].flatten(1)[:, :5] + self.w

    V, NC = 4, 3
    g = torch.Generator().manual_seed(5)
    clips = torch.rand(V * NC, 5, generator=g)
    labels = torch.arange(V * NC) // NC % 5
    loader = [([clips[i:i + 4]], labels[i:i + 4], torch.arange(i, min(i + 4, V * NC)), {}) for i in range(0, V * NC, 4)]
    meter = meters.TestMeter(V, NC, 5, len(loader))
    st = perform_test(loader, Scores(), meter, None)
    assert st["complete"] and torch.allclose(meter.video_preds, clips.view(V, NC, 5).sum(1), atol=1e-6)


@pytest.mark.gpu
def test_fg_ari_on_the_device():
    z = fixture()
    d = torch.device("cuda:0")
    true, pred = torch.from_numpy(z["true"]).to(d), torch.from_numpy(z["pred"]).to(d)
    assert abs(metrics.evaluate_ari(true[:, 1:], pred) - float(z["ari_fg"])) < 1e-12


@pytest.mark.gpu
def test_slot_eval_epoch_scores_the_encoder_masks():
    """steve_eval_net.py:75-132 on a tiny STEVE: the loop's FG-ARI equals evaluate_ari on model.encode's masks."""
    from test_gpu_steve import _steve_small
    from focus_amd.train import slot_eval_epoch
    cfg, m = _steve_small(False)
    m = m.to("cuda:0")
    g = torch.Generator().manual_seed(0)
    video = torch.rand(2, 2, 3, 16, 16, generator=g)
    seg = torch.randint(0, 4, (2, 2, 16, 16), generator=g)
    true = torch.nn.functional.one_hot(seg, 4).permute(0, 1, 4, 2, 3).unsqueeze(3).float()     # [B,T,S,1,H,W]
    torch.manual_seed(1)
    mean, std = slot_eval_epoch([(video, true)], m)
    torch.manual_seed(1)
    with torch.no_grad():
        _, _, pm = m.eval().encode(video.to("cuda:0"))
    want = 100 * metrics.evaluate_ari(true.permute(0, 2, 1, 3, 4, 5)[:, 1:].flatten(start_dim=2),
                                      pm.permute(0, 2, 1, 3, 4, 5).flatten(start_dim=2))
    assert abs(mean - want) < 1e-9 and std == 0.0 and -100.0 <= mean <= 100.0
