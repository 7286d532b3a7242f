"""Multi-view test ensembling (mirror of the arithmetic of slowfast/utils/meters.py:235-410 `TestMeter`; its timers and
JSON logging are plumbing and are not reproduced).  Every video is sampled as NUM_ENSEMBLE_VIEWS x NUM_SPATIAL_CROPS
clips; clip i belongs to video i // num_clips; predictions are summed (or max-ed) per video and top-k accuracy is
taken over the videos."""
import torch

from . import metrics


class TestMeter(object):
    def __init__(self, num_videos, num_clips, num_cls, overall_iters, multi_label=False, ensemble_method="sum"):
        if multi_label:
            raise NotImplementedError("multi-label mAP (meters.py:385-389) belongs to the AVA/Charades families")
        if ensemble_method not in ("sum", "max"):
            raise NotImplementedError("Ensemble Method {} is not supported".format(ensemble_method))
        self.num_clips = num_clips
        self.overall_iters = overall_iters
        self.ensemble_method = ensemble_method
        self.video_preds = torch.zeros((num_videos, num_cls))
        self.video_labels = torch.zeros((num_videos)).long()
        self.clip_count = torch.zeros((num_videos)).long()
        self.stats = {}

    def reset(self):
        self.clip_count.zero_()
        self.video_preds.zero_()
        self.video_labels.zero_()

    def update_stats(self, preds, labels, clip_ids):
        """meters.py:300-332, vectorised: one index_add_ / index_reduce_ per batch instead of a Python loop per clip."""
        preds, labels = preds.detach().float().cpu(), labels.detach().cpu().long()
        vid = (clip_ids.detach().cpu().long() // self.num_clips)
        seen = self.video_labels[vid] > 0
        assert torch.equal(self.video_labels[vid][seen], labels[seen]), "clips of one video disagree on its label"
        self.video_labels[vid] = labels
        if self.ensemble_method == "sum":
            self.video_preds.index_add_(0, vid, preds)
        else:
            self.video_preds.index_reduce_(0, vid, preds, "amax", include_self=True)
        self.clip_count.index_add_(0, vid, torch.ones_like(vid))

    def finalize_metrics(self, ks=(1, 5)):
        """meters.py:372-410 -> {'split': 'test_final', 'top1_acc': 'xx.xx', ...}; `complete` says whether every video
        received exactly num_clips clips (the reference only logs a warning)."""
        self.stats = {"split": "test_final", "complete": bool(torch.all(self.clip_count == self.num_clips))}
        correct = metrics.topks_correct(self.video_preds, self.video_labels, ks)
        for k, x in zip(ks, correct):
            self.stats["top{}_acc".format(k)] = "{:.{prec}f}".format(float(x / self.video_preds.size(0)) * 100.0, prec=2)
        return self.stats
