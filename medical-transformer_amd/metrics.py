"""Drop-in for the reference's metrics.py.  LogNLLLoss.forward is plain mean cross entropy
(reference metrics.py:17-20; the log() line is commented out there) and runs as the HIP kernel pair
medt_ce_fwd / medt_ce_bwd.  The classwise helpers are imported by train.py:23 but never called."""
import torch
from torch.nn.modules.loss import _WeightedLoss

import medt_amd

EPSILON = 1e-32


class LogNLLLoss(_WeightedLoss):
    __constants__ = ["weight", "reduction", "ignore_index"]

    def __init__(self, weight=None, size_average=None, reduce=None, reduction=None, ignore_index=-100):
        super().__init__(weight, size_average, reduce, reduction)
        self.ignore_index = ignore_index

    def forward(self, y_input, y_target):
        if self.weight is not None:
            raise NotImplementedError("class weights are never used by train.py (criterion = LogNLLLoss())")
        return medt_amd.cross_entropy(y_input, y_target, self.ignore_index)


def classwise_iou(output, gt):
    dims = (0, *range(2, len(output.shape)))
    onehot = torch.zeros_like(output).scatter_(1, gt[:, None, :], 1)
    inter = output * onehot
    union = output + onehot - inter
    return (inter.sum(dim=dims).float() + EPSILON) / (union.sum(dim=dims) + EPSILON)


def classwise_f1(output, gt):
    eps = 1e-20
    n = output.shape[1]
    pred = torch.argmax(output, dim=1)
    tp = torch.tensor([((pred == i) * (gt == i)).sum() for i in range(n)]).float()
    sel = torch.tensor([(pred == i).sum() for i in range(n)]).float()
    rel = torch.tensor([(gt == i).sum() for i in range(n)]).float()
    precision, recall = (tp + eps) / (sel + eps), (tp + eps) / (rel + eps)
    return 2 * (precision * recall) / (precision + recall)


def make_weighted_metric(classwise_metric):
    def weighted_metric(output, gt, weights=None):
        if weights is not None and len(weights) != output.shape[1]:
            raise ValueError("The number of weights must match with the number of classes")
        return classwise_metric(output, gt).cpu()          # the reference computes weights and ignores them too
    return weighted_metric


jaccard_index = make_weighted_metric(classwise_iou)
f1_score = make_weighted_metric(classwise_f1)


# --------------------------------------------------------------------------- #
# Dataset scores without MATLAB
# --------------------------------------------------------------------------- #
def segmentation_scores(counts):
    """Per-image F1 / IoU / pixel accuracy from (N,4) {tp, fp, fn, tn} counts, with the conventions of the reference's
    scoring scripts (performancemetrics_monuseg.m:68-78): F = 2tp/(2tp+fp+fn), IoU = tp/(tp+fp+fn),
    PA = tp/(tp+fn) (their `tp/ttp`), and an image without a single true positive scores 1 on all three."""
    c = counts.detach().to("cpu").double()
    tp, fp, fn = c[:, 0], c[:, 1], c[:, 2]
    one = tp == 0
    f1 = 2 * tp / (2 * tp + fp + fn).clamp_min(1)
    iou = tp / (tp + fp + fn).clamp_min(1)
    pa = tp / (tp + fn).clamp_min(1)
    f1[one], iou[one], pa[one] = 1.0, 1.0, 1.0
    return f1, iou, pa
