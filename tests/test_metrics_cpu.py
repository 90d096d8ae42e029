"""metrics.py on the CPU: the scoring conventions of performancemetrics_monuseg.m and the reference's classwise
metrics (metrics.py:23-57), the latter against the reference module itself where it is checked out."""
import pytest
import torch

import metrics
from oracle import ref_loader


def test_segmentation_scores_follow_the_matlab_script():
    torch.manual_seed(1)
    counts = torch.randint(0, 500, (7, 4), dtype=torch.int32)
    counts[2, 0] = 0                                        # no true positive: the script scores the image 1 / 1 / 1
    counts[5] = torch.tensor([0, 0, 0, 1000])
    f1, iou, pa = metrics.segmentation_scores(counts)
    for n, (tp, fp, fn, tn) in enumerate(counts.tolist()):
        if tp == 0:
            assert f1[n].item() == iou[n].item() == pa[n].item() == 1.0
        else:
            # the script's counters: "fp" = missed foreground, "fn" = false alarm, uni = union, ttp = ground truth
            uni, ttp = tp + fp + fn, tp + fn
            assert abs(f1[n].item() - 2 * tp / (2 * tp + fp + fn)) < 1e-12
            assert abs(iou[n].item() - tp / uni) < 1e-12
            assert abs(pa[n].item() - tp / ttp) < 1e-12


@pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")
def test_classwise_metrics_match_reference():
    ref = ref_loader.load_metrics()
    torch.manual_seed(2)
    out = torch.rand(3, 2, 9, 11)
    gt = torch.randint(0, 2, (3, 9, 11))
    assert torch.allclose(metrics.classwise_iou(out, gt), ref.classwise_iou(out, gt), rtol=0, atol=0)
    assert torch.allclose(metrics.classwise_f1(out, gt), ref.classwise_f1(out, gt), rtol=0, atol=0)
    assert torch.equal(metrics.jaccard_index(out, gt), ref.jaccard_index(out, gt).cpu())
    assert torch.equal(metrics.f1_score(out, gt), ref.f1_score(out, gt).cpu())
