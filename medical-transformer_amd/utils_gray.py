"""Drop-in for the reference's utils_gray.py: one-channel reads and the `>= 127` mask threshold
(reference utils_gray.py:151,159-160,222)."""
from medt_amd import data as _d
from medt_amd.data import JointTransform2D, Logger, MetricList, chk_mkdir, correct_dims, to_long_tensor  # noqa: F401


class ImageToImage2D(_d.ImageToImage2D):
    def __init__(self, dataset_path, joint_transform=None, one_hot_mask=False):
        super().__init__(dataset_path, joint_transform, one_hot_mask, gray=True)


class Image2D(_d.Image2D):
    def __init__(self, dataset_path, transform=None):
        super().__init__(dataset_path, transform, gray=True)
