"""Drop-in for the reference's utils.py (colour datasets): same public names, PIL/numpy back end
(medt_amd/data.py) because cv2 / torchvision / skimage are not available on the MI355X image."""
from medt_amd.data import (JointTransform2D, ImageToImage2D, Image2D, Logger, MetricList, chk_mkdir,  # noqa: F401
                           correct_dims, to_long_tensor)
