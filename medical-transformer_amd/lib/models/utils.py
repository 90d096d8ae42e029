"""reference lib/models/utils.py:4-5 -- the class name must survive for
state_dict / isinstance compatibility; the weight is consumed by the HIP
1x1-convolution kernel, this module's own forward is never called."""
import torch.nn as nn


class qkv_transform(nn.Conv1d):
    """Conv1d(k=1, bias=False) holder of the (2C, C, 1) qkv projection weight."""
