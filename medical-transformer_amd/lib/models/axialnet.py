"""MI355X-native drop-in for the reference's lib/models/axialnet.py.

Same public names, constructor signatures, attribute names, parameter/buffer
registration order and state_dict layout as the reference (so checkpoints,
`optimizer` parameter order and `sum(p.numel())` match -- SURVEY.md 8b), but
the modules only *hold* parameters: forward/backward of the attention layers
run as hand-written gfx950 kernels behind libmedt_hip.so (include/medt_abi.h).

Reference map:
  AxialAttention / _dynamic / _wopos   lib/models/axialnet.py:19-258
  AxialBlock / _dynamic / _wopos       lib/models/axialnet.py:262-391
  ResAxialAttentionUNet, medt_net      lib/models/axialnet.py:397-711
  axialunet, gated, MedT, logo         lib/models/axialnet.py:714-728
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import qkv_transform

import medt_amd
from medt_amd import net as _net

__all__ = ["AxialAttention", "AxialAttention_dynamic", "AxialAttention_wopos", "AxialBlock", "AxialBlock_dynamic",
           "AxialBlock_wopos", "ResAxialAttentionUNet", "medt_net", "axialunet", "gated", "MedT", "logo", "conv1x1",
           "qkv_transform"]


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


# --------------------------------------------------------------------------- #
# attention layers
# --------------------------------------------------------------------------- #
class _AxialAttentionBase(nn.Module):
    """Parameter holder + dispatch for the three attention flavours."""
    _has_pos = True       # relative-position tables, BN2d(3G), BN1d(2C)
    _gated = False        # f_qr / f_kr / f_sve / f_sv
    _gate_init = (0.1, 0.1, 0.1, 1.0)      # f_qr, f_kr, f_sve, f_sv (:124-127)
    _gate_mode = 0        # 1: sigmoid(f) multiplies (model_codes.AxialAttention_gated_sig)

    def __init__(self, in_planes, out_planes, groups=8, kernel_size=56, stride=1, bias=False, width=False):
        assert (in_planes % groups == 0) and (out_planes % groups == 0)
        super().__init__()
        self.in_planes, self.out_planes, self.groups = in_planes, out_planes, groups
        self.group_planes = out_planes // groups
        self.kernel_size, self.stride, self.bias, self.width = kernel_size, stride, bias, width
        self.bn_groups = 1            # >1 only for the batched LoGo patches (set by medt_net)

        pos = self._has_pos
        self.qkv_transform = qkv_transform(in_planes, out_planes * 2, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn_qkv = nn.BatchNorm1d(out_planes * 2)
        self.bn_similarity = nn.BatchNorm2d(groups * 3 if pos else groups)
        self.bn_output = nn.BatchNorm1d(out_planes * 2 if pos else out_planes)
        if self._gated:
            for name, val in zip(("f_qr", "f_kr", "f_sve", "f_sv"), self._gate_init):
                setattr(self, name, nn.Parameter(torch.tensor(val), requires_grad=False))
        self._gate_modules(in_planes)         # (model_codes.AxialAttention_gated_data registers its gate MLP here)
        if pos:
            self.relative = nn.Parameter(torch.randn(self.group_planes * 2, kernel_size * 2 - 1), requires_grad=True)
            rows = torch.arange(kernel_size).unsqueeze(1)
            cols = torch.arange(kernel_size).unsqueeze(0)
            self.register_buffer("flatten_index", (rows - cols + kernel_size - 1).view(-1))
        if stride > 1:
            self.pooling = nn.AvgPool2d(stride, stride=stride)     # kept for module-tree parity; fused in the kernel
        self.reset_parameters()

    def _gate_modules(self, in_planes):
        pass

    def _gates(self, x):
        """(f_qr, f_kr, f_sve, f_sv) for medt_amd.axial_attention, or None."""
        return (self.f_qr, self.f_kr, self.f_sve, self.f_sv) if self._gated else None

    def reset_parameters(self):
        self.qkv_transform.weight.data.normal_(0, math.sqrt(1. / self.in_planes))
        if self._has_pos:
            nn.init.normal_(self.relative, 0., math.sqrt(1. / self.group_planes))

    def forward(self, x):
        return self.run(x, self.bn_groups, False)

    def run(self, x, bn_groups=1, out_relu=False, pre=None):
        L = x.shape[3] if self.width else x.shape[2]
        if self._has_pos and L != self.kernel_size:
            # same failure class as the reference's einsum shape error (e.g. `logo` at 256)
            raise RuntimeError(f"axial attention built for sequence length {self.kernel_size}, got {L}")
        gates = self._gates(x)
        return medt_amd.axial_attention(
            x, self.qkv_transform.weight, self.bn_qkv, self.bn_similarity, self.bn_output,
            self.relative if self._has_pos else None, gates, self.groups, self.width, self.stride,
            self.training, bn_groups, out_relu, self._gate_mode, pre)


class AxialAttention(_AxialAttentionBase):
    pass


class AxialAttention_dynamic(_AxialAttentionBase):
    _gated = True


class AxialAttention_wopos(_AxialAttentionBase):
    _has_pos = False


# --------------------------------------------------------------------------- #
# residual blocks
# --------------------------------------------------------------------------- #
class _AxialBlockBase(nn.Module):
    expansion = 2
    _attention = AxialAttention
    _extra_conv1 = False          # AxialBlock_wopos registers an unused biased 1x1 `conv1`

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None, kernel_size=56):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        width = int(planes * (base_width / 64.))
        self.conv_down = conv1x1(inplanes, width)
        if self._extra_conv1:
            self.conv1 = nn.Conv2d(width, width, kernel_size=1)
        self.bn1 = norm_layer(width)
        self.hight_block = self._attention(width, width, groups=groups, kernel_size=kernel_size)
        self.width_block = self._attention(width, width, groups=groups, kernel_size=kernel_size, stride=stride,
                                           width=True)
        self.conv_up = conv1x1(width, planes * self.expansion)
        self.bn2 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        return _net.axial_block_forward(self, x)


class AxialBlock(_AxialBlockBase):
    pass


class AxialBlock_dynamic(_AxialBlockBase):
    _attention = AxialAttention_dynamic


class AxialBlock_wopos(_AxialBlockBase):
    _attention = AxialAttention_wopos
    _extra_conv1 = True


# --------------------------------------------------------------------------- #
# networks
# --------------------------------------------------------------------------- #
class _AxialNetBase(nn.Module):
    def _make_layer(self, block, planes, blocks, kernel_size=56, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        stages = [block(self.inplanes, planes, stride, downsample, groups=self.groups, base_width=self.base_width,
                        dilation=previous_dilation, norm_layer=norm_layer, kernel_size=kernel_size)]
        self.inplanes = planes * block.expansion
        if stride != 1:
            kernel_size = kernel_size // 2
        for _ in range(1, blocks):
            stages.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation, norm_layer=norm_layer, kernel_size=kernel_size))
        return nn.Sequential(*stages)

    def _common_init(self, groups, width_per_group, replace_stride_with_dilation, norm_layer, s):
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = int(64 * s)
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None "
                             "or a 3-element tuple, got {}".format(replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        return norm_layer, replace_stride_with_dilation

    def forward(self, x):
        return self._forward_impl(x)


class ResAxialAttentionUNet(_AxialNetBase):
    def __init__(self, block, layers, num_classes=2, zero_init_residual=True, groups=8, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, s=0.125, img_size=128, imgchan=3):
        super().__init__()
        norm_layer, rswd = self._common_init(groups, width_per_group, replace_stride_with_dilation, norm_layer, s)
        self.conv1 = nn.Conv2d(imgchan, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv2 = nn.Conv2d(self.inplanes, 128, kernel_size=3, stride=1, padding=1, bias=False)
        self.conv3 = nn.Conv2d(128, self.inplanes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.bn2 = norm_layer(128)
        self.bn3 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, int(128 * s), layers[0], kernel_size=(img_size // 2))
        self.layer2 = self._make_layer(block, int(256 * s), layers[1], stride=2, kernel_size=(img_size // 2),
                                       dilate=rswd[0])
        self.layer3 = self._make_layer(block, int(512 * s), layers[2], stride=2, kernel_size=(img_size // 4),
                                       dilate=rswd[1])
        self.layer4 = self._make_layer(block, int(1024 * s), layers[3], stride=2, kernel_size=(img_size // 8),
                                       dilate=rswd[2])
        self.decoder1 = nn.Conv2d(int(1024 * 2 * s), int(1024 * 2 * s), kernel_size=3, stride=2, padding=1)
        self.decoder2 = nn.Conv2d(int(1024 * 2 * s), int(1024 * s), kernel_size=3, stride=1, padding=1)
        self.decoder3 = nn.Conv2d(int(1024 * s), int(512 * s), kernel_size=3, stride=1, padding=1)
        self.decoder4 = nn.Conv2d(int(512 * s), int(256 * s), kernel_size=3, stride=1, padding=1)
        self.decoder5 = nn.Conv2d(int(256 * s), int(128 * s), kernel_size=3, stride=1, padding=1)
        self.adjust = nn.Conv2d(int(128 * s), num_classes, kernel_size=1, stride=1, padding=0)
        self.soft = nn.Softmax(dim=1)

    def _forward_impl(self, x):
        return _net.unet_forward(self, x)


class medt_net(_AxialNetBase):
    def __init__(self, block, block_2, layers, num_classes=2, zero_init_residual=True, groups=8, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, s=0.125, img_size=128, imgchan=3):
        super().__init__()
        norm_layer, rswd = self._common_init(groups, width_per_group, replace_stride_with_dilation, norm_layer, s)
        # global branch
        self.conv1 = nn.Conv2d(imgchan, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv2 = nn.Conv2d(self.inplanes, 128, kernel_size=3, stride=1, padding=1, bias=False)
        self.conv3 = nn.Conv2d(128, self.inplanes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.bn2 = norm_layer(128)
        self.bn3 = norm_layer(self.inplanes)
        self.bn1 = norm_layer(self.inplanes)       # re-assigned in the reference too (:532,:536): consumes RNG-free init
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, int(128 * s), layers[0], kernel_size=(img_size // 2))
        self.layer2 = self._make_layer(block, int(256 * s), layers[1], stride=2, kernel_size=(img_size // 2),
                                       dilate=rswd[0])
        self.decoder4 = nn.Conv2d(int(512 * s), int(256 * s), kernel_size=3, stride=1, padding=1)
        self.decoder5 = nn.Conv2d(int(256 * s), int(128 * s), kernel_size=3, stride=1, padding=1)
        self.adjust = nn.Conv2d(int(128 * s), num_classes, kernel_size=1, stride=1, padding=0)
        self.soft = nn.Softmax(dim=1)
        # local (patch) branch -- note self.inplanes is 64 here, hence the 64-wide local stem
        self.conv1_p = nn.Conv2d(imgchan, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv2_p = nn.Conv2d(self.inplanes, 128, kernel_size=3, stride=1, padding=1, bias=False)
        self.conv3_p = nn.Conv2d(128, self.inplanes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1_p = norm_layer(self.inplanes)
        self.bn2_p = norm_layer(128)
        self.bn3_p = norm_layer(self.inplanes)
        self.relu_p = nn.ReLU(inplace=True)
        img_size_p = img_size // 4
        self.layer1_p = self._make_layer(block_2, int(128 * s), layers[0], kernel_size=(img_size_p // 2))
        self.layer2_p = self._make_layer(block_2, int(256 * s), layers[1], stride=2, kernel_size=(img_size_p // 2),
                                         dilate=rswd[0])
        self.layer3_p = self._make_layer(block_2, int(512 * s), layers[2], stride=2, kernel_size=(img_size_p // 4),
                                         dilate=rswd[1])
        self.layer4_p = self._make_layer(block_2, int(1024 * s), layers[3], stride=2, kernel_size=(img_size_p // 8),
                                         dilate=rswd[2])
        self.decoder1_p = nn.Conv2d(int(1024 * 2 * s), int(1024 * 2 * s), kernel_size=3, stride=2, padding=1)
        self.decoder2_p = nn.Conv2d(int(1024 * 2 * s), int(1024 * s), kernel_size=3, stride=1, padding=1)
        self.decoder3_p = nn.Conv2d(int(1024 * s), int(512 * s), kernel_size=3, stride=1, padding=1)
        self.decoder4_p = nn.Conv2d(int(512 * s), int(256 * s), kernel_size=3, stride=1, padding=1)
        self.decoder5_p = nn.Conv2d(int(256 * s), int(128 * s), kernel_size=3, stride=1, padding=1)
        self.decoderf = nn.Conv2d(int(128 * s), int(128 * s), kernel_size=3, stride=1, padding=1)
        self.adjust_p = nn.Conv2d(int(128 * s), num_classes, kernel_size=1, stride=1, padding=0)
        self.soft_p = nn.Softmax(dim=1)

    def _forward_impl(self, x):
        return _net.medt_forward(self, x)


# --------------------------------------------------------------------------- #
# factories (reference :714-728) -- `pretrained` is accepted and ignored there too
# --------------------------------------------------------------------------- #
def axialunet(pretrained=False, **kwargs):
    return ResAxialAttentionUNet(AxialBlock, [1, 2, 4, 1], s=0.125, **kwargs)


def gated(pretrained=False, **kwargs):
    return ResAxialAttentionUNet(AxialBlock_dynamic, [1, 2, 4, 1], s=0.125, **kwargs)


def MedT(pretrained=False, **kwargs):
    return medt_net(AxialBlock_dynamic, AxialBlock_wopos, [1, 2, 4, 1], s=0.125, **kwargs)


def logo(pretrained=False, **kwargs):
    return medt_net(AxialBlock, AxialBlock, [1, 2, 4, 1], s=0.125, **kwargs)
