"""ResNet-18-shaped feature extractor for 360 degree range images (reference src/models/resnet_modified.py).

Same topology and parameter names as the reference so that its checkpoints load unchanged
(``conv1``, ``layer{1..4}.{i}.conv{1,2}``, ``layer{2,3,4}.0.downsample.0``, ``fc``): no normalisation layers,
tanh (or relu) activations, every 3x3 convolution sees one wrapped column on each side of W and one zero row on
each side of H, strides (1,2) in the stem/pool/layer2/layer3 and (2,2) in layer4.

Two GPU paths compute it (``cnn_impl``; both are compared with each other and with torch-CPU in the tests):
  * fp32 tensors on shapes that tile (the reference's full-size network on 64/128-ring images): three autograd
    Functions on channels-last activations -- ``ring_conv.RingStem`` (conv1 + activation + max-pooling), ``RingSegment``
    for layer1..layer4 (fused Winograd / direct MFMA convolutions with the wrap-around as addressing and the elementwise tail
    in their epilogues; one Function in a single process, cut per layer under DDP so that the gradient all-reduce overlaps
    the backward) and ``MeanHW`` before ``fc``; inside ``torch.autocast`` the same structure in bf16 / fp16
    (``RingSegmentH``: csrc/convh.hip, wgradh.hip);
  * everything else (narrow test networks, autocast, dropout): the modules below -- library convolutions on inputs that
    travel in wrapped form, with activation (+ residual add) and the wrap-around padding as ONE fused HIP elementwise op
    (``ring_ops.ring_act_pad``) instead of the reference's separate tanh, add and three-copy F.pad per layer, and the
    stem's activation + padding + max-pooling + padding as one more (``ring_ops.ring_act_pool_pad``).
"""
import torch

from . import ring_conv
from .ring_ops import ring_act_pad, ring_act_pool_pad


class RingConv2d(torch.nn.Conv2d):
    """Bias-free 3x3 convolution of a 360-degree image whose input is ALREADY wrapped by one column on each side of W
    (see ring_ops.ring_act_pad); H gets ordinary zero padding.  Parameters are those of nn.Conv2d (key
    ``<name>.weight``).  Reference: F.pad(..., (1,1,0,0), 'circular') followed by Conv2d(padding=(1,0)),
    resnet_modified.py:97-98,162-168."""

    def __init__(self, in_planes, out_planes, stride=1):
        super().__init__(in_planes, out_planes, kernel_size=3, stride=stride, padding=(1, 0), bias=False)


class BasicBlock(torch.nn.Module):
    """Two ring convolutions with a residual connection (reference resnet_modified.py:136-177).  Activations travel
    between convolutions in wrapped ("padded") form: the block takes a padded, activated input and returns a padded,
    activated output, and each of its two elementwise stages -- activation, and residual add + activation -- is fused
    with the wrap-around padding of the next convolution's input."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, activation_fct="relu"):
        super().__init__()
        self.conv1 = RingConv2d(inplanes, planes, stride=stride)
        self.conv2 = RingConv2d(planes, planes)
        self.activation = torch.nn.ReLU(inplace=True) if activation_fct == "relu" else torch.nn.Tanh()
        self.act_name = "relu" if activation_fct == "relu" else "tanh"
        self.downsample = downsample
        self.stride = stride
        self.pad_out = True             # the last block of the network returns an unpadded tensor

    def forward(self, p_in):
        p_mid = ring_act_pad(self.conv1(p_in), self.act_name, pad=True)
        v = self.conv2(p_mid)
        if self.downsample is None:
            shortcut = p_in                                         # padded tensor: its interior is the residual
        else:
            shortcut = self.downsample(p_in[..., 1:-1])             # 1x1 strided convolution of the unpadded input
        return ring_act_pad(v, self.act_name, pad=self.pad_out, residual=shortcut)


class ResNetModified(torch.nn.Module):
    def __init__(self, in_channels, num_outputs, use_dropout=False, layers=(2, 2, 2, 2),
                 factor_fewer_resnet_channels=1, activation_fct="relu", impl="auto"):
        super().__init__()
        self.activation_fct = activation_fct
        # "auto": the channels-last HIP trunk (ring_conv.RingTrunk: fp32 MFMA convolutions with fused epilogues) whenever
        # the tensors are fp32 on the GPU and the shapes tile, the module path (library convolutions + fused ring ops)
        # otherwise; "modules" forces the latter, "hip" makes unsupported shapes an error.
        self.impl = impl
        self.use_dropout = bool(use_dropout)
        widths = [int(c / factor_fewer_resnet_channels) for c in (64, 128, 256, 512)]
        self.inplanes = widths[0]
        self.dropout_values = torch.nn.Dropout(p=0.2) if use_dropout else torch.nn.Identity()
        self.dropout_channels = torch.nn.Dropout2d(p=0.2) if use_dropout else torch.nn.Identity()
        self.conv1 = RingConv2d(in_channels, self.inplanes, stride=(1, 2))
        self.relu = torch.nn.ReLU(inplace=True)
        self.tanh = torch.nn.Tanh()
        self.maxpool = torch.nn.MaxPool2d(kernel_size=3, stride=(1, 2), padding=(1, 0))
        self.layer1 = self._make_layer(widths[0], layers[0], stride=1)
        self.layer2 = self._make_layer(widths[1], layers[1], stride=(1, 2))
        self.layer3 = self._make_layer(widths[2], layers[2], stride=(1, 2))
        self.layer4 = self._make_layer(widths[3], layers[3], stride=(2, 2))
        self.avgpool = torch.nn.AdaptiveAvgPool2d((1, 1))
        self.fc = torch.nn.Linear(widths[3] * BasicBlock.expansion, num_outputs)
        self.layer4[-1].pad_out = False
        for m in self.modules():                       # resnet_modified.py:64-66
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity=activation_fct)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * BasicBlock.expansion:
            downsample = torch.nn.Sequential(
                torch.nn.Conv2d(self.inplanes, planes * BasicBlock.expansion, kernel_size=1, stride=stride, bias=False))
        stack = [BasicBlock(self.inplanes, planes, stride=stride, downsample=downsample, activation_fct=self.activation_fct)]
        self.inplanes = planes * BasicBlock.expansion
        stack += [BasicBlock(self.inplanes, planes, activation_fct=self.activation_fct) for _ in range(1, blocks)]
        return torch.nn.Sequential(*stack)

    def _trunk_blocks(self):
        blocks, weights = [], []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                stride = blk.stride if isinstance(blk.stride, tuple) else (blk.stride, blk.stride)
                has_ds = blk.downsample is not None
                blocks.append((blk.conv1.in_channels, blk.conv1.out_channels, tuple(stride), has_ds))
                weights += [blk.conv1.weight, blk.conv2.weight] + ([blk.downsample[0].weight] if has_ds else [])
        return tuple(blocks), weights

    def trunk_weights_channels_last(self):
        """Store the trunk's convolution weights as ``[K][k][k][C]`` (torch channels_last): the HIP trunk then reads the
        parameters and writes their gradients in place.  Shapes, names and values of the state_dict do not change; conv1
        (8 input channels, re-laid out per call: 4.6 kB) keeps the default layout."""
        for w in self._trunk_blocks()[1]:
            w.data = w.data.contiguous(memory_format=torch.channels_last)
        return self

    def _note_module_path(self, x):
        """One line per (shape, dtype, mode) whenever a CUDA input takes the module path (library convolutions + ring ops) instead of
        the HIP stem + trunk: a 5x slower path must not be taken silently (narrow test networks, dropout, widths not divisible by 4)."""
        if not x.is_cuda or self.impl == "modules":
            return
        key = (tuple(x.shape), x.dtype, torch.is_autocast_enabled(), self.training and self.use_dropout)
        seen = self.__dict__.setdefault("_module_path_noted", set())
        if key not in seen:
            seen.add(key)
            why = ("dropout is active" if key[3] else "autocast shapes do not tile (csrc/convh.hip)" if key[2]
                   else "channel counts are not multiples of 64, or the width is not a multiple of 4")
            print(f"[delora_amd] CNN input {key[0]} {str(x.dtype).replace('torch.', '')} runs on the MODULE path (library convolutions), "
                  f"not on the HIP stem + trunk: {why}", flush=True)

    def hip_path_takes(self, H, W, in_channels=8, batch=1):
        """Whether an ``[N,in_channels,H,W]`` fp32 CUDA input would run on the channels-last HIP stem + trunk (shape test
        only: ``ring_conv.supported`` / ``stem_supported``: full-width channel counts, a width divisible by 4).  True for BASELINE's
        images and for the reference's shipped 64 x 720 and 64 x 512 (tiles hang over the edges of maps that do not divide)."""
        if self.impl == "modules" or W % 4:
            return False
        C0 = self.conv1.out_channels
        return (ring_conv.stem_supported((batch, in_channels, H, W), C0)
                and ring_conv.supported((batch, H, W // 4, C0), self._trunk_blocks()[0]))

    def hip_trunk_applicable(self, x_pooled_shape_nhwc, x):
        """The HIP trunk runs fp32 CUDA tensors, no dropout, outside autocast, on shapes that tile."""
        if self.impl == "modules" or not x.is_cuda or x.dtype != torch.float32 or torch.is_autocast_enabled():
            return False
        if self.use_dropout and self.training:
            return False
        return ring_conv.supported(x_pooled_shape_nhwc, self._trunk_blocks()[0])

    def hip_half_applicable(self, x):
        """The half-precision HIP trunk (``ring_conv.RingTrunkH``) runs CUDA inputs inside ``torch.autocast`` (fp16 / bf16), no
        dropout, on shapes that tile; returns the autocast dtype or None."""
        if self.impl == "modules" or not x.is_cuda or not torch.is_autocast_enabled():
            return None
        if self.use_dropout and self.training:
            return None
        dtype = torch.get_autocast_dtype("cuda")
        N, Cin, Hin, Win = x.shape
        C0 = self.conv1.out_channels
        if dtype not in ring_conv.DTYPE_CODE or Win % 4 or not ring_conv.stem_supported((N, Cin, Hin, Win), C0):
            return None
        return dtype if ring_conv.supported_h((N, Hin, Win // 4, C0), self._trunk_blocks()[0]) else None

    def pooled_features(self, x):
        """The globally pooled feature ``[N,C']`` (fp32; the input of ``fc``) when the whole CNN runs on the HIP stem + trunk -- fp32, or
        half precision inside autocast -- else None.  Returns (feat, last feature map as NCHW view or None)."""
        act = "relu" if self.activation_fct == "relu" else "tanh"
        N, Cin, Hin, Win = x.shape
        C0 = self.conv1.out_channels
        half = self.hip_half_applicable(x)
        if half is not None:
            # autocast: fp32 stem (8 input channels: 0.4 ms), then layer1..layer4 + pooling on the half-precision MFMA kernels
            blocks, weights = self._trunk_blocks()
            with torch.autocast("cuda", enabled=False):
                x0 = ring_conv.RingStem.apply(x.float(), self.conv1.weight, ring_conv.ACT[act])          # [N,H,W/4,C0] fp32
                feat = ring_conv.RingTrunkH.apply(x0, ring_conv.ACT[act], blocks, half, *weights)     # [N,C'] fp32
            return feat, None
        if (self.hip_trunk_applicable((N, Hin, Win // 4, C0), x) and Win % 4 == 0
                and ring_conv.stem_supported(tuple(x.shape), C0)):
            # channels-last from the first layer on: stem (conv1 + act + pool) and layer1..layer4 on the HIP kernels
            blocks, weights = self._trunk_blocks()
            x0 = ring_conv.RingStem.apply(x, self.conv1.weight, ring_conv.ACT[act])              # [N,H,W/4,C0]
            x4 = ring_conv.RingTrunk.apply(x0, ring_conv.ACT[act], blocks, *weights)          # [N,H',W',C']
            return ring_conv.MeanHW.apply(x4), x4.permute(0, 3, 1, 2)
        return None, None

    def forward(self, x):
        act = "relu" if self.activation_fct == "relu" else "tanh"
        x = self.dropout_values(x)
        N, Cin, Hin, Win = x.shape
        C0 = self.conv1.out_channels
        feat, x4 = self.pooled_features(x)
        if feat is not None:
            with torch.autocast("cuda", enabled=False):
                out = self.dropout_values(self.fc(feat))
            return [None, None, None, x4, out]
        p = ring_act_pad(x, "none", pad=True)
        p = ring_act_pool_pad(self.conv1(p), act)                    # act + wrap + self.maxpool + wrap, fused
        N, C0, H0, Wp = p.shape
        if self.hip_trunk_applicable((N, H0, Wp - 2, C0), p):
            # channels-last trunk: layer1..layer4 as one autograd Function on the fp32 matrix cores
            blocks, weights = self._trunk_blocks()
            x0 = p[..., 1:-1].permute(0, 2, 3, 1).contiguous()
            x4 = ring_conv.RingTrunk.apply(x0, ring_conv.ACT[act], blocks, *weights)          # [N,H',W',C']
            out = self.dropout_values(self.fc(ring_conv.MeanHW.apply(x4)))
            return [None, None, None, x4.permute(0, 3, 1, 2), out]
        if self.impl == "hip":
            raise RuntimeError(f"cnn_impl 'hip': the HIP trunk does not support input {tuple(x.shape)} / dtype {x.dtype}")
        self._note_module_path(x)
        p1 = self.layer1(p)
        p2 = self.layer2(p1)
        p3 = self.dropout_channels(self.layer3(p2))
        x4 = self.layer4(p3)                                         # unpadded (last block)
        out = self.dropout_values(self.fc(torch.flatten(self.avgpool(x4), 1)))
        # the reference returns the four feature maps as well; they are views of the padded tensors here
        return [p1[..., 1:-1], p2[..., 1:-1], p3[..., 1:-1], x4, out]
