"""ResNet-18-shaped feature extractor for 360 degree range images (reference src/models/resnet_modified.py).

Same topology and parameter names as the reference so that its checkpoints load unchanged
(``conv1``, ``layer{1..4}.{i}.conv{1,2}``, ``layer{2,3,4}.0.downsample.0``, ``fc``): no normalisation layers,
tanh (or relu) activations, every 3x3 convolution sees one wrapped column on each side of W and one zero row on
each side of H, strides (1,2) in the stem/pool/layer2/layer3 and (2,2) in layer4.  The wrap-around is folded
into the convolution module (``RingConv2d``) instead of a separate padding call per layer.
"""
import torch
import torch.nn.functional as F


class RingConv2d(torch.nn.Conv2d):
    """Bias-free convolution whose input is first wrapped by one column on each side of W (circular) while H
    gets ordinary zero padding; parameters are those of the wrapped nn.Conv2d (key ``<name>.weight``).
    Reference: F.pad(..., (1,1,0,0), 'circular') followed by Conv2d(padding=(1,0)), resnet_modified.py:97-98,162-168."""

    def __init__(self, in_planes, out_planes, stride=1):
        super().__init__(in_planes, out_planes, kernel_size=3, stride=stride, padding=(1, 0), bias=False)

    def forward(self, x):
        return super().forward(F.pad(x, (1, 1, 0, 0), mode="circular"))


def _activation(name):
    return torch.nn.ReLU(inplace=True) if name == "relu" else torch.nn.Tanh()


class BasicBlock(torch.nn.Module):
    """Two ring convolutions with a residual connection (reference resnet_modified.py:136-177)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, activation_fct="relu"):
        super().__init__()
        self.conv1 = RingConv2d(inplanes, planes, stride=stride)
        self.conv2 = RingConv2d(planes, planes)
        self.activation = _activation(activation_fct)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.activation(self.conv1(x))
        y = self.conv2(y)
        y = y + shortcut
        return self.activation(y)


class ResNetModified(torch.nn.Module):
    def __init__(self, in_channels, num_outputs, use_dropout=False, layers=(2, 2, 2, 2),
                 factor_fewer_resnet_channels=1, activation_fct="relu"):
        super().__init__()
        self.activation_fct = activation_fct
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

    def forward(self, x):
        act = self.relu if self.activation_fct == "relu" else self.tanh
        x = act(self.conv1(self.dropout_values(x)))
        x = self.maxpool(F.pad(x, (1, 1, 0, 0), mode="circular"))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.dropout_channels(self.layer3(x2))
        x4 = self.layer4(x3)
        out = self.dropout_values(self.fc(torch.flatten(self.avgpool(x4), 1)))
        return [x1, x2, x3, x4, out]
