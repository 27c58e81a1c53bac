"""Visual (Conv3d + ResNet-18) and audio (1-D ResNet-18) front-ends with the reference's module tree and
``state_dict`` keys (frontend/resnet.py, frontend/resnet1d.py).

STATUS ("what is native"): the convolution / BatchNorm arithmetic of the front-ends still runs
through ATen here while the implicit-GEMM HIP kernels for them are being brought up; this file is the one place
in the hot path that is not yet served by libavsr_hip.so."""
import torch
import torch.nn.functional as F
from torch import nn


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def downsample_basic_block(inplanes, outplanes, stride):
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False),
                         nn.BatchNorm2d(outplanes))


def _act(relu_type, planes):
    if relu_type != "swish":
        raise NotImplementedError("only relu_type='swish' (the reference model) is implemented")
    return nn.SiLU(inplace=True)


class BasicBlock(nn.Module):
    """frontend/resnet.py:38-98."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="swish"):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = _act(relu_type, planes)
        self.relu2 = _act(relu_type, planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu2(out + res)


def _make_layer(block, inplanes, planes, blocks, stride, relu_type, down):
    ds = None
    if stride != 1 or inplanes != planes * block.expansion:
        ds = down(inplanes=inplanes, outplanes=planes * block.expansion, stride=stride)
    layers = [block(inplanes, planes, stride, ds, relu_type=relu_type)]
    layers += [block(planes * block.expansion, planes, relu_type=relu_type) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class ResNet(nn.Module):
    """frontend/resnet.py:101-166."""

    def __init__(self, block, layers, relu_type="swish"):
        super().__init__()
        self.inplanes = 64
        self.relu_type = relu_type
        self.downsample_block = downsample_basic_block
        self.layer1 = _make_layer(block, 64, 64, layers[0], 1, relu_type, downsample_basic_block)
        self.layer2 = _make_layer(block, 64, 128, layers[1], 2, relu_type, downsample_basic_block)
        self.layer3 = _make_layer(block, 128, 256, layers[2], 2, relu_type, downsample_basic_block)
        self.layer4 = _make_layer(block, 256, 512, layers[3], 2, relu_type, downsample_basic_block)
        self.inplanes = 512
        self.avgpool = nn.AdaptiveAvgPool2d(1)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.avgpool(x).flatten(1)


def threeD_to_2D_tensor(x):
    b, c, t, h, w = x.shape
    return x.transpose(1, 2).reshape(b * t, c, h, w)


class Conv3dResNet(nn.Module):
    """frontend/resnet.py:175-233: (B,T,1,88,88) -> (B,T,512)."""

    def __init__(self, backbone_type="resnet", relu_type="swish"):
        super().__init__()
        self.backbone_type = backbone_type
        self.frontend_nout = 64
        self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type=relu_type)
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, self.frontend_nout, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(self.frontend_nout),
            _act(relu_type, self.frontend_nout),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)),
        )

    def forward(self, xs_pad):
        xs = xs_pad.transpose(2, 1)
        B = xs.size(0)
        xs = self.frontend3D(xs)
        Tn = xs.shape[2]
        xs = self.trunk(threeD_to_2D_tensor(xs))
        return xs.view(B, Tn, xs.size(1))


def video_resnet():
    return Conv3dResNet()


# ------------------------------------------------------------------------------------------------ audio
def conv3x3_1d(in_planes, out_planes, stride=1):
    return nn.Conv1d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def downsample_basic_block_1d(inplanes, outplanes, stride):
    return nn.Sequential(nn.Conv1d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False),
                         nn.BatchNorm1d(outplanes))


class BasicBlock1D(nn.Module):
    """frontend/resnet1d.py:38-99."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="swish"):
        super().__init__()
        self.conv1 = conv3x3_1d(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm1d(planes)
        self.relu1 = _act(relu_type, planes)
        self.relu2 = _act(relu_type, planes)
        self.conv2 = conv3x3_1d(planes, planes)
        self.bn2 = nn.BatchNorm1d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu2(out + res)


class ResNet1D(nn.Module):
    """frontend/resnet1d.py:100-201."""

    def __init__(self, block, layers, relu_type="swish", a_upsample_ratio=1):
        super().__init__()
        self.inplanes = 64
        self.relu_type = relu_type
        self.downsample_block = downsample_basic_block_1d
        self.a_upsample_ratio = a_upsample_ratio
        self.conv1 = nn.Conv1d(1, 64, kernel_size=80, stride=4, padding=38, bias=False)
        self.bn1 = nn.BatchNorm1d(64)
        self.relu = _act(relu_type, 64)
        self.layer1 = _make_layer(block, 64, 64, layers[0], 1, relu_type, downsample_basic_block_1d)
        self.layer2 = _make_layer(block, 64, 128, layers[1], 2, relu_type, downsample_basic_block_1d)
        self.layer3 = _make_layer(block, 128, 256, layers[2], 2, relu_type, downsample_basic_block_1d)
        self.layer4 = _make_layer(block, 256, 512, layers[3], 2, relu_type, downsample_basic_block_1d)
        self.inplanes = 512
        self.avgpool = nn.AvgPool1d(kernel_size=20 // a_upsample_ratio, stride=20 // a_upsample_ratio)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        return self.avgpool(self.layer4(self.layer3(self.layer2(self.layer1(x)))))


class Conv1dResNet(nn.Module):
    """frontend/resnet1d.py:204-234: (B,S,1) -> (B,S//640,512)."""

    def __init__(self, relu_type="swish", a_upsample_ratio=1):
        super().__init__()
        self.a_upsample_ratio = a_upsample_ratio
        self.trunk = ResNet1D(BasicBlock1D, [2, 2, 2, 2], relu_type=relu_type, a_upsample_ratio=a_upsample_ratio)

    def forward(self, xs_pad):
        n = xs_pad.size(1) // 640 * 640
        return self.trunk(xs_pad[:, :n, :].transpose(1, 2)).transpose(1, 2)


def audio_resnet():
    return Conv1dResNet()
