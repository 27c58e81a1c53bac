"""Visual (Conv3d + ResNet-18) and audio (1-D ResNet-18) front-ends with the reference's module tree and
``state_dict`` keys (frontend/resnet.py, frontend/resnet1d.py).

The torch ``nn.Conv*`` / ``nn.BatchNorm*`` children are parameter containers only: activations are kept
channels-last ([N, H, W, C], activation dtype) and every convolution / BatchNorm / pooling step, forward and
backward, runs in libavsr_hip.so (implicit-GEMM MFMA convolutions, auto_avsr_amd.functional.BasicBlockFn /
StemFn / AvgPoolFn).  The blocks therefore exchange ``(tensor, (N, H, W, C))`` pairs instead of NCHW tensors;
``Conv3dResNet`` / ``Conv1dResNet`` keep the reference's input and output conventions."""
import torch
from torch import nn

from . import functional as AF


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

    def forward(self, xd):
        x, (N, H, W, C) = xd
        out = AF.basic_block(x, (N, H, W, C), self.stride, self.training, self.conv1, self.bn1, self.conv2, self.bn2,
                             self.downsample)
        return out, (N, out.shape[1], out.shape[2], out.shape[3])


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

    def forward(self, xd):
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            with AF.component(f"trunk{i + 1}"):  # (mixed numerical mode: forward arithmetic per stage, AF.MIXED_POLICY)
                xd = layer(xd)
        x, (N, H, W, C) = xd
        return AF.avg_pool(x, N, H * W, C)  # AdaptiveAvgPool2d(1) + flatten -> (N, 512) f32


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
        B, Tn, C1, H, W = xs_pad.shape
        assert C1 == 1, "expects (B, T, 1, H, W) grayscale clips"
        conv, bn = self.frontend3D[0], self.frontend3D[1]
        geom = (B, Tn, H, W) + tuple(conv.kernel_size) + (conv.stride[1],) + tuple(conv.padding)
        with AF.component("stem"):
            x = AF.stem(xs_pad.reshape(B, Tn, H, W).float(), conv, bn, geom, pool=True)  # (B*T, 22, 22, 64)
        feats = self.trunk((x, (B * Tn, x.shape[1], x.shape[2], x.shape[3])))
        return feats.view(B, Tn, feats.size(1))


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

    def forward(self, xd):
        x, (N, H, W, C) = xd
        out = AF.basic_block(x, (N, H, W, C), self.stride, self.training, _As2d(self.conv1), self.bn1,
                             _As2d(self.conv2), self.bn2,
                             None if self.downsample is None else (_As2d(self.downsample[0]), self.downsample[1]))
        return out, (N, out.shape[1], out.shape[2], out.shape[3])


class _As2d:
    """View of a Conv1d's weight as a (Cout, Cin, 1, K) 2-D kernel (the 1-D trunk runs as H = 1 images)."""

    def __init__(self, conv):
        self.weight = conv.weight.unsqueeze(2)


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
        """x: (B, S) f32 waveform -> (B, S // 640, 512) f32."""
        B, S = x.shape
        k, st, pd = self.conv1.kernel_size[0], self.conv1.stride[0], self.conv1.padding[0]
        geom = (B, 1, 1, S, 1, 1, k, st, 0, 0, pd)
        with AF.component("astem"):
            h = AF.stem(x, self.conv1, self.bn1, geom, pool=False)  # (B, 1, S/4, 64)
        xd = (h, (B, 1, h.shape[2], h.shape[3]))
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            with AF.component(f"atrunk{i + 1}"):  # (mixed numerical mode: forward arithmetic per stage, AF.MIXED_POLICY)
                xd = layer(xd)
        h, (N, H, W, C) = xd
        win = self.avgpool.kernel_size[0] if isinstance(self.avgpool.kernel_size, tuple) else self.avgpool.kernel_size
        groups = B * (W // win)
        if W % win:
            raise ValueError("audio length must be a multiple of 640 samples")
        return AF.avg_pool(h, groups, win, C).view(B, W // win, C)


class Conv1dResNet(nn.Module):
    """frontend/resnet1d.py:204-234: (B,S,1) -> (B,S//640,512)."""

    def __init__(self, relu_type="swish", a_upsample_ratio=1):
        super().__init__()
        self.a_upsample_ratio = a_upsample_ratio
        self.trunk = ResNet1D(BasicBlock1D, [2, 2, 2, 2], relu_type=relu_type, a_upsample_ratio=a_upsample_ratio)

    def forward(self, xs_pad):
        n = xs_pad.size(1) // 640 * 640
        return self.trunk(xs_pad[:, :n, 0].float().contiguous())


def audio_resnet():
    return Conv1dResNet()
