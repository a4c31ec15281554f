"""TEST INFRASTRUCTURE — never imported by the product path.

Torch-only stand-ins for the two third-party backbones the reference pulls in
and that are NOT installed in this image:

  * torchvision.models.resnet50      (call site /root/reference/src/student_model.py:16-20)
  * timm.create_model('vit_small_patch16_224', pretrained=True, num_classes=0)
                                      (call site /root/reference/src/teacher_model.py:36)

Both versions are un-pinned by the reference (no requirements/lock file).  The
published architectures are restated here from the papers / public module
layouts: ResNet-50 "v1.5" (stride on the 3x3 conv of each bottleneck) with the
torchvision child order conv1,bn1,relu,maxpool,layer1..4,avgpool,fc, and
ViT-S/16 (12 blocks, width 384, 6 heads, MLP 1536, pre-norm, LayerNorm eps 1e-6,
exact-erf GELU, class token + learned position embedding, final norm) with
timm's parameter names.  `install()` registers them in sys.modules so that the
reference's own files import unchanged; parity at the torchvision/timm boundary
itself is "unpinned" (nothing in the reference pins it) and is anchored on torch
CPU primitives instead.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- ResNet-50
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self._inplanes = 64
        self.layer1 = self._stage(64, 3, 1)
        self.layer2 = self._stage(128, 4, 2)
        self.layer3 = self._stage(256, 6, 2)
        self.layer4 = self._stage(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)

    def _stage(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self._inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self._inplanes, planes * 4, 1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self._inplanes, planes, stride, ds)]
        self._inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self._inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class _Weights:
    IMAGENET1K_V1 = "IMAGENET1K_V1"
    IMAGENET1K_V2 = "IMAGENET1K_V2"
    DEFAULT = "DEFAULT"


def resnet50(weights=None, **kw):
    # pretrained weights cannot be fetched offline; random init, callers overwrite
    return ResNet50()


# ----------------------------------------------------------------------------- MobileNetV2
# torchvision.models.mobilenet_v2 (call site /root/reference/src/student_model_compact.py:19-22: `.features`, 1280 channels).
# Module layout and state_dict names follow torchvision: features[0] = Conv2dNormActivation(3, 32, 3, stride 2) as
# Sequential(conv, bn, ReLU6); features[1..17] = InvertedResidual with `.conv` = Sequential([expand 1x1 CNA unless t = 1],
# depthwise 3x3 CNA, project 1x1 Conv2d, BatchNorm2d), residual when stride 1 and cin == cout; features[18] = CNA(320, 1280, 1).
MBV2_SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def _cna(cin, cout, k, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout),
                         nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, t):
        super().__init__()
        hid = cin * t
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        if t != 1:
            layers.append(_cna(cin, hid, 1))
        layers += [_cna(hid, hid, 3, stride, groups=hid), nn.Conv2d(hid, cout, 1, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        feats = [_cna(3, 32, 3, 2)]
        cin = 32
        for t, c, n, s in MBV2_SETTINGS:
            for i in range(n):
                feats.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_cna(cin, 1280, 1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))

    def forward(self, x):
        return self.classifier(self.features(x).mean((2, 3)))


def mobilenet_v2(weights=None, **kw):
    return MobileNetV2()


# ----------------------------------------------------------------------------- ViT-S/16
class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, dim * mlp_ratio)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch=16):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformerS16(nn.Module):
    def __init__(self, dim=384, depth=12, heads=6):
        super().__init__()
        self.num_features = dim
        self.embed_dim = dim
        self.patch_embed = _PatchEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, 197, dim) * 0.02)
        self.blocks = nn.Sequential(*[_Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        return self.norm(self.blocks(x))

    def forward(self, x):
        return self.forward_features(x)[:, 0]


def create_model(name, pretrained=False, num_classes=0, **kw):
    if name != "vit_small_patch16_224":
        raise ValueError(f"stand-in only restates vit_small_patch16_224, got {name}")
    return VisionTransformerS16()


def install():
    """Register `torchvision.models` and `timm` stand-ins (idempotent)."""
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvm.resnet50 = resnet50
        tvm.ResNet50_Weights = _Weights
        tvm.mobilenet_v2 = mobilenet_v2
        tvm.MobileNet_V2_Weights = _Weights
        tv.models = tvm
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
    if "timm" not in sys.modules:
        tm = types.ModuleType("timm")
        tm.create_model = create_model
        sys.modules["timm"] = tm
