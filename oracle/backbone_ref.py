"""CPU restatement (plain torch, fp32, functional) of the reference's SMAP inference forward.

TEST INFRASTRUCTURE ONLY (see oracle/smap_oracle.c header): used by tests/, smoke() and
bench.py's cpu_baseline leg; never imported by the product package.

Follows model/smap.py layer by layer from a state_dict, nothing fused, nothing commuted:
  conv_bn_relu   smap.py:13-45      Bottleneck      smap.py:48-77    ResNet_top smap.py:80-92
  downsample     smap.py:140-154    Upsample_unit   smap.py:210-241  Upsample_module smap.py:273-286
  SMAP.forward   smap.py:403-419 (inference branch; heads the branch never returns are skipped)
Pinned by tests/test_oracle_cpu.py against tests/golden/backbone_small.npz, which was produced
by importing the reference model itself (tests/golden/gen_golden.py).
"""
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)


def _cbr(sd, p, x, stride=1, relu=True):
    w = sd[p + ".conv.weight"]
    x = F.conv2d(x, w, sd[p + ".conv.bias"], stride=stride, padding=w.shape[-1] // 2)
    x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                     sd[p + ".bn.bias"], False, 0.0, 1e-5)
    return F.relu(x) if relu else x


def _bottleneck(sd, p, x, stride):
    out = _cbr(sd, p + ".conv_bn_relu1", x)
    out = _cbr(sd, p + ".conv_bn_relu2", out, stride=stride)
    out = _cbr(sd, p + ".conv_bn_relu3", out, relu=False)
    if (p + ".downsample.conv.weight") in sd:
        x = _cbr(sd, p + ".downsample", x, stride=stride, relu=False)
    return F.relu(out + x)


def _up(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def smap_forward(sd, imgs, stage_num=3):
    """sd: reference-keyed state_dict (fp32 CPU tensors); imgs [B,3,H,W] -> (hms, det_d, root_d)."""
    H, W = imgs.shape[-2] // 4, imgs.shape[-1] // 4
    x = _cbr(sd, "top.conv", imgs, stride=2)
    x = F.max_pool2d(x, 3, 2, 1)
    skip1 = skip2 = None
    res = res_d = res_rd = None
    for s in range(stage_num):
        last = s == stage_num - 1
        feats = []
        for li, n in enumerate(LAYERS):
            for j in range(n):
                x = _bottleneck(sd, f"stage{s}.downsample.layer{li + 1}.{j}", x, 2 if (li > 0 and j == 0) else 1)
            if skip1 is not None:
                x = x + skip1[li] + skip2[li]
            feats.append(x)
        out = None
        n1, n2, cross = [None] * 4, [None] * 4, None
        res = {}
        for ind, xin in enumerate(reversed(feats)):
            u = f"stage{s}.upsample.up{ind + 1}"
            o = _cbr(sd, u + ".u_skip", xin, relu=False)
            if ind > 0:
                o = o + _cbr(sd, u + ".up_conv", _up(out, xin.shape[-2:]), relu=False)
            out = F.relu(o)
            if not last:
                n1[3 - ind] = _cbr(sd, u + ".skip1", xin)
                n2[3 - ind] = _cbr(sd, u + ".skip2", out)
                if ind == 3:
                    cross = _cbr(sd, u + ".cross_conv", out)
            elif ind >= 1:
                res[ind + 1] = _up(_cbr(sd, u + ".res_conv2", _cbr(sd, u + ".res_conv1", out), relu=False), (H, W))
                if ind == 3:
                    res_d = _up(_cbr(sd, u + ".res_d_conv2", _cbr(sd, u + ".res_d_conv1", out), relu=False), (H, W))
                    res_rd = _up(_cbr(sd, u + ".res_rd_conv2", _cbr(sd, u + ".res_rd_conv1", out), relu=False), (H, W))
        skip1, skip2, x = n1, n2, cross
    return res[4] + res[3] + res[2], res_d, res_rd
