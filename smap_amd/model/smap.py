"""`model.smap.SMAP` for MI355X: same constructor, parameters/buffers (1876 state_dict keys)
and inference `forward` contract as the reference (model/smap.py:313-421), with the forward
pass executed by the HIP engine (smap_amd/engine.py -> libsmap_hip.so).

The module tree exists to own the checkpoint: strict `load_state_dict` of a reference
checkpoint works because every parameter / buffer carries the reference's key
(`top.conv.{conv,bn}.*`, `stageK.downsample.layerL.i.conv_bn_reluJ.*`, `stageK.upsample.upU.<head>.*`).
There is no PyTorch forward here: `forward` needs eval mode and a ROCm device and raises
otherwise (training - smap.py:355-401 - is out of scope).
"""
import os

import torch
import torch.nn as nn

from ..engine import BackboneEngine, LAYERS, PLANES

DEFAULT_PRECISION = "x3"


class ConvBN(nn.Module):
    """Parameter holder for the reference's conv_bn_relu (smap.py:13-45): `.conv` + `.bn`."""

    def __init__(self, cin, cout, k, stride, relu):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2)
        self.bn = nn.BatchNorm2d(cout)
        self.has_relu = relu


class _Block(nn.Module):          # Bottleneck (smap.py:48-77)
    def __init__(self, cin, planes, stride, shortcut):
        super().__init__()
        self.conv_bn_relu1 = ConvBN(cin, planes, 1, 1, True)
        self.conv_bn_relu2 = ConvBN(planes, planes, 3, stride, True)
        self.conv_bn_relu3 = ConvBN(planes, planes * 4, 1, 1, False)
        self.downsample = shortcut


class _Down(nn.Module):           # ResNet_downsample_module (smap.py:95-154)
    def __init__(self):
        super().__init__()
        cin = 64
        for li, (planes, n) in enumerate(zip(PLANES, LAYERS)):
            stride = 1 if li == 0 else 2
            blocks = []
            for j in range(n):
                sc = None
                if j == 0 and (stride != 1 or cin != planes * 4):
                    sc = ConvBN(cin, planes * 4, 1, stride, False)      # built before the block, as upstream
                blocks.append(_Block(cin, planes, stride if j == 0 else 1, sc))
                cin = planes * 4
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        for m in self.modules():                                        # smap.py:111-117
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class _Unit(nn.Module):           # Upsample_unit (smap.py:157-241)
    def __init__(self, ind, cin, heads, chl, gen_skip, gen_cross):
        super().__init__()
        self.u_skip = ConvBN(cin, chl, 1, 1, False)
        if ind > 0:
            self.up_conv = ConvBN(chl, chl, 1, 1, False)
        if gen_skip:
            self.skip1 = ConvBN(cin, cin, 1, 1, True)
            self.skip2 = ConvBN(chl, cin, 1, 1, True)
        if ind == 3 and gen_cross:
            self.cross_conv = ConvBN(chl, 64, 1, 1, True)
        self.res_conv1 = ConvBN(chl, chl, 1, 1, True)
        self.res_conv2 = ConvBN(chl, heads[0], 3, 1, False)
        self.res_d_conv1 = ConvBN(chl, chl, 1, 1, True)
        self.res_d_conv2 = ConvBN(chl, heads[1], 3, 1, False)
        self.res_rd_conv1 = ConvBN(chl, chl, 1, 1, True)
        self.res_rd_conv2 = ConvBN(chl, 1, 3, 1, False)


class _Up(nn.Module):             # Upsample_module (smap.py:244-286)
    def __init__(self, heads, chl, gen_skip, gen_cross):
        super().__init__()
        for ind, cin in enumerate((2048, 1024, 512, 256)):
            setattr(self, f"up{ind + 1}", _Unit(ind, cin, heads, chl, gen_skip, gen_cross))


class _Stage(nn.Module):          # Single_stage_module (smap.py:289-310)
    def __init__(self, heads, chl, gen_skip, gen_cross):
        super().__init__()
        self.downsample = _Down()
        self.upsample = _Up(heads, chl, gen_skip, gen_cross)


class _Top(nn.Module):            # ResNet_top (smap.py:80-92)
    def __init__(self):
        super().__init__()
        self.conv = ConvBN(3, 64, 7, 2, True)


class SMAP(nn.Module):
    def __init__(self, cfg, run_efficient=False, **kwargs):
        super().__init__()
        if kwargs.get("zero_init_residual"):
            raise NotImplementedError("zero_init_residual is a training-time option (smap.py:119-122)")
        self.stage_num = cfg.MODEL.STAGE_NUM
        self.keypoint_num = cfg.DATASET.KEYPOINT.NUM
        self.paf_num = cfg.DATASET.PAF.NUM
        self.kpt_paf_num = self.keypoint_num + 2 * self.paf_num
        self.output_shape = tuple(cfg.OUTPUT_SHAPE)
        self.upsample_chl_num = cfg.MODEL.UPSAMPLE_CHANNEL_NUM
        self.ohkm, self.topk, self.ctf = cfg.LOSS.OHKM, cfg.LOSS.TOPK, cfg.LOSS.COARSE_TO_FINE
        if self.upsample_chl_num % 64:
            raise ValueError("UPSAMPLE_CHANNEL_NUM must be a multiple of 64 for the MFMA conv engine")
        self.top = _Top()
        for i in range(self.stage_num):
            last = i == self.stage_num - 1
            setattr(self, f"stage{i}", _Stage([self.kpt_paf_num, self.paf_num], self.upsample_chl_num,
                                              gen_skip=not last, gen_cross=not last))
        self._engines = {}
        self.weights_generation = 0
        # arithmetic of the HIP engine (smap_amd/engine.py): "x3" = fp16 hi/lo pairs + three MFMAs per K step, the mode
        # that reproduces the reference's fp32 forward (3D joints within 1e-3 m end to end); "f16" = plain fp16 storage,
        # ~2.7x faster, ~1e-3 relative error on the maps (0.3 cm mean joint error at 3 m, tests/test_e2e_parity_gpu.py)
        self.precision = os.environ.get("SMAP_PRECISION", DEFAULT_PRECISION)

    # -- engine cache: one device-resident schedule per (B,H,W,device); dropped whenever the
    #    weights are (re)loaded or moved.  After editing parameters in place call invalidate_engine().
    def invalidate_engine(self):
        self._engines = {}
        self.weights_generation = getattr(self, "weights_generation", 0) + 1     # long-lived users (PosePipeline) check it

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_engine()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate_engine()
        return super()._apply(fn, *a, **k)

    def engine(self, B, H, W, device, flip_pair=None, scaled_hms=False):
        """flip_pair: 43 channel indices (FLIP_ORDER + 15 + PAF FLIP_CHANNEL) -> an engine that runs the flip-TTA of
        test.py:55-70 inside its schedule (B frames in, merged maps of B frames out).  scaled_hms: the engine writes hms / 255 | / 127,
        the maps as test.py:111-112 hands them to dapalib (the pipelines); forward() keeps the raw maps the reference returns."""
        key = (B, H, W, str(device), self.precision, tuple(flip_pair) if flip_pair is not None else None, bool(scaled_hms))
        if key not in self._engines:
            if (H // 4, W // 4) != self.output_shape:
                raise ValueError(f"input {H}x{W} does not match cfg.OUTPUT_SHAPE {self.output_shape} (stride 4)")
            self._engines[key] = BackboneEngine(self.state_dict(), B, H, W, device, self.stage_num,
                                                self.upsample_chl_num, self.kpt_paf_num, self.paf_num,
                                                precision=self.precision, flip_pair=flip_pair, scaled_hms=scaled_hms)
        return self._engines[key]

    def forward(self, imgs, valids=None, labels=None, rdepth=None):
        if valids is not None or labels is not None:
            raise NotImplementedError("the training branch (smap.py:355-401) is out of scope of smap_amd")
        if self.training:
            raise RuntimeError("smap_amd.SMAP runs eval-mode inference only: call model.eval() first")
        if not imgs.is_cuda:
            raise RuntimeError("smap_amd.SMAP.forward needs the images on a ROCm GPU; there is no CPU fallback")
        if imgs.dim() != 4 or imgs.shape[1] != 3:
            raise ValueError(f"imgs must be [B,3,H,W], got {tuple(imgs.shape)}")
        B, _, H, W = imgs.shape
        eng = self.engine(B, H, W, imgs.device)
        # a fresh output buffer per call (caching allocator: no device copy), so the caller owns what it gets back --
        # the reference returns new tensors too -- while the engine's arena is reused by the next forward
        out = eng.new_output()
        res = eng.run(imgs.float(), out=out)
        # the engine's arithmetic keeps fp16's RANGE: an activation beyond 65504 ends as NaN in the maps and sets the status word
        # behind them.  Checking it costs a host sync, so direct callers opt in (PosePipeline checks every batch it collects anyway).
        if os.environ.get("SMAP_CHECK_FINITE"):
            eng.raise_if_nonfinite(out)
        return res
