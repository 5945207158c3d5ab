"""Torch (CPU or GPU, fp32 math) interpreter of the engine's op list (smap_amd.engine.Graph).

TEST INFRASTRUCTURE ONLY.  Two uses:
  * quantize=False: executes the schedule with the un-rounded folded weights -> checks the
    WIRING of the schedule (folding, dead-head removal, commuted up_conv, merged heads,
    skip adds) against the reference's golden outputs, on CPU, without a GPU;
  * quantize=True: rounds exactly where the HIP engine rounds (fp16 weights, fp16 activation
    storage after each op, fp32 heads) -> tight per-tensor check of the HIP kernels.
"""
import torch
import torch.nn.functional as F

from smap_amd.engine import OP_CONV, OP_STEM, OP_MAXPOOL, OP_UPADD, OP_HEADSUM, OP_STEMPOOL, OP_TAPSUM


def _q(x, on):
    return x.to(torch.float16).to(torch.float32) if on else x


def run_graph(g, imgs, quantize, keep=False):
    """g: Graph built with keep_ref=True.  Returns (hms, det_d, root_d[, dict name->NCHW tensor]).
    quantize=False computes in the dtype of `imgs` (fp32, or fp64 for a tight reference)."""
    dev = imgs.device
    dt = torch.float32 if quantize or imgs.dtype != torch.float64 else torch.float64
    blob = g.weight_blob()
    T = {}
    outs = {}
    up = lambda x, size: F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    for op in g.ops:
        p = op.p
        if op.kind in (OP_STEM, OP_STEMPOOL):
            w, b = p["w_ref"].to(dt).to(dev), p["b_ref"].to(dt).to(dev)
            x0 = imgs.to(dt)
            if quantize:                                 # the stem kernel rounds image and weights to fp16
                x0, w = _q(x0, True), _q(w, True)
            y = _q(F.relu(F.conv2d(x0, w, b, stride=2, padding=3)), quantize)
            T[op.out.name] = F.max_pool2d(y, 3, 2, 1) if op.kind == OP_STEMPOOL else y
        elif op.kind == OP_MAXPOOL:
            T[op.out.name] = F.max_pool2d(T[op.inp.name], 3, 2, 1)
        elif op.kind == OP_CONV:
            cin, cout, k = p["Cin"], p["Cout"], p["ksize"]
            if "head" in p:                              # whole Bottleneck (csrc/convb.hip): the leading 1x1 + ReLU in front of the 3x3; its
                hd = p["head"]                           # output is rounded like a stored activation (split precision: exact here)
                assert not quantize, "the whole-block kernel exists in split precision only (interpreted un-quantized)"
                x = _q(F.relu(F.conv2d(T[op.inp.name], hd["w_ref"].to(dt).to(dev), hd["b_ref"].to(dt).to(dev))), quantize)
            else:
                x = T[op.inp.name][:, p["in_c_off"]:p["in_c_off"] + cin]
            cin2 = p["cat"]["cin"] if "cat" in p else 0       # a second input concatenated along K (Graph.conv_cat): the block's shortcut conv
            if quantize:
                K = k * k * cin + cin2
                from smap_amd.engine import unpack_conv_weights      # the blob holds pre-tiled weight blocks
                wk = unpack_conv_weights(blob[p["w_off"]:p["w_off"] + p["cout_pad"] * K * 2].view(torch.float16), p["tile"], False,
                                         k, cin + cin2 if cin2 else cin, p["cout_pad"], pairs=p["w_pairs"])[0]
                w = wk[:cout, :k * k * cin].float().reshape(cout, k, k, cin).permute(0, 3, 1, 2).contiguous()
                w2 = wk[:cout, k * k * cin:].float().reshape(cout, cin2, 1, 1) if cin2 else None
                b = blob[p["bias_off"]:p["bias_off"] + p["cout_pad"] * 4].view(torch.float32)[:cout].clone()      # (cat: b1 + b2)
                b2 = None
            else:
                w, b = p["w_ref"].to(dt), p["b_ref"].to(dt)
                w2, b2 = (p["cat"]["w_ref"].to(dt), p["cat"]["b_ref"].to(dt)) if cin2 else (None, None)
            y = F.conv2d(x, w.to(dev), b.to(dev), stride=p["stride"], padding=p["pad"])
            if cin2 and p["cat"].get("relusum"):         # relu(conv1(x)) + relu(conv2(x2)): two activated convs, one tensor (Graph.conv_relusum)
                if quantize:
                    b2 = blob[p["cat"]["bias_off"]:p["cat"]["bias_off"] + p["cout_pad"] * 4].view(torch.float32)[:cout].clone()
                y = F.relu(y) + F.relu(F.conv2d(T[op.aux2.name][:, :cin2], w2.to(dev), b2.to(dev)))
            elif cin2:
                st2 = p["cat"]["stride"]
                y = y + F.conv2d(T[op.aux2.name][:, :cin2, ::st2, ::st2], w2.to(dev), b2.to(dev) if b2 is not None else None)
            if "tail" in p:                              # fused Bottleneck tail (csrc/convf.hip): relu(3x3) -> 1x1; the 3x3's
                tl = p["tail"]                           # output is rounded like a stored activation (hi|lo split = exact here)
                y = _q(F.relu(y), quantize)
                if quantize:
                    from smap_amd.engine import unpack_halo_rows, TAIL_BN
                    wk = unpack_halo_rows(blob[tl["w_off"]:tl["w_off"] + tl["cout_pad"] * cout * 2].view(torch.float16), TAIL_BN[p["tile"]],
                                          1, cout, tl["cout_pad"], False)[0]
                    w1 = wk[:tl["cout"]].float().view(tl["cout"], cout, 1, 1)
                    b1 = blob[tl["bias_off"]:tl["bias_off"] + tl["cout_pad"] * 4].view(torch.float32)[:tl["cout"]].clone()
                else:
                    w1, b1 = tl["w_ref"].to(dt), tl["b_ref"].to(dt)
                y = F.conv2d(y, w1.to(dev), b1.to(dev))
            if "short" in p:                             # first block of a layer: + the 1x1 shortcut conv of the block's input
                sh = p["short"]
                y = y + F.conv2d(T[op.inp.name], sh["w_ref"].to(dt).to(dev), sh["b_ref"].to(dt).to(dev))
            if op.res is not None:
                y = y + T[op.res.name]
            if op.aux:                                   # fused relu(u_skip(x) + bilinear(up_conv@low))
                y = y + up(T[op.aux[0].name], y.shape[-2:])
            if p["relu"]:
                y = F.relu(y)
            if op.add1 is not None:
                y = y + T[op.add1.name]
            if op.add2 is not None:
                y = y + T[op.add2.name]
            if "tap" in p:                               # tap-dot epilogue: nine per-pixel dot products of the (never stored) activation
                w3 = p["tap"]["w_ref"].to(dt).to(dev)    # [1, C, 3, 3] -> nine 1x1 convs, channel k = tap 3 kh + kw
                y = F.conv2d(y, w3[0].permute(1, 2, 0).reshape(9, -1, 1, 1))
            T[op.out.name] = y if p["out_fp32"] else _q(y, quantize)
            for sg, t in zip(p.get("segs", []), op.outs):        # N segments: further 1x1 convs on the same input, one output tensor each
                if quantize:
                    ws = wk[sg["n0"]:sg["n0"] + sg["cout"]].float().view(sg["cout"], k, k, cin).permute(0, 3, 1, 2).contiguous()
                    bs = blob[p["bias_off"]:p["bias_off"] + p["cout_pad"] * 4].view(torch.float32)[sg["n0"]:sg["n0"] + sg["cout"]].clone()
                else:
                    ws, bs = sg["w_ref"].to(dt), sg["b_ref"].to(dt)
                ys = F.conv2d(x, ws.to(dev), bs.to(dev))
                T[t.name] = _q(F.relu(ys) if sg["relu"] else ys, quantize)
        elif op.kind == OP_UPADD:
            a, t = T[op.inp.name], T[op.aux[0].name]
            y = a + up(t, a.shape[-2:])
            T[op.out.name] = _q(F.relu(y) if p["relu"] else y, quantize)
        elif op.kind == OP_TAPSUM:                       # out[y, x] = b + sum over taps of t[y + kh - 1, x + kw - 1][3 kh + kw]
            t = T[op.aux[0].name][:g.frames]
            tp = F.pad(t, (1, 1, 1, 1))
            H_, W_ = t.shape[-2:]
            y = p["b_ref"].to(dt).to(dev).reshape(1, 1, 1, 1) + sum(tp[:, kh * 3 + kw:kh * 3 + kw + 1, kh:kh + H_, kw:kw + W_] for kh in range(3) for kw in range(3))
            outs[p["ext_off"]] = y
        elif op.kind == OP_HEADSUM:
            size = (g.out_h, g.out_w)
            y = None
            for t in op.aux:
                u = up(T[t.name][:, :p["Cout"]], size)
                y = u if y is None else y + u
            if p.get("scale_hms"):                       # test.py:111-112 fused into the head sum (smap_op.scale_hms)
                y = y.clone()
                y[:, :p["n_kpt"]] /= 255
                y[:, p["n_kpt"]:] /= 127
            outs[p["ext_off"]] = y
    lay = g.out_layout
    r = (outs[lay["hms"][0]], outs[lay["det_d"][0]], outs[lay["root_d"][0]])
    return r + (T,) if keep else r
