#!/usr/bin/env python3
"""Where do the CUs spend a pipelined step?  Runs the bench's two-backbones-in-flight pipeline with the timeline build of the
library (every workgroup of every conv launch leaves start / end on the device-wide 100 MHz clock + the CU it ran on) and
reconstructs, for a window of whole steps, without a profiler in the way:
  * per-CU occupancy (fraction of the window with at least one conv workgroup resident, mean resident workgroups),
  * per-stream launch spans and the gaps between dependent launches,
  * how much of the window has launches of BOTH streams active.

    python tools/build_ablate.py --timeline
    SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_timeline.so python tools/trace_pipeline.py [--steps 4] [--depth 2]
"""
import argparse
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

SLICE = 4 * (1 + 16384)
TICK_US = 0.01                   # s_memrealtime: 100 MHz


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--cap", type=int, default=2000)
    args = ap.parse_args()
    assert "timeline" in os.environ.get("SMAP_HIP_LIB", ""), "run with SMAP_HIP_LIB=<libsmap_hip_timeline.so>"
    dev = torch.device("cuda:0")
    buf = torch.zeros(args.cap * SLICE, dtype=torch.int64, device=dev)
    os.environ["SMAP_TIMELINE_PTR"] = str(buf.data_ptr())
    os.environ["SMAP_TIMELINE_CAP"] = "0"                      # nothing recorded during warm-up
    from helpers import make_cfg
    from model.smap import SMAP
    from smap_amd import lib as L
    from smap_amd.pipeline import PosePipeline
    from exps.stage3_root2.config import cfg as run_cfg
    lib = L.load()
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval().to(dev)
    B, H, W = 8, 512, 832
    pipe = PosePipeline(net, run_cfg, B, H, W, dev, depth=args.depth)
    imgs = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1234)).to(dev)
    cams = np.tile(np.array([1.0, 832, 512, 832, 512, 832, 832, 416, 256], np.float64), (B, 1))
    tags = [f"f{i}" for i in range(B)]
    for _ in range(6):
        pipe.submit(imgs, cams, tags)
    pipe.flush()
    torch.cuda.synchronize()
    os.environ["SMAP_TIMELINE_CAP"] = str(args.cap)
    lib.smap_timeline_reset()
    for _ in range(args.steps + 2 * args.depth):
        pipe.submit(imgs, cams, tags)
    pipe.flush()
    torch.cuda.synchronize()
    n = lib.smap_timeline_count()
    raw = buf[:n * SLICE].cpu().numpy().reshape(n, 1 + 16384, 4)
    launches = []
    for k in range(n):
        rows = raw[k, 1:]
        rows = rows[rows[:, 0] != 0]
        if len(rows) == 0:
            continue
        op, _, stream, tile = raw[k, 0]
        launches.append(dict(op=int(op), stream=int(stream), tile=int(tile), rows=rows,
                             s=int(rows[:, 0].min()), e=int(rows[:, 1].max())))
    streams = sorted({l["stream"] for l in launches})
    per_fwd = max(l["op"] for l in launches) + 1
    print(f"{n} conv launches recorded, {len(launches)} with stamps, {len(streams)} backbone streams, ops per forward {per_fwd}")
    # window: from the start of the (depth+1)-th forward to the end of the (depth+steps)-th: whole steps in steady state
    starts = sorted(l["s"] for l in launches if l["op"] == min(x["op"] for x in launches))
    t0, t1 = starts[args.depth], starts[args.depth + args.steps]
    win = (t1 - t0) * TICK_US
    print(f"window: {args.steps} steps = {win:.1f} us  ({win / args.steps:.1f} us per step)")
    inwin = [l for l in launches if l["e"] > t0 and l["s"] < t1]
    # ---- per-stream spans and gaps
    for st in streams:
        ls = sorted((l for l in inwin if l["stream"] == st), key=lambda l: l["s"])
        span = sum(min(l["e"], t1) - max(l["s"], t0) for l in ls) * TICK_US
        gaps = [(b["s"] - a["e"]) * TICK_US for a, b in zip(ls, ls[1:])]
        gpos = [g for g in gaps if g > 0]
        print(f"stream {st & 0xffff:04x}: {len(ls)} launches, launch spans cover {100 * span / win:.1f} % of the window; "
              f"gap between consecutive launches: median {np.median(gaps):.2f} us, mean {np.mean(gaps):.2f} us, "
              f"sum of positive gaps {sum(gpos):.0f} us ({100 * sum(gpos) / win:.1f} %)")
    both = 0
    iv = {st: [(max(l["s"], t0), min(l["e"], t1)) for l in inwin if l["stream"] == st] for st in streams}
    if len(streams) >= 2:
        a, b = iv[streams[0]], iv[streams[1]]
        ev = [(s, 1, 0) for s, e in a] + [(e, -1, 0) for s, e in a] + [(s, 1, 1) for s, e in b] + [(e, -1, 1) for s, e in b]
        ev.sort()
        cnt, last = [0, 0], t0
        for t, d, w in ev:
            if cnt[0] > 0 and cnt[1] > 0:
                both += t - last
            cnt[w] += d
            last = t
        anyt = union_len(a + b)
        print(f"launches of both streams active: {100 * both * TICK_US / win:.1f} % of the window; of at least one: "
              f"{100 * anyt * TICK_US / win:.1f} %")
    # ---- per-CU occupancy
    cu_iv = collections.defaultdict(list)
    wg_time = 0
    for l in inwin:
        r = l["rows"]
        s = np.maximum(r[:, 0], t0)
        e = np.minimum(r[:, 1], t1)
        ok = e > s
        hw, xcc = r[:, 2], r[:, 3] & 0xf
        key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
        for k_, s_, e_ in zip(key[ok], s[ok], e[ok]):
            cu_iv[int(k_)].append((int(s_), int(e_)))
        wg_time += int((e[ok] - s[ok]).sum())
    busy = np.array([union_len(v) for v in cu_iv.values()]) * TICK_US
    print(f"CUs seen: {len(cu_iv)}; a CU has >= 1 conv workgroup resident {100 * busy.mean() / win:.1f} % of the window "
          f"(min {100 * busy.min() / win:.1f} %, max {100 * busy.max() / win:.1f} %); mean resident conv workgroups per CU "
          f"{wg_time * TICK_US / win / max(len(cu_iv), 1):.2f}")


if __name__ == "__main__":
    main()
