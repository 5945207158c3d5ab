#!/usr/bin/env python3
"""SMAP inference hot path benchmark (BASELINE.json metric: frames/sec at 3x512x832).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of B=8 synthetic frames per GPU
(BASELINE config 3): SMAP backbone forward (HIP engine) -> /255,/127 scaling -> depth-aware
association -> 3D lifting -> device-to-host copy of the poses -> result records, run as a
two-stream pipeline (smap_amd/pipeline.py: the post-processing of batch k overlaps the backbone
of batch k+1; the pipeline is drained inside the timed region, so K steps = K complete batches); with N>1
every rank processes its own batch (weak scaling, frames shard with no data-path collective)
and the step ends with the RCCL all_gather of the per-frame JSON records (config 4).
Inputs are resident in HBM before the timed region.  Weights are the default random init
(torch.manual_seed(0)); with them the pelvis heat-map has no peaks, so grouping of the
network's own output is trivial -- to keep representative association work inside the timed
region, every step ALSO associates + lifts a resident batch of synthetic 8-person scenes.
"""
import argparse
import gc
import json
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALG_GFLOP_PER_FRAME = 300.628      # SURVEY.md 8d / BASELINE.md 3: inference-live conv FLOPs (2*MAC)
PEAK_F16_TFLOPS = 2500.0           # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
H, W = 512, 832


def records_from(pred_2d, pred_3d, root_z, counts, tag):
    out = []
    for i, P in enumerate(counts):
        if P == 0:
            continue                   # test.py:131-132
        out.append({"pred_2d": pred_2d[i, :P].tolist(), "pred_3d": pred_3d[i, :P].tolist(),
                    "root_d": root_z[i, :P].tolist(), "image_path": f"{tag}/{i}", "gt_3d": [], "gt_2d": []})
    return out


def usable_cpus(cap=64):
    """CPUs this process may really use: affinity mask, cgroup quota, capped (an OpenMP team larger
    than the quota makes every barrier a scheduling round-trip)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, cap))


_CPU_CHILD = r"""
import sys, time, os, json, torch
sys.path.insert(0, sys.argv[1])
from oracle.backbone_ref import smap_forward
from smap_amd.model.smap import SMAP
from helpers import make_cfg
threads, budget = int(sys.argv[2]), float(sys.argv[3])
torch.set_num_threads(threads)
torch.manual_seed(0)
sd = {k: v.float() for k, v in SMAP(make_cfg((128, 208))).state_dict().items()}
x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(1234))
with torch.no_grad():
    t0 = time.time(); smap_forward(sd, x); warm = time.time() - t0
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < budget and n < 10):
        smap_forward(sd, x); n += 1
    print(json.dumps({"sec_per_frame": (time.time() - t0) / n, "n": n, "warm": warm, "threads": torch.get_num_threads()}))
"""


def cpu_baseline(scenes, budget_s=12.0):
    """Reference CPU path restated (oracle/): torch-CPU backbone (oracle/backbone_ref.py) + C
    association + lifting (oracle/smap_oracle.c), on a bounded sample of the same workload.
    The backbone runs in a child process under a hard timeout so that a mis-sized OpenMP team on
    the box's host CPU cannot stall the bench."""
    import subprocess
    from oracle import oracle_lib as O
    threads = usable_cpus(32)
    bb = None
    try:
        r = subprocess.run([sys.executable, "-c", _CPU_CHILD, ROOT, str(threads), str(budget_s)],
                           capture_output=True, text=True, timeout=150, cwd=ROOT,
                           env={**os.environ, "PYTHONPATH": os.pathsep.join([ROOT, os.path.join(ROOT, "tests"),
                                                                             os.path.join(ROOT, "tests", "golden")])})
        bb = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:          # timeout / parse error: report it, never hang the bench
        bb = {"error": repr(e)[:200]}
    cam = np.array([1.0, 832, 512, 832, 512, 832, 832, 416, 256], np.float64)
    det = np.zeros((14, 128, 208), np.float32)
    t0, m = time.time(), 0
    for _ in range(3):
        for hms, rd in scenes[:8]:
            bodys, _, _ = O.connect(hms, rd)
            O.lift(bodys, det, rd, cam)
            m += 1
    t_as = (time.time() - t0) / max(m, 1)
    out = {"unit": "frames/sec", "cores": threads, "kind": "port", "assoc_ms": t_as * 1e3}
    if "sec_per_frame" in bb:
        out["value"] = 1.0 / (bb["sec_per_frame"] + t_as)
        out["backbone_ms"] = bb["sec_per_frame"] * 1e3
        out["sample"] = (f"{bb['n']} x backbone fwd of 1x3x{H}x{W} (oracle/backbone_ref.py, torch CPU fp32, "
                         f"{bb['threads']} threads) + {m} x association+lift of a synthetic 8-person frame "
                         f"(oracle/smap_oracle.c, 1 thread)")
    else:
        out["value"] = None
        out["sample"] = f"backbone child failed: {bb.get('error')}"
    return out


def forward_only(args, dev, rank, world, B):
    """BASELINE configs[1]: SMAP forward only (HIP conv engine), inputs resident, K forwards back to back on one
    stream, timed with HIP events around the K schedules and with the host clock around barrier + synchronize."""
    from helpers import make_cfg
    from model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval().to(dev)
    net.precision = args.precision
    eng = net.engine(B, H, W, dev)
    imgs = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    out = eng.new_output()
    if args.graph:
        replay = eng.capture(out)
        eng_run = lambda x, out=None: replay(x)
    else:
        eng_run = eng.run
    for _ in range(args.warmup):
        eng_run(imgs, out=out)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        eng_run(imgs, out=out)
    e1.record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ev = e0.elapsed_time(e1) * 1e-3
        achieved = ALG_GFLOP_PER_FRAME * B * args.steps / ev / 1e3
        print(json.dumps({
            "metric": "frames/sec at 3x512x832 (SMAP backbone forward only)", "value": B * world * args.steps / dt,
            "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"batch={B} x 3x512x832 per GPU, SMAP forward only, no association "
                                   f"(BASELINE configs[1] when batch=1)", "frames_per_step": B * world,
                       "launch": "one HIP graph per forward" if args.graph else "kernel by kernel"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS, "traffic": None,
                         "kernel": "all backbone launches of the schedule (HIP events around the K schedules)",
                         "algorithmic_gflop_per_frame": ALG_GFLOP_PER_FRAME}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


class _StandInPipeline:
    """The submit()/flush() protocol of smap_amd.pipeline.PosePipeline without a GPU: records of the batch submitted
    `depth` calls earlier, made of deterministic numbers."""

    def __init__(self, depth, B, rank):
        self.depth, self.B, self.rank, self.q, self.k = depth, B, rank, [], 0

    def _records(self, k):
        rng = np.random.default_rng(1000 * self.rank + k)
        return [{"pred_2d": rng.normal(0, 100, (3, 15, 4)).astype(np.float32).tolist(),
                 "pred_3d": rng.normal(0, 100, (3, 15, 4)).tolist(), "root_d": rng.normal(300, 50, 3).tolist(),
                 "image_path": f"r{self.rank}/b{k}/f{i}", "gt_3d": [], "gt_2d": []} for i in range(self.B)]

    def submit(self):
        self.q.append(self._records(self.k))
        self.k += 1
        return self.q.pop(0) if len(self.q) > self.depth else None

    def flush(self):
        out = [r for recs in self.q for r in recs]
        self.q = []
        return out or None


def dry_run(args):
    """bench.py's distributed control flow on CPU (gloo): same sequence of collectives per rank as the GPU run."""
    from smap_amd.dist import gather_bytes
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        dist.init_process_group("gloo")
    pipe = _StandInPipeline(args.depth, args.batch, rank)
    last = [None]

    def finish(recs):
        if recs is None:
            return
        last[0] = [recs] if world == 1 else gather_bytes(pickle.dumps(recs, protocol=pickle.HIGHEST_PROTOCOL), "cpu")

    for _ in range(args.warmup):
        finish(pipe.submit())
    finish(pipe.flush())
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        finish(pipe.submit())
    finish(pipe.flush())
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    got = last[0] if world == 1 else [pickle.loads(b) for b in last[0]]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU): control flow only", "value": args.batch * world * args.steps / dt,
                          "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "stand-in pipeline", "ranks_in_last_gather": len(got),
                                     "last_paths": [g[-1]["image_path"] for g in got]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=2,
                    help="backbones in flight per GPU (2 = two streams/arenas: batch k+1 fills the CUs that batch k's "
                         "low-resolution layers leave idle)")
    ap.add_argument("--forward-only", action="store_true",
                    help="BASELINE configs[1] (use with --batch 1): time the backbone forward alone, no association / "
                         "lifting / result records; prints its own JSON line")
    ap.add_argument("--graph", action="store_true",
                    help="with --forward-only: replay the schedule as one HIP graph (BackboneEngine.capture) instead of "
                         "launching its ~208 kernels one by one")
    ap.add_argument("--refine", action="store_true",
                    help="BASELINE configs[4]: also run the RefineNet post-refinement (model/refinenet.py) on every pose")
    ap.add_argument("--precision", choices=("f16", "x3"), default=os.environ.get("SMAP_PRECISION", "f16"),
                    help="backbone arithmetic: x3 = fp16 hi/lo pairs + three MFMAs per K step (meets the reference's fp32 "
                         "results end to end); f16 = fp16 storage (fast mode, ~1e-3 relative error on the maps)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the same control flow (pipeline protocol, per-step gather, MAX over ranks, one JSON line on "
                         "rank 0) with a stand-in pipeline and the gloo backend -- what the multi-rank CPU test runs")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm; one process per GPU

    import dapalib
    from helpers import make_cfg, synth_scene
    from model.smap import SMAP
    from smap_amd.dist import gather_bytes

    B = args.batch
    if args.forward_only:
        return forward_only(args, dev, rank, world, B)
    from smap_amd.pipeline import PosePipeline
    from exps.stage3_root2.config import cfg as run_cfg
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval().to(dev)
    net.precision = args.precision
    refine_w = None
    if args.refine:
        from model.refinenet import RefineNet
        torch.manual_seed(1)
        refine_w = RefineNet().eval().folded(dev)
    pipe = PosePipeline(net, run_cfg, B, H, W, dev, refine_weights=refine_w, n_extra=1, depth=args.depth)
    imgs = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    scenes = [synth_scene(8, seed=1000 * rank + i)[:2] for i in range(B)]
    s_hms = torch.from_numpy(np.stack([s[0] for s in scenes])).to(dev)
    s_rd = torch.from_numpy(np.stack([s[1] for s in scenes])).to(dev)
    cams = np.tile(np.array([1.0, 832, 512, 832, 512, 832, 832, 416, 256], np.float64), (B, 1))
    tags = [f"r{rank}/f{i}" for i in range(B)]
    last = [None]

    def finish(recs):
        """Records of a completed batch -> (RCCL) all_gather of every rank's serialised records to every rank, on the comm
        stream (BASELINE configs[3]).  Serialisation is pickle, like the reference's gather helper (lib/utils/comm.py:
        57-59); the gathered payloads stay bytes -- decoding 8 ranks' records on every rank every step is not part of
        the path (rank 0 writes the JSON file once, after the run: test.py:147-151).  Nothing to do at N = 1."""
        if recs is None:
            return
        if world == 1:
            last[0] = recs
            return
        payload = pickle.dumps(recs, protocol=pickle.HIGHEST_PROTOCOL)
        with torch.cuda.stream(pipe.s_comm):
            last[0] = gather_bytes(payload, dev)

    host = {"submit": 0.0, "finish": 0.0}

    def step(timed):
        # backbone(k) on one stream; association+lift+D2H of batch k on another; records of batch k-1 on the host
        t0 = time.perf_counter()
        recs = pipe.submit(imgs, cams, tags, extra=[("synth", s_hms, s_rd, None)], time_backbone=timed)
        t1 = time.perf_counter()
        finish(recs)
        if timed:
            host["submit"] += t1 - t0
            host["finish"] += time.perf_counter() - t1
            host.setdefault("steps", []).append((t1 - t0) * 1e3)

    for _ in range(args.warmup):
        step(False)
    finish(pipe.flush())
    # The set-up leaves ~2 M long-lived Python objects behind (state dict, folded weights, op lists); a full cyclic-GC
    # pass over them costs ~35 ms and lands on some step or other of the timed loop depending on K and W.  Freeze them
    # into the permanent generation: collections during the loop then only look at what the loop itself allocates.
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    t_loop = time.perf_counter() - t0
    finish(pipe.flush())                      # the K-th batch is complete inside the timed region
    t_flush = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bb_ms = [e0.elapsed_time(e1) for e0, e1 in pipe.bb_events]
    # idle time of a backbone stream between two consecutive schedules (batch k-depth's end -> batch k's start)
    ev = pipe.bb_events
    gap_ms = [ev[k - args.depth][1].elapsed_time(ev[k][0]) for k in range(args.depth, len(ev))]
    last = [last[0]] if world == 1 else [pickle.loads(b) for b in last[0]]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        frames = B * world * args.steps
        fps = frames / dt
        bb = float(np.mean(bb_ms)) * 1e-3                        # HIP-event span of one backbone schedule
        if args.depth > 1:       # backbones overlap: rate = algorithmic work of the timed region / its duration
            achieved = ALG_GFLOP_PER_FRAME * B * args.steps / dt / 1e3
        else:
            achieved = ALG_GFLOP_PER_FRAME * B / bb / 1e3        # TFLOP/s over the whole backbone schedule
        traffic, traffic_src = None, None                        # HBM bytes per batch from committed PMC passes
        tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tj) and B == 8:
            t = json.load(open(tj))
            traffic = t["hbm_read_bytes_per_batch"] + t["hbm_write_bytes_per_batch"]
            traffic_src = t["source"]
        out = {
            "metric": "frames/sec at 3x512x832 (SMAP backbone + depth-aware association + 3D lifting)",
            "value": fps, "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"batch={B} x 3x512x832 per GPU, full SMAP + depth-aware PAF association "
                                   f"+ lifting{' + RefineNet (configs[4])' if args.refine else ''} "
                                   f"(BASELINE configs[2]; configs[3] when n_gpus=8)",
                       "frames_per_step": B * world, "persons_in_last_step": sum(len(r) for r in last),
                       "arithmetic": "backbone fp16 storage / fp32 MFMA accumulate, heads fp32; association fp32 (+f64 "
                                     "where the reference is); lifting f64",
                       "pipeline": f"post-processing of batch k overlaps later backbones; {args.depth} backbone(s) in flight",
                       "host_ms_per_step": {k: v / args.steps * 1e3 for k, v in host.items() if k != "steps"},
                       "host_submit_ms_quantiles": [float(np.percentile(host["steps"], q)) for q in (0, 10, 50, 90, 100)],
                       "host_submit_slowest_step": int(np.argmax(host["steps"])),
                       "host_timeline_ms": {"submit_loop_done": t_loop * 1e3, "flush_done": t_flush * 1e3, "total": dt * 1e3}},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "conv_igemm_kernel (all backbone launches; HIP events: per-schedule span below, "
                                   "rate = algorithmic FLOPs of the timed region / its duration when depth > 1)",
                         "backbone_ms_per_batch": bb * 1e3,
                         "backbone_stream_idle_ms_between_batches": float(np.mean(gap_ms)) if gap_ms else None,
                         "first_backbone_start_to_last_end_ms": ev[0][0].elapsed_time(max(ev[-1][1], ev[-2][1], key=lambda e: ev[0][0].elapsed_time(e))) if len(ev) > 1 else None,
                         "algorithmic_gflop_per_frame": ALG_GFLOP_PER_FRAME},
        }
        if not args.no_cpu_baseline and world == 1:              # reported at N = 1 only (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(scenes)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
