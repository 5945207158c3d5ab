#!/usr/bin/env python3
"""SMAP inference hot path benchmark (BASELINE.json metric: frames/sec at 3x512x832).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of B=8 synthetic frames per GPU (BASELINE configs[2]):
SMAP backbone forward (HIP engine) -> /255,/127 scaling -> depth-aware association -> 3D lifting -> device-to-host
copy of the poses -> result records, run as a multi-stream pipeline (smap_amd/pipeline.py: the post-processing of batch
k overlaps later backbones; the pipeline is drained inside the timed region, so K steps = K complete batches).
With N > 1 every rank processes its own batches (weak scaling, frames shard with no data-path collective) and the run
ends -- inside the timed region -- with ONE RCCL all_gather of every rank's result records (configs[3]; the reference
gathers once per run as well: exps/stage3_root2/test.py + lib/utils/comm.py:47-87).
Inputs are resident in HBM before the timed region.

Arithmetic (--precision): "x3" (default) = fp16 hi/lo pairs with three MFMAs per K step: the mode whose results meet
the reference's fp32 path end to end (3D joints within 1e-3 m, same peaks and limbs: `config.e2e_parity`, computed in
this very run against the CPU reference on the same images); "f16" = fp16 storage, ~2x faster, ~0.3 cm mean joint error.

Workload: recipe weights with calibrated heads (benchkit/workload.py: ~24 peaks per key-point channel, ~22 skeletons
per frame, root depth ~3 m), so the association of the NETWORK's own output is real work; in addition every step
associates + lifts a resident batch of synthetic scenes whose person count rotates over K in {0, 2, 8, 20}
(SURVEY.md 8d config 3), timed per K with HIP events on the post-processing stream.
"""
import argparse
import gc
import json
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALG_GFLOP_PER_FRAME = 300.628      # SURVEY.md 8d / BASELINE.md 3: inference-live conv FLOPs (2*MAC)
PEAK_F16_TFLOPS = 2500.0           # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
SURVEY_GB_PER_FRAME_FP16 = 1.854   # minimum unfused activation + weight traffic per frame, fp16 (3.707 GB at fp32 size)
PEAK_HBM_GBPS = 8000.0             # HBM3E (same guide); the measured ceilings are lower and reported beside it
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
# Reference path on the host cores (oracle/: fp32 torch-CPU SMAP forward + C association + lifting), frame by frame on
# the SAME images and weights as the GPU run.  Serves two purposes with one pass: the cpu_baseline timing and the
# per-frame results the end-to-end parity figures are computed from (benchkit/parity.py).
import sys, time, os, json, pickle, torch, numpy as np
root, threads, frames, seed, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
flip = "flip" in sys.argv[6:]
refine = None
if "refine" in sys.argv[6:]:                                   # BASELINE configs[4]: the same RefineNet (seed 1 default init) folded for the oracle
    from smap_amd.model.refinenet import RefineNet
    torch.manual_seed(1)
    wt, bs = RefineNet().eval().folded("cpu")
    refine = ([w.t().contiguous().numpy() for w in wt], [b.numpy() for b in bs])
sys.path.insert(0, root)
from benchkit import parity
from benchkit.workload import make_cfg, people_state_dict, PEOPLE_CAM
from smap_amd.model.smap import SMAP
torch.set_num_threads(threads)
torch.manual_seed(0)
sd = people_state_dict(SMAP(make_cfg((128, 208))).state_dict(), "smooth")
x = torch.randn(8, 3, 512, 832, generator=torch.Generator().manual_seed(seed))[:frames]
cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (frames, 1))
fp = None
if flip:                                                       # the reference's mirror tables (dataset/data_settings.py:22,33-34)
    from exps.stage3_root2.config import cfg
    fp = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [cfg.DATASET.KEYPOINT.NUM + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
parity.reference_path(sd, x[:1], cams[:1], flip_pair=fp, refine=refine)       # warm-up (oneDNN primitive cache, page faults)
t0 = time.time()
ref = parity.reference_path(sd, x, cams, flip_pair=fp, refine=refine)   # --flip: the reference's two forwards + channel loop (test.py:55-70)
dt = (time.time() - t0) / frames
# association + lifting alone: single thread (the reference's own path: its OpenMP pragmas are commented out,
# association.cpp:79,100) and frame-parallel over the cores
from oracle import oracle_lib as O
from concurrent.futures import ThreadPoolExecutor
def assoc(r):
    b, _, _ = O.connect(r["hms"], r["root_d"], 2, True)
    O.lift(b, r["det_d"], r["root_d"], cams[0])
t0 = time.time()
for r in ref: assoc(r)
t_as1 = (time.time() - t0) / frames
with ThreadPoolExecutor(min(threads, frames)) as ex:
    t0 = time.time(); list(ex.map(assoc, ref * 4)); t_asn = (time.time() - t0) / (4 * frames)
pickle.dump({"ref": ref, "sec_per_frame": dt, "assoc_1thread_ms": t_as1 * 1e3, "assoc_all_cores_ms": t_asn * 1e3,
             "threads": torch.get_num_threads(), "frames": frames}, open(out, "wb"))
"""


def cpu_reference(frames, seed, flip=False, refine=False):
    """Runs _CPU_CHILD under a hard timeout (a mis-sized OpenMP team on the box's host CPU must not stall the bench).
    Returns its result dict or {"error": ...}."""
    import subprocess
    import tempfile
    threads = usable_cpus(32)
    out = os.path.join(tempfile.mkdtemp(prefix="smap_bench_"), "ref.pkl")
    try:
        r = subprocess.run([sys.executable, "-c", _CPU_CHILD, ROOT, str(threads), str(frames), str(seed), out] + (["flip"] if flip else []) + (["refine"] if refine else []),
                           capture_output=True, text=True, timeout=240, cwd=ROOT)
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        res = pickle.load(open(out, "rb"))
        res["cores"] = threads
        return res
    except Exception as e:          # timeout / parse error: report it, never hang the bench
        return {"error": repr(e)[:200]}


def forward_only(args, dev, rank, world, B):
    """BASELINE configs[1]: SMAP forward only (HIP conv engine), inputs resident, K forwards back to back on one
    stream, timed with HIP events around the K schedules and with the host clock around barrier + synchronize."""
    from benchkit.workload import make_cfg
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
        t = torch.tensor([dt], dtype=torch.float64, device=getattr(args, "cdev", dev))
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
                       "launch": "one HIP graph per forward" if args.graph else "kernel by kernel",
                       "conv_launches_per_forward": sum(1 for op in eng.graph.ops if op.kind == 0),
                       "split_k_launches": sum(1 for op in eng.graph.ops if op.p.get("ksplit", 1) > 1)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS, "traffic": None,
                         "flops_executed_per_algorithmic_flop": 3 if args.precision == "x3" else 1,
                         "pipe_frac": (3 if args.precision == "x3" else 1) * achieved / PEAK_F16_TFLOPS,
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
    """bench.py's distributed control flow on CPU (gloo): same sequence of collectives per rank as the GPU run -- records
    collected over the timed steps, ONE gather at the end of the run, barrier, MAX all-reduce, one JSON line on rank 0."""
    from smap_amd.dist import gather_bytes
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        dist.init_process_group("gloo")
    pipe = _StandInPipeline(args.depth, args.batch, rank)
    for _ in range(args.warmup):
        pipe.submit()
    pipe.flush()
    collected = []
    if world > 1:
        dist.barrier()
    t0, cpu0 = time.perf_counter(), time.process_time()
    for _ in range(args.steps):
        collected.extend(pipe.submit() or [])
    collected.extend(pipe.flush() or [])
    got = [collected]
    if world > 1:
        got = [pickle.loads(b) for b in gather_bytes(pickle.dumps(collected, protocol=pickle.HIGHEST_PROTOCOL), "cpu")]
        dist.barrier()
    dt = time.perf_counter() - t0
    cpu_ms = (time.process_time() - cpu0) / args.steps * 1e3
    per_rank_cpu = [cpu_ms]
    per_rank_dt = [dt]
    if world > 1:
        dts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]      # every rank's own clock (the GPU run reports the same block)
        dist.all_gather(dts, torch.tensor([dt], dtype=torch.float64))
        per_rank_dt = [float(x.item()) for x in dts]
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        hc = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]       # host CPU per step of EVERY rank, as the GPU run reports it
        dist.all_gather(hc, torch.tensor([cpu_ms], dtype=torch.float64))
        per_rank_cpu = [float(h.item()) for h in hc]
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU): control flow only", "value": args.batch * world * args.steps / dt,
                          "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "stand-in pipeline", "ranks_in_gather": len(got),
                                     "host_ms_per_step": {"process_cpu_per_rank": per_rank_cpu},
                                     "host_cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                                     "records_per_rank": [len(g) for g in got],
                                     "per_rank": {"frames_per_sec": [args.batch * args.steps / x for x in per_rank_dt],
                                                  "ms_per_step_max": max(per_rank_dt) / args.steps * 1e3,
                                                  "ms_per_step_min": min(per_rank_dt) / args.steps * 1e3},
                                     "first_paths": [g[0]["image_path"] for g in got],
                                     "last_paths": [g[-1]["image_path"] for g in got]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def thread_cpu_seconds():
    """{tid: (name, cpu seconds)} of every native thread of this process (/proc): who burns the host while the GPU works."""
    out = {}
    try:
        tck = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            f = open(f"/proc/self/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tck)
    except Exception:
        pass
    return out


def pin_host_threads(local, world):
    """One process per GPU on a shared host: give each rank its own slice of the allowed CPUs and a matching thread
    count, so that eight submit threads + their OpenMP/MKL pools do not fight over the same cores."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        if world > 1 and len(cpus) >= 2 * world:
            per = len(cpus) // world
            os.sched_setaffinity(0, cpus[local * per:(local + 1) * per])
            cpus = cpus[local * per:(local + 1) * per]
        n = max(1, min(len(cpus), 16))
    except (AttributeError, OSError):
        n = 4
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    torch.set_num_threads(n)
    return n


def spawn_or_check_world(args):
    """--gpus N is the contract, WORLD_SIZE is what a launcher set.  `python bench.py --gpus 8` without a launcher re-executes
    itself under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1); a launcher whose world size
    disagrees with --gpus is an error -- a line that says n_gpus: 1 for a --gpus 8 request would be worse than no line."""
    world = os.environ.get("WORLD_SIZE")
    if world is None and args.gpus > 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without a launcher: re-executing as %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    if int(world or 1) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world or 1}: launch with --nproc-per-node {args.gpus} "
                         f"(or drop the launcher: bench.py spawns its own ranks)")
    if not args.dry_run and not os.environ.get("SMAP_BENCH_SHARE_GPU"):
        import torch as _t
        if _t.cuda.is_available() and _t.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {_t.cuda.device_count()} GPU(s) are visible")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU reference pass (no cpu_baseline, no config.e2e_parity): kernel experiments only")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct resident 8-frame batches the timed steps rotate over (1 = the same frames every step, as rounds 1-5 ran)")
    ap.add_argument("--launch-frames", type=int, default=16,
                    help="frames per backbone launch the pipeline aims for: consecutive steps' batches are coalesced up to it "
                         "(0 = one launch per step)")
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
    ap.add_argument("--flip", action="store_true",
                    help="flip-TTA (test.py:55-70, the reference's shipped --do_flip 1): every frame also runs mirrored, inside "
                         "the same schedule (2B-frame batch; stem reads the mirror by index, head sum merges)")
    ap.add_argument("--precision", choices=("f16", "x3"), default=os.environ.get("SMAP_PRECISION", "x3"),
                    help="backbone arithmetic: x3 = fp16 hi/lo pairs + three MFMAs per K step (meets the reference's fp32 "
                         "results end to end); f16 = fp16 storage (fast mode, ~1e-3 relative error on the maps)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the same control flow (pipeline protocol, end-of-run gather, MAX over ranks, one JSON line on "
                         "rank 0) with a stand-in pipeline and the gloo backend -- what the multi-rank CPU test runs")
    args = ap.parse_args()
    spawn_or_check_world(args)
    if args.dry_run:
        return dry_run(args)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    host_threads = pin_host_threads(local, world)
    # Test hooks (tests/test_entry_gpu.py runs this very function with two ranks on a ONE-GPU box): SMAP_BENCH_SHARE_GPU=1
    # puts every rank on cuda:0, SMAP_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); the
    # collectives then run on CPU tensors.  The driver's multi-GPU runs use neither.
    if os.environ.get("SMAP_BENCH_SHARE_GPU"):
        local = 0
    backend = os.environ.get("SMAP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective payloads live
    # SMAP_FORCE_GATHER=1 (tests/test_entry_gpu.py): a ONE-rank run that still initialises the process group and sends its records through
    # the end-of-run all_gather -- the RCCL path of configs[3] on a one-GPU box
    force_gather = world == 1 and os.environ.get("SMAP_FORCE_GATHER", "") == "1"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm; one process per GPU
        else:
            dist.init_process_group(backend)
    elif force_gather:
        from smap_amd.dist import init_single_rank_group
        init_single_rank_group(backend, dev)

    from benchkit.workload import PEOPLE_CAM, make_cfg, people_state_dict, synth_scene
    from model.smap import SMAP
    from smap_amd.dist import gather_bytes

    B = args.batch
    args.cdev = cdev
    if args.forward_only:
        return forward_only(args, dev, rank, world, B)
    from smap_amd.pipeline import make_pipeline
    from exps.stage3_root2.config import cfg as run_cfg
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    net.load_state_dict(people_state_dict(net.state_dict(), "smooth"))
    net = net.to(dev)
    net.precision = args.precision
    refine_w = None
    if args.refine:
        from model.refinenet import RefineNet
        torch.manual_seed(1)
        refine_w = RefineNet().eval().folded(dev)
    SEED = 1234                      # frame 0 of batch 0 on rank 0 is the frame the heads were calibrated on
    # The timed steps ROTATE over `--rotate` (default 4) distinct resident batches (round 5 ran the same 8 frames every step: 40 MB of input
    # that never left the 256 MB Infinity Cache); with outputs and arenas > 256 MB are in play per step.  Rank r draws its own frames
    # (seed + 104729 r): the gathered records then prove rank order by CONTENT.  Batch 0 of rank 0 = the frames the CPU reference runs.
    NB = max(1, args.rotate)

    def batch_images(j):
        x = torch.randn(8, 3, H, W, generator=torch.Generator().manual_seed(SEED + 7919 * j + 104729 * rank))
        return (x[:B] if B <= 8 else x.repeat((B + 7) // 8, 1, 1, 1)[:B]).to(dev)
    img_batches = [batch_images(j) for j in range(NB)]
    imgs = img_batches[0]
    cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (B, 1))

    # ---- end-to-end parity of THIS configuration (outside the timed region; rank 0, N = 1 reports it).  The CPU child
    #      computes the reference path while the GPU side is being set up.
    parity_out, cpu_ref, hip_frames = None, None, None
    want_ref = rank == 0 and world == 1 and not args.no_cpu_baseline
    if want_ref:
        import concurrent.futures as cf
        nref = min(B, 8)
        ref_future = cf.ThreadPoolExecutor(1).submit(cpu_reference, nref, SEED, args.flip, args.refine)
        from benchkit import parity
        # the parity block reports THIS configuration: with --flip both sides run the flip-TTA (HIP: mirror + merge inside the schedule)
        fp = (list(run_cfg.DATASET.KEYPOINT.FLIP_ORDER) + [run_cfg.DATASET.KEYPOINT.NUM + c for c in run_cfg.DATASET.PAF.FLIP_CHANNEL]) if args.flip else None
        hip_frames = parity.hip_path(net, imgs[:nref] if B == nref else imgs, cams, flip_pair=fp, refine=refine_w)[:nref] if B >= nref else None

    # batches of <= 8 frames are run two (or more) to a backbone launch: --launch-frames (smap_amd/pipeline.py::make_pipeline)
    pipe = make_pipeline(net, run_cfg, B, H, W, dev, launch_frames=args.launch_frames, refine_weights=refine_w, n_extra=1,
                         depth=args.depth, numpy_records=True, do_flip=args.flip)
    KS = (0, 2, 8, 20)               # synthetic scenes, person count rotating over the steps
    synth = {}
    for K in KS:
        sc = [synth_scene(K, seed=1000 * rank + 10 * K + i)[:2] for i in range(B)]
        synth[K] = (torch.from_numpy(np.stack([s_[0] for s_ in sc])).to(dev), torch.from_numpy(np.stack([s_[1] for s_ in sc])).to(dev))
    batch_tags = [[f"r{rank}/b{j}/f{i}" for i in range(B)] for j in range(NB)]
    collected = []                   # this rank's records of the timed steps (gathered once, at the end of the run)
    # the LAST timed launch (whose maps the parity block reads back) starts with batch 0: step k runs batch (k - first) mod NB
    group = max(1, pipe.frames_per_launch // (B * (2 if args.flip else 1)))
    first = (args.steps - group) % NB if args.steps >= group else 0

    host = {"submit": 0.0}
    step_no = [0]

    def step(timed):
        # backbone(k) on one stream; association+lift+D2H of batch k on another; records of earlier batches on the host
        K = KS[step_no[0] % len(KS)]
        j = (step_no[0] - first) % NB
        step_no[0] += 1
        t0 = time.perf_counter()
        recs = pipe.submit(img_batches[j], cams, batch_tags[j], extra=[(f"synthK{K}", synth[K][0], synth[K][1], None)], time_backbone=timed)
        if timed:
            dt = time.perf_counter() - t0
            host["submit"] += dt
            host.setdefault("steps", []).append(dt * 1e3)
            if recs:
                collected.extend(recs)

    if want_ref:
        cpu_ref = ref_future.result()         # the CPU pass ran beside the GPU set-up; it is over before anything is timed
    for _ in range(args.warmup):
        step(False)
    pipe.flush()
    step_no[0] = 0
    # The set-up leaves ~2 M long-lived Python objects behind (state dict, folded weights, op lists); a full cyclic-GC
    # pass over them costs ~35 ms and lands on some step or other of the timed loop depending on K and W.  Freeze them
    # into the permanent generation: collections during the loop then only look at what the loop itself allocates.
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    from benchkit.clocks import ClockSampler
    sampler = ClockSampler(local)             # shader clock + package power the box holds during the timed region (sysfs reads, 50 Hz)
    sampler.__enter__()
    cpu0 = time.process_time()
    thr0 = thread_cpu_seconds()
    wait0 = pipe.wait_s
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    t_loop = time.perf_counter() - t0
    wait_loop = pipe.wait_s - wait0
    tail = pipe.flush()                       # the K-th batch is complete inside the timed region
    if tail:
        collected.extend(tail)
    t_flush = time.perf_counter() - t0
    gathered = None
    if world > 1 or force_gather:             # ONE gather per run: every rank's records to every rank (RCCL, comm stream)
        payload = pickle.dumps(collected, protocol=pickle.HIGHEST_PROTOCOL)
        with torch.cuda.stream(pipe.s_comm):
            gathered = gather_bytes(payload, cdev)
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sampler.__exit__()
    clocks = sampler.summary(t0, t0 + dt)
    cpu_ms_per_step = (time.process_time() - cpu0) / args.steps * 1e3
    thr1 = thread_cpu_seconds()
    busiest = sorted(((n, (c - thr0.get(t, (n, 0.0))[1]) / args.steps * 1e3) for t, (n, c) in thr1.items()), key=lambda x: -x[1])[:4]
    bb_ms = [e0.elapsed_time(e1) for e0, e1 in pipe.bb_events]
    # idle time of a backbone stream between two consecutive schedules (batch k-depth's end -> batch k's start)
    ev = pipe.bb_events
    gap_ms = [ev[k - args.depth][1].elapsed_time(ev[k][0]) for k in range(args.depth, len(ev))]
    post_us = {}
    for tag, p0, p1 in pipe.post_events:
        post_us.setdefault(tag, []).append(p0.elapsed_time(p1) * 1e3)
    # ---- what the TIMED launches produced (pipelined, depth `--depth`, coalesced): the maps of the last timed launch are still
    #      in its output buffers; association + lifting re-run on them outside the timed region give the per-frame data the
    #      parity block compares with the CPU reference (and must reproduce the timed records bit for bit)
    timed_frames = None
    if want_ref:
        maps = pipe.last_maps()
        if maps is not None:
            nfr = min(nref, maps[0].shape[0])
            timed_frames = parity.frames_from_maps(maps[0][:nfr], maps[1][:nfr], maps[2][:nfr], cams[:nfr], refine=refine_w)
    # ---- the same configuration with ONE backbone launch per step (--launch-frames 0), timed in the same process
    fps_lf0 = None
    small = getattr(pipe, "_small", None)
    if world == 1 and small is not None and not os.environ.get("SMAP_BENCH_NO_LF0"):
        def step0():
            K = KS[step_no[0] % len(KS)]
            j = step_no[0] % NB
            step_no[0] += 1
            small.submit(img_batches[j], cams, batch_tags[j], extra=[(f"synthK{K}", synth[K][0], synth[K][1], None)])
        for _ in range(args.warmup):
            step0()
        small.flush()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step0()
        small.flush()
        torch.cuda.synchronize()
        fps_lf0 = B * args.steps / (time.perf_counter() - t1)
    per_rank_host = [cpu_ms_per_step]
    per_rank_dt = [dt]
    if world > 1:
        dts = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]      # every rank's own clock, for the line's diagnostics
        dist.all_gather(dts, torch.tensor([dt], dtype=torch.float64, device=cdev))
        per_rank_dt = [float(x.item()) for x in dts]
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        hc = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(hc, torch.tensor([cpu_ms_per_step], dtype=torch.float64, device=cdev))
        per_rank_host = [float(h.item()) for h in hc]
    rc = 0
    if rank == 0:
        frames = B * world * args.steps
        fps = frames / dt
        bb = float(np.mean(bb_ms)) * 1e-3                        # HIP-event span of one backbone schedule
        if args.depth > 1:       # backbones overlap: rate = algorithmic work of the timed region / its duration
            achieved = ALG_GFLOP_PER_FRAME * B * args.steps / dt / 1e3
        else:
            achieved = ALG_GFLOP_PER_FRAME * pipe.frames_per_launch / bb / 1e3        # TFLOP/s over the whole backbone schedule
        x3 = args.precision == "x3"
        step_s = dt / args.steps if args.depth > 1 else bb * B / pipe.frames_per_launch
        fpl = pipe.frames_per_launch                                             # input frames per backbone launch
        alg_bytes = pipe.engine.alg_bytes_per_batch * B / fpl                    # bytes per launch schedule x launches per step
        # Counters come from committed PMC passes (their own rocprofv3 runs).  They are quoted ONLY while the sources that decide the kernels
        # (csrc/*, the header, the tile tables, the schedule builder: benchkit/buildhash.py) hash to what the passes were measured on;
        # otherwise the fields are null and `counters` says which build the files belong to.
        from benchkit.buildhash import counters_for_build
        traffic, traffic_src = None, None                        # HBM bytes per batch
        t, traffic_info = counters_for_build(os.path.join(ROOT, "profiles", "hbm_traffic_x3.json" if x3 else "hbm_traffic.json"))
        if t is not None and B == 8:
            traffic = t["hbm_read_bytes_per_batch"] + t["hbm_write_bytes_per_batch"]
            traffic_src = t["source"]
        mfma_ctr, mfma_src = None, None                         # MFMA pipe utilisation by counters (separate PMC pass, depth 1)
        t, mfma_info = counters_for_build(os.path.join(ROOT, "profiles", "mfma_utilisation_x3.json" if x3 else "mfma_utilisation.json"))
        if t is not None and B == 8:
            mfma_ctr, mfma_src = t["pipe_utilisation"], t["source"]
        mfma_ctr16, mfma16_info = None, None                    # the same pass over the 16-frame launches the default pipeline issues
        if x3:
            t, mfma16_info = counters_for_build(os.path.join(ROOT, "profiles", "mfma_utilisation_x3_16_frames.json"))
            if t is not None and B == 8 and fpl == t.get("frames_per_launch"):
                mfma_ctr16 = t["pipe_utilisation"]

        def hbm_view(nbytes):                                   # bytes per step -> GB/s over the timed region and its share of the HBM peak
            g = nbytes / step_s / 1e9
            return {"bytes_per_step": nbytes, "achieved": g, "frac": g / PEAK_HBM_GBPS}
        n_rec = len(collected) if gathered is None else sum(len(pickle.loads(b)) for b in gathered)
        # the timed steps rotate over NB resident batches (and four synthetic scenes): records with one image_path must be identical, bit
        # for bit, whatever ran next to them on the GPU -- a guard the overlapped pipeline lacked until round 3 (EXPERIMENTS R3.6)
        n_rem = getattr(pipe, "remainder_records", 0)            # an odd step count leaves one batch to the batch-sized schedule, whose
        groups = {}                                              # kernels sum in other orders: compared with itself only
        for r in (collected[:len(collected) - n_rem] if n_rem else collected):
            key = (np.asarray(r["pred_2d"]).tobytes(), np.asarray(r["pred_3d"]).tobytes(), np.asarray(r["root_d"]).tobytes())
            groups.setdefault(r["image_path"], set()).add(key)
        identical = all(len(v) == 1 for v in groups.values())
        out = {
            "metric": "frames/sec at 3x512x832 (SMAP backbone + depth-aware association + 3D lifting)",
            "value": fps, "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16x3" if x3 else "f16", "data": "synthetic",
            "config": {"workload": f"batch={B} x 3x512x832 per GPU, full SMAP + depth-aware PAF association "
                                   f"+ lifting{' + RefineNet (configs[4])' if args.refine else ''}{' + flip-TTA' if args.flip else ''} "
                                   f"(BASELINE configs[2]; configs[3] when n_gpus=8)",
                       "frames_per_step": B * world, "records_in_run": n_rec,
                       "resident_batches_in_rotation": NB, "input_bytes_in_rotation": NB * B * 3 * H * W * 4,
                       # N > 1: every rank's own timed region (the headline divides by the MAX), so that a slow rank shows in ONE record
                       "per_rank": {"frames_per_sec": [B * args.steps / x for x in per_rank_dt],
                                    "ms_per_step_max": max(per_rank_dt) / args.steps * 1e3, "ms_per_step_min": min(per_rank_dt) / args.steps * 1e3,
                                    "frame_seeds": [SEED + 104729 * r for r in range(world)]},
                       "timed_steps_reproduce": {"identical_records_per_frame_across_steps": identical, "frames_checked": len(groups),
                                                 "records_checked": len(collected) - n_rem,
                                                 "frames_with_variants": sorted(k for k, v in groups.items() if len(v) > 1)[:8]},
                       "arithmetic": ("backbone: fp16 hi/lo pairs (22 significant bits), three fp16 MFMAs per K step, fp32 "
                                      "accumulate = the reference's fp32 results to ~3e-6 relative" if x3 else
                                      "backbone: fp16 storage / fp32 MFMA accumulate (~2e-3 relative on the maps)") +
                                     "; heads fp32; association fp32 (+f64 where the reference is); lifting f64",
                       "weights": "by-key recipe, stage-2 heads calibrated to ~24 peaks per key-point channel (benchkit/workload.py)",
                       "precision_modes": "x3 (this line unless --precision f16): meets the reference's fp32 results end to end "
                                          "(e2e_parity below); f16: ~2x the frames/s, ~0.36 cm mean / 1.06 cm max joint error at 3 m "
                                          "(profiles/r2_final_bench_f16.json) -- not within the 1e-3 m of the north star",
                       "pipeline": f"post-processing of batch k overlaps later backbones; {args.depth} backbone(s) in flight; "
                                   f"{pipe.frames_per_launch} frames per backbone launch"
                                   + (f" (= {pipe.frames_per_launch // B} consecutive steps' batches coalesced; --launch-frames 0: one launch "
                                      f"per step)" if pipe.frames_per_launch > B else "")
                                   + "; one end-of-run gather of the records",
                       "frames_per_launch": pipe.frames_per_launch,
                       # layer1's Bottlenecks and layer2's identity Bottlenecks as ONE launch each (csrc/convb.hip, convc.hip), the shared-input 1x1s
                       # of the Upsample_units as one launch with several outputs: conv launches per forward 206 -> 164 -> 152
                       "whole_block_launches": sum(1 for op in pipe.engine.graph.ops if "head" in op.p),
                       "conv_launches_per_forward": sum(1 for op in pipe.engine.graph.ops if op.kind == 0),
                       "merged_1x1_launches": sum(1 for op in pipe.engine.graph.ops if op.outs),
                       # the same steps with ONE backbone launch per step (no coalescing: a batch's records are not held
                       # back for its group), timed in this process right after the headline region
                       "value_launch_frames_0": fps_lf0 if fps_lf0 is not None else (fps if pipe.frames_per_launch == B * (2 if args.flip else 1) else None),
                       "added_latency_steps_by_coalescing": (pipe.frames_per_launch // (B * (2 if args.flip else 1)) - 1),
                       "ranks_in_gather": len(gathered) if gathered is not None else 1,
                       # did the records go through torch.distributed (backend "nccl" = RCCL) and come back byte for byte?
                       "gather": ({"backend": dist.get_backend(), "payload_bytes": len(payload), "payload_device": str(cdev),
                                   "own_payload_returned_identical": gathered[rank] == payload} if gathered is not None else None),
                       "association_lift_us_per_launch": {k: float(np.median(v)) for k, v in sorted(post_us.items())},
                       # submit_wall = enqueue (the host's own work: launches, H2D of the cameras, record building) + backpressure_wait
                       # (asleep until the GPU has finished the batch submitted `depth` steps earlier: a closed loop must wait somewhere)
                       "host_ms_per_step": {"submit_wall": host["submit"] / args.steps * 1e3,
                                            "enqueue_and_records": (host["submit"] - wait_loop) / args.steps * 1e3,
                                            "backpressure_wait": wait_loop / args.steps * 1e3, "process_cpu_per_rank": per_rank_host,
                                            "threads": host_threads,
                                            "busiest_threads_cpu_ms": [[n, round(v, 2)] for n, v in busiest]},
                       "host_submit_ms_quantiles": [float(np.percentile(host["steps"], q)) for q in (0, 10, 50, 90, 100)],
                       "host_timeline_ms": {"submit_loop_done": t_loop * 1e3, "flush_done": t_flush * 1e3, "total": dt * 1e3}},
            # Headline = SURVEY.md 8d: the backbone's bounding roof is the matrix pipe, achieved = 300.628 GFLOP x frames / time against the
            # dense fp16 MFMA peak (`frac`); the split-precision mode EXECUTES three MFMAs per algorithmic one (`pipe_frac`), and the
            # counter figure of the committed PMC pass stands beside it.  `hbm` is the other view of the same region: bytes / time
            # against the 8 TB/s HBM peak, with SURVEY 8d's unfused per-frame traffic (1.854 GB fp16, 3.707 GB fp32-sized) AND this
            # build's own count (x3 storage = fp32-sized; a fused launch counts what IT must move: x + out + weights).
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS,
                         "flops_executed_per_algorithmic_flop": 3 if x3 else 1,
                         "pipe_frac": (3 if x3 else 1) * achieved / PEAK_F16_TFLOPS,
                         # the chip clocks to its power budget: what the box held in the timed region, and pipe_frac against the MFMA
                         # peak AT that clock (2.5 PFLOP/s is the peak at 2.4 GHz)
                         "clocks": clocks,
                         "pipe_frac_at_measured_clock": ((3 if x3 else 1) * achieved / PEAK_F16_TFLOPS / clocks["clock_share_of_max"]) if clocks else None,
                         "pipe_frac_counters": mfma_ctr, "pipe_frac_counters_source": mfma_src,
                         # busy cycles / available cycles at whatever clock the pass ran at; `pipe_frac` prices time against 2.5 PFLOP/s
                         "pipe_frac_counters_this_launch_size": mfma_ctr16,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_commit": traffic_info.get("measured_on_commit"),
                         "counters_match_build": bool(traffic_info["counters_match_build"] and mfma_info["counters_match_build"]),
                         "counters": {"traffic": traffic_info, "pipe_frac_counters": mfma_info, "pipe_frac_counters_this_launch_size": mfma16_info},
                         "kernel": "conv_igemm_kernel / conv3x3_halo_kernel / convp_kernel / bottleneck_kernel / bottleneck128_kernel (all backbone launches; HIP "
                                   "events: per-schedule span below, rate = algorithmic work of the timed region / its duration when depth > 1)",
                         "hbm": {"peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                 "survey_8d_fp16": hbm_view(SURVEY_GB_PER_FRAME_FP16 * 1e9 * B),
                                 "survey_8d_fp32_sized": hbm_view(2 * SURVEY_GB_PER_FRAME_FP16 * 1e9 * B),
                                 "this_schedule": hbm_view(alg_bytes),
                                 "measured_traffic": hbm_view(traffic) if traffic else None,
                                 "measured_ceilings_GBps": {"hbm_read": 5300, "hbm_write": 6200, "hbm_mixed": 5200,
                                                            "source": "profiles/r2_v17_ubench_hbm_read_write_mix.log"}},
                         "backbone_ms_per_launch": bb * 1e3,
                         "backbone_stream_idle_ms_between_batches": float(np.mean(gap_ms)) if gap_ms else None,
                         "algorithmic_gflop_per_frame": ALG_GFLOP_PER_FRAME},
        }
        if want_ref:
            if "error" in cpu_ref:
                out["cpu_baseline"] = {"value": None, "unit": "frames/sec", "cores": cpu_ref.get("cores"), "kind": "port",
                                       "sample": "reference child failed: " + cpu_ref["error"]}
            else:
                # e2e_parity: the TIMED path (maps of the last timed launch: pipelined, `--depth` backbones in flight, coalesced
                # launches) vs the CPU reference on the same frames; e2e_parity_serial: one serial forward through the public API
                ms = parity.compare(hip_frames, cpu_ref["ref"])
                m = parity.compare(timed_frames, cpu_ref["ref"]) if timed_frames is not None else None
                if m is not None:
                    # ... and the timed RECORDS are these very frames' results (bit for bit): frame i of batch 0 <-> image_path r0/b0/f{i}
                    by_path = {}
                    for r in collected:
                        by_path.setdefault(r["image_path"], r)
                    same = True
                    for i, fr in enumerate(timed_frames):
                        r = by_path.get(f"r{rank}/b0/f{i}")
                        if r is None:
                            same = same and len(fr["p3"]) == 0
                        else:
                            same = same and (np.array_equal(np.asarray(r["pred_3d"]), fr["p3"]) and np.array_equal(np.asarray(r["pred_2d"]), fr["p2"])
                                             and np.array_equal(np.asarray(r["root_d"]), fr["rz"]))
                    m["timed_records_equal_these_frames"] = bool(same)
                    m["path"] = (f"timed region: depth {args.depth}, {pipe.frames_per_launch} frames per launch, maps of the last timed launch")
                out["config"]["e2e_parity"] = m if m is not None else ms
                out["config"]["e2e_parity_serial"] = ms
                m = out["config"]["e2e_parity"]
                out["config"]["e2e_mpjpe_cm"], out["config"]["peak_match"] = m["mpjpe_cm"], m["peak_match"]
                out["cpu_baseline"] = {
                    "value": 1.0 / cpu_ref["sec_per_frame"], "unit": "frames/sec", "cores": cpu_ref["cores"], "kind": "port",
                    "backbone_plus_assoc_ms_per_frame": cpu_ref["sec_per_frame"] * 1e3,
                    "assoc_lift_ms_per_frame_1thread": cpu_ref["assoc_1thread_ms"],
                    "assoc_lift_ms_per_frame_all_cores_frame_parallel": cpu_ref["assoc_all_cores_ms"],
                    "sample": f"{cpu_ref['frames']} frames of the same batch: oracle/backbone_ref.py (torch CPU fp32, "
                              f"{cpu_ref['threads']} threads) + oracle/smap_oracle.c association + lifting (1 thread, as the "
                              f"reference) per frame; association also frame-parallel over the cores"}
        print(json.dumps(out), flush=True)
        if not identical:            # records of one frame differ between timed steps: the overlapped pipeline is not deterministic
            print("bench.py: timed steps did not reproduce each other's records: " + json.dumps(out["config"]["timed_steps_reproduce"]),
                  file=sys.stderr, flush=True)
            rc = 3
    if world > 1:
        rcs = torch.tensor([rc], dtype=torch.int32, device=cdev)
        dist.all_reduce(rcs, op=dist.ReduceOp.MAX)
        rc = int(rcs.item())
    if dist.is_initialized():
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
