"""SMAP inference entry point on MI355X (reference: exps/stage3_root2/test.py).

Same command line (test.py:156-177) and the same result JSON
(`<exp>_<mode>_<data_mode>_<suffix>.json` with `model_pattern` and `3d_pairs`, test.py:32-38,147-151):

    export PROJECT_HOME=/path/to/repo ; export PYTHONPATH=$PYTHONPATH:$PROJECT_HOME
    python test.py -p SMAP_model.pth -t run_inference -d test [-rp RefineNet.pth] \
           --batch_size 16 --do_flip 1 --dataset_path /path/to/images

Per batch everything stays on the GPU: HIP backbone -> optional flip-TTA merge -> /255,/127 ->
batched association -> batched lifting (-> RefineNet); only the poses come back, and the
post-processing of batch k overlaps the backbone of batch k+1 (smap_amd/pipeline.py).  Launched under
`torch.distributed.run` the image list is split in contiguous per-rank blocks
(lib/utils/dataloader.py:80-85) and the records are gathered with one RCCL all_gather.

The ground-truth modes (test.py:73-95,142-143) run on the same pipeline: `-t generate_result` registers the
persons to the annotations of cfg.TEST.JSON_PATH on the device (smap_register_gt) and writes one record per frame
with the ground truth attached (the input of lib/eval/convert.py); `-t generate_train -d generation|test` writes
one record per matched person, the RefineNet training pairs (dataset/p2p_dataset.py)."""
import argparse
import json
import logging
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Subset

from model.smap import SMAP
from model.refinenet import RefineNet
from dataset.custom_dataset import CustomDataset
from smap_amd.dist import gather_records, shard_range
from exps.stage3_root2.config import cfg
from smap_amd.pipeline import PosePipeline, make_pipeline
from exps.stage3_root2.test_util import default_cams
from smap_amd.records import annotation_camera, kept_annotations, to_jsonable


def get_logger(name, log_dir, filename):
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger(name)
    if not logger.handlers:
        logger.setLevel(logging.INFO)
        fmt = logging.Formatter("%(asctime)s %(levelname)s %(message)s")
        for h in (logging.StreamHandler(), logging.FileHandler(os.path.join(log_dir, filename))):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger


class DevicePreprocLoader:
    """Batches of (imgs on the device, names, scales) with decode on the host and resize / pad /
    normalise in one HIP kernel per image (smap_amd/preprocess.py).  The decodes run AHEAD of the consumer on a small thread pool
    (PIL's decoders and numpy's file reads release the GIL): at ~800 frames/s of engine, one thread decoding a 1080p JPEG in ~10 ms
    would be the whole run (profiles/r5_cli_e2e.json: 62 frames/s with one thread, 333 with 8, 461 with 32).  SMAP_DECODE_THREADS
    (default: up to 16 of the allowed CPUs; 1 = decode in the consumer's thread, the round-4 behaviour).
    SMAP_DECODE_PROCS=<n>: the decoders as n WORKER PROCESSES (python -m dataset.decode: numpy + PIL only) writing into one shared-memory
    block, one slot per worker; the pool threads of this process then only talk to their worker and copy its slot into page-locked memory.
    What threads cannot scale past is the part of a decode that holds the interpreter lock (PIL's packers, file objects): ~600 frames/s on
    the bench's image mix, below the engine (EXPERIMENTS R6.11).  SMAP_DECODE_SLOT_MB (default 16): a larger frame is decoded in-thread."""

    def __init__(self, dataset, indices, batch_size, cfg, device):
        self.ds, self.idx, self.bs, self.cfg, self.device = dataset, list(indices), batch_size, cfg, device
        try:
            allowed = len(os.sched_getaffinity(0))
        except AttributeError:
            allowed = os.cpu_count() or 1
        self.threads = int(os.environ.get("SMAP_DECODE_THREADS", "0")) or max(1, min(16, allowed))
        self.procs = int(os.environ.get("SMAP_DECODE_PROCS", "0"))
        if self.procs > 0:
            self.threads = self.procs                        # one pool thread per worker process
        self.slot_bytes = int(os.environ.get("SMAP_DECODE_SLOT_MB", "16")) << 20

    def __len__(self):
        return (len(self.idx) + self.bs - 1) // self.bs

    def __iter__(self):
        from smap_amd.preprocess import preprocess_batch
        if self.threads <= 1:
            for s in range(0, len(self.idx), self.bs):
                raws, names = zip(*[self.ds.raw(i) for i in self.idx[s:s + self.bs]])
                imgs, scales = preprocess_batch(raws, self.cfg.INPUT.MEANS, self.cfg.INPUT.STDS, self.device)
                yield imgs, list(names), scales
            return
        import collections
        import contextlib
        from concurrent.futures import ThreadPoolExecutor
        ahead = max(2 * self.threads, 3 * self.bs)           # images being decoded or waiting: bounds the host memory (~6 MB per 1080p frame)

        def pinned(img):
            # ... leave the frame in PAGE-LOCKED memory (torch's caching host allocator recycles the blocks): the consumer's
            # upload is then an asynchronous DMA instead of a blocking pageable copy (~1 ms per 1080p frame of the consumer's time)
            buf = torch.empty(img.shape, dtype=torch.uint8, pin_memory=True)
            np.copyto(buf.numpy(), img)                      # (one copy, GIL released; `img` may be a read-only view of the decoder's bytes)
            return buf

        def decode(i):
            img, name = self.ds.raw(i)
            return pinned(img), name
        workers = contextlib.ExitStack()
        if self.procs > 0:
            decode = self._process_decoders(workers, pinned)
        with workers, ThreadPoolExecutor(self.threads) as ex:
            todo, futs = iter(self.idx), collections.deque()

            def fill():
                while len(futs) < ahead:
                    try:
                        futs.append(ex.submit(decode, next(todo)))
                    except StopIteration:
                        return
            fill()
            while futs:
                n = min(self.bs, len(futs))
                got = [futs.popleft().result() for _ in range(n)]       # in submission order: frame order is kept
                fill()
                raws, names = zip(*got)
                imgs, scales = preprocess_batch(raws, self.cfg.INPUT.MEANS, self.cfg.INPUT.STDS, self.device)
                yield imgs, list(names), scales


    def _process_decoders(self, stack, pinned):
        """Start SMAP_DECODE_PROCS workers (dataset/decode.py) over one shared-memory block; -> decode(i) for the pool threads.  A pool
        thread takes a free worker and its slot per frame: it writes "<slot>\\t<path>", blocks on the answer (no interpreter lock held), copies the
        slot into page-locked memory.  `stack` closes the workers' pipes, waits for them and unlinks the block when the iteration ends."""
        import queue
        import subprocess
        import sys
        from multiprocessing import shared_memory
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        shm = shared_memory.SharedMemory(create=True, size=self.procs * self.slot_bytes)
        view = np.frombuffer(shm.buf, np.uint8)
        procs = [subprocess.Popen([sys.executable, "-m", "dataset.decode", shm.name, str(self.slot_bytes)], cwd=root, text=True, bufsize=1,
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=dict(os.environ, OMP_NUM_THREADS="1"))
                 for _ in range(self.procs)]
        free = queue.Queue()
        for k in range(self.procs):
            free.put(k)

        def close():
            nonlocal view
            for p in procs:
                try:
                    p.stdin.close()
                except Exception:
                    pass
            for p in procs:
                try:
                    p.wait(timeout=5)
                except Exception:
                    p.kill()
            view = None
            shm.close()
            shm.unlink()
        stack.callback(close)

        def decode(i):
            k = free.get()                                   # a worker and its slot, for this frame (as many pool threads as workers: no wait)
            try:
                p = procs[k]
                path = self.ds.image_list[i].rstrip()
                name = path.replace(self.ds.dataset_path, "").lstrip("/")
                p.stdin.write(f"{k}\t{path}\n")
                p.stdin.flush()
                ans = p.stdout.readline().split(None, 3)
                if len(ans) < 3:
                    raise RuntimeError(f"decode worker {k} died on {path}")
                h, w = int(ans[1]), int(ans[2])
                if h == -1:                                  # larger than a slot: here, in this thread
                    return pinned(self.ds.raw(i)[0]), name
                if h < 0:
                    raise RuntimeError(f"decode worker: {path}: {ans[3] if len(ans) > 3 else 'failed'}")
                n = h * w * 3
                return pinned(view[k * self.slot_bytes:k * self.slot_bytes + n].reshape(h, w, 3)), name
            finally:
                free.put(k)
        return decode


class _DryRunPipeline:
    """Stand-in for PosePipeline in `--dry_run 1` (no GPU, no checkpoint): the same submit / flush contract -- records of
    the batch submitted `depth` calls earlier -- with one fake person per frame.  What the rehearsal exercises is everything
    AROUND the device work: CLI, image listing, the contiguous per-rank split, ragged last batches, the end-of-run gather
    (gloo instead of RCCL) and the result file; `python -m torch.distributed.run --nproc-per-node 8 test.py --dry_run 1 ...`."""

    def __init__(self, model, cfg, batch, H, W, device, refine_w=None, depth=2, **kw):
        self.B, self.depth, self.q = batch, max(1, int(depth)), []

    def submit(self, imgs, cams, tags, annotations=None):
        from smap_amd.records import frame_record
        p2 = np.zeros((1, 15, 4), np.float32)
        self.q.append([frame_record(p2, np.zeros((1, 15, 4)), np.full((1,), float(imgs[i].mean())), t, None, as_lists=False)
                       for i, t in enumerate(tags) if t is not None])
        return self.q.pop(0) if len(self.q) > self.depth else None

    def flush(self):
        out = [r for recs in self.q for r in recs]
        self.q = []
        return out or None


def generate_3d_point_pairs(model, refine_model, data_loader, cfg, logger, device, output_dir="", pipeline_cls=None):
    os.makedirs(output_dir, exist_ok=True)
    if pipeline_cls is None:       # batches of <= 8 frames share a backbone launch (smap_amd/pipeline.py::make_pipeline, SMAP_LAUNCH_FRAMES)
        pipeline_cls = lambda m, c, b, h, w, d, rw, **kw: make_pipeline(m, c, b, h, w, d, refine_weights=rw, **kw)
    if model is not None:
        model.eval()
    refine_w = None
    if refine_model is not None:
        refine_model.eval()
        refine_w = refine_model.folded(device)
    result = dict()
    result["model_pattern"] = cfg.DATASET.NAME
    result["3d_pairs"] = []
    rank = dist.get_rank() if dist.is_initialized() else 0
    it = data_loader
    if rank == 0:
        try:
            from tqdm import tqdm
            it = tqdm(data_loader)
        except ImportError:
            pass
    pipe = None
    dropped = []                     # frames without a result: their maps came out non-finite

    def drain(recs):
        if recs:
            result["3d_pairs"].extend(recs)

    def retire(p):                   # a pipeline that is replaced or finished: everything in flight + the frames it dropped
        drain(p.flush())
        dropped.extend(getattr(p, "dropped_frames", []))

    import time
    clock = {"loader_s": 0.0, "submit_s": 0.0, "frames": 0, "t0": time.perf_counter()}     # where the host's time goes (SMAP_CLI_TIMING)
    batches = iter(it)
    while True:
        t_ = time.perf_counter()
        try:
            batch = next(batches)
        except StopIteration:
            break
        clock["loader_s"] += time.perf_counter() - t_
        t_sub = time.perf_counter()
        annotations = None
        if cfg.TEST_MODE == "run_inference":
            imgs, img_path, scales = batch
            cams = default_cams(scales, len(imgs))
        else:                                                    # test.py:50-51,73-95
            imgs, meta_data, img_path, scales = batch
            annotations = [kept_annotations(m.numpy(), cfg.DATASET.ROOT_IDX) for m in meta_data]
            cams = [annotation_camera(a, s) if len(a) else [1.0] * 9 for a, s in zip(annotations, scales)]
        imgs = imgs.to(device, non_blocking=True).float().contiguous()
        img_path = list(img_path)
        if pipe is not None and len(imgs) < pipe.B:              # ragged last batch: pad with copies of its last frame and
            pad = pipe.B - len(imgs)                             # drop their records (tag None) -- no second engine / arena
            imgs = torch.cat([imgs, imgs[-1:].expand(pad, -1, -1, -1)], 0).contiguous()
            cams = np.concatenate([np.asarray(cams, np.float64), np.repeat(np.asarray(cams, np.float64)[-1:], pad, 0)], 0)
            img_path = img_path + [None] * pad
            if annotations is not None:
                annotations = list(annotations) + [annotations[-1]] * pad
        if pipe is None or pipe.B != len(imgs):
            if pipe is not None:
                retire(pipe)
            pipe = pipeline_cls(model, cfg, len(imgs), imgs.shape[-2], imgs.shape[-1], device, refine_w,
                                do_flip=bool(cfg.DO_FLIP), record_mode=cfg.TEST_MODE, numpy_records=True,
                                depth=int(os.environ.get("SMAP_PIPELINE_DEPTH", 2)))   # two backbones in flight (+19 %)
        with torch.no_grad():
            drain(pipe.submit(imgs, cams, list(img_path), annotations=annotations))
        clock["submit_s"] += time.perf_counter() - t_sub
        if "first_submit_s" not in clock:                        # builds the engine (schedule, weight packing, plan, arenas): seconds, once
            clock["first_submit_s"] = time.perf_counter() - t_sub
        clock["frames"] += sum(1 for p_ in img_path if p_ is not None)
    t_ = time.perf_counter()
    if pipe is not None:
        retire(pipe)
    clock["flush_s"] = time.perf_counter() - t_
    clock["loop_s"] = time.perf_counter() - clock.pop("t0")
    if os.environ.get("SMAP_CLI_TIMING") and rank == 0:
        # steady-state split of the run_inference loop: loader_s = waiting for the next batch (decode [+ host resize] + H2D + GPU
        # pre-processing enqueue), submit_s = enqueue + back-pressure of the pipeline + record building, flush_s = draining the tail
        steady = clock["loop_s"] - clock.get("first_submit_s", 0.0)
        clock.update(frames_per_s=clock["frames"] / clock["loop_s"] if clock["loop_s"] > 0 else None,
                     frames_per_s_after_engine_build=clock["frames"] / steady if steady > 0 else None, wait_gpu_s=getattr(pipe, "wait_s", None))
        with open(os.environ["SMAP_CLI_TIMING"], "w") as f:
            json.dump(clock, f)
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SMAP_FORCE_GATHER", "") == "1"):
        parts = gather_records(result["3d_pairs"], device)
        result["3d_pairs"] = [r for part in parts for r in part]                # rank order == frame order
        dropped = [n for part in gather_records(dropped, device) for n in part]
    if dropped:
        # the reference's fp32 forward has no such failure: say so where the caller will look (the result file and the exit status)
        result["dropped_frames"] = list(dropped)
        logger.warning("{} frame(s) have no result (non-finite maps: an activation exceeded the fp16 range, INTEGRATION.md section 5): {}".format(
            len(dropped), dropped[:20]))
    if rank == 0:
        dir_name = os.path.split(os.path.split(os.path.realpath(__file__))[0])[1]
        name = os.path.join(output_dir, "{}_{}_{}_{}.json".format(dir_name, cfg.TEST_MODE, cfg.DATA_MODE,
                                                                  cfg.JSON_SUFFIX_NAME))
        result["3d_pairs"] = to_jsonable(result["3d_pairs"])          # ndarray -> nested lists, once, at the end of the run
        with open(name, "w") as f:
            json.dump(result, f)
        logger.info("Pairs writed to {}".format(name))
    return result


def main():
    # the schedule of a (checkpoint, shape, arithmetic) is built once and kept on disk (smap_amd/engine.py plan cache): the second start of
    # this command loads it through smap_plan_create_from_blob instead of re-packing the weights (SMAP_PLAN_CACHE=0 switches it off)
    os.environ.setdefault("SMAP_PLAN_CACHE", os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.expanduser("~/.cache")), "smap_amd"))
    parser = argparse.ArgumentParser()
    parser.add_argument("--test_mode", "-t", type=str, default="run_inference",
                        choices=["generate_train", "generate_result", "run_inference"])
    parser.add_argument("--data_mode", "-d", type=str, default="test", choices=["test", "generation"])
    parser.add_argument("--SMAP_path", "-p", type=str, default="log/SMAP.pth", help="Path to SMAP model")
    parser.add_argument("--RefineNet_path", "-rp", type=str, default="",
                        help="Path to RefineNet model, empty means without RefineNet")
    parser.add_argument("--batch_size", type=int, default=1, help="Batch_size of test")
    parser.add_argument("--do_flip", type=float, default=0, help="Set to 1 if do flip when test")
    parser.add_argument("--dataset_path", type=str, default="", help='Image dir path of "run_inference" test mode')
    parser.add_argument("--json_name", type=str, default="", help="Add a suffix to the result json.")
    parser.add_argument("--precision", type=str, default="", choices=["", "x3", "f16"],
                        help="(addition) backbone arithmetic: x3 (default) = fp16 hi/lo pairs, three MFMAs per K step -- "
                             "reproduces the reference's fp32 forward; f16 = fp16 storage, ~2x faster, ~1e-3 relative error")
    parser.add_argument("--dry_run", type=int, default=0,
                        help="1: rehearse the run without a GPU or a checkpoint (stand-in pipeline, gloo gather): what a "
                             "multi-rank launch does around the device work -- split, ragged batches, gather, result file")
    parser.add_argument("--device_preprocess", type=int, default=0,
                        help="(addition) 1: resize/pad/normalise on the GPU (smap_preprocess) instead of in the dataset")
    args = parser.parse_args()
    cfg.TEST_MODE = args.test_mode
    cfg.DATA_MODE = args.data_mode
    cfg.REFINE = len(args.RefineNet_path) > 0
    cfg.DO_FLIP = args.do_flip
    cfg.JSON_SUFFIX_NAME = args.json_name
    cfg.TEST.IMG_PER_GPU = args.batch_size

    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dry = bool(args.dry_run)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("gloo" if dry else "nccl")
    elif world == 1 and os.environ.get("SMAP_FORCE_GATHER", "") == "1" and not dist.is_initialized():
        from smap_amd.dist import init_single_rank_group       # one rank, real collectives: the RCCL path on a one-GPU box (tests)
        init_single_rank_group("gloo" if dry else "nccl")
    if not dry:
        torch.cuda.set_device(local)
    os.makedirs(cfg.TEST_DIR, exist_ok=True)
    logger = get_logger(cfg.DATASET.NAME, cfg.TEST_DIR, "test_log_{}.txt".format(args.test_mode))

    device = torch.device("cpu") if dry else torch.device(cfg.MODEL.DEVICE, local)
    model = None
    if not dry:
        model = SMAP(cfg, run_efficient=cfg.RUN_EFFICIENT)
        model.to(device)
        if args.precision:
            model.precision = args.precision

    if args.test_mode != "run_inference":
        from lib.utils.dataloader import get_test_loader
        data_loader = get_test_loader(cfg, num_gpu=world, local_rank=dist.get_rank() if world > 1 else 0,
                                      stage=args.data_mode)
        dataset = indices = None
    else:
        dataset = CustomDataset(cfg, args.dataset_path)
        indices = range(len(dataset))
        if world > 1:
            st, ed = shard_range(len(dataset), world, dist.get_rank())
            indices = range(st, ed)
    if dataset is None:
        pass
    elif args.device_preprocess:
        data_loader = DevicePreprocLoader(dataset, indices, args.batch_size, cfg, torch.device(cfg.MODEL.DEVICE, local))
    else:
        data_loader = DataLoader(Subset(dataset, indices) if world > 1 else dataset, batch_size=args.batch_size,
                                 shuffle=False)

    if dry:
        generate_3d_point_pairs(None, None, data_loader, cfg, logger, device, output_dir=os.path.join(cfg.OUTPUT_DIR, "result"),
                                pipeline_cls=_DryRunPipeline)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    refine_model = RefineNet().to(device) if cfg.REFINE else None
    if os.path.exists(args.SMAP_path):
        state_dict = torch.load(args.SMAP_path, map_location=lambda storage, loc: storage)
        model.load_state_dict(state_dict["model"])
        if refine_model is not None:
            if os.path.exists(args.RefineNet_path):
                refine_model.load_state_dict(torch.load(args.RefineNet_path, map_location="cpu"))
            else:
                logger.info("No such RefineNet checkpoint of {}".format(args.RefineNet_path))
                return
        result = generate_3d_point_pairs(model, refine_model, data_loader, cfg, logger, device,
                                         output_dir=os.path.join(cfg.OUTPUT_DIR, "result"))
        if dist.is_initialized():
            dist.destroy_process_group()
        if result.get("dropped_frames"):
            return 2                 # the result file is written, but it lacks frames the reference would have produced
    else:
        logger.info("No such checkpoint of SMAP {}".format(args.SMAP_path))
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
