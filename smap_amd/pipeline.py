"""Two-stream inference pipeline for the SMAP hot path on one GPU.

The reference's loop (exps/stage3_root2/test.py:43-145) is strictly serial per frame: forward,
`.cpu()` syncs, one `dapalib.connect` per frame, numpy lifting.  Here a batch flows through two
HIP streams so that the device never waits for the host:

    stream "bb"   : SMAP backbone of batch k+1                      (smap_plan_run)
    stream "post" : nms -> paf -> group -> lift [-> RefineNet] -> async D2H of batch k   (/255, /127: in the backbone's head sum)
    host          : builds the `3d_pairs` records of batch k from pinned memory meanwhile

Network outputs and pinned result buffers are per slot, so batch k's post-processing is independent
of later forwards.  With `depth=2` two backbones are in flight on two streams (two arenas, shared
weights): while batch k sits in its low-resolution, latency-bound layers (layer3/4: 100-400
workgroups per launch) batch k+1 streams its HBM-bound high-resolution layers on the idle CUs.
`submit()` returns the records of the batch submitted `depth` calls earlier (None until then);
`flush()` returns everything still in flight, in order.

`make_pipeline` puts a `CoalescedPipeline` in front when the caller's batches are small: consecutive batches are run as ONE
backbone launch of up to 16 frames (the 32x52 and 16x26 levels of the network have too few workgroups per launch at 8 frames
to fill 256 CUs), same protocol, same records, same order.
"""
import os

import numpy as np
import torch

from . import dapalib
from .records import frame_record, train_records

NJ, MAXP = 15, 127
MAXG = 64                       # annotations per frame the registration kernel accepts (cfg.DATASET.MAX_PEOPLE = 20)


def pack_layout(B):
    """Byte offsets of (p3, rz, p2, counts) in a result pack of B frames, and its size (8-byte aligned fields)."""
    n3, nz, n2 = B * MAXP * NJ * 4 * 8, B * MAXP * 8, B * MAXP * NJ * 4 * 4
    return 0, n3, n3 + nz, n3 + nz + n2, (n3 + nz + n2 + 4 * B + 7) // 8 * 8


def pack_views(buf, B):
    """dict(p3, rz, p2, counts): typed views of a uint8 result pack (host or device)."""
    o3, oz, o2, oc, end = pack_layout(B)
    return dict(p3=buf[o3:oz].view(torch.float64).view(B, MAXP, NJ, 4), rz=buf[oz:o2].view(torch.float64).view(B, MAXP),
                p2=buf[o2:oc].view(torch.float32).view(B, MAXP, NJ, 4), counts=buf[oc:oc + 4 * B].view(torch.int32))


class _Slot:
    def __init__(self, engine, device, n_extra, B):
        # one output buffer per backbone launch (engine.B frames each); with several launches per batch the maps of the
        # whole batch are gathered into B-frame tensors behind them
        self.outs = [engine.new_output() for _ in range(B // engine.B)]
        self.out = self.outs[0]
        if len(self.outs) == 1:
            self.hms, self.det_d, self.root_d = engine.views(self.out)   # B frames, also with flip-TTA (merged in the schedule)
        else:
            h, w = engine.h, engine.w
            self.hms = torch.empty((B, engine.kpt_paf, h, w), dtype=torch.float32, device=device)
            self.det_d = torch.empty((B, engine.paf, h, w), dtype=torch.float32, device=device)
            self.root_d = torch.empty((B, 1, h, w), dtype=torch.float32, device=device)
        # results of one association + lifting pass over B frames: ONE page-locked buffer [p3 f64 | rz f64 | p2 f32 | counts i32] per result
        # set, so that a whole-batch pass reaches the host in one copy (round 5: four; each is a blit kernel + a launch gap on the post stream)
        self.pack_bytes = pack_layout(B)[-1]
        self.host_pack = [torch.empty((self.pack_bytes,), dtype=torch.uint8).pin_memory() for _ in range(1 + n_extra)]
        self.host = [pack_views(hp, B) for hp in self.host_pack]
        self.status = engine.out_floats                                      # index of the engine's status words in `out`
        self.status_words = engine.status_words
        # host copy of every launch's status words (word f // 31: bit 0 = non-finite maps, bit 1 + f % 31 = frame f of the launch)
        self.status_host = torch.zeros((len(self.outs), engine.status_words), dtype=torch.float32).pin_memory()
        self.p2_f64 = None               # ground-truth modes: the f64 pred_2d (allocated on first use)
        self.ev_bb = torch.cuda.Event()
        self.ev_post = torch.cuda.Event()       # the host waits on this one: PosePipeline._wait
        self.meta = None
        self.busy = False


class PosePipeline:
    def __init__(self, model, cfg, batch, H, W, device, refine_weights=None, n_extra=0, do_flip=False, depth=1,
                 record_mode="run_inference", numpy_records=False, max_frames_per_launch=None, strict_nonfinite=None):
        """record_mode: test.py's -t: "run_inference" (no ground truth), "generate_result" (one record per frame
        with the annotations attached) or "generate_train" (one record per matched person); the last two need
        `annotations=` in submit()."""
        assert record_mode in ("run_inference", "generate_result", "generate_train")
        self.record_mode = record_mode
        # frames whose maps came out non-finite (fp16 range exceeded): dropped with a RuntimeWarning and listed in
        # `dropped_frames` by default; strict_nonfinite=True (env SMAP_STRICT_NONFINITE=1) raises at collection instead
        self.strict_nonfinite = bool(int(os.environ.get("SMAP_STRICT_NONFINITE", "0"))) if strict_nonfinite is None else bool(strict_nonfinite)
        self.dropped_frames = []
        self.as_lists = not numpy_records      # numpy_records: records carry ndarray copies (records.to_jsonable at the end)
        self.device = torch.device(device)
        self.cfg = cfg
        self.B, self.do_flip = batch, bool(do_flip)
        # flip-TTA (test.py:55-70): frames and their mirror images run as ONE 2B batch inside the engine's schedule -- the
        # stem reads the mirrored image by index, the head sum merges the mirrored maps (no ATen cat/flip, no merge pass)
        kpt = cfg.DATASET.KEYPOINT.NUM
        self.flip_pair = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [kpt + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
        # frames per backbone launch: the whole batch, or the largest divisor of it that the engine accepts (ArenaTooLarge: one tensor
        # beyond a 4 GiB addressing window -- 53+ frames in split precision -- or an arena beyond the memory budget, SMAP_MAX_ARENA_BYTES /
        # 90 % of the device memory shared by 4 arenas: every backbone in flight has its own)
        from .engine import ArenaTooLarge
        limit = max_frames_per_launch or int(os.environ.get("SMAP_MAX_FRAMES_PER_LAUNCH", "0")) or batch
        self.chunk = None
        for parts in range(1, batch + 1):
            if batch % parts or batch // parts > limit:
                continue
            try:
                # (scaled_hms: the head sum of the schedule writes hms / 255 | / 127 itself -- test.py:111-112 -- no pass over the maps in _post)
                self.engine = model.engine(batch // parts, H, W, self.device, flip_pair=self.flip_pair if do_flip else None, scaled_hms=True)
                self.chunk = batch // parts
                break
            except ArenaTooLarge as exc:
                if parts == batch:
                    raise
                import logging                          # not silently: the chunking decides the throughput
                logging.getLogger("smap_amd").warning("PosePipeline: %d frames do not run as one backbone launch (%s); trying %d launches per batch",
                                                      batch // parts, exc, parts + 1)
        self._model, self._generation = model, model.weights_generation      # a reload / .to() after this point makes the
                                                                             # pipeline stale: submit() refuses to run on old weights
        self.refine = refine_weights
        self.depth = max(1, int(depth))
        self.engines = [self.engine] + [self.engine.sibling() for _ in range(self.depth - 1)]
        prio = [int(x) for x in os.environ.get("SMAP_BB_STREAM_PRIORITIES", "").split(",") if x.strip()]     # experiment hook
        self.s_bbs = [torch.cuda.Stream(self.device, priority=prio[i % len(prio)]) if prio else torch.cuda.Stream(self.device)
                      for i in range(self.depth)]
        self.s_bb = self.s_bbs[0]
        self.s_post = torch.cuda.Stream(self.device)
        self.s_comm = torch.cuda.Stream(self.device)      # result gather (RCCL) never queues behind compute
        self.nslots = self.depth + 1
        self.slots = [_Slot(self.engine, self.device, n_extra, batch) for _ in range(self.nslots)]
        self.frames_per_launch = self.chunk
        self.k = 0
        self.wait_s = 0.0                # host time spent WAITING for the GPU inside submit() / flush() (sleeping, not enqueueing)
        self.bb_events = []              # (start, end) HIP events of timed backbone runs
        self.post_events = []            # (tag, start, end) HIP events of timed association+lifting passes (time_backbone=True)

    # -- device side -------------------------------------------------------------------------
    def _post(self, slot, idx, hms, det_d, root_d, cams, scale, gt=None, row0=0):
        """Association + lifting of one set of maps on the post stream; results -> pinned memory (rows row0.. of result set idx:
        a coalesced launch hands its callers' extra maps over batch by batch, where they lie).
        gt = (gt_roots [B,G,2], gt_counts [B]) on the device: register the persons to the annotations first
        (test_util.py:18-42) and lift in the f64 flavour of the ground-truth modes."""
        if scale:
            dapalib.scale_hms_(hms)                                             # test.py:111-112
        n = hms.shape[0]
        if gt is None and row0 == 0 and n == self.B:
            # a whole batch, no ground truth: every kernel writes into ONE device buffer laid out like the slot's page-locked pack -> one copy
            dpack = torch.empty((slot.pack_bytes,), dtype=torch.uint8, device=hms.device)
            v = pack_views(dpack, n)
            bodys, counts = dapalib.connect_batch(hms, root_d, self.cfg.DATASET.ROOT_IDX, True, counts=v["counts"])
            p3_lift = v["p3"] if self.refine is None else torch.empty_like(v["p3"])
            p2, p3, rz = dapalib.lift_batch(bodys, counts, det_d, root_d, cams, out=(v["p2"], p3_lift, v["rz"]))
            if self.refine is not None:
                dapalib.refine_batch(p2, p3, counts, *self.refine, out=v["p3"])
            slot.host_pack[idx].copy_(dpack, non_blocking=True)
            return
        bodys, counts = dapalib.connect_batch(hms, root_d, self.cfg.DATASET.ROOT_IDX, True)
        if gt is not None:
            bodys, counts = dapalib.register_gt_batch(bodys, counts, *gt)
        p2, p3, rz = dapalib.lift_batch(bodys, counts, det_d, root_d, cams, gt_mode=gt is not None)
        if self.refine is not None:
            p3 = dapalib.refine_batch(p2, p3, counts, *self.refine)
        h = slot.host[idx]
        if gt is not None:
            if slot.p2_f64 is None:
                slot.p2_f64 = torch.empty(tuple(p2.shape), dtype=torch.float64).pin_memory()
            slot.p2_f64.copy_(p2, non_blocking=True)
        else:
            h["p2"][row0:row0 + n].copy_(p2, non_blocking=True)
        for k, t in (("p3", p3), ("rz", rz), ("counts", counts)):
            h[k][row0:row0 + n].copy_(t, non_blocking=True)

    def submit(self, imgs, cams, tags, extra=(), time_backbone=False, annotations=None):
        """imgs [B,3,H,W] fp32 on the device -- or a list of equally sized tensors that together hold the B frames (read where
        they are: smap_plan_run_inputs); cams [B,9] float64 (host array); tags: B image names.
        extra: tuples (tag_prefix, hms, root_d, det_d) of already-scaled maps to associate as well (bench only); each of the three
        may be a list of tensors covering the B frames in order (associated part by part, no concatenation).  annotations (ground-truth modes): B arrays [G_i,15,C] of the KEPT annotations of each
        frame (records.kept_annotations; G_i may be 0 -- the frame is skipped, test.py:81-82).
        Returns the record list of the previous batch or None."""
        if self._model.weights_generation != self._generation:
            raise RuntimeError("the model's weights were reloaded or moved after this PosePipeline was built; build a new one")
        gt = None
        if self.record_mode != "run_inference":
            if annotations is None or len(annotations) != len(tags):
                raise ValueError("record_mode %r needs one annotation array per frame" % self.record_mode)
            gmax = max(1, max(len(a) for a in annotations))
            if gmax > MAXG:
                raise ValueError("at most %d annotations per frame" % MAXG)
            roots = np.zeros((len(tags), gmax, 2), np.float32)
            for i, a in enumerate(annotations):
                if len(a):
                    roots[i, :len(a)] = np.asarray(a)[:, self.cfg.DATASET.ROOT_IDX, :2]
            gt = (torch.from_numpy(roots).to(self.device, non_blocking=True),
                  torch.tensor([len(a) for a in annotations], dtype=torch.int32).to(self.device, non_blocking=True))
        slot = self.slots[self.k % self.nslots]
        eng, s_bb = self.engines[self.k % self.depth], self.s_bbs[self.k % self.depth]
        ready = self._collect(slot) if slot.busy else None       # the batch submitted nslots calls ago ...
        cams_d = torch.as_tensor(np.asarray(cams), dtype=torch.float64).to(self.device, non_blocking=True)
        cur = torch.cuda.current_stream(self.device)
        s_bb.wait_stream(cur)                          # imgs were produced on the caller's stream
        self.s_post.wait_stream(cur)
        img_parts = list(imgs) if isinstance(imgs, (list, tuple)) else [imgs]
        if len(img_parts) > 1 and (len(img_parts) > 8 or len({tuple(t.shape) for t in img_parts}) > 1 or
                                   (len(slot.outs) > 1 and len(img_parts) % len(slot.outs))):
            img_parts = [torch.cat(img_parts)]         # shapes the stem's buffer table cannot express: gather (never on the bench path)
        for t in img_parts:
            t.record_stream(s_bb)
        cams_d.record_stream(self.s_post)
        if gt is not None:
            gt[0].record_stream(self.s_post)
            gt[1].record_stream(self.s_post)
        as_parts = lambda t: list(t) if isinstance(t, (list, tuple)) else [t]
        for _, e_hms, e_rd, e_dd in extra:               # the caller may drop its references
            for t in (e_hms, e_rd, e_dd):
                if t is not None:
                    for q in as_parts(t):
                        q.record_stream(self.s_post)
        with torch.cuda.stream(s_bb):
            if time_backbone:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if len(slot.outs) == 1:
                eng.run(img_parts if len(img_parts) > 1 else img_parts[0], out=slot.out)
            else:                                      # the batch in engine-sized launches, one after the other on this stream
                c = eng.B
                per = len(img_parts) // len(slot.outs)
                for j, o in enumerate(slot.outs):
                    src = img_parts[0][j * c:(j + 1) * c] if len(img_parts) == 1 else img_parts[j * per:(j + 1) * per]
                    hm, dd, rd = eng.run(src[0] if isinstance(src, list) and len(src) == 1 else src, out=o)
                    slot.hms[j * c:(j + 1) * c].copy_(hm, non_blocking=True)
                    slot.det_d[j * c:(j + 1) * c].copy_(dd, non_blocking=True)
                    slot.root_d[j * c:(j + 1) * c].copy_(rd, non_blocking=True)
            if time_backbone:
                e1.record()
                self.bb_events.append((e0, e1))
            slot.ev_bb.record()
        with torch.cuda.stream(self.s_post):
            self.s_post.wait_event(slot.ev_bb)
            def timed_post(tag, *a, **k):
                if not time_backbone:
                    return self._post(*a, **k)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
                self._post(*a, **k)
                p1.record()
                self.post_events.append((tag if isinstance(tag, str) else "+".join(sorted(set(tag))), p0, p1))
            for j, o in enumerate(slot.outs):
                slot.status_host[j].copy_(o[slot.status:slot.status + slot.status_words], non_blocking=True)
            timed_post("network", slot, 0, slot.hms, slot.det_d, slot.root_d, cams_d, scale=False, gt=gt)   # (scaled by the head sum)
            for j, (tag, hms, rd, dd) in enumerate(extra):
                row = 0
                hp, rp = as_parts(hms), as_parts(rd)
                dp = [None] * len(hp) if dd is None else as_parts(dd)
                for h_, r_, d_ in zip(hp, rp, dp):       # part by part where the maps lie (leading-dimension slices are views)
                    n_ = h_.shape[0]
                    timed_post(tag if isinstance(tag, str) else tag[row], slot, 1 + j, h_, slot.det_d[row:row + n_] if d_ is None else d_, r_, cams_d[row:row + n_],
                               scale=False, row0=row)
                    row += n_
            slot.ev_post.record()
        slot.meta = (list(tags), [t for t, *_ in extra], annotations)
        slot.busy = True
        self.k += 1
        # ... or, more eagerly, the oldest batch once `depth` newer ones are queued behind it
        if ready is None:
            old = self.slots[(self.k - 1 - self.depth) % self.nslots] if self.k > self.depth else None
            if old is not None and old.busy:
                ready = self._collect(old)
        return ready

    def flush(self):
        out = None
        for j in range(self.nslots):                   # oldest first
            s = self.slots[(self.k + j) % self.nslots]
            if s.busy:
                r = self._collect(s)
                out = r if out is None else out + r
        return out

    def last_maps(self):
        """(hms, det_d, root_d) of the most recent submit(), as its association read them (hms already /255,/127-scaled in
        place): device tensors that stay valid until the next submit().  Call after flush().  bench.py compares what the TIMED
        launches produced with the CPU reference through these."""
        if self.k == 0:
            return None
        s = self.slots[(self.k - 1) % self.nslots]
        assert not s.busy, "flush() first"
        return s.hms, s.det_d, s.root_d

    # -- host side ---------------------------------------------------------------------------
    @staticmethod
    def _wait(ev, poll_s=2e-4):
        """Wait for a HIP event WITHOUT spinning: hipEventSynchronize busy-polls the signal (also for events created with
        the blocking flag on this ROCm), which makes the submit thread ~100 % of a core per rank; a query loop with 0.2 ms
        sleeps costs ~2 % and at most 0.2 ms of latency, which the second batch in flight hides."""
        import time
        while not ev.query():
            time.sleep(poll_s)

    def _collect(self, slot):
        import time
        t0 = time.perf_counter()
        self._wait(slot.ev_post)
        self.wait_s += time.perf_counter() - t0          # back-pressure: the host sleeping until the GPU has finished a batch
        tags, extra_tags, annotations = slot.meta
        words = slot.status_host.view(torch.int32).tolist()                   # [launch][word]
        if any(w[0] & 1 for w in words):
            # The reference's fp32 forward has no such failure; its loop would carry on with the other frames (the result file
            # is only written at the end of the run, test.py:147-151).  So: the frames whose maps are not finite have no
            # result and are dropped with a warning, the other frames of the launch keep theirs; strict mode raises instead.
            c = self.engine.B                                                   # output frames per launch
            bad = [j * c + f for j, w in enumerate(words) for f in range(c) if (w[f // 31] >> (1 + f % 31)) & 1]
            names = [tags[i] for i in bad if tags[i] is not None]
            if self.strict_nonfinite:
                slot.busy = False
                raise RuntimeError("SMAP backbone produced non-finite maps for %s: an activation exceeded the fp16 range (65504) "
                                   "that the engine's arithmetic keeps (INTEGRATION.md section 5); these frames have no valid "
                                   "result" % names)
            import warnings
            warnings.warn("SMAP backbone: non-finite maps (an activation exceeded the fp16 range, INTEGRATION.md section 5); "
                          "no result for %s, the other frames of the launch are kept" % names, RuntimeWarning, stacklevel=2)
            self.dropped_frames.extend(names)
            tags = [None if i in set(bad) else t for i, t in enumerate(tags)]   # a None tag = no record (as for padding frames)
        recs = []
        for idx, h in enumerate(slot.host):
            counts = h["counts"].numpy()
            gt_mode = idx == 0 and annotations is not None
            p2 = slot.p2_f64.numpy() if gt_mode else h["p2"].numpy()
            p3, rz = h["p3"].numpy(), h["rz"].numpy()
            for i, P in enumerate(counts):
                P = int(P)
                if P == 0 or tags[i] is None:                                   # tag None = padding frame of a ragged last batch
                    continue                                                    # test.py:81-82,131-132
                pref = extra_tags[idx - 1] if idx else None                     # one prefix, or one per frame (coalesced batches)
                name = tags[i] if idx == 0 else f"{pref if isinstance(pref, str) else pref[i]}/{tags[i]}"
                if gt_mode and self.record_mode == "generate_train":            # test.py:142-143
                    recs.extend(train_records(p2[i, :P], p3[i, :P], rz[i, :P], np.asarray(annotations[i]),
                                              self.cfg.DATASET.ROOT_IDX, as_lists=self.as_lists))
                else:
                    recs.append(frame_record(p2[i, :P], p3[i, :P], rz[i, :P], name,
                                             np.asarray(annotations[i]) if gt_mode else None, as_lists=self.as_lists))
        slot.busy = False
        return recs


class CoalescedPipeline:
    """PosePipeline for callers with SMALL batches: `group` consecutive submit() calls run as ONE backbone launch of
    group * batch frames.  The low-resolution layers of the backbone have 100-200 workgroups per launch at 8 frames and are
    bound by what ONE workgroup per CU can stream from L2 into LDS; at 16 frames per launch the same layers fill the chip
    (790 vs 747 frames/s at batch 8, profiles/r3_frames_per_launch.log).  Same protocol as PosePipeline -- submit() returns
    the records of earlier batches (whole groups at a time, in submission order) or None, flush() everything outstanding;
    a trailing incomplete group runs through a `batch`-sized pipeline (built up front).  Costs latency (a batch waits
    for its group), never order or results: frames are independent."""

    def __init__(self, model, cfg, batch, H, W, device, group, **kw):
        assert group >= 2
        self.B, self.group = batch, group
        self.inner = PosePipeline(model, cfg, batch * group, H, W, device, **kw)
        # the batch-sized pipeline for an incomplete trailing group: built NOW (a second engine of the smaller batch: ~1 s, one more
        # arena), not inside somebody's flush() -- bench.py's flush is inside its timed region
        self._small = PosePipeline(model, cfg, batch, H, W, device, **kw)
        self._pending = []
        self._timed = False
        self.remainder_records = 0                       # set by flush()

    # what callers read off a pipeline
    engine = property(lambda self: self.inner.engine)
    chunk = property(lambda self: self.inner.chunk)
    depth = property(lambda self: self.inner.depth)
    s_comm = property(lambda self: self.inner.s_comm)
    bb_events = property(lambda self: self.inner.bb_events)          # one entry per LAUNCH (group * batch frames)
    post_events = property(lambda self: self.inner.post_events)
    frames_per_launch = property(lambda self: self.inner.chunk)
    wait_s = property(lambda self: self.inner.wait_s + self._small.wait_s)
    # frames dropped for non-finite maps, by either schedule (PosePipeline.dropped_frames), and the strict switch of both
    dropped_frames = property(lambda self: self.inner.dropped_frames + self._small.dropped_frames)
    strict_nonfinite = property(lambda self: self.inner.strict_nonfinite,
                                lambda self, v: (setattr(self.inner, "strict_nonfinite", bool(v)), setattr(self._small, "strict_nonfinite", bool(v)))[0])

    def last_maps(self):
        """Maps of the most recent COALESCED launch (group * batch frames, in submission order), or None if none ran."""
        return self.inner.last_maps()

    @staticmethod
    def _merge(pending):
        """The group's batches as ONE submit of the inner pipeline WITHOUT copying a map or an image: images and extra maps go
        down as lists of the callers' tensors (the stem reads up to 8 input buffers; the association runs part by part)."""
        imgs = [p[0] for p in pending]
        cams = np.concatenate([np.asarray(p[1]) for p in pending])
        tags = [t for p in pending for t in p[2]]
        extra = []
        for j in range(len(pending[0][3])):
            parts = [p[3][j] for p in pending]
            assert all((q[3] is None) == (parts[0][3] is None) for q in parts)
            prefixes = [q[0] for q, p in zip(parts, pending) for _ in p[2]]
            extra.append((prefixes if len(set(prefixes)) > 1 else prefixes[0], [q[1] for q in parts],
                          [q[2] for q in parts], None if parts[0][3] is None else [q[3] for q in parts]))
        ann = None if pending[0][4] is None else [a for p in pending for a in p[4]]
        return imgs, cams, tags, extra, ann

    def submit(self, imgs, cams, tags, extra=(), time_backbone=False, annotations=None):
        assert len(tags) == self.B and all(len(extra) == len(p[3]) for p in self._pending)
        self._pending.append((imgs, cams, list(tags), list(extra), annotations))
        self._timed = self._timed or time_backbone
        if len(self._pending) < self.group:
            return None
        imgs, cams, tags, extra, ann = self._merge(self._pending)
        self._pending, timed, self._timed = [], self._timed, False
        return self.inner.submit(imgs, cams, tags, extra=extra, time_backbone=timed, annotations=ann)

    def flush(self):
        out = self.inner.flush()
        self.remainder_records = 0                       # how many of the records below came from the batch-sized schedule (its tile
        n0 = len(out) if out else 0                      # choices differ from the coalesced one's: equal to ~1e-6, not bit for bit)
        if self._pending:                                # an incomplete group: after everything older, batch by batch
            for imgs, cams, tags, extra, ann in self._pending:
                r = self._small.submit(imgs, cams, tags, extra=extra, annotations=ann)
                if r:
                    out = r if out is None else out + r
            self._pending, self._timed = [], False
            r = self._small.flush()
            if r:
                out = r if out is None else out + r
            self.remainder_records = (len(out) if out else 0) - n0
        return out


def make_pipeline(model, cfg, batch, H, W, device, launch_frames=None, **kw):
    """PosePipeline, or CoalescedPipeline when the caller's batch (x2 with flip-TTA) is at most half of `launch_frames`
    (default 16, env SMAP_LAUNCH_FRAMES; 0 / 1 = one launch per submitted batch)."""
    lf = int(os.environ.get("SMAP_LAUNCH_FRAMES", "16")) if launch_frames is None else int(launch_frames)
    per = batch * (2 if kw.get("do_flip") else 1)
    group = lf // per if per > 0 else 0
    if group >= 2 and kw.get("record_mode", "run_inference") == "run_inference":
        return CoalescedPipeline(model, cfg, batch, H, W, device, group, **kw)
    return PosePipeline(model, cfg, batch, H, W, device, **kw)
