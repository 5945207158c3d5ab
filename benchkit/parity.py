"""End-to-end parity checker: the HIP path (backbone -> association -> lifting) against the reference path restated on the
CPU in fp32 (oracle/backbone_ref.py -> oracle/smap_oracle.c) ON THE SAME IMAGES AND WEIGHTS.

CHECKER ONLY (imports the oracle): used by tests/test_e2e_parity_gpu.py, __graft_entry__.smoke() and, outside the timed
region, by bench.py to fill `config.e2e_parity`.  Reference path: exps/stage3_root2/test.py:50-134.

What is compared, per frame (north_star: "peak indices / limb assignments bit-exact, 3D joints within 1e-3 m"):
  peaks    the NMS candidates of the 15 key-point channels.  A candidate of one path is MATCHED when the other path has a
           candidate of the same channel within 0.5 heat-map pixel (the 7x7 centroid moves by ~1e-3 px under a 1e-3
           relative perturbation of the map; a different integer peak is >= 1 px away).  peak_match = matched / max(nA, nB).
  persons  skeletons are paired through their root joint (same 0.5 px rule).  person_match = paired / max(PA, PB).
  limbs    for paired skeletons, a joint agrees when it is absent in both or within 0.5 px in both: limb_match.
  3D       over paired skeletons and joints present in both: |dX,dY,dZ| in cm -> mpjpe_cm (mean Euclidean), max_joint_err_cm,
           and root_z_max_err_cm.
  ties     "Same peaks" between two floating-point forwards is only defined down to the forwards' own resolution: a candidate
           that clears the 0.2 threshold, or beats its best neighbour, by less than the maps' rounding noise is decided by the
           summation order (the fp32 CPU forward is itself ~1e-6 of the map scale away from an fp64 one; cuDNN's would be
           too).  So every peak present in only one path is looked up in the REFERENCE map: its decision margin
           |v - max(threshold, best neighbour)| relative to the map scale (decided on the UNCAPPED candidate sets; a tie in
           a channel with more than 127 candidates also shifts the cap: peaks_cap_shifted).  peaks_differing counts them,
           peaks_differing_max_margin is the largest such margin (a genuinely different peak would show ~1e-2), and
           peaks_clear_mismatch counts those above NEAR_TIE = 1e-6 (a third of the split-precision map error; the ties observed
           on MI355X have margins of 3e-8 .. 9e-8): the number that
           must be ZERO.
  lifter   generate_relZ (test_util.py:60-86) reads the depth maps at ROUNDED positions: the root at int(x), int(y) and ten
           np.round(linspace(src, dst)) samples per limb, each then mapped to a heat-map pixel by the nearest x4 upsampling
           (index // 4).  Those index functions are step functions of the sub-pixel peak coordinates, which differ between two
           floating-point forwards by ~1e-6 px: a sample that sits that close to a step lands on the neighbouring depth pixel
           in one path -- the lifter's counterpart of a peak tie, a DISCRETE event and not accumulated rounding.  Every joint
           beyond 0.01 cm is therefore traced back: the sampled pixel indices of both paths are recomputed for the limbs
           on its chain to the root (chain_bones, test_util.py:45-57); the joint is a LIFTER TIE when some index differs there
           while (1) the two paths' sample coordinates are within the centroid noise the MEASURED map difference allows
           (`centroid_bound`: |d centroid| <= 2 eps sum|x - c| / sum(w) over the peak's 7x7 window, eps = max |d map| of the
           frame, + the fp32 accumulation term; x 4 = network px; a sample between two peaks moves by at most the larger of
           their two bounds) -- so it is the step, not the coordinate, that moved the sample -- and (2) its error is no larger
           than what the moved samples can cause: the depth maps' own difference between the two pixels each moved sample
           read, summed over the chain (mean and percentile clamp of test_util.py:80-85 are 1-Lipschitz in the samples), plus
           the measured root-depth difference (`lifter_tie_cap`).  The derivation is checked on EVERY matched peak of the
           comparison: the measured coordinate difference must stay under the peak's bound (centroid_noise_max_over_bound <=
           1).  joints_over_0.1cm_unexplained -- a joint off by more than 1e-3 m WITHOUT such a straddled step, or beyond
           its cap -- is the number that must be ZERO; lifter_ties / lifter_tie_max_coord_diff_px are reported beside it.
"""
import numpy as np
import torch

NJ = 15
TOL_PX = 0.5
NEAR_TIE = 1e-6          # decision margin (relative to the key-point map scale) below which a peak is a floating-point tie
THRESHOLD = 0.2          # association.cpp:55 nms threshold on the /255-scaled maps
MAXP = 127
# (rounds 3-4 used a constant LIFT_TIE_PX here, 1e-4 and then 5e-4 network px, chosen after looking at the cases; round 5 derives
#  the bound per peak from the measured map difference: centroid_bounds below.)
STRIDE = 4
# association.cpp:23-25 jointPairs = cfg.DATASET.PAF.VECTOR (limb k: src -> dst); chain_bones (test_util.py:45-57) walks
# them from the root (joint 2): joint 0 from limb 1 (reversed), joint 1 from limb 0, then dst from src for k >= 2
LIMBS = [(0, 1), (0, 2), (0, 9), (9, 10), (10, 11), (0, 3), (3, 4), (4, 5), (2, 12), (12, 13), (13, 14), (2, 6), (6, 7), (7, 8)]


def _chain_limbs(root_idx=2):
    """joint -> list of limb indices whose depth samples its Z is built from (chain_bones order)."""
    parent = {0: (1, root_idx), 1: (0, 0)}                      # joint: (limb, parent joint)
    for k in range(2, len(LIMBS)):
        parent[LIMBS[k][1]] = (k, LIMBS[k][0])
    chains = {}
    for j in range(NJ):
        c, cur = [], j
        while cur != root_idx:
            k, cur = parent[cur]
            c.append(k)
        chains[j] = c
    return chains


_CHAINS = _chain_limbs()


def lift_sample_pixels(body):
    """body [15,4] fp32 in HEAT-MAP pixels (dapalib.connect's output) -> (root (iy, ix) heat-map pixel, root coords,
    {limb k: (idx [10,2] heat-map pixels, coords [10,2] network pixels)}) exactly as smap_oracle_lift / generate_relZ index
    the depth maps: x4 in fp32, f64 linspace, np.round (half to even), nearest x4 = index // 4."""
    b = body[:, :2].astype(np.float32) * np.float32(STRIDE)
    root = (int(b[2, 1]) // STRIDE, int(b[2, 0]) // STRIDE)
    out = {}
    for k, (s, d) in enumerate(LIMBS):
        if not (body[s, 3] > 0 and body[d, 3] > 0):
            continue
        xs = np.linspace(float(b[s, 0]), float(b[d, 0]), 10)
        ys = np.linspace(float(b[s, 1]), float(b[d, 1]), 10)
        idx = np.stack([np.round(ys).astype(np.int64) // STRIDE, np.round(xs).astype(np.int64) // STRIDE], 1)
        out[k] = (idx, np.stack([ys, xs], 1))
    return root, b[2, ::-1].astype(np.float64), out


def centroid_bounds(kp_ref, peaks_ref, eps):
    """How far can writeResultKernel's 7x7 centroid (nmsBase.cu:95-128: sum of x * s over the window pixels with s > 0, divided by
    their sum) move when every map value moves by at most `eps`?  First order: |dc| <= eps * sum|x_j - c| / S with S the window's
    positive mass (a value crossing zero enters or leaves with |s| <= eps: the same form); doubled for the second-order terms, plus
    the rounding of the fp32 accumulation itself when its inputs change (24 * 2^-24 * coordinate, see below).  kp_ref [15,H,W]: reference key-point maps (scaled); peaks_ref
    [15,128,3]: the reference peak list.  Returns {(channel, x as float32 bits, y bits): bound in HEAT-MAP pixels}."""
    out = {}
    C, H, W = kp_ref.shape
    cand = {}
    for (c, y, x) in peak_pixels(kp_ref, cap=True):
        cand.setdefault(c, []).append((y, x))
    for c, pts in cand.items():
        pts.sort()                                               # raster order = the order of the peak list
        n = int(peaks_ref[c, 0, 0])
        for i, (y, x) in enumerate(pts[:n]):
            y0, y1, x0, x1 = max(0, y - 3), min(H, y + 4), max(0, x - 3), min(W, x + 4)
            win = kp_ref[c, y0:y1, x0:x1].astype(np.float64)
            pos = win > -eps
            S = float(np.where(win > 0, win, 0.0).sum())
            if S <= 49 * eps:
                b = 7.0
            else:
                xs = np.arange(x0, x1, dtype=np.float64)[None, :] + np.zeros_like(win)
                ys = np.arange(y0, y1, dtype=np.float64)[:, None] + np.zeros_like(win)
                cx, cy = float(peaks_ref[c, 1 + i, 0]) - 0.5, float(peaks_ref[c, 1 + i, 1]) - 0.5
                t = max(float(np.abs(xs - cx)[pos].sum()), float(np.abs(ys - cy)[pos].sum()))
                # fp32 accumulation: xAcc and sAcc each collect <= 49 roundings of relative size 2^-24; their differences between two
                # runs on nearly equal inputs add like a random walk (sqrt(49) = 7), twice (numerator and denominator), relative to
                # the coordinate itself, with a factor 1.7 in hand: 24 (SURVEY 8c measured 4.6e-5 px between an FMA and a non-FMA
                # build of the same kernel = 0.16 of this term at x = 200; tests/test_parity_cpu.py: 0.35 under 3e-6 map noise)
                b = 2.0 * eps * t / S + 24 * 2.0 ** -24 * max(cx, cy, 1.0)
            out[(c, np.float32(peaks_ref[c, 1 + i, 0]).tobytes(), np.float32(peaks_ref[c, 1 + i, 1]).tobytes())] = b
    return out


def peak_pixels(kp, cap=True):
    """kp [15,H,W] scaled key-point maps -> set of (c, y, x): nmsRegisterKernel's rule (strict > threshold and all 8
    neighbours, interior pixels only); cap: the first 127 per channel in raster order, as writeResultKernel keeps them
    (nmsBase.cu:10-60,137-175); cap=False: every pixel the rule accepts."""
    v = torch.from_numpy(np.ascontiguousarray(kp, np.float32))
    nb = torch.nn.functional.unfold(v[:, None], 3).view(v.shape[0], 9, v.shape[1] - 2, v.shape[2] - 2)
    centre, others = nb[:, 4], torch.cat([nb[:, :4], nb[:, 5:]], 1).max(1).values
    mask = ((centre > THRESHOLD) & (centre > others)).numpy()
    out = set()
    for c in range(mask.shape[0]):
        ys, xs = np.nonzero(mask[c])
        for y, x in list(zip(ys, xs))[:MAXP if cap else None]:
            out.add((c, int(y) + 1, int(x) + 1))
    return out


def decision_margin(kp_ref, c, y, x):
    """|value - max(threshold, best neighbour)| of pixel (c, y, x) in the reference key-point maps, relative to their scale."""
    v = float(kp_ref[c, y, x])
    win = kp_ref[c, y - 1:y + 2, x - 1:x + 2].astype(np.float64).copy()
    win[1, 1] = -np.inf
    return abs(v - max(THRESHOLD, float(win.max()))) / max(float(np.abs(kp_ref).max()), 1e-30)


def reference_path(sd, imgs, cams, root_idx=2, threads=None, refine=None, flip_pair=None):
    """imgs [B,3,H,W] fp32 CPU, cams [B,9] -> list of per-frame dicts (peaks, bodys, p2, p3, rz, hms, det_d, root_d).
    refine = (W[5] [out,in], b[5]) BN-folded RefineNet weights (numpy): p3 becomes the refined pose (BASELINE configs[4],
    test_util.py:102-131).  flip_pair (43 channel indices): the flip-TTA of test.py:55-70 -- a second forward on the
    mirrored image, merged by the reference's channel loop (restated here in torch fp32, same operation order)."""
    from oracle import oracle_lib as O
    from oracle.backbone_ref import smap_forward
    if threads:
        torch.set_num_threads(threads)
    outs = []
    with torch.no_grad():
        for i in range(imgs.shape[0]):                      # frame by frame: bounded memory, same numbers (eval-mode BN)
            hms, det_d, root_d = smap_forward(sd, imgs[i:i + 1])
            if flip_pair is not None:
                hf = smap_forward(sd, torch.flip(imgs[i:i + 1], [-1]))[0]
                flipped = torch.flip(hf, [-1])[:, list(flip_pair)]
                sign = torch.ones(len(flip_pair))
                sign[NJ::2] = -1.0                             # PAF-x channels change sign under mirroring
                hms = hms + flipped * sign.view(1, -1, 1, 1)
                hms[:, NJ:] *= 0.5                             # only the PAF channels are averaged (test.py:66-69)
            hms = hms[0].clone()
            hms[:NJ] /= 255                                  # test.py:111-112
            hms[NJ:] /= 127
            h, d, r = hms.numpy(), det_d[0].numpy(), root_d[0, 0].numpy()
            bodys, peaks, _ = O.connect(h, r, root_idx, True)
            p2, p3, rz = O.lift(bodys, d, r, cams[i])
            if refine is not None and len(bodys):
                p3 = O.refine(p2, p3, refine[0], refine[1])
            outs.append(dict(peaks=peaks, bodys=bodys, p2=p2, p3=p3, rz=rz, hms=h, det_d=d, root_d=r))
    return outs


def hip_path(net, imgs_dev, cams, root_idx=2, refine=None, flip_pair=None):
    """The product path through its public pieces (model.smap.SMAP -> dapalib batch entry points), per-frame numpy.
    refine = RefineNet.folded(device) -> the refined pose replaces p3; flip_pair -> the engine that runs the flip-TTA inside
    its schedule (what PosePipeline(do_flip=True) uses)."""
    from smap_amd import dapalib
    if flip_pair is not None:
        B, _, H, W = imgs_dev.shape
        hms, det_d, root_d = net.engine(B, H, W, imgs_dev.device, flip_pair=flip_pair).run(imgs_dev.float().contiguous())
    else:
        hms, det_d, root_d = net(imgs_dev)
    hms = hms.clone()
    dapalib.scale_hms_(hms)
    return frames_from_maps(hms, det_d, root_d, cams, root_idx=root_idx, refine=refine)


def frames_from_maps(hms, det_d, root_d, cams, root_idx=2, refine=None):
    """Already SCALED maps on the device ([B,43,h,w], [B,14,h,w], [B,1,h,w]) -> per-frame dicts: association + lifting
    (+ RefineNet) through the product's batch entry points.  bench.py feeds it the output buffers of its TIMED launches
    (pipelined, coalesced), so that the parity block describes what was timed."""
    from smap_amd import dapalib
    bodys, counts, peaks, _ = dapalib.connect_batch(hms, root_d, root_idx, True, return_intermediate=True)
    p2, p3, rz = dapalib.lift_batch(bodys, counts, det_d, root_d, cams)
    if refine is not None:
        p3 = dapalib.refine_batch(p2, p3, counts, *refine)
    torch.cuda.synchronize()
    counts = counts.cpu().numpy()
    outs = []
    for i, P in enumerate(counts):
        P = int(P)
        outs.append(dict(peaks=peaks[i].cpu().numpy(), bodys=bodys[i, :P].cpu().numpy(), p2=p2[i, :P].cpu().numpy(),
                         p3=p3[i, :P].cpu().numpy(), rz=rz[i, :P].cpu().numpy(), hms=hms[i].cpu().numpy(),
                         det_d=det_d[i].cpu().numpy(), root_d=root_d[i, 0].cpu().numpy()))
    return outs


def _pair(a, b, tol=TOL_PX):
    """Greedy one-to-one pairing of 2-D points a [n,2], b [m,2] within `tol` (Chebyshev): list of (i, j)."""
    if len(a) == 0 or len(b) == 0:
        return []
    d = np.abs(a[:, None, :] - b[None, :, :]).max(-1)
    pairs, used = [], set()
    for i in np.argsort(d.min(1)):
        for j in np.argsort(d[i]):
            if d[i, j] > tol:
                break
            if j not in used:
                used.add(j)
                pairs.append((int(i), int(j)))
                break
    return pairs


def compare(hip, ref, root_idx=2):
    """hip / ref: lists of per-frame dicts as above -> metrics dict (plain Python floats/ints)."""
    n_pk = m_pk = 0
    n_pe = m_pe = 0
    n_j = m_j = 0
    errs, rz_errs = [], []
    big_unexplained, lifter_ties, tie_diffs, after_peak_tie, tie_events = [], 0, [], 0, set()
    noise_px, noise_ratio, tie_bounds, tie_over_cap, tie_errs, root_steps = [0.0], [0.0], [], 0, [], 0
    rz_same_pixel = []
    worst_frame = None
    margins = []
    cap_shifted = n_cand = 0
    maps = {"hms": 0.0, "det_d": 0.0, "root_d": 0.0}
    for f, (a, b) in enumerate(zip(hip, ref)):
        for k in maps:                                                  # backbone: max |d| / max |ref| per output
            maps[k] = max(maps[k], float(np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)))
        if "hms" in a and "hms" in b:                                   # peaks present in one path only: how close was the call?
            # decided on the UNCAPPED sets: with > 127 candidates in a channel one flipped near-tie early in the raster
            # also moves the 127-cap, i.e. swaps a perfectly clear peak in or out at the END of the list -- a consequence
            # of the tie, counted separately (peaks_cap_shifted), not a second decision that differs
            ua, ub = peak_pixels(a["hms"][:NJ], cap=False), peak_pixels(b["hms"][:NJ], cap=False)
            margins.extend(decision_margin(b["hms"][:NJ], *p) for p in ua ^ ub)
            n_cand += len(ub)
            pa, pb = peak_pixels(a["hms"][:NJ]), peak_pixels(b["hms"][:NJ])
            cap_shifted += sum(1 for p in pa ^ pb if p in ua and p in ub)
        # centroid noise the measured map difference allows, per reference peak (see `centroid_bounds`), and the measured one
        eps_f = float(np.abs(a["hms"][:NJ].astype(np.float64) - b["hms"][:NJ]).max()) if ("hms" in a and "hms" in b) else 0.0
        cb_f = centroid_bounds(b["hms"][:NJ], b["peaks"], eps_f) if "hms" in b else {}
        cb_default = max(cb_f.values()) if cb_f else 0.0

        def bound_of(c, xy):
            # a coordinate that is not in the reference peak list (hand-made frames of the unit tests): the fp32 accumulation term alone
            return cb_f.get((c, np.float32(xy[0]).tobytes(), np.float32(xy[1]).tobytes()),
                            max(cb_default, 24 * 2.0 ** -24 * max(float(xy[0]), float(xy[1]), 1.0)))
        for c in range(NJ):
            na, nb = int(a["peaks"][c, 0, 0]), int(b["peaks"][c, 0, 0])
            n_pk += max(na, nb)
            prs = _pair(a["peaks"][c, 1:1 + na, :2], b["peaks"][c, 1:1 + nb, :2])
            m_pk += len(prs)
            for i, j in prs:
                d = float(np.abs(a["peaks"][c, 1 + i, :2].astype(np.float64) - b["peaks"][c, 1 + j, :2]).max())
                if d < 0.05:                                      # the same integer peak (a neighbouring candidate is >= 1 px away)
                    noise_px.append(d)
                    noise_ratio.append(d / max(bound_of(c, b["peaks"][c, 1 + j, :2]), 1e-30))
        A, Bo = a["bodys"], b["bodys"]
        n_pe += max(len(A), len(Bo))
        ia = [i for i in range(len(A)) if A[i, root_idx, 3] > 0]
        ib = [i for i in range(len(Bo)) if Bo[i, root_idx, 3] > 0]
        pr = _pair(A[ia][:, root_idx, :2] if ia else np.zeros((0, 2)), Bo[ib][:, root_idx, :2] if ib else np.zeros((0, 2)))
        m_pe += len(pr)
        for i, j in pr:
            pa, pb = ia[i], ib[j]
            va, vb = A[pa, :, 3] > 0, Bo[pb, :, 3] > 0
            close = np.abs(A[pa, :, :2] - Bo[pb, :, :2]).max(-1) <= TOL_PX
            agree = (~va & ~vb) | (va & vb & close)
            n_j += NJ
            m_j += int(agree.sum())
            both = va & vb & close
            if both.any():
                e = np.linalg.norm(a["p3"][pa, both, :3] - b["p3"][pb, both, :3], axis=-1)
                errs.extend(e.tolist())
                if worst_frame is None or e.max() > worst_frame[1]:
                    worst_frame = (f, float(e.max()))
                if e.max() > 1e-2:                                       # trace the discrete events of the lifter (see "lifter")
                    ra, ca, sa = lift_sample_pixels(A[pa])
                    rb, cb, sb = lift_sample_pixels(Bo[pb])
                    root_step = float(np.abs(ca - cb).max()) if ra != rb else None
                    root_bound = STRIDE * bound_of(root_idx, Bo[pb, root_idx, :2])
                    drz = abs(float(a["rz"][pa]) - float(b["rz"][pb]))
                    for j, ej in zip(np.nonzero(both)[0], e):
                        if ej <= 1e-2:
                            continue
                        chain = _CHAINS[int(j)]
                        # a joint hanging off a joint the two paths do NOT share inherits that difference (the limb is sampled
                        # between other end points): a consequence of the peak tie that changed the skeleton, counted apart
                        if any(not (agree[LIMBS[k][0]] and agree[LIMBS[k][1]]) for k in chain):
                            after_peak_tie += 1
                            continue
                        # coordinate gaps at the straddled steps of the chain, each relative to ITS bound (network px): a sample
                        # between two peaks moves by at most the larger of the two centroids' bounds; cap = what the moved samples
                        # can cause (the reference depth map's difference between the two pixels each of them read, 1-Lipschitz mean)
                        diffs, ratios = [], []
                        cap = drz
                        if root_step is not None:
                            diffs.append(root_step)
                            ratios.append(root_step / max(root_bound, 1e-30))
                        for k in chain:
                            if k in sa and k in sb:
                                moved = (sa[k][0] != sb[k][0]).any(1)
                                if moved.any():
                                    s_, d_ = LIMBS[k]
                                    bk = STRIDE * max(bound_of(s_, Bo[pb, s_, :2]), bound_of(d_, Bo[pb, d_, :2]))
                                    gap = float(np.abs(sa[k][1][moved] - sb[k][1][moved]).max())
                                    diffs.append(gap)
                                    ratios.append(gap / max(bk, 1e-30))
                                    tie_bounds.append(bk)
                                    ia_, ib_ = sa[k][0][moved], sb[k][0][moved]
                                    dm = b["det_d"][k].astype(np.float64) if ("det_d" in b and b["det_d"].shape[0] > k) else None
                                    if dm is not None and max(ia_[:, 0].max(), ib_[:, 0].max()) < dm.shape[0] and max(ia_[:, 1].max(), ib_[:, 1].max()) < dm.shape[1] \
                                            and min(ia_.min(), ib_.min()) >= 0:
                                        cap += float(np.abs(dm[ia_[:, 0], ia_[:, 1]] - dm[ib_[:, 0], ib_[:, 1]]).max())
                                    else:
                                        cap = float("inf")              # no depth map to look the step up in: no cap
                        # X, Y follow Z through the back-projection ((x - cx) z / f: |x - cx| / f < 1 for any lens this wide), + slack
                        # for the continuous part (the maps' own 3e-6)
                        cap = 1.8 * cap + 0.01
                        if diffs and max(ratios) <= 1.0 and ej <= cap:
                            lifter_ties += 1
                            tie_events.add((f, int(pb)))             # one straddled step moves every joint further down the chain
                            tie_diffs.append(max(diffs))
                            tie_errs.append(float(ej))
                        elif ej > 0.1:
                            tie_over_cap += int(bool(diffs) and max(ratios) <= 1.0)
                            big_unexplained.append((f, int(pb), int(j), float(ej), diffs, ratios, cap))
            rz_errs.append(abs(float(a["rz"][pa]) - float(b["rz"][pb])))
            # the root depth is read at int(x), int(y) of the root joint // 4 (test_util.py:62-66): the SAME pixel in both paths unless
            # the root coordinate straddles an integer (a "root step": one more lifter tie, bounded like the others)
            ra_, ca_, _ = lift_sample_pixels(A[pa])
            rb_, cb_, _ = lift_sample_pixels(Bo[pb])
            if ra_ == rb_:
                rz_same_pixel.append(rz_errs[-1])
            else:
                root_steps += 1
    errs = np.asarray(errs) if errs else np.zeros((0,))
    return {
        "frames": len(hip), "peaks_ref": int(sum(int(b["peaks"][c, 0, 0]) for b in ref for c in range(NJ))),
        "peak_match": m_pk / n_pk if n_pk else 1.0, "peaks_unmatched": int(n_pk - m_pk),
        "peaks_differing": len(margins), "peaks_differing_max_margin": float(max(margins)) if margins else 0.0,
        "peaks_clear_mismatch": int(sum(m > NEAR_TIE for m in margins)), "peaks_cap_shifted": int(cap_shifted),
        "peak_candidates_ref": int(n_cand),        # every pixel the NMS rule accepts in the reference maps (before the 127-cap)
        "persons_ref": int(sum(len(b["bodys"]) for b in ref)), "person_match": m_pe / n_pe if n_pe else 1.0,
        "limb_match": m_j / n_j if n_j else 1.0, "joints_compared": int(errs.size),
        "mpjpe_cm": float(errs.mean()) if errs.size else 0.0, "max_joint_err_cm": float(errs.max()) if errs.size else 0.0,
        "p99_joint_err_cm": float(np.percentile(errs, 99)) if errs.size else 0.0,
        # a joint beyond 0.01 cm with identical 2D pixel is a DISCRETE event of the lifter, not accumulated rounding: one of the
        # ten np.round(linspace) sample positions of its limb (test_util.py:74-76) sat within ~1e-6 px of .5 (sub-pixel peak
        # coordinates differ by that much) and landed on the neighbouring depth pixel.  Counted, like the near-tie peaks.
        "joints_over_0.01cm": int((errs > 1e-2).sum()) if errs.size else 0,
        "joints_over_0.1cm": int((errs > 0.1).sum()) if errs.size else 0,
        # ... of which NOT explained by a straddled index step of the lifter (must be 0), and the explained events
        "joints_over_0.1cm_unexplained": len(big_unexplained), "unexplained_examples": big_unexplained[:4],
        "joints_moved_after_peak_tie": int(after_peak_tie),
        "lifter_ties": int(lifter_ties), "lifter_tie_events": len(tie_events),      # joints moved / skeletons with a straddled step
        "lifter_tie_max_coord_diff_px": float(max(tie_diffs)) if tie_diffs else 0.0,
        # the DERIVED bound those gaps were held to (network px; per limb: 4 x the larger of its two peaks' centroid bounds) and the
        # check of the derivation on every matched peak of the comparison: measured centroid difference / its bound (must be <= 1)
        "lifter_tie_bound_px_min_max": [float(min(tie_bounds)), float(max(tie_bounds))] if tie_bounds else None,
        "lifter_tie_max_joint_err_cm": float(max(tie_errs)) if tie_errs else 0.0,
        "lifter_ties_over_cap": int(tie_over_cap),
        "centroid_noise_max_px": float(max(noise_px)), "centroid_noise_max_over_bound": float(max(noise_ratio)),
        "centroid_peaks_checked": len(noise_px) - 1,
        "root_z_max_err_cm": float(max(rz_errs)) if rz_errs else 0.0,
        "root_z_max_err_cm_same_pixel": float(max(rz_same_pixel)) if rz_same_pixel else 0.0, "root_steps": int(root_steps),
        "root_z_mean_cm": float(np.mean([r for b in ref for r in b["rz"]])) if any(len(b["rz"]) for b in ref) else 0.0,
        "map_rel_err_max": maps,
    }


def run(net, sd, imgs, cam=None, threads=None):
    """One call for tests / smoke / bench: net = model.smap.SMAP on the GPU with weights `sd`; imgs [B,3,H,W] fp32 CPU."""
    from .workload import PEOPLE_CAM
    cams = np.tile(np.asarray(PEOPLE_CAM if cam is None else cam, np.float64), (imgs.shape[0], 1))
    dev = next(net.parameters()).device
    hip = hip_path(net, imgs.to(dev), cams)
    ref = reference_path(sd, imgs, cams, threads=threads)
    return compare(hip, ref), hip, ref
