"""The end-to-end parity CHECKER (benchkit/parity.py) on CPU: both "paths" are the oracle, one of them on maps perturbed by the
amount the split-precision backbone differs from the fp32 reference (3e-6 of the map scale).  What is pinned here is the checker's
own arithmetic: the DERIVED centroid-noise bound holds on every peak, joints moved by a straddled rounding step are classified as
lifter ties within their cap, and a genuinely wrong coordinate is not."""
import os

import numpy as np
import pytest

from benchkit import parity
from benchkit.workload import PEOPLE_CAM, synth_scene
from oracle import oracle_lib as O


def _frames(n, eps, seed0=100, shift=None):
    rng = np.random.default_rng(7)
    ref, hip = [], []
    cam = np.asarray(PEOPLE_CAM, np.float64)
    for f in range(n):
        hms, rdepth, _, _ = synth_scene(6 + f, seed=seed0 + f)
        yy, xx = np.mgrid[0:128, 0:208].astype(np.float32)
        det_d = np.stack([10.0 * np.sin(xx / (5.0 + k) + k) * np.cos(yy / (7.0 + k)) for k in range(14)]).astype(np.float32)
        root_d = (rdepth * 2.0).astype(np.float32)

        def path(h, wrong=None):
            bodys, peaks, _ = O.connect(h, root_d, 2, True)
            if wrong is not None and len(bodys):
                bodys = bodys.copy()
                bodys[0, 5, 0] += wrong                       # a wrong coordinate: NOT a tie
            p2, p3, rz = O.lift(bodys, det_d, root_d, cam)
            return dict(peaks=peaks, bodys=bodys, p2=p2, p3=p3, rz=rz, hms=h, det_d=det_d, root_d=root_d)
        ref.append(path(hms))
        noisy = (hms + rng.uniform(-eps, eps, hms.shape).astype(np.float32) * (np.abs(hms) > 0)).astype(np.float32)
        hip.append(path(noisy, wrong=shift if f == 0 else None))
    return hip, ref


def test_centroid_bound_holds_on_every_peak_and_ties_are_classified():
    hip, ref = _frames(6, 3e-6)
    m = parity.compare(hip, ref)
    assert m["persons_ref"] >= 30 and m["centroid_peaks_checked"] >= 300
    assert 0.0 < m["centroid_noise_max_over_bound"] <= 1.0, m
    assert m["centroid_noise_max_px"] < 1e-3
    assert m["joints_over_0.1cm_unexplained"] == 0 and m["lifter_ties_over_cap"] == 0, m
    assert m["root_z_max_err_cm_same_pixel"] <= 1e-3 * max(m["root_z_mean_cm"], 1.0)
    # identical inputs: nothing to classify
    z = parity.compare(ref, ref)
    assert z["max_joint_err_cm"] == 0 and z["lifter_ties"] == 0 and z["centroid_noise_max_px"] == 0


def test_larger_map_noise_moves_samples_and_stays_explained():
    """1e-3 of the map scale (the fp16 mode's error level): centroids move by ~1e-3 px, depth samples straddle rounding steps --
    every joint beyond 1e-3 m must be traced to a step within the bound that noise level allows, and stay under its cap."""
    hip, ref = _frames(8, 1e-3, seed0=300)
    m = parity.compare(hip, ref)
    assert m["centroid_noise_max_over_bound"] <= 1.0, m
    assert m["lifter_ties"] >= 1 and m["lifter_ties_over_cap"] == 0, m
    assert m["lifter_tie_max_coord_diff_px"] <= m["lifter_tie_bound_px_min_max"][1]
    assert m["lifter_tie_max_joint_err_cm"] < 1.0


def test_a_wrong_coordinate_is_not_a_tie():
    hip, ref = _frames(2, 3e-6, seed0=500, shift=0.3)       # 0.3 heat-map px = 1.2 network px off on one joint of one path
    m = parity.compare(hip, ref)
    assert m["joints_over_0.1cm_unexplained"] >= 1 or m["max_joint_err_cm"] <= 0.1, m
    if m["max_joint_err_cm"] > 0.1:
        assert m["joints_over_0.1cm_unexplained"] >= 1


def _scene_paths(seed, eps, tamper=None):
    """One frame through both paths; `tamper(bodys, peaks)` edits the HIP side's skeletons / peak table before lifting."""
    rng = np.random.default_rng(seed)
    cam = np.asarray(PEOPLE_CAM, np.float64)
    hms, rdepth, _, _ = synth_scene(8, seed=seed)
    yy, xx = np.mgrid[0:128, 0:208].astype(np.float32)
    det_d = np.stack([10.0 * np.sin(xx / (5.0 + k) + k) * np.cos(yy / (7.0 + k)) for k in range(14)]).astype(np.float32)
    root_d = (rdepth * 2.0).astype(np.float32)

    def path(h, edit=None):
        bodys, peaks, _ = O.connect(h, root_d, 2, True)
        if edit is not None:
            bodys, peaks = bodys.copy(), peaks.copy()
            edit(bodys, peaks)
        p2, p3, rz = O.lift(bodys, det_d, root_d, cam)
        return dict(peaks=peaks, bodys=bodys, p2=p2, p3=p3, rz=rz, hms=h, det_d=det_d, root_d=root_d)
    noisy = (hms + rng.uniform(-eps, eps, hms.shape).astype(np.float32) * (np.abs(hms) > 0)).astype(np.float32)
    return path(noisy, tamper), path(hms)


def test_a_small_wrong_coordinate_that_moves_a_depth_sample_is_never_called_a_tie():
    """ADVERSARIAL for the tie classifier: a joint coordinate wrong by 2e-3 .. 5e-2 heat-map px -- 10 to 500 times the centroid
    noise the measured map difference allows, still far too small to see in the 2D output -- on skeletons whose limbs then sample the depth
    maps at another pixel.  Looks exactly like a lifter tie (an index step, a joint off by millimetres to centimetres) except for the size of
    the coordinate difference: every such joint beyond 0.1 cm must come out UNEXPLAINED, whatever joint / person / shift."""
    flagged = moved = 0
    for seed in (610, 611, 612, 613):
        for shift in (2e-3, 1e-2, 5e-2):
            for joint in (1, 4, 7, 10, 13):
                def tamper(bodys, peaks, joint=joint, shift=shift):
                    for p in range(len(bodys)):
                        if bodys[p, joint, 3] > 0:                      # a joint that exists: x and y off by `shift`
                            bodys[p, joint, 0] += shift
                            bodys[p, joint, 1] -= shift
                hip, ref = _scene_paths(seed, 3e-6, tamper)
                m = parity.compare([hip], [ref])
                if m["max_joint_err_cm"] > 0.1:
                    moved += 1
                    assert m["joints_over_0.1cm_unexplained"] >= 1 and m["lifter_ties"] < m["joints_over_0.1cm"] + m["lifter_ties"], (seed, shift, joint, m)
                    flagged += 1
                else:
                    assert m["lifter_ties_over_cap"] == 0
    assert moved >= 3 and flagged == moved, (moved, flagged)                # the sweep does reach the interesting case


def test_a_wrong_peak_coordinate_breaks_the_derived_centroid_bound():
    """The same for the peak table: a centroid off by 2e-3 px (the map noise allows ~1e-4) must show as measured / bound > 1 -- the check
    that validates the derivation on every matched peak is the one that has to see it."""
    def tamper(bodys, peaks):
        ch = int(np.argmax(peaks[:, 0, 0]))                             # a channel with peaks: move its first centroid
        peaks[ch, 1, 0] += 2e-3
    hip, ref = _scene_paths(700, 3e-6, tamper)
    m = parity.compare([hip], [ref])
    assert m["centroid_noise_max_over_bound"] > 1.0, m
    clean_hip, clean_ref = _scene_paths(700, 3e-6)
    assert parity.compare([clean_hip], [clean_ref])["centroid_noise_max_over_bound"] <= 1.0


def test_counter_files_are_quoted_only_for_the_build_they_were_measured_on(tmp_path):
    """bench.py's roofline.traffic / pipe_frac_counters come from committed PMC passes; benchkit/buildhash.py ties them to a hash of the
    kernel sources, the header, the tile tables and the schedule builder: one flipped byte in a .hip file and the figures are withheld."""
    import json
    import shutil
    from benchkit import buildhash as H
    root = tmp_path / "tree"
    for f in H.hashed_files():
        dst = root / os.path.relpath(f, H.ROOT)
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(f, dst)
    assert len(H.hashed_files(str(root))) == len(H.hashed_files()) >= 12
    h0 = H.source_hash(str(root))
    assert h0 == H.source_hash()                                       # same bytes, same hash, wherever the tree lies
    cj = tmp_path / "hbm_traffic_x3.json"
    json.dump({"hbm_read_bytes_per_batch": 1.0, "hbm_write_bytes_per_batch": 2.0, "source": "test", "build_hash": h0, "commit": "abc"}, open(cj, "w"))
    t, info = H.counters_for_build(str(cj), str(root))
    assert t is not None and info["counters_match_build"] and info["measured_on_commit"] == "abc"
    victim = root / "smap_amd" / "csrc" / "conv.hip"
    raw = bytearray(victim.read_bytes())
    raw[len(raw) // 2] ^= 1
    victim.write_bytes(bytes(raw))
    t, info = H.counters_for_build(str(cj), str(root))
    assert t is None and not info["counters_match_build"] and info["measured_on_build"] == h0 != info["this_build"]
    t, info = H.counters_for_build(str(tmp_path / "missing.json"), str(root))
    assert t is None and not info["present"]
    # a counter file from before the stamp existed (no build_hash) is never quoted
    json.dump({"hbm_read_bytes_per_batch": 1.0, "hbm_write_bytes_per_batch": 2.0, "source": "old"}, open(cj, "w"))
    assert H.counters_for_build(str(cj))[0] is None


def _one_peak_maps(delta_rel=0.0, bump=None):
    """Reference and "HIP" frames over one hand-made key-point map: a plateau of two neighbouring pixels whose values differ by
    `delta_rel` of the map scale (the reference prefers the left one), the HIP side with the order flipped; `bump` = (value) adds a
    genuinely different local maximum to the HIP side only."""
    cam = np.asarray(PEOPLE_CAM, np.float64)
    hms, rdepth, _, _ = synth_scene(4, seed=900)
    root_d = (rdepth * 2.0).astype(np.float32)
    yy, xx = np.mgrid[0:128, 0:208].astype(np.float32)
    det_d = np.stack([10.0 * np.sin(xx / (5.0 + k) + k) * np.cos(yy / (7.0 + k)) for k in range(14)]).astype(np.float32)
    scale = float(np.abs(hms[:15]).max())
    ref_h, hip_h = hms.copy(), hms.copy()
    c, y, x = 3, 100, 180                                              # an empty corner of channel 3
    assert float(np.abs(hms[c, y - 3:y + 4, x - 3:x + 5]).max()) < 0.2 * scale     # (background only)
    top = np.float32(0.5 * scale)
    ref_h[c, y, x], ref_h[c, y, x + 1] = top, np.float32(top - delta_rel * scale)
    hip_h[c, y, x], hip_h[c, y, x + 1] = np.float32(top - delta_rel * scale), top
    if bump is not None:
        hip_h[5, 20, 190] = np.float32(bump * scale)                   # channel 5, nothing near it in the reference

    def path(h):
        bodys, peaks, _ = O.connect(h, root_d, 2, True)
        p2, p3, rz = O.lift(bodys, det_d, root_d, cam)
        return dict(peaks=peaks, bodys=bodys, p2=p2, p3=p3, rz=rz, hms=h, det_d=det_d, root_d=root_d)
    return path(hip_h), path(ref_h)


def test_a_flipped_plateau_is_a_tie_only_below_the_margin_and_a_new_maximum_never_is():
    """ADVERSARIAL for the PEAK classifier: (a) two neighbouring pixels 2.5e-7 of the map scale apart (four fp32 steps), winner flipped in one path -- the
    situation the split-precision backbone produces a few times per 10 000 candidates -- counts as differing peaks, none of them clear;
    (b) the same flip with the pixels 1e-4 apart (no rounding of the backbone explains that) must be a CLEAR mismatch; (c) a local
    maximum that exists in one path only, 30 % of the map scale high, must be a clear mismatch whatever else agrees."""
    hip, ref = _one_peak_maps(2.5e-7)
    m = parity.compare([hip], [ref])
    assert m["peaks_differing"] == 2 and m["peaks_clear_mismatch"] == 0 and m["peaks_differing_max_margin"] < parity.NEAR_TIE, m
    hip, ref = _one_peak_maps(1e-4)
    m = parity.compare([hip], [ref])
    assert m["peaks_differing"] == 2 and m["peaks_clear_mismatch"] == 2, m
    hip, ref = _one_peak_maps(0.0, bump=0.3)                            # (exact plateau: neither path keeps either pixel -- strict >)
    m = parity.compare([hip], [ref])
    assert m["peaks_clear_mismatch"] >= 1 and m["peaks_differing_max_margin"] > 0.1, m


def test_a_peak_table_that_lost_a_peak_of_its_own_maps_is_a_mismatch_not_a_tie():
    """ADVERSARIAL: identical maps in both paths (so no decision can be called close), but the HIP side's peak TABLE lacks one peak its
    maps contain -- what a broken scan / rank step of the NMS kernel would produce.  peak_match must drop below 1 and the missing
    peak must be counted unmatched; nothing in the tie accounting may absorb it."""
    def tamper(bodys, peaks):
        ch = int(np.argmax(peaks[:, 0, 0]))
        n = int(peaks[ch, 0, 0])
        peaks[ch, 1:n] = peaks[ch, 2:n + 1]                             # drop the first centroid of the fullest channel
        peaks[ch, n] = 0
        peaks[ch, 0, 0] = n - 1
    hip, ref = _scene_paths(820, 0.0, tamper)
    m = parity.compare([hip], [ref])
    assert m["peaks_differing"] == 0 and m["peaks_clear_mismatch"] == 0             # the maps agree everywhere
    assert m["peak_match"] < 1.0 and m["peaks_unmatched"] >= 1, m
