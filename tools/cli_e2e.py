#!/usr/bin/env python3
"""End-to-end rate of the SHIPPED command line on a folder of images:

    python exps/stage3_root2/test.py -t run_inference -d test --batch_size 8 --dataset_path <folder> [--device_preprocess 1]

decode -> [host resize | H2D of the uint8 image + GPU pre-processing] -> backbone -> association -> lifting -> records -> JSON, timed
inside the CLI's own loop (SMAP_CLI_TIMING) and around the whole process.  The folder is generated here: N images of two source
resolutions -- the bench's 8 frames as 832x512 PNGs and as 1664x1024 JPEGs (every pixel doubled: the pre-processing's 2x shrink gives
the frame back) -- so that the network sees the workload the bench line is quoted on (~20 skeletons per frame: association, lifting
and record building do real work); a second folder holds the same pictures as .npy (no decoder).

    python tools/cli_e2e.py [--images 256] [--out profiles/r5_cli_e2e.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def bench_frames_as_images(cfg_means, cfg_stds):
    """The bench's 8 input frames (randn, seed 1234: the frames the workload's heads were calibrated on, ~20 skeletons each) as uint8
    BGR images of the network size: pixel = (x * std + mean) * 255, rounded and clipped -- what a decoder hands the pre-processing."""
    x = torch.randn(8, 3, 512, 832, generator=torch.Generator().manual_seed(1234))
    mean = torch.tensor(cfg_means).view(1, 3, 1, 1)
    std = torch.tensor(cfg_stds).view(1, 3, 1, 1)
    return ((x * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()     # [8,512,832,3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cli_e2e.json"))
    args = ap.parse_args()
    from PIL import Image
    from benchkit.workload import make_cfg, people_state_dict
    from smap_amd.model.smap import SMAP
    tmp = tempfile.mkdtemp(prefix="smap_cli_e2e_")
    enc, raw = os.path.join(tmp, "encoded"), os.path.join(tmp, "npy")
    os.makedirs(enc), os.makedirs(raw)
    from exps.stage3_root2.config import cfg as run_cfg
    t0 = time.perf_counter()
    base = bench_frames_as_images(list(run_cfg.INPUT.MEANS), list(run_cfg.INPUT.STDS))
    sizes = []
    # two source resolutions: 832x512 (no resize) as PNG, and 1664x1024 = every pixel doubled (the 2x shrink of the pre-processing is the
    # 2x2 box mean: it returns the original frame) as JPEG quality 98 (4:4:4); 8 + 8 files are written, the rest of the folder links to them
    written = {}
    for i in range(args.images):
        f, big = i % 8, (i // 8) % 2
        ext = "jpg" if big else "png"
        if (f, big) not in written:
            img = base[f] if not big else np.repeat(np.repeat(base[f], 2, axis=0), 2, axis=1)
            path = os.path.join(enc, f"src_{f}_{big}.{ext}.data")
            if big:
                Image.fromarray(img[:, :, ::-1]).save(path, format="JPEG", quality=98, subsampling=0)      # 4:4:4: the frame comes back within 0.6 grey levels
            else:
                Image.fromarray(img[:, :, ::-1]).save(path, format="PNG", compress_level=1)
            npy = os.path.join(tmp, f"src_{f}_{big}.npy")          # (outside the folder: the loader lists every *.npy below it)
            np.save(npy, img)
            written[(f, big)] = (path, npy)
            sizes.append(os.path.getsize(path))
        os.symlink(written[(f, big)][0], os.path.join(enc, f"im{i:05d}.{ext}"))
        os.symlink(written[(f, big)][1], os.path.join(raw, f"im{i:05d}.npy"))
    gen_s = time.perf_counter() - t0
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = people_state_dict(net.state_dict(), "smooth")
    torch.save({"model": sd}, os.path.join(tmp, "SMAP.pth"))
    runs = []
    cases = [("encoded jpg/png, GPU pre-processing, 16 decode threads (default)", enc, ["--device_preprocess", "1"], {}),
             ("encoded jpg/png, GPU pre-processing, 8 decode threads", enc, ["--device_preprocess", "1"], {"SMAP_DECODE_THREADS": "8"}),
             ("encoded jpg/png, GPU pre-processing, 32 decode threads", enc, ["--device_preprocess", "1"], {"SMAP_DECODE_THREADS": "32"}),
             ("encoded jpg/png, GPU pre-processing, 16 decode PROCESSES (SMAP_DECODE_PROCS)", enc, ["--device_preprocess", "1"], {"SMAP_DECODE_PROCS": "16"}),
             ("encoded jpg/png, GPU pre-processing, 32 decode PROCESSES", enc, ["--device_preprocess", "1"], {"SMAP_DECODE_PROCS": "32"}),
             ("encoded jpg/png, GPU pre-processing, ONE decode thread (round-4 loader)", enc, ["--device_preprocess", "1"], {"SMAP_DECODE_THREADS": "1"}),
             (".npy frames (no decoder), GPU pre-processing", raw, ["--device_preprocess", "1"], {}),
             ("encoded jpg/png, host pre-processing (the reference's DataLoader path), first 128 images", enc + "_few", [], {})]
    os.makedirs(enc + "_few")
    for n in sorted(os.listdir(enc)):
        if n.startswith("im") and int(n[2:7]) < 128:
            os.symlink(os.path.realpath(os.path.join(enc, n)), os.path.join(enc + "_few", n))
    for name, folder, extra, env_extra in cases:
        timing = os.path.join(tmp, "timing.json")
        # the plan cache of this run lives in the temporary folder: the FIRST case builds the schedules (first_submit_s = seconds of weight
        # packing), every later case is a "second start" that loads them through smap_plan_create_from_blob
        env = dict(os.environ, PROJECT_HOME=tmp, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), SMAP_CLI_TIMING=timing,
                   SMAP_PLAN_CACHE=os.path.join(tmp, "plan_cache"), **env_extra)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "exps", "stage3_root2", "test.py"), "-p", os.path.join(tmp, "SMAP.pth"),
                            "-t", "run_inference", "-d", "test", "--batch_size", str(args.batch), "--dataset_path", folder, "--json_name", "e2e"] + extra,
                           capture_output=True, text=True, env=env, cwd=tmp, timeout=1500)
        wall = time.perf_counter() - t0
        rec = {"case": name, "returncode": r.returncode, "process_wall_s": wall, "plan_cache": "cold (schedules built and stored)" if not runs else "warm (second start)"}
        if r.returncode == 0 and os.path.exists(timing):
            rec.update(json.load(open(timing)))
            out = os.path.join(tmp, "model_logs", "stage3_root2", "result", "stage3_root2_run_inference_test_e2e.json")
            res = json.load(open(out))
            rec["records_in_result_file"] = len(res["3d_pairs"])
            rec["result_file_MB"] = os.path.getsize(out) / 1e6
        else:
            rec["stderr_tail"] = r.stderr[-1500:]
        runs.append(rec)
        print(json.dumps(rec), flush=True)
    out = {"images": args.images, "batch_size": args.batch, "sources": "the bench's 8 frames: 832x512 PNG and 1664x1024 JPEG q98 4:4:4 (pixel-doubled), alternating in groups of 8",
           "mean_file_KB": float(np.mean(sizes)) / 1e3, "generation_s": gen_s, "host_cpus_allowed": len(os.sched_getaffinity(0)),
           "gpu": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None, "runs": runs}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
