#!/usr/bin/env python3
"""End-to-end rate of the SHIPPED command line on a folder of images:

    python exps/stage3_root2/test.py -t run_inference -d test --batch_size 8 --dataset_path <folder> [--device_preprocess 1]

decode -> [host resize | H2D of the uint8 image + GPU pre-processing] -> backbone -> association -> lifting -> records -> JSON, timed
inside the CLI's own loop (SMAP_CLI_TIMING) and around the whole process.  The folder is generated here: N images, half 1920x1080
JPEG (quality 90), half 1280x720 PNG, smooth synthetic content (decode cost of photographs, not of noise); a second folder holds the
same pictures as .npy (no decoder).  Weights: the bench's workload (benchkit/workload.py), so every frame carries ~20 skeletons.

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


def picture(rng, h, w):
    """Smooth colour fields + a few soft blobs + mild grain: compresses like a photograph."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        a, b, p, q = rng.uniform(0.002, 0.01, 4)
        img[..., c] = 110 + 60 * np.sin(a * xx + p * 50) * np.cos(b * yy + q * 50)
    for _ in range(12):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(20, 120)
        img += rng.uniform(-60, 60, 3) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))[..., None]
    img += rng.normal(0, 3, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cli_e2e.json"))
    args = ap.parse_args()
    from PIL import Image
    from benchkit.workload import make_cfg, people_state_dict
    from smap_amd.model.smap import SMAP
    tmp = tempfile.mkdtemp(prefix="smap_cli_e2e_")
    enc, raw = os.path.join(tmp, "encoded"), os.path.join(tmp, "npy")
    os.makedirs(enc), os.makedirs(raw)
    rng = np.random.default_rng(7)
    t0 = time.perf_counter()
    base = [picture(rng, 1080, 1920) for _ in range(4)] + [picture(rng, 720, 1280) for _ in range(4)]
    sizes = []
    for i in range(args.images):
        img = np.roll(base[(i % 2) * 4 + (i // 2) % 4], (i * 37) % 200, axis=1)       # BGR in memory; PIL writes RGB
        if i % 2 == 0:
            path = os.path.join(enc, f"im{i:04d}.jpg")
            Image.fromarray(img[:, :, ::-1]).save(path, quality=90)
        else:
            path = os.path.join(enc, f"im{i:04d}.png")
            Image.fromarray(img[:, :, ::-1]).save(path, compress_level=3)
        sizes.append(os.path.getsize(path))
        np.save(os.path.join(raw, f"im{i:04d}.npy"), img)
    gen_s = time.perf_counter() - t0
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = people_state_dict(net.state_dict(), "smooth")
    torch.save({"model": sd}, os.path.join(tmp, "SMAP.pth"))
    runs = []
    cases = [("encoded jpg/png, GPU pre-processing, decode threads (default)", enc, ["--device_preprocess", "1"], {}),
             ("encoded jpg/png, GPU pre-processing, ONE decode thread (round-4 loader)", enc, ["--device_preprocess", "1"], {"SMAP_DECODE_THREADS": "1"}),
             ("encoded jpg/png, host pre-processing (the reference's DataLoader path)", enc, [], {}),
             (".npy frames (no decoder), GPU pre-processing", raw, ["--device_preprocess", "1"], {})]
    for name, folder, extra, env_extra in cases:
        timing = os.path.join(tmp, "timing.json")
        env = dict(os.environ, PROJECT_HOME=tmp, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), SMAP_CLI_TIMING=timing, **env_extra)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "exps", "stage3_root2", "test.py"), "-p", os.path.join(tmp, "SMAP.pth"),
                            "-t", "run_inference", "-d", "test", "--batch_size", str(args.batch), "--dataset_path", folder, "--json_name", "e2e"] + extra,
                           capture_output=True, text=True, env=env, cwd=tmp, timeout=1500)
        wall = time.perf_counter() - t0
        rec = {"case": name, "returncode": r.returncode, "process_wall_s": wall}
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
    out = {"images": args.images, "batch_size": args.batch, "sources": "half 1920x1080 JPEG q90, half 1280x720 PNG (synthetic smooth content)",
           "mean_file_KB": float(np.mean(sizes)) / 1e3, "generation_s": gen_s, "host_cpus_allowed": len(os.sched_getaffinity(0)),
           "gpu": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None, "runs": runs}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
