"""Image file -> uint8 HxWx3 BGR array (what cv2.imread(path, IMREAD_COLOR) hands the reference: dataset/custom_dataset.py:33 upstream),
and the same as a WORKER PROCESS for the decode-ahead loader of exps/stage3_root2/test.py (SMAP_DECODE_PROCS).

Only numpy and PIL are imported here: a worker starts in ~0.3 s and holds no torch, no GPU context.

    python -m dataset.decode <shared memory name> <slot bytes>
reads lines "<slot>\\t<path>" on stdin, decodes the file into bytes [slot * slot_bytes, ...) of the shared memory block and answers
"<slot> <height> <width>" on stdout ("<slot> -1 -1": the frame does not fit a slot -- the caller decodes it itself; "<slot> -2 -2
<message>": the decoder raised)."""
import sys

import numpy as np


def read_bgr(path):
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image, ImageOps
    # cv2.imread(IMREAD_COLOR) applies the EXIF orientation; PIL does not by itself.  BGR order comes out of PIL's own packer (one C
    # pass, 1.9 ms for a 1664x1024 frame): numpy's reversed-stride copy of the same bytes walks a 3-element inner loop and took 9 ms,
    # as long as the JPEG decode itself (EXPERIMENTS R6.11).  The result is a read-only array over PIL's bytes.
    im = ImageOps.exif_transpose(Image.open(path)).convert("RGB")
    w, h = im.size
    return np.frombuffer(im.tobytes("raw", "BGR"), np.uint8).reshape(h, w, 3)


def worker(shm_name, slot_bytes):
    from multiprocessing import shared_memory
    try:                                        # (the parent owns the block: keep this process's resource tracker out of it)
        shm = shared_memory.SharedMemory(name=shm_name, track=False)
    except TypeError:                           # Python < 3.13 has no `track`
        from multiprocessing import resource_tracker
        shm = shared_memory.SharedMemory(name=shm_name)
        try:
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
    buf = np.frombuffer(shm.buf, np.uint8)
    try:
        for line in sys.stdin:
            slot, _, path = line.rstrip("\n").partition("\t")
            slot = int(slot)
            try:
                img = read_bgr(path)
                if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                    raise ValueError("not an HxWx3 uint8 image")
                if img.nbytes > slot_bytes:
                    print(slot, -1, -1, flush=True)
                    continue
                np.copyto(buf[slot * slot_bytes:slot * slot_bytes + img.nbytes].reshape(img.shape), img)
                print(slot, img.shape[0], img.shape[1], flush=True)
            except Exception as e:             # the caller raises it in the consumer's thread
                print(slot, -2, -2, repr(e).replace("\n", " "), flush=True)
    finally:
        del buf
        shm.close()


if __name__ == "__main__":
    worker(sys.argv[1], int(sys.argv[2]))
