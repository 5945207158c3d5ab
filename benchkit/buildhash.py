"""What the committed counter files under profiles/ were measured ON: a hash of the sources that decide which kernels run and how
(smap_amd/csrc/*, include/smap_hip.h, the tile tables, the schedule builder).  tools/prof_traffic.py and tools/prof_mfma.py stamp it into
profiles/hbm_traffic_x3.json / mfma_utilisation_x3.json; bench.py quotes those counters only while the hash of the tree it runs from is
the same, and prints null otherwise -- a kernel change can no longer leave stale counters in the bench line."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hashed_files(root=ROOT):
    pats = ("smap_amd/csrc/*.hip", "smap_amd/csrc/*.h", "include/smap_hip.h", "smap_amd/tile_table*.json", "smap_amd/engine.py")
    return sorted(f for p in pats for f in glob.glob(os.path.join(root, p)))


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for f in hashed_files(root):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def stamp(root=ROOT):
    """Fields the profiling tools add to the JSON they write (the commit comes from the visit script: the GPU box has no .git)."""
    return {"build_hash": source_hash(root), "commit": os.environ.get("SMAP_GIT_COMMIT") or None}


def counters_for_build(path, root=ROOT):
    """(dict or None, info): the counter file when it was measured on THIS tree's kernels, else None; info says why."""
    if not os.path.exists(path):
        return None, {"file": os.path.basename(path), "present": False, "counters_match_build": False}
    t = json.load(open(path))
    ok = t.get("build_hash") == source_hash(root)
    return (t if ok else None), {"file": os.path.basename(path), "present": True, "counters_match_build": ok,
                                 "measured_on_build": t.get("build_hash"), "measured_on_commit": t.get("commit"), "this_build": source_hash(root)}
