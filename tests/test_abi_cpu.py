"""CPU suite: the C-ABI library loads and exports every symbol include/smap_hip.h declares;
the host-side mirrors validate arguments and refuse to run without the GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from helpers import make_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "smap_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smap_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from smap_amd import lib as L
    so = ctypes.CDLL(L.SO_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/smap_hip.h but not exported"
    assert sorted(L.SYMBOLS) == names
    # ... and the converse: the dynamic symbol table holds NOTHING but the header's functions (-fvisibility=hidden + the linker
    # version script of smap_amd/build.py: no mangled internals, no debug helpers, no __hip_cuid_* markers)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == names, sorted(set(exported) ^ set(names))
    L.load()
    assert "gfx950" in L.version()
    assert so.smap_sizeof_op() == ctypes.sizeof(L.SmapOp)


def test_plan_blob_round_trip_and_workspace_bytes():
    """The serialised schedule (Graph.blob -> smap_plan_create_from_blob, no GPU involved in either): header, ops and weight
    section come back as written, the library sizes the buffers from the ops alone (smap_workspace_bytes) within what the
    header states, and damaged blobs are refused."""
    import ctypes as C
    from recipe import recipe_state_dict
    from smap_amd import lib as L
    from smap_amd.engine import Graph
    from smap_amd.model.smap import SMAP
    lib = L.load()
    torch.manual_seed(0)
    sd = recipe_state_dict(SMAP(make_cfg((16, 24))).state_dict())
    for precision, flip in (("x3", None), ("f16", list(range(43)))):
        g = Graph(sd, 2, 64, 96, precision=precision, flip_pair=flip)
        g.allocate()
        blob = g.blob()
        plan, info = C.c_void_p(), L.BlobInfo()
        assert lib.smap_plan_create_from_blob(blob, len(blob), C.byref(plan), C.byref(info)) == 0
        assert (info.frames, info.H, info.W, info.out_h, info.out_w) == (2, 64, 96, 16, 24)
        assert (info.n_hms, info.n_det, info.n_root, info.precision) == (43, 14, 1, int(precision == "x3"))
        assert info.arena_bytes == g.arena_bytes and info.out_bytes == g.out_bytes + 4 and info.status_off == g.out_bytes
        assert (info.hms_off, info.det_off, info.root_off) == (0, 2 * 43 * 16 * 24 * 4, 2 * 57 * 16 * 24 * 4)
        assert blob[info.weights_offset:info.weights_offset + info.weights_bytes] == g.weight_blob().numpy().tobytes()
        ar, ob = C.c_int64(), C.c_int64()
        assert lib.smap_workspace_bytes(plan, C.byref(ar), C.byref(ob)) == 0
        assert ob.value == info.out_bytes and 0.5 * g.arena_bytes < ar.value <= g.arena_bytes     # ops touch (almost) all of the arena
        lib.smap_plan_destroy(plan)
        bad = bytearray(blob)
        bad[0:1] = b"X"
        assert lib.smap_plan_create_from_blob(bytes(bad), len(bad), C.byref(plan), None) == -1      # magic
        assert lib.smap_plan_create_from_blob(blob, len(blob) // 2, C.byref(plan), None) == -1      # truncated
        hdr = L.BlobHeader.from_buffer_copy(blob[:C.sizeof(L.BlobHeader)])
        assert hdr.version == L.BLOB_VERSION == 2
        hdr.version = 1                                                                              # a blob of an older smap_op meaning, same struct size
        old = bytes(hdr) + blob[C.sizeof(hdr):]
        assert lib.smap_plan_create_from_blob(old, len(old), C.byref(plan), None) == -1
        hdr = L.BlobHeader.from_buffer_copy(blob[:C.sizeof(L.BlobHeader)])
        hdr.arena_bytes = 4096                                                                       # header smaller than what the ops touch
        lied = bytes(hdr) + blob[C.sizeof(hdr):]
        assert lib.smap_plan_create_from_blob(lied, len(lied), C.byref(plan), None) == -1
        # ... the sizes in `info` (what the header tells a host to allocate from) are held to the same ops
        hdr = L.BlobHeader.from_buffer_copy(blob[:C.sizeof(L.BlobHeader)])
        hdr.info.out_bytes = 16
        lied = bytes(hdr) + blob[C.sizeof(hdr):]
        assert lib.smap_plan_create_from_blob(lied, len(lied), C.byref(plan), None) == -1
        # offsets + sizes that wrap int64, and a weight section that ends before the packed weights of an op do
        for field, val in (("ops_offset", 2 ** 63 - 8), ("weights_offset", 2 ** 63 - 8), ("weights_bytes", 2 ** 63 - 1)):
            hdr = L.BlobHeader.from_buffer_copy(blob[:C.sizeof(L.BlobHeader)])
            setattr(hdr, field, val)
            lied = bytes(hdr) + blob[C.sizeof(hdr):]
            assert lib.smap_plan_create_from_blob(lied, len(lied), C.byref(plan), None) == -1, field
        hdr = L.BlobHeader.from_buffer_copy(blob[:C.sizeof(L.BlobHeader)])
        nops, so = hdr.n_ops, C.sizeof(L.SmapOp)
        ops = (L.SmapOp * nops).from_buffer_copy(blob[hdr.ops_offset:hdr.ops_offset + nops * so])
        last_w = max(range(nops), key=lambda i: ops[i].w_off if ops[i].kind == 0 else -1)        # the conv whose weights lie last in the section
        for field in ("w_off", "bias_off") + (("head_w_off", "tail_w_off", "tail_bias_off", "head_bias_off") if precision == "x3" else ()):
            i = last_w if field in ("w_off", "bias_off") else next(i for i in range(nops) if ops[i].head_cin > 0)
            cut = bytearray(blob)
            o = L.SmapOp.from_buffer_copy(cut[hdr.ops_offset + i * so:hdr.ops_offset + (i + 1) * so])
            setattr(o, field, hdr.weights_bytes - 16)                                            # START inside the section, extent beyond its end
            cut[hdr.ops_offset + i * so:hdr.ops_offset + (i + 1) * so] = bytes(o)
            assert lib.smap_plan_create_from_blob(bytes(cut), len(cut), C.byref(plan), None) == -1, field


@pytest.mark.parametrize("spec", ["", "64:90", "64:91", "64:91+64:93", "64:91,128:94+64:93"])
def test_library_accepts_the_full_size_schedules(monkeypatch, spec):
    """smap_plan_create on the benchmarked schedules (8 and 16 frames at 512x832, with and without the whole-Bottleneck launches
    of csrc/convb.hip): what the GPU box will be asked to run validates here, without a GPU."""
    import ctypes as C
    from recipe import recipe_state_dict
    from smap_amd import lib as L
    from smap_amd.engine import Graph
    from smap_amd.model.smap import SMAP
    spec, _, first = spec.partition("+")
    monkeypatch.setenv("SMAP_BLOCK", spec)
    monkeypatch.setenv("SMAP_BLOCK_FIRST", first)
    torch.manual_seed(0)
    sd = SMAP(make_cfg((128, 208))).state_dict()
    for B in (8, 16):
        g = Graph(sd, B, 512, 832, precision="x3")
        g.allocate()
        assert sum(1 for op in g.ops if "head" in op.p) == ((9 if first else 6) if spec else 0) + (9 if "128:" in spec else 0)
        h = C.c_void_p()
        assert L.load().smap_plan_create(g.emit(), len(g.ops), C.byref(h)) == 0, (spec, B)
        L.load().smap_plan_destroy(h)


def test_missing_library_fails_loudly(monkeypatch):
    from smap_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "SO_PATH", "/nonexistent/libsmap_hip.so")
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        L.load()


def test_dapalib_argument_validation():
    import dapalib
    hms = torch.zeros(43, 128, 208)
    with pytest.raises(ValueError, match="GPU"):
        dapalib.connect(hms, torch.zeros(128, 208))
    with pytest.raises(ValueError, match="shape"):
        dapalib.connect(torch.zeros(42, 128, 208), torch.zeros(128, 208))
    with pytest.raises(ValueError, match="float32"):
        dapalib.extract(torch.zeros(43, 128, 208, dtype=torch.float64))


def test_smap_forward_has_no_cpu_path():
    from model.smap import SMAP
    net = SMAP(make_cfg((16, 24)))
    with pytest.raises(RuntimeError, match="eval"):
        net(torch.zeros(1, 3, 64, 96))
    net.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 64, 96))
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 64, 96), valids=torch.zeros(1), labels=torch.zeros(1))


def test_product_never_imports_oracle():
    """The oracle and benchkit (workloads + the end-to-end checker, which imports the oracle) are test / bench
    infrastructure: nothing under the product tree may reference them."""
    bad = []
    for base in ("smap_amd", "model", "exps", "dataset", "lib", "dapalib.py"):
        p = os.path.join(ROOT, base)
        files = [p] if os.path.isfile(p) else [os.path.join(d, f) for d, _, fs in os.walk(p) for f in fs
                                               if f.endswith((".py", ".hip", ".h", ".cpp"))]
        for f in files:
            if re.search(r"^\s*(from|import)\s+(oracle|benchkit)\b|oracle_lib|smap_oracle", open(f).read(), flags=re.M):
                bad.append(f)
    assert not bad, bad


def test_no_kernel_reads_the_dispatch_packet_or_uses_scratch(tmp_path):
    """A rule this code base learned the hard way (EXPERIMENTS.md R3.6): a run-time-indexed private array in a kernel is promoted
    to LDS by the compiler and addressed through the AQL dispatch packet -- the head sum written that way returned wrong values
    in 10-30 % of the launches that overlapped another stream's kernels, and never in a serial test.  Checked on the compiled
    code of EVERY kernel of the library (device-only assembly, hipcc cross-compiles without a GPU): no dispatch-packet
    pointer, no scratch, no dynamic stack."""
    import re
    import subprocess
    from smap_amd import build as B
    procs = []
    for src, extra in B.SOURCES:
        out = tmp_path / (src + ".s")
        cmd = [B._hipcc()] + [f for f in B.COMMON if f not in ("-fPIC",)] + extra + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
        procs.append((src, out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    n = 0
    for src, out, p in procs:
        _, err = p.communicate(timeout=1200)
        assert p.returncode == 0, (src, err[-2000:])
        txt = out.read_text()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
            name, body = m.group(1), m.group(2)
            for field in ("amdhsa_user_sgpr_dispatch_ptr", "amdhsa_private_segment_fixed_size", "amdhsa_uses_dynamic_stack"):
                v = re.search(r"\.%s (\d+)" % field, body)
                assert v is not None and int(v.group(1)) == 0, (src, name, field, v and v.group(1))
            n += 1
    assert n >= 100          # conv.hip alone instantiates ~90 kernels


def test_status_words_follow_the_frame_count():
    """SMAP_STATUS_WORDS(frames) int32 words behind the maps (one bit per output frame, 31 per word): the packer, the blob header and the
    library's own sizing agree for launches of 1, 31, 32 and 40 frames."""
    import ctypes as C
    from recipe import recipe_state_dict
    from smap_amd import lib as L
    from smap_amd.engine import Graph
    from smap_amd.model.smap import SMAP
    lib = L.load()
    torch.manual_seed(0)
    sd = recipe_state_dict(SMAP(make_cfg((16, 24))).state_dict())
    for frames, words in ((1, 1), (31, 1), (32, 2), (40, 2)):
        g = Graph(sd, frames, 64, 96, precision="x3")
        g.allocate()
        assert g.status_words == words
        blob = g.blob()
        plan, info = C.c_void_p(), L.BlobInfo()
        assert lib.smap_plan_create_from_blob(blob, len(blob), C.byref(plan), C.byref(info)) == 0
        assert info.out_bytes == g.out_bytes + 4 * words and info.status_off == g.out_bytes
        ob = C.c_int64()
        assert lib.smap_workspace_bytes(plan, None, C.byref(ob)) == 0 and ob.value == info.out_bytes
        lib.smap_plan_destroy(plan)
