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
    L.load()
    assert "gfx950" in L.version()
    assert so.smap_sizeof_op() == ctypes.sizeof(L.SmapOp)


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
