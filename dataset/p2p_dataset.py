"""Import-path shim: `from dataset.p2p_dataset import P2PDataset` (exps/refinenet_root2, tests/golden/gen_golden_gt.py) keeps
resolving; the class lives in tools/p2p_dataset.py (RefineNet training input, outside the inference hot path)."""
from tools.p2p_dataset import P2PDataset  # noqa: F401
