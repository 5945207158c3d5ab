"""Shared test helpers (re-exported from benchkit/workload.py, which bench.py uses too)."""
from benchkit.workload import *  # noqa: F401,F403
from benchkit.workload import NJ, NL, PAIRS, TEMPLATE_CM, make_cfg, noise_scene, synth_scene  # noqa: F401
