"""Bench / test support shared by bench.py, __graft_entry__.smoke() and tests/: synthetic workloads, the deterministic
weight recipe and the end-to-end parity checker.  NOT part of the product (smap_amd/, model/, dapalib.py, exps/ never
import it); benchkit.parity imports the oracle and is therefore checker-only code."""
