"""Shader clock and package power of one GPU while a timed region runs, from the amdgpu hwmon files (world-readable sysfs: no rocm-smi
process, no root).  MI355X clocks to its power budget: under this bench the package sits at its ~1.4 kW cap and the shader clock at ~2.0 GHz,
not at the 2.4 GHz the MFMA peak is quoted at (MI355X_MICROARCH.md, "DVFS give-back"; profiles/r6_v21_clocks_power_under_bench.txt), so the
bench line carries what the box actually held.  Everything here is best effort: a missing file gives None, never an error."""
import glob
import os
import threading
import time

MAX_SCLK_MHZ = 2400.0          # the clock the dense-MFMA peak is quoted at


def _hwmon_dir(device_index=0):
    """hwmon directory of torch's cuda:<device_index>, matched by PCI address; the only amdgpu hwmon if the address is not available."""
    cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    cands = [c for c in cands if os.path.exists(os.path.join(c, "freq1_input"))]
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        for c in cands:
            if os.path.realpath(os.path.join(c, "..", "..")).endswith(addr):
                return c
    except Exception:
        pass
    return cands[0] if len(cands) == 1 else None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


class ClockSampler:
    """with ClockSampler(dev_index) as s: <timed region>; s.summary() -> {"sclk_mhz": mean, "power_w": mean, ...} or None."""

    def __init__(self, device_index=0, period_s=0.02):
        self.dir = _hwmon_dir(device_index)
        self.period = period_s
        self.samples = []
        self._stop = threading.Event()
        self._thr = None

    def _loop(self):
        f = os.path.join(self.dir, "freq1_input")
        pw = next((os.path.join(self.dir, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(self.dir, n))), None)
        while not self._stop.is_set():
            hz = _read_int(f)
            uw = _read_int(pw) if pw else None
            if hz:
                self.samples.append((time.perf_counter(), hz / 1e6, uw / 1e6 if uw else None))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.dir:
            self._thr = threading.Thread(target=self._loop, name="clock-sampler", daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=1.0)
        return False

    def summary(self, t0=None, t1=None):
        s = [x for x in self.samples if (t0 is None or x[0] >= t0) and (t1 is None or x[0] <= t1)]
        if not s:
            return None
        mhz = sorted(x[1] for x in s)
        pw = [x[2] for x in s if x[2] is not None]
        return {"sclk_mhz": sum(mhz) / len(mhz), "sclk_mhz_min": mhz[0], "sclk_mhz_max": mhz[-1],
                "power_w": (sum(pw) / len(pw)) if pw else None, "samples": len(s),
                "clock_share_of_max": sum(mhz) / len(mhz) / MAX_SCLK_MHZ,
                "source": "amdgpu hwmon freq1_input / power1_average, sampled every %d ms inside the timed region" % int(self.period * 1e3)}
