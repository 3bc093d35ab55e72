#!/usr/bin/env python3
"""GPU clock / power / temperature telemetry beside a timing.

Every timing tool of this repository (bench.py, tools/probe.py, the A/B scripts) records the shader clock, the socket power
and the temperature of the GPU it runs on WHILE the timed region runs, so that two timings can be compared only when their
clock states agree.  Two sources, the first that works:

  * sysfs hwmon of the amdgpu device (`freq1_input` = current sclk in Hz, `power1_average` / `power1_input` in uW,
    `temp1_input` in m°C) — a file read, sampled every 5 ms by a thread;
  * `rocm-smi --showclocks --showpower --showtemp --json` — a process per sample (~0.2-0.4 s each), sampled back to back.

    with Sampler(device=0) as tm:
        ... timed region ...
    tm.summary() -> {"source": ..., "n": ..., "sclk_mhz": {"min","median","max"}, "power_w": {...}, "temp_c": {...}}

`comparable(a, b, tol=0.02)` is what the A/B scripts call: False when the median sclk of two summaries differs by more than
2 % (or when either has no clock samples).
"""
from __future__ import annotations

import glob
import json
import os
import statistics
import subprocess
import threading
import time


def _pci_bdf_of(device: int):
    """PCI address ("0000:75:00.0") of HIP device `device`, through torch when it is already imported; else None."""
    import sys
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(device)
            return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pass
    return None


def _hwmon_dir(device: int):
    """hwmon directory of the amdgpu card that is HIP device `device` (matched by PCI address: a container sees the sysfs
    nodes of every GPU of the host, not only its own), else of the `device`-th card with a freq1_input, else None."""
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(d, "freq1_input")):
            cands.append(d)
    bdf = _pci_bdf_of(device)
    if bdf:
        for d in cands:
            if os.path.realpath(os.path.join(d, "device")).endswith(bdf):
                return d
    if 0 <= device < len(cands):
        return cands[device]
    return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def _smi_sample(device: int):
    """One rocm-smi call -> (sclk MHz, power W, temp C), None where a field is missing."""
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
        rec = json.loads(r.stdout)
        card = rec.get(f"card{device}") or next(iter(rec.values()))
    except Exception:
        return None, None, None
    sclk = power = temp = None
    for k, v in card.items():
        kl = k.lower()
        try:
            if "sclk" in kl and "clock" in kl and sclk is None:
                sclk = float(str(v).strip("()").lower().replace("mhz", ""))
            elif "power" in kl and "(w)" in kl and power is None:
                power = float(v)
            elif "temperature" in kl and temp is None and ("junction" in kl or "edge" in kl or "hotspot" in kl):
                temp = float(v)
        except ValueError:
            pass
    return sclk, power, temp


class Sampler:
    def __init__(self, device: int = 0, period_s: float = 0.005):
        self.device, self.period = device, period_s
        self.samples = []          # (t, sclk_mhz, power_w, temp_c)
        self._stop = threading.Event()
        self._thr = None
        self.hw = _hwmon_dir(device)
        self.source = "sysfs-hwmon" if self.hw else "rocm-smi"

    def _one(self):
        if self.hw:
            f = _read_int(os.path.join(self.hw, "freq1_input"))
            p = _read_int(os.path.join(self.hw, "power1_average"))
            if p is None:
                p = _read_int(os.path.join(self.hw, "power1_input"))
            t = None
            for k in (1, 2, 3):      # temp1 = edge where it exists; this box has temp2 (junction) and temp3 (memory) only
                t = _read_int(os.path.join(self.hw, f"temp{k}_input"))
                if t is not None:
                    break
            return (f / 1e6 if f else None, p / 1e6 if p else None, t / 1e3 if t else None)
        return _smi_sample(self.device)

    def _run(self):
        while not self._stop.is_set():
            s = self._one()
            self.samples.append((time.perf_counter(),) + tuple(s))
            if self.hw:
                self._stop.wait(self.period)

    def start(self):
        self._stop.clear()
        self._thr = threading.Thread(target=self._run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=15)
        return self

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()

    def summary(self, t0: float | None = None, t1: float | None = None):
        """Statistics over the samples taken in [t0, t1] (perf_counter times; default: all)."""
        rows = [s for s in self.samples if (t0 is None or s[0] >= t0) and (t1 is None or s[0] <= t1)]

        def stat(idx):
            v = [r[idx] for r in rows if r[idx] is not None]
            if not v:
                return None
            return {"min": round(min(v), 1), "median": round(statistics.median(v), 1), "max": round(max(v), 1)}
        return {"source": self.source, "n": len(rows), "sclk_mhz": stat(1), "power_w": stat(2), "temp_c": stat(3)}


def comparable(a: dict, b: dict, tol: float = 0.02, power_tol: float = 0.10) -> bool:
    """True when two telemetry summaries ran in the same clock state: median shader clock within `tol` AND — when both carry
    power samples — median socket power within `power_tol`.  The power test is not a luxury: after 20 ms of idle time the
    reported sclk of an MI355X is back at 2.39 GHz at once while the socket power (and the frame rate: -18 %) takes tens of
    frames to return to the sustained figure (profiles/r04_clock_states.txt)."""
    try:
        fa, fb = a["sclk_mhz"]["median"], b["sclk_mhz"]["median"]
    except (KeyError, TypeError):
        return False
    if abs(fa - fb) > tol * max(fa, fb):
        return False
    try:
        pa, pb = a["power_w"]["median"], b["power_w"]["median"]
    except (KeyError, TypeError):
        return True
    return abs(pa - pb) <= power_tol * max(pa, pb)


if __name__ == "__main__":
    import sys
    dur = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    with Sampler() as tm:
        time.sleep(dur)
    print(json.dumps(tm.summary()))
