"""Board power and shader clock of the GPU a process runs on, sampled by a host thread from read-only sysfs files
(hwmon power1_average / power1_input in microwatts, freq1_input in Hz) - nothing is queued on any HIP stream and nothing is written.

bench.py brackets every timed region with `PowerSampler.window()` so that the driver's record carries the package power and sclk each
number was measured at (a ViT-B/16 forward sits at the 1.4 kW cap, vit_small below it).  On a host where the files are
absent or unreadable the sampler reports `{"samples": 0}` and the bench line says so - it never fails the run."""
from __future__ import annotations

import glob
import os
import threading
import time


def device_pci_address(index: int = 0):
    """dddd:bb:dd.f of torch's device `index`, or None (then every card of the host is sampled and the busiest one in a window is reported)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        return None


class PowerSampler(threading.Thread):
    def __init__(self, pci=None, period_s: float = 0.02):
        super().__init__(daemon=True)
        self.period = period_s
        self.cards = self._find(pci) or self._find(None)
        self.samples, self._halt = [], threading.Event()

    @staticmethod
    def _find(pci):
        cards = []
        for b in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            addr = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(b))))
            if pci is not None and addr.lower() != pci.lower():
                continue
            files = {}
            for key, names in (("power", ("power1_average", "power1_input")), ("sclk", ("freq1_input",)), ("cap", ("power1_cap",))):
                for n in names:
                    if os.path.exists(os.path.join(b, n)):
                        files[key] = os.path.join(b, n)
                        break
            if "power" in files or "sclk" in files:
                cards.append(files)
        return cards

    def run(self):
        while not self._halt.is_set():
            row = {"t": time.perf_counter(), "cards": []}
            for files in self.cards:
                c = {}
                for k, f in files.items():
                    if k == "cap":
                        continue
                    try:
                        with open(f) as fh:
                            c[k] = float(fh.read().strip())
                    except (OSError, ValueError):
                        pass
                row["cards"].append(c)
            self.samples.append(row)
            if len(self.samples) > 400_000:               # (~2 h at 50 Hz: a long-lived process keeps the recent half)
                del self.samples[:200_000]
            self._halt.wait(self.period)

    def stop(self):
        self._halt.set()

    def cap_w(self):
        for files in self.cards:
            try:
                with open(files["cap"]) as fh:
                    return round(float(fh.read().strip()) * 1e-6, 1)
            except (KeyError, OSError, ValueError):
                continue
        return None

    def mean(self, t0: float, t1: float, skip_frac: float = 0.0):
        """Mean power (W) and sclk (MHz) of the samples in [t0 + skip_frac (t1 - t0), t1] for the card that drew the most power there."""
        t0 = t0 + skip_frac * (t1 - t0)
        sel = [s for s in self.samples if t0 <= s["t"] <= t1]
        best = {}
        for ci in range(len(self.cards)):
            pw = [s["cards"][ci]["power"] for s in sel if "power" in s["cards"][ci]]
            ck = [s["cards"][ci]["sclk"] for s in sel if "sclk" in s["cards"][ci]]
            if pw and (not best or sum(pw) / len(pw) * 1e-6 > best["power_w"]):
                best = {"power_w": round(sum(pw) / len(pw) * 1e-6, 1), "power_w_max": round(max(pw) * 1e-6, 1)}
                if ck:
                    best["sclk_mhz"] = round(sum(ck) / len(ck) * 1e-6, 1)
                    best["sclk_mhz_min"] = round(min(ck) * 1e-6, 1)
            elif not pw and ck and not best:
                best = {"sclk_mhz": round(sum(ck) / len(ck) * 1e-6, 1)}
        best["samples"] = len(sel)
        return best

    class _Window:
        def __init__(self, sampler):
            self.s, self.t0, self.t1 = sampler, None, None

        def __enter__(self):
            self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            self.t1 = time.perf_counter()
            return False

        def result(self, skip_frac: float = 0.0):
            return self.s.mean(self.t0, self.t1, skip_frac)

    def window(self):
        return PowerSampler._Window(self)


_sampler = None


def sampler(index: int = 0) -> PowerSampler:
    """The process-wide sampler thread (started on first use)."""
    global _sampler
    if _sampler is None:
        _sampler = PowerSampler(device_pci_address(index))
        _sampler.start()
    return _sampler
