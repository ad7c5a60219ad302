"""Board telemetry beside a timed leg of bench.py: average socket power and shader clock while the leg ran, and the MFMA rate
the part sustains on this box right now (csrc/probe.hip).  The bf16 kernels of this path run against the part's POWER limit
(DESIGN.md section 7): their rate moves with the box and its thermal state, and a record that carries {avg_W, avg_sclk_MHz,
measured_ceiling} explains such a difference instead of leaving it to prose.  Measurement aid only -- nothing on the data path.
"""
from __future__ import annotations

import glob
import os
import threading
import time
from typing import Dict, Optional

import torch


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _card_dir(device_index: int) -> Optional[str]:
    """/sys/class/drm/cardN/device of the torch device: matched by PCI bus id when torch exposes it, else the first card that has
    an amdgpu hwmon power file (1-GPU boxes)."""
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    want = None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:                                         # noqa: BLE001 - older torch: no PCI ids
        pass
    with_power = [c for c in cards if glob.glob(os.path.join(c, "hwmon", "hwmon*", "power1_*"))]
    if want:
        for c in with_power:
            if os.path.basename(os.path.realpath(c)).lower().startswith(want):
                return c
    return with_power[device_index] if device_index < len(with_power) else (with_power[0] if with_power else None)


class BoardSampler:
    """Background thread reading hwmon `power1_average` (or `power1_input`; microwatts) and the shader clock (`freq1_input` in Hz,
    or the starred line of `pp_dpm_sclk`) of the device's card every `period` seconds between start() and stop()."""

    def __init__(self, device_index: int = 0, period: float = 0.025):
        self.period = period
        self.card = _card_dir(device_index)
        self.power_file = self.freq_file = self.dpm_file = None
        self.cap_w = None                       # the board's power cap (hwmon power1_cap), what a power-limited leg runs against
        if self.card:
            hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*")))
            for h in hw:
                for n in ("power1_average", "power1_input"):
                    if self.power_file is None and _read(os.path.join(h, n)) not in (None, ""):
                        self.power_file = os.path.join(h, n)
                if self.freq_file is None and _read(os.path.join(h, "freq1_input")) not in (None, ""):
                    self.freq_file = os.path.join(h, "freq1_input")
                try:
                    self.cap_w = int(_read(os.path.join(h, "power1_cap"))) * 1e-6
                except (TypeError, ValueError):
                    pass
            d = os.path.join(self.card, "pp_dpm_sclk")
            if _read(d):
                self.dpm_file = d
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.watts, self.mhz = [], []

    @property
    def available(self) -> bool:
        return self.power_file is not None or self.freq_file is not None or self.dpm_file is not None

    def _sample(self):
        if self.power_file:
            v = _read(self.power_file)
            try:
                self.watts.append(int(v) * 1e-6)
            except (TypeError, ValueError):
                pass
        mhz = None
        if self.freq_file:
            v = _read(self.freq_file)
            try:
                mhz = int(v) * 1e-6
            except (TypeError, ValueError):
                mhz = None
        if mhz is None and self.dpm_file:
            for ln in (_read(self.dpm_file) or "").splitlines():
                if ln.rstrip().endswith("*"):
                    try:
                        mhz = float(ln.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
                    except (IndexError, ValueError):
                        mhz = None
        if mhz is not None:
            self.mhz.append(mhz)

    def _run(self):
        while not self._stop.is_set():
            self._sample()
            self._stop.wait(self.period)

    def start(self):
        self.watts, self.mhz = [], []
        self._stop.clear()
        if self.available:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def stop(self) -> Dict:
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        mean = lambda v: round(sum(v) / len(v), 1) if v else None
        return {"avg_W": mean(self.watts), "max_W": round(max(self.watts), 1) if self.watts else None,
                "cap_W": self.cap_w, "avg_sclk_MHz": mean(self.mhz), "samples": max(len(self.watts), len(self.mhz)),
                "source": {"power": self.power_file, "sclk": self.freq_file or self.dpm_file} if self.available else
                          "no amdgpu hwmon / pp_dpm_sclk file readable on this box"}

    def __enter__(self):
        return self.start()

    def __exit__(self, *a):
        self.last = self.stop()


PROBE_KINDS = {"bf16_random": 0, "bf16_constant": 1, "f32_random": 2, "f32_constant": 3}


def mfma_probe(kind: str, target_ms: float = 50.0, device_index: int = 0) -> Dict:
    """Run csrc/probe.hip's register-resident MFMA loop on every CU for about `target_ms` and return the sustained rate:
    {"TFLOP/s", "ms", "sclk_MHz" (shader clocks / 100 MHz reference ticks seen by workgroup 0), "operands"}."""
    from ..lib import call, query
    k = PROBE_KINDS[kind]
    dev = torch.device("cuda", device_index)
    wgs = max(1, query("tag_device_cu_count"))
    clocks = torch.zeros(3, dtype=torch.int64, device=dev)

    def run(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("tag_mfma_probe", k, iters, wgs, 7, clocks.data_ptr())
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    iters = 2000
    ms = run(iters)                                            # calibration launch (also the warm-up)
    iters = max(256, int(iters * target_ms / max(ms, 1e-3)))
    ms = run(iters)
    c = clocks.cpu().tolist()
    flop = query("tag_mfma_probe_flop", k, iters, wgs)
    return {"TFLOP/s": round(flop / (ms * 1e-3) / 1e12, 1), "ms": round(ms, 2),
            "sclk_MHz": round(c[0] / c[1] * 100.0, 0) if c[1] > 0 else None,
            "operands": kind, "workgroups": wgs, "what": "register-resident MFMA loop on every CU, no LDS / memory traffic"}
