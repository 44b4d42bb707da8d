"""Shader clock and board power of the GPU while a benchmark's timed region runs (bench.py: `clock_mhz_mean`, `power_w_mean`).

A power-capped MI355X runs a matrix-bound kernel anywhere between ~1.5 and 2.4 GHz; two boxes of the pool differ by up to 10 % on the
same binary.  Without the clock next to it a bench line cannot tell box variance from a regression, so the line carries both.

The sampler is a SEPARATE PROCESS (`python -m vattention_amd.telemetry --device N --interval S`): it prints one JSON sample per line
(wall-clock stamp, MHz, W) until its stdin closes; the benchmark keeps the samples whose stamps fall inside its timed region.  Nothing
of it runs in the benchmark's process (no GIL share, no second user of the driver libraries there).  Sources, first that answers:
amdsmi (gpu_metrics: current_gfxclk / per-XCD current_gfxclks, current_socket_power) and the amdgpu sysfs / hwmon files.
"""
from __future__ import annotations

import glob
import json
import os
import subprocess
import sys
import time


def _sysfs_cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.exists(os.path.join(dev, "pp_dpm_sclk")) or glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            out.append(dev)
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class _Sysfs:
    name = "sysfs"

    def __init__(self, index):
        cards = _sysfs_cards()
        if not cards:
            raise RuntimeError("no amdgpu sysfs device")
        self.dev = cards[min(index, len(cards) - 1)]
        hw = glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))
        self.hw = hw[0] if hw else None
        if self.sample()[0] is None and self.sample()[1] is None:
            raise RuntimeError("sysfs has neither clock nor power")

    def sample(self):
        mhz = w = None
        if self.hw:
            v = _read(os.path.join(self.hw, "freq1_input"))
            if v and v.isdigit():
                mhz = int(v) / 1e6
            for f in ("power1_input", "power1_average"):
                v = _read(os.path.join(self.hw, f))
                if v and v.isdigit():
                    w = int(v) / 1e6
                    break
        if mhz is None:
            v = _read(os.path.join(self.dev, "pp_dpm_sclk"))
            if v:
                for line in v.splitlines():
                    if line.rstrip().endswith("*"):
                        try:
                            mhz = float(line.split(":")[1].strip().split("M")[0])
                        except (IndexError, ValueError):
                            pass
        return mhz, w


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        if not hs:
            raise RuntimeError("amdsmi: no processors")
        self.h = hs[min(index, len(hs) - 1)]
        if self.sample() == (None, None):
            raise RuntimeError("amdsmi answers neither clock nor power")

    @staticmethod
    def _num(x):
        try:
            x = float(x)
        except (TypeError, ValueError):
            return None
        return x if 0 < x < 65535 else None        # 0xFFFF = "not supported" in gpu_metrics

    def sample(self):
        mhz = w = None
        try:
            g = self.m.amdsmi_get_gpu_metrics_info(self.h)
            per = [self._num(x) for x in (g.get("current_gfxclks") or [])]
            per = [x for x in per if x]
            mhz = sum(per) / len(per) if per else self._num(g.get("current_gfxclk"))
            w = self._num(g.get("current_socket_power")) or self._num(g.get("average_socket_power"))
        except Exception:      # noqa: BLE001
            pass
        if mhz is None:
            try:
                c = self.m.amdsmi_get_clock_info(self.h, self.m.AmdSmiClkType.GFX)
                mhz = self._num(c.get("clk") or c.get("cur_clk"))
            except Exception:      # noqa: BLE001
                pass
        if w is None:
            try:
                pw = self.m.amdsmi_get_power_info(self.h)
                w = self._num(pw.get("current_socket_power")) or self._num(pw.get("average_socket_power"))
            except Exception:      # noqa: BLE001
                pass
        return mhz, w


def open_source(index=0):
    errs = []
    for cls in (_AmdSmi, _Sysfs):
        try:
            return cls(index)
        except Exception as e:      # noqa: BLE001
            errs.append("%s: %s" % (cls.name, e))
    raise RuntimeError("; ".join(errs))


def _serve(index, interval):
    """child process: one sample per line until stdin closes"""
    try:
        src = open_source(index)
    except Exception as e:      # noqa: BLE001
        print(json.dumps({"error": str(e)}), flush=True)
        return
    print(json.dumps({"source": src.name}), flush=True)
    import select
    while True:
        t = time.time()
        mhz, w = src.sample()
        print(json.dumps({"t": t, "mhz": mhz, "w": w}), flush=True)
        r, _, _ = select.select([sys.stdin], [], [], interval)
        if r and not sys.stdin.readline():
            return


class Sampler:
    """with Sampler(device) as s: ...; s.window(t0, t1) -> {"clock_mhz_mean", "clock_mhz_min", "power_w_mean", "power_w_max", "samples", "source"}
    (t0 / t1 from time.time()).  A box without a readable source yields {"source": None, "error": ...}: the benchmark never fails on it."""

    def __init__(self, device=0, interval=0.05):
        self.device, self.interval = device, interval
        self.proc = None
        self.samples = []
        self.source = None
        self.error = None

    def __enter__(self):
        try:
            self.proc = subprocess.Popen([sys.executable, "-m", "vattention_amd.telemetry", "--device", str(self.device), "--interval", str(self.interval)],
                                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        except OSError as e:
            self.error = str(e)
            return self
        # the child's lines are drained as they come (a reader thread that only sleeps in read()): left in the pipe they fill its 64 KiB
        # after about a thousand samples and the child blocks in print — the means would then cover the head of a long region only
        import threading
        self._lines = []
        self._reader = threading.Thread(target=self._drain, args=(self.proc.stdout, self._lines), daemon=True)
        self._reader.start()
        return self

    @staticmethod
    def _drain(pipe, lines):
        try:
            for line in pipe:
                lines.append(line)
        except Exception:      # noqa: BLE001
            pass

    def stop(self):
        if self.proc is None:
            return
        try:
            self.proc.stdin.close()
            self._reader.join(timeout=10)
            self.proc.wait(timeout=10)
        except Exception:      # noqa: BLE001
            self.proc.kill()
        self.proc = None
        for line in list(self._lines):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if "source" in d:
                self.source = d["source"]
            elif "error" in d:
                self.error = d["error"]
            elif "t" in d:
                self.samples.append(d)

    def __exit__(self, *exc):
        self.stop()

    def window(self, t0, t1):
        self.stop()
        inside = [s for s in self.samples if t0 <= s["t"] <= t1]
        mhz = [s["mhz"] for s in inside if s.get("mhz")]
        w = [s["w"] for s in inside if s.get("w")]
        out = {"source": self.source, "samples": len(inside), "interval_s": self.interval}
        if self.error:
            out["error"] = self.error
        if mhz:
            out.update(clock_mhz_mean=round(sum(mhz) / len(mhz), 1), clock_mhz_min=round(min(mhz), 1), clock_mhz_max=round(max(mhz), 1))
        if w:
            out.update(power_w_mean=round(sum(w) / len(w), 1), power_w_max=round(max(w), 1))
        return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--interval", type=float, default=0.05)
    ap.add_argument("--probe", action="store_true", help="print what every source answers once, then exit")
    a = ap.parse_args()
    if a.probe:
        for cls in (_AmdSmi, _Sysfs):
            try:
                s = cls(a.device)
                print(cls.name, "->", s.sample())
            except Exception as e:      # noqa: BLE001
                print(cls.name, "unavailable:", e)
    else:
        _serve(a.device, a.interval)
