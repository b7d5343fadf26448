#!/usr/bin/env python3
"""Samples the GPU's clock / power / temperature while a command runs: clock_watch.py <out file> -- cmd ...
Reads the amdgpu hwmon / sysfs files directly (a `rocm-smi` process per sample stalls for seconds once the GPU is busy:
round 5's first log had 13 lines for a 100 s run) -- one line per 100 ms:  t_seconds sclk_MHz power_W temp_C  [mclk_MHz]"""
import glob
import os
import subprocess
import sys
import threading
import time


def first(paths):
    for p in paths:
        g = sorted(glob.glob(p))
        if g:
            return g[0]
    return None


def read_num(path, scale=1.0):
    try:
        return float(open(path).read().split()[0]) * scale
    except Exception:
        return float("nan")


def active_mhz(path):
    """pp_dpm_sclk: lines like '1: 2100Mhz *'"""
    try:
        for line in open(path):
            if "*" in line:
                return float(line.split(":")[1].strip().split("M")[0])
    except Exception:
        pass
    return float("nan")


def main():
    out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    dev = first(["/sys/class/drm/card*/device/hwmon/hwmon*"])
    card = os.path.dirname(os.path.dirname(dev)) if dev else first(["/sys/class/drm/card*/device"])
    f_sclk = first([dev + "/freq1_input"]) if dev else None
    f_mclk = first([dev + "/freq2_input"]) if dev else None
    f_pow = first([dev + "/power1_average", dev + "/power1_input"]) if dev else None
    f_temp = first([dev + "/temp2_input", dev + "/temp1_input"]) if dev else None  # (temp2 = junction where present)
    f_dpm = card + "/pp_dpm_sclk" if card else None
    stop = threading.Event()
    t0 = time.time()

    def sample():
        with open(out, "w") as f:
            f.write("# t_s sclk_MHz power_W temp_C mclk_MHz   (hwmon: %s)\n" % dev)
            while not stop.is_set():
                sclk = read_num(f_sclk, 1e-6) if f_sclk else active_mhz(f_dpm) if f_dpm else float("nan")
                f.write("%.2f %.0f %.1f %.1f %.0f\n" % (time.time() - t0, sclk, read_num(f_pow, 1e-6) if f_pow else float("nan"),
                                                      read_num(f_temp, 1e-3) if f_temp else float("nan"), read_num(f_mclk, 1e-6) if f_mclk else float("nan")))
                f.flush()
                stop.wait(0.1)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=2)
    sys.exit(rc)


if __name__ == "__main__":
    main()
