#!/usr/bin/env python3
"""Does a kernel's time move with the GPU's clock / power / temperature?  Scans one workload back to back for <seconds>
(no finalize in between: the GPU never idles), keeps every scan's hipEvent time with its wall-clock time, samples every amdgpu
hwmon of the box every 50 ms in a thread of the same process, and prints one line per second: scans, median / max scan ms,
and -- for the device whose clock the load moved -- shader clock, power, temperature.  The evidence VERDICT r4 (item 3d) asked
for beside config 4's kernel trace.   usage: clock_scan.py <workload substring> <seconds> [idle seconds before]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth


def rd(path, scale):
    try:
        return float(open(path).read().split()[0]) * scale
    except Exception:
        return float("nan")


def main():
    name = [k for k in synth.WORKLOADS if sys.argv[1] in k][0]
    seconds = float(sys.argv[2])
    idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    devs = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(h + "/freq1_input"):
            temp = h + "/temp2_input" if os.path.exists(h + "/temp2_input") else h + "/temp1_input"
            power = h + "/power1_average" if os.path.exists(h + "/power1_average") else h + "/power1_input"
            devs.append((h, h + "/freq1_input", power, temp))
    # which of them is OURS: the PCI address HIP reports for device 0 against the cards' sysfs links (a box may show other
    # tenants' GPUs in sysfs that this process cannot open)
    ours = None
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            bus = buf.value.decode().lower()
            for i, (h, _, _, _) in enumerate(devs):
                if bus in os.path.realpath(os.path.dirname(os.path.dirname(h))).lower():
                    ours = i
    except Exception:
        pass
    samples = []  # (t, [(sclk, power, temp) per device])
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.time(), [(rd(f, 1e-6), rd(p, 1e-6), rd(t, 1e-3)) for _, f, p, t in devs]))
            stop.wait(0.05)

    wl = synth.WORKLOADS[name]
    ctx = sybil_amd.Context(0)
    t = ctx.synth_table("t", synth.SEED, wl["rows"], 0, wl["rows"], synth.synth_cols(wl["columns"]))
    t.compact()
    q = t.query(**wl["query"])
    q.scan(); ctx.sync()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(idle)
    t0 = time.time()
    scans = []
    while time.time() - t0 < seconds:
        q.scan(); ctx.sync()
        scans.append((time.time(), q.stats()["scan_ms"]))
    stop.set()
    th.join(timeout=1)
    # the device the load moved: the largest spread of its shader clock
    if devs:
        spread = [max(s[1][i][0] for s in samples) - min(s[1][i][0] for s in samples) for i in range(len(devs))]
        d = ours if ours is not None else max(range(len(devs)), key=lambda i: spread[i])
    print("# %s: %d scans in %.1f s, back to back; hwmon devices: %d, shown: %s (%s)" % (
        name, len(scans), seconds, len(devs), devs[d][0] if devs else None, "matched by PCI address" if ours is not None else "largest clock spread: may be another tenant's"))
    print("# second  scans  scan_ms median / min / max   sclk_MHz median (min-max)   power_W median (max)   temp_C median (max)")
    start = samples[0][0] if samples else t0
    for sec in range(int(time.time() - start) + 1):
        ms = sorted(m for (tt, m) in scans if sec <= tt - start < sec + 1)
        ss = [s[1][d] for s in samples if sec <= s[0] - start < sec + 1] if devs else []
        if not ss:
            continue
        ck, pw, tp = sorted(x[0] for x in ss), sorted(x[1] for x in ss), sorted(x[2] for x in ss)
        print("%3d  %5d  %s   %5.0f (%4.0f-%4.0f)   %5.0f (%5.0f)   %4.1f (%4.1f)" % (
            sec, len(ms), ("%.3f / %.3f / %.3f" % (ms[len(ms) // 2], ms[0], ms[-1])) if ms else "   (idle)          ",
            ck[len(ck) // 2], ck[0], ck[-1], pw[len(pw) // 2], pw[-1], tp[len(tp) // 2], tp[-1]))


if __name__ == "__main__":
    main()
