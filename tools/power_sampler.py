#!/usr/bin/env python
"""Socket-power / shader-clock sampler for one MI355X (GPU box only): a stand-alone process that appends

    t_unix, power_W, sclk_MHz

lines to --out at --hz until it receives SIGTERM (tools/power_per_kernel.py starts and stops it by PID).  Sources, first that
answers: librocm_smi64 through ctypes (rsmi_dev_current_socket_power_get / rsmi_dev_power_ave_get / rsmi_dev_power_get;
rsmi_dev_gpu_clk_freq_get), then the amdgpu hwmon files in sysfs.  The first line of the file names the source that is being read.
"""
import argparse
import ctypes as C
import glob
import signal
import sys
import time


class Freqs(C.Structure):       # rsmi_frequencies_t (rocm_smi.h: RSMI_MAX_NUM_FREQUENCIES = 33)
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


def rsmi_sources(dev):
    try:
        lib = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
    except OSError:
        return None, None, "no librocm_smi64"
    if lib.rsmi_init(C.c_uint64(0)) != 0:
        return None, None, "rsmi_init failed"
    u64 = C.c_uint64()
    power = None
    name = ""
    if hasattr(lib, "rsmi_dev_current_socket_power_get") and lib.rsmi_dev_current_socket_power_get(dev, C.byref(u64)) == 0 and u64.value:
        power = lambda: (lib.rsmi_dev_current_socket_power_get(dev, C.byref(u64)), u64.value * 1e-6)[1]
        name = "rsmi_dev_current_socket_power_get"
    elif lib.rsmi_dev_power_ave_get(dev, 0, C.byref(u64)) == 0 and u64.value:
        power = lambda: (lib.rsmi_dev_power_ave_get(dev, 0, C.byref(u64)), u64.value * 1e-6)[1]
        name = "rsmi_dev_power_ave_get"
    elif hasattr(lib, "rsmi_dev_power_get"):
        typ = C.c_int()
        if lib.rsmi_dev_power_get(dev, C.byref(u64), C.byref(typ)) == 0 and u64.value:
            power = lambda: (lib.rsmi_dev_power_get(dev, C.byref(u64), C.byref(typ)), u64.value * 1e-6)[1]
            name = "rsmi_dev_power_get(type %d)" % typ.value
    fr = Freqs()
    clk = None
    if lib.rsmi_dev_gpu_clk_freq_get(dev, 0, C.byref(fr)) == 0 and fr.num_supported:
        def clk():
            lib.rsmi_dev_gpu_clk_freq_get(dev, 0, C.byref(fr))
            return fr.frequency[min(fr.current, 32)] * 1e-6
        name += " + rsmi_dev_gpu_clk_freq_get"
    return power, clk, name


def sysfs_sources():
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"):
        files = sorted(glob.glob(pat))
        if files:
            f = files[0]
            fq = f.rsplit("/", 1)[0] + "/freq1_input"

            def rd(path, scale):
                try:
                    with open(path) as fh:
                        return float(fh.read()) * scale
                except (OSError, ValueError):
                    return float("nan")
            return (lambda: rd(f, 1e-6)), (lambda: rd(fq, 1e-6)), "sysfs " + f
    return None, None, "no hwmon power file"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--hz", type=float, default=50.0)
    ap.add_argument("--dev", type=int, default=0)
    args = ap.parse_args()
    power, clk, name = rsmi_sources(args.dev)
    if power is None:
        p2, c2, n2 = sysfs_sources()
        power, name = p2, name + "; " + n2
        clk = clk or c2
    stop = []
    signal.signal(signal.SIGTERM, lambda *_: stop.append(1))
    with open(args.out, "w") as f:
        f.write("# source: %s\n" % name)
        if power is None:
            f.write("# NO POWER SOURCE\n")
            return 1
        period = 1.0 / args.hz
        nxt = time.time()
        while not stop:
            t = time.time()
            f.write("%.4f,%.1f,%.0f\n" % (t, power(), clk() if clk else float("nan")))
            nxt += period
            d = nxt - time.time()
            if d > 0:
                time.sleep(d)
            else:
                nxt = time.time()
        f.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
