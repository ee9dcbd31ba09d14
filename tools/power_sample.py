#!/usr/bin/env python3
"""Run a command as a child process and sample the GPU's power / shader clock from sysfs (hwmon) meanwhile - no GPU call in
this process.  Answers "is this workload power-limited": a package power that sits at the cap with a shader clock below the
maximum while the kernels run.
Usage: python3 tools/power_sample.py [--dt 0.02] -- <command ...>      (prints one JSON line after the child's output)"""
import glob, json, os, subprocess, sys, threading, time


def rd(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sensors():
    out = {"power": [], "freq": [], "cap": [], "dpm": [], "temp": []}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for n in ("power1_average", "power1_input"):
                if rd(os.path.join(hw, n)) is not None:
                    out["power"].append(os.path.join(hw, n))
                    break
            for n in ("freq1_input",):
                if rd(os.path.join(hw, n)) is not None:
                    out["freq"].append(os.path.join(hw, n))
            if rd(os.path.join(hw, "power1_cap")) is not None:
                out["cap"].append(os.path.join(hw, "power1_cap"))
            for n in ("temp2_input", "temp1_input"):
                if rd(os.path.join(hw, n)) is not None:
                    out["temp"].append(os.path.join(hw, n))
                    break
        if rd(os.path.join(card, "pp_dpm_sclk")) is not None:
            out["dpm"].append(os.path.join(card, "pp_dpm_sclk"))
    return out


def main():
    args = sys.argv[1:]
    dt = 0.02
    if args and args[0] == "--dt":
        dt = float(args[1])
        args = args[2:]
    if args and args[0] == "--":
        args = args[1:]
    s = sensors()
    samples = []
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            t = time.time()
            row = {"t": t}
            if s["power"]:
                v = rd(s["power"][0])
                row["w"] = float(v) / 1e6 if v else None
            if s["freq"]:
                v = rd(s["freq"][0])
                row["mhz"] = float(v) / 1e6 if v else None
            if s["temp"]:
                v = rd(s["temp"][0])
                row["c"] = float(v) / 1e3 if v else None
            samples.append(row)
            time.sleep(dt)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(args)
    stop.set()
    th.join()
    ws = sorted(r["w"] for r in samples if r.get("w") is not None)
    fs = sorted(r["mhz"] for r in samples if r.get("mhz") is not None)
    q = lambda a, p: a[min(len(a) - 1, int(p * len(a)))] if a else None
    busy = [r for r in samples if r.get("w") is not None and ws and r["w"] > 0.6 * ws[-1]]   # samples taken while the kernels run
    bf = sorted(r["mhz"] for r in busy if r.get("mhz") is not None)
    bw = sorted(r["w"] for r in busy)
    rep = {"sensors": {k: v[:1] for k, v in s.items()}, "n_samples": len(samples), "rc": rc,
           "power_cap_w": (float(rd(s["cap"][0])) / 1e6 if s["cap"] else None),
           "power_w": {"min": q(ws, 0), "p50": q(ws, 0.5), "p90": q(ws, 0.9), "max": q(ws, 1.0)},
           "sclk_mhz": {"min": q(fs, 0), "p10": q(fs, 0.1), "p50": q(fs, 0.5), "max": q(fs, 1.0)},
           "busy": {"n": len(busy), "power_w_p50": q(bw, 0.5), "sclk_mhz_p10": q(bf, 0.1), "sclk_mhz_p50": q(bf, 0.5), "sclk_mhz_p90": q(bf, 0.9)},
           "dpm_sclk": (rd(s["dpm"][0]) if s["dpm"] else None)}
    print(json.dumps(rep))
    return rc


if __name__ == "__main__":
    sys.exit(main())
