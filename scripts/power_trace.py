"""Power / shader-clock trace of the GPU while one workload loops ("power-bound" needs evidence on file).

  python scripts/power_trace.py --out gpurun_out/power_<name>.json --seconds 8 -- <command ...>

Starts <command> (which should loop for longer than the trace), waits --settle seconds, then samples at ~10 Hz from the
amdgpu hwmon / sysfs files (package power in W, current shader clock in MHz, junction temperature), falling back to
`rocm-smi --showpower --showclocks --json`.  The trace and its summary (median / min / max) go to --out; the workload's
last lines are kept with it.
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import time


def sysfs_sources():
    """every amdgpu card with a power and a shader-clock file (the box may expose idle GPUs next to the one in use: all are
    sampled and the card drawing the most power during the trace is reported)"""
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if not hw:
            continue
        hw = hw[0]
        p = [f for f in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(f)]
        f = [x for x in (os.path.join(hw, "freq1_input"),) if os.path.exists(x)]
        t = [x for x in (os.path.join(hw, "temp2_input"), os.path.join(hw, "temp1_input")) if os.path.exists(x)]
        if p and f:
            out.append(dict(power=p[0], sclk=f[0], temp=t[0] if t else None, card=card))
    return out


def read_num(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return None


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        power = next((float(v) for k, v in card.items() if "Power" in k and re.match(r"^[\d.]+$", str(v))), None)
        sclk = next((float(re.sub(r"[^\d.]", "", str(v))) for k, v in card.items() if k.startswith("sclk clock speed")), None)
        return power, sclk, None
    except Exception:
        return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--settle", type=float, default=6.0)
    ap.add_argument("--hz", type=float, default=10.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    log = open(a.out + ".log", "w")
    proc = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT)
    src = sysfs_sources()
    time.sleep(a.settle)
    per_card = [[] for _ in src]
    smi = []
    t_end = time.time() + a.seconds
    while time.time() < t_end and proc.poll() is None:
        t0 = time.time()
        for i, c in enumerate(src):
            p, f, tj = read_num(c["power"]), read_num(c["sclk"]), read_num(c["temp"]) if c["temp"] else None
            per_card[i].append(dict(t=round(t0, 3), power_w=None if p is None else round(p / 1e6, 1), sclk_mhz=None if f is None else round(f / 1e6, 0),
                                    temp_c=None if tj is None else round(tj / 1e3, 1)))
        if not src:
            p, f, tj = smi_sample()
            smi.append(dict(t=round(t0, 3), power_w=p, sclk_mhz=f, temp_c=tj))
        time.sleep(max(0.0, 1.0 / a.hz - (time.time() - t0)))
    if src:
        med = [sorted(s_["power_w"] or 0 for s_ in c)[len(c) // 2] if c else 0 for c in per_card]
        best = max(range(len(src)), key=lambda i: med[i])
        samples, src = per_card[best], dict(src[best], cards_sampled=len(per_card), median_power_per_card=med)
    else:
        samples = smi
    alive = proc.poll() is None
    if alive:
        proc.terminate()
        try:
            proc.wait(timeout=10)
        except Exception:
            proc.kill()
    log.close()

    def stats(key):
        v = sorted(s[key] for s in samples if s.get(key) is not None)
        return None if not v else dict(median=v[len(v) // 2], min=v[0], max=v[-1], n=len(v))
    res = dict(command=" ".join(cmd), source=("sysfs %s (the busiest of %d cards; median W per card %s)" % (src["card"], src["cards_sampled"], src["median_power_per_card"])) if src else "rocm-smi --json", workload_still_running_at_end=alive,
               power_w=stats("power_w"), sclk_mhz=stats("sclk_mhz"), temp_c=stats("temp_c"), samples=samples,
               workload_tail=open(a.out + ".log").read()[-1500:])
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("command", "source", "power_w", "sclk_mhz", "temp_c", "workload_still_running_at_end")}))


if __name__ == "__main__":
    main()
