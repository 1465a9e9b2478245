"""Run a command while sampling rocm-smi (sclk, socket power) every 0.25 s; print min/median/max.
usage: power_during.py <cmd...>"""
import re, subprocess, sys, threading, time
samples, stop = [], False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            p = re.search(r"Power \(W\): ([\d.]+)", out)
            if s and p:
                samples.append((time.time(), int(s.group(1)), float(p.group(1))))
        except Exception:
            pass
        time.sleep(0.25)
t = threading.Thread(target=sampler); t.start()
rc = subprocess.call(sys.argv[1:])
stop = True; t.join()
busy = [x for x in samples if x[2] > 600]
for name, xs in (("all", samples), ("busy(>600W)", busy)):
    if xs:
        c = sorted(x[1] for x in xs); p = sorted(x[2] for x in xs)
        print(f"[power] {name}: n={len(xs)} sclk MHz min/med/max {c[0]}/{c[len(c)//2]}/{c[-1]}  power W min/med/max {p[0]:.0f}/{p[len(p)//2]:.0f}/{p[-1]:.0f}", file=sys.stderr)
sys.exit(rc)
