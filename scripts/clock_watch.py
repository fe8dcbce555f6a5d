#!/usr/bin/env python3
"""Sample the card's shader and memory clock levels (sysfs) every few ms while a command runs; print the distinct readings with
the time they were first seen.  usage: clock_watch.py <command ...>"""
import glob
import subprocess
import sys
import time

files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")) + sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk"))
print("clock files:", files, flush=True)
p = subprocess.Popen(sys.argv[1:])
t0 = time.perf_counter()
last = None
n = 0
while p.poll() is None:
    cur = []
    for f in files:
        try:
            cur.append(",".join(l.strip() for l in open(f) if "*" in l))
        except OSError as e:
            cur.append(repr(e))
    cur = " | ".join(cur)
    if cur != last and n < 400:
        print(f"[clock] {1e3 * (time.perf_counter() - t0):9.1f} ms  {cur}", flush=True)
        last = cur
        n += 1
    time.sleep(0.002)
sys.exit(p.returncode)
