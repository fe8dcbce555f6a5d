"""A/B of the batch-norm launch fusions on whole steps: python scripts/ab_bn.py [bench.py args...]
Runs bench.py's step loop in-process for each setting (fold limit, one-launch few-row norm) and prints ms/step."""
import json, os, subprocess, sys
args = sys.argv[1:]
for name, env in [("fold 0, small off", {"MINK_BN_FOLD": "0", "MINK_BN_SMALL": "0"}), ("fold 128 (bytes rule), small off", {"MINK_BN_FOLD": "128", "MINK_BN_SMALL": "0"}),
                  ("fold 0, small on", {"MINK_BN_FOLD": "0", "MINK_BN_SMALL": "1"}),
                  ("fold 128 (bytes rule), small on", {"MINK_BN_FOLD": "128", "MINK_BN_SMALL": "1"})]:
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-kernel-timing"] + args, env=e, capture_output=True, text=True)
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print(f"[{' '.join(args)}] {name}: {d['ms_per_step']:.3f} ms/step, forward only {d['roofline']['forward_only']['ms']:.3f}", flush=True)
    except Exception:
        print(name, "failed", r.stderr[-500:])
