#!/bin/bash
# bench timeline under a list of env settings: one line per setting with the phase durations (ms)
for cfg in "$@"; do
  env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --timeline 2>&1 | python -c "
import sys, json, re
t = {}
ms = None
for l in sys.stdin:
    m = re.match(r'\[bench\]\s+(\w+)\s+gpu\s+(-?[\d.]+)', l)
    if m: t[m.group(1)] = float(m.group(2))
    if l.startswith('{'): ms = json.loads(l)['ms_per_step']
print('[%s] step %.3f | stem_fwd %.3f  mid(4..2) %.3f  l1_bwd %.3f  stem_bwd %.3f  boundary %.3f' % ('$cfg', ms, t['stem_forward'] - t['stem_forward_begin'], t.get('early_grads', 0) - t['stem_forward'], t['stem_backward_begin'] - t.get('early_grads', 0), t['stem_backward_end'] - t['stem_backward_begin'], t['next_step_begin'] - t['stem_backward_end'] + t['stem_forward_begin']))
"
done
