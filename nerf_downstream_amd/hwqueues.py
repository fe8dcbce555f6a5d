"""How many hardware queues the HIP runtime may hand to this process (GPU_MAX_HW_QUEUES), decided from the streams a rank will
keep busy -- call before anything loads the HIP runtime (`import torch`).  No torch import here.

What was measured (MI355X, ROCm 7.2; scripts/cliff_profile*.sh, profiles/r05_queue_cliff.txt): the device runs FOUR hardware
queues side by side.  A rank's step keeps four streams busy -- compute, map preparation, weight gradients (+ the shortcut branch),
and the process group's own -- and as long as each of them has a queue of its own and no FIFTH queue carries work, a step costs
what it costs without the data-parallel machinery (+2 %).  Round 4 issued the collectives from a fifth stream (event waits and
records only): with GPU_MAX_HW_QUEUES <= 7 that stream happened to share a queue with an idle one; from 8 on it got a queue of its
own, and with five active queues a step took 1.5x (Mink-ResNet14, 16 scenes: 3.73 -> 5.8 ms) to 3x (Mink-ResNet34, 4 scenes:
4.1 -> 12.9 ms) as long -- at 8, 9, ... 16 alike, whichever queue ids the streams got, and back to 3.79 / 4.49 ms at 8 and at
16 once the fifth stream's work rode on the weight-gradient stream.  Since round 5 the collectives are issued from inside the
backward call on the weight-gradient stream (minkowski/trunk.py, mink_set_block_done_hook): four busy queues by construction.

Rule: busy streams <= 4 and GPU_MAX_HW_QUEUES >= busy streams + the idle ones created before them (7 covers torch's own);
with the round-4 launch stream (MINK_DP_LAUNCH=stream) a value of 8 or more is on the wrong side and is refused."""
import os
import sys

SAFE = 7


def busy_streams(data_parallel):
    n = 3  # compute, map preparation, weight gradients (+ shortcut branch)
    if data_parallel:
        n += 1  # the process group's stream
        if os.environ.get("MINK_DP_LAUNCH", "call") == "stream":
            n += 1  # the bucket-launch stream of round 4
    return n


def configure(data_parallel=True):
    """Set GPU_MAX_HW_QUEUES for this process; returns the value in force, or None when the runtime was loaded before this call
    with nothing inherited (its default then applies and the environment is left alone).  An inherited value is kept unless it
    is known to be on the wrong side (five busy streams and room for each to get a queue of its own)."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        try:
            cur_n = int(cur)
        except ValueError:
            raise ValueError(f"GPU_MAX_HW_QUEUES={cur!r} is not an integer") from None
    if "torch" in sys.modules and cur is None:
        # (the runtime has read its environment already: say so rather than pretend)
        print("[hwqueues] torch was imported before hwqueues.configure(): GPU_MAX_HW_QUEUES is not applied", file=sys.stderr)
        return None
    n_busy = busy_streams(data_parallel)
    if cur is None:
        os.environ["GPU_MAX_HW_QUEUES"] = str(SAFE)
    elif n_busy > 4 and cur_n > SAFE and os.environ.get("MINK_HWQUEUES_KEEP") != "1":  # (KEEP: measurement runs of the cliff itself)
        print(f"[hwqueues] GPU_MAX_HW_QUEUES={cur} with {n_busy} busy streams puts a fifth queue to work (1.5-3x per step): using {SAFE}",
              file=sys.stderr)
        os.environ["GPU_MAX_HW_QUEUES"] = str(SAFE)
    return int(os.environ["GPU_MAX_HW_QUEUES"])
