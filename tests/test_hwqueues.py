"""nerf_downstream_amd/hwqueues.py: how many hardware queues a rank asks the HIP runtime for (no GPU, no torch needed)."""
import importlib
import os
import sys

import pytest


def _configure(env, torch_loaded=False, **kw):
    old = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "MINK_DP_LAUNCH", "MINK_HWQUEUES_KEEP")}
    hidden = None if torch_loaded else sys.modules.pop("torch", None)  # (the test session has imported torch: a launcher has not)
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        from nerf_downstream_amd import hwqueues

        importlib.reload(hwqueues)
        return hwqueues.configure(**kw), hwqueues.busy_streams(kw.get("data_parallel", True))
    finally:
        if hidden is not None:
            sys.modules["torch"] = hidden
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_default_is_seven_and_four_streams_are_busy():
    q, busy = _configure({})
    assert q == 7 and busy == 4  # compute, map preparation, weight gradients (+ branch), the process group's own


def test_an_inherited_value_is_kept_when_only_four_queues_carry_work():
    assert _configure({"GPU_MAX_HW_QUEUES": "16"})[0] == 16  # (measured: 3.72 / 3.74 / 3.74 ms at 7 / 8 / 16)


def test_the_round4_launch_stream_is_refused_more_than_seven_queues():
    q, busy = _configure({"GPU_MAX_HW_QUEUES": "8", "MINK_DP_LAUNCH": "stream"})
    assert busy == 5 and q == 7  # a fifth busy queue with room for a queue of its own: 1.5-3x per step
    assert _configure({"GPU_MAX_HW_QUEUES": "8", "MINK_DP_LAUNCH": "stream", "MINK_HWQUEUES_KEEP": "1"})[0] == 8  # measurement runs
    assert _configure({"GPU_MAX_HW_QUEUES": "6", "MINK_DP_LAUNCH": "stream"})[0] == 6


def test_after_the_runtime_is_loaded_nothing_is_pretended():
    """torch imported first and nothing inherited: the runtime never read a value -- None comes back and the environment stays as it was."""
    old = os.environ.pop("GPU_MAX_HW_QUEUES", None)
    try:
        sys.modules.setdefault("torch", sys)  # (any module object: only the name is looked at)
        from nerf_downstream_amd import hwqueues

        assert hwqueues.configure() is None and "GPU_MAX_HW_QUEUES" not in os.environ
    finally:
        if sys.modules.get("torch") is sys:
            del sys.modules["torch"]
        if old is not None:
            os.environ["GPU_MAX_HW_QUEUES"] = old


def test_an_inherited_value_that_is_not_a_number_is_named():
    with pytest.raises(ValueError, match="GPU_MAX_HW_QUEUES"):
        _configure({"GPU_MAX_HW_QUEUES": "many"}, data_parallel=False)
