"""Files this project did not write are loaded without executing anything from them (nerf_downstream_amd/safe_load.py):
the Plenoxel `last.ckpt` of a CO3D scene, reference / own training checkpoints, ScanNet's `scene_scales.data`."""
import os
import pickle

import numpy as np
import pytest
import torch

from nerf_downstream_amd.safe_load import load_checkpoint_file, load_plain_pickle


class _Boom:
    """What a hostile pickle looks like: unpickling it calls an arbitrary function."""

    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_plenoxel_checkpoint_layout_loads(tmp_path):
    n = 50
    ck = {"state_dict": {"model.links_idx": torch.arange(n, dtype=torch.int32), "model.density_data": torch.rand(n, 1),
                         "model.sh_data": torch.randint(0, 255, (n, 27), dtype=torch.uint8)},
          "model.sh_data_scale": np.float32(0.031), "model.sh_data_min": np.full(27, -3.5, np.float32),
          "epoch": 9, "hyper_parameters": {"reso": [256, 256, 256], "name": "scene"}}
    p = tmp_path / "last.ckpt"
    torch.save(ck, p)
    got = load_checkpoint_file(p)
    assert torch.equal(got["state_dict"]["model.sh_data"], ck["state_dict"]["model.sh_data"])
    assert float(got["model.sh_data_scale"]) == pytest.approx(0.031) and np.array_equal(got["model.sh_data_min"], ck["model.sh_data_min"])
    assert got["hyper_parameters"]["reso"] == [256, 256, 256]


def test_hostile_checkpoint_is_refused(tmp_path):
    p = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {"w": torch.zeros(2)}, "callback": _Boom()}, p)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_file(p)


def test_trainer_checkpoint_round_trip(tmp_path):
    """What train.py saves (Lightning key layout, optimizer state, a PolyLR scheduler whose lambda is an object) loads
    through the weights-only path."""
    from nerf_downstream_amd import gin_lite as gin
    from nerf_downstream_amd.co3d_3d.src.modules.optim import PolyLR
    from nerf_downstream_amd.co3d_3d.train import load_checkpoint, save_checkpoint

    gin.clear_config()
    gin.parse_config_files_and_bindings([], ["train.max_steps=100"])
    try:
        model = torch.nn.Linear(4, 3)
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
        model(torch.randn(2, 4)).sum().backward()
        opt.step()
        sched = PolyLR(opt)
        sched.step()
        p = tmp_path / "last.ckpt"
        save_checkpoint(p, model, opt, sched, step=7, epoch=1, best=3.5, batch_in_epoch=2)
        model2 = torch.nn.Linear(4, 3)
        opt2 = torch.optim.SGD(model2.parameters(), lr=0.1, momentum=0.9)
        sched2 = PolyLR(opt2)
        ck = load_checkpoint(p, model2, opt2, sched2)
    finally:
        gin.clear_config()
    assert ck["global_step"] == 7 and ck["batch_in_epoch"] == 2
    assert torch.equal(model2.weight, model.weight)
    assert torch.equal(opt2.state_dict()["state"][0]["momentum_buffer"], opt.state_dict()["state"][0]["momentum_buffer"])
    assert sched2.last_epoch == sched.last_epoch


def test_plain_pickle_loader():
    scales = {"scene0000_00": 1.25, "scene0001_00": np.float64(0.5), "n": [1, 2, (3, "x")], "a": np.arange(3)}
    got = load_plain_pickle(pickle.dumps(scales))
    assert got["scene0000_00"] == 1.25 and float(got["scene0001_00"]) == 0.5 and got["n"] == [1, 2, (3, "x")]
    assert np.array_equal(got["a"], np.arange(3))
    with pytest.raises(pickle.UnpicklingError):
        load_plain_pickle(pickle.dumps({"x": _Boom()}))
    with pytest.raises(pickle.UnpicklingError):
        load_plain_pickle(pickle.dumps(torch.nn.Linear))  # any other global, callable or not
