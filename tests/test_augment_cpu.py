"""Scene augmentation, host side: the Philox generator of the oracle against the
Random123 known-answer vectors, the rotation matrix against the reference's expm form, the folded
per-scene program against a stage-by-stage evaluation of the reference's formulas, and the loader
plumbing (dataset -> collate) that carries the drawn programs to the model."""
import random

import numpy as np
import pytest
import torch

from nerf_downstream_amd.co3d_3d.src.data import transforms as T
from oracle import augment as OA


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = OA.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_rotation_matrix_is_the_reference_expm():
    from scipy.linalg import expm, norm

    rng = np.random.default_rng(0)
    for _ in range(10):
        axis, theta = rng.normal(size=3), rng.uniform(-7, 7)
        want = expm(np.cross(np.eye(3), axis / norm(axis) * theta))  # reference transforms.py:334-336
        np.testing.assert_allclose(T.rotation_matrix(axis, theta), want, atol=1e-12)


def _aug3():
    return T.Compose([T.RandomRotation(upright_axis="y"), T.RandomAffine(upright_axis="y"),
                      T.CoordinateDropout(application_ratio=0.9), T.RandomHorizontalFlip(upright_axis="y"),
                      T.CoordinateUniformTranslation(max_translation=0.2), T.CoordinateJitter(), T.RandomScale(scale_ratio=0.4),
                      T.RandomFeatureJitter(start_ind=4, feature_dim=27)])


def test_folded_program_equals_stagewise_evaluation():
    random.seed(3), np.random.seed(3)
    rng = np.random.default_rng(1)
    seen = set()
    for trial in range(40):
        stages = _aug3().draw()
        seen.update(s[0] for s in stages)
        n = 500
        xyz = rng.integers(0, 128, (n, 3)).astype(np.float32)
        feats = rng.normal(size=(n, 28)).astype(np.float32)
        stream, seed = int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 63))
        P = T.compile_program(stages)
        coords = np.concatenate([np.zeros((n, 1), np.float32), xyz], 1)
        oc, of = OA.augment_batch(coords, feats, [0, n], P[None], [stream], seed, T.raw_columns(["density", "sh"]))
        r = OA.philox4x32_10(np.arange(n, dtype=np.uint32), 0, stream, 0, seed & 0xFFFFFFFF, seed >> 32)
        keep = OA.u01(r[0]) >= np.float32(P[T.AUG["DROPOUT"]])
        ju = np.stack([OA.u01(r[1]), OA.u01(r[2]), OA.u01(r[3])], 1)
        geo = [s for s in stages if s[0] != "feature_jitter"]
        want = OA.stagewise(xyz, geo, keep=keep, jitter_u=ju)
        assert oc.shape == (int(keep.sum()), 4) and of.shape == (int(keep.sum()), 28)
        np.testing.assert_allclose(oc[:, 1:], want, atol=2e-3)  # fp32 folded form vs float64 stage by stage
        fj = [s for s in stages if s[0] == "feature_jitter"]
        d = of - feats[keep]
        assert np.all(d[:, 0] == 0)  # density (raw column 3) is outside [4, 31)
        if fj:  # (normal - 0.5) * std on the 27 SH columns (reference transforms.py:36-39)
            assert abs(d[:, 1:].mean() / fj[0][1] + 0.5) < 0.05 and abs(d[:, 1:].std() / fj[0][1] - 1) < 0.05
        else:
            assert np.all(d == 0)
    assert seen == {"linear", "translate", "dropout", "flip", "jitter", "feature_jitter"}


def test_program_rejects_what_the_device_form_cannot_express():
    with pytest.raises(NotImplementedError):
        T.compile_program([("jitter", 1.0), ("flip", (0, 2))])
    with pytest.raises(NotImplementedError):
        T.compile_program([("flip", (0,)), ("flip", (2,))])
    with pytest.raises(ValueError):
        T.compile_program([("dropout", 1.0)])
    P = T.compile_program([("flip", (0, 2)), ("dropout", 0.2)])  # dropout after the flip: max over all voxels
    assert P[T.AUG["FLIP_ALL"]] == 1 and T.compile_program([("dropout", 0.2), ("flip", (0,))])[T.AUG["FLIP_ALL"]] == 0
    with pytest.raises(RuntimeError):
        T.RandomScale()(None, None, None)  # no CPU execution path


def test_loader_carries_programs(tmp_path, monkeypatch):
    from nerf_downstream_amd import gin_lite as gin
    from nerf_downstream_amd.co3d_3d.src.data.co3d import Co3DDatasetBase
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink

    rng = np.random.default_rng(0)
    root = tmp_path / "co3d"
    (tmp_path / "filelist").mkdir()
    names = []
    for i, n in enumerate((40, 25)):
        d = root / f"plenoxel_co3d_s{i}"
        d.mkdir(parents=True)
        np.savez(d / "data.npz", links=np.sort(rng.choice(128 ** 3, n, replace=False)).astype(np.int32),
                 density=rng.normal(size=(n, 1)).astype(np.float32), sh=rng.integers(0, 256, (n, 27), dtype=np.uint8),
                 sh_scale=np.float32(0.01), sh_min=np.float32(-1.0), reso=np.array([128, 128, 128]))
        names.append(f"apple s{i}")
    for phase in ("train", "test"):
        (tmp_path / "filelist" / f"{phase}.txt").write_text("\n".join(names) + "\n")
    monkeypatch.chdir(tmp_path)
    gin.clear_config()
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    gin.parse_config_file(os.path.join(here, "..", "nerf_downstream_amd", "co3d_3d", "configs", "co3d_aug3.gin"))
    try:
        for compact in (False, True):
            ds = Co3DDatasetBase("train", data_root=str(root), features=["density", "sh"], compact=compact)
            assert [type(t).__name__ for t in ds.transformations.transforms][:2] == ["RandomRotation", "RandomAffine"]
            assert ds.transformations.transforms[0].upright_axis == 1 and ds.transformations.transforms[6].scale_ratio == 0.40
            batch = collate_mink([ds[0], ds[1]])
            assert batch["aug_params"].shape == (2, T.AUG["PARAMS"]) and batch["aug_params"].dtype == torch.float32
            assert batch["aug_streams"].dtype == torch.int32 and batch["aug_streams"].shape == (2,)
            assert batch["scene_offsets"].tolist() == [0, 40, 65] and isinstance(batch["aug_seed"], int)
            assert list(batch["feature_names"]) == ["density", "sh"]
        assert Co3DDatasetBase("test", data_root=str(root)).transformations is None  # eval list is empty
        with pytest.raises(NotImplementedError):
            Co3DDatasetBase("train", data_root=str(root), train_transformations=["ElasticDistortion"])
    finally:
        gin.clear_config()
