"""-m gpu: batch augmentation on the device (`mink_augment_scenes`) vs the numpy
restatement: surviving voxels and their order identical, coordinates BIT-EXACT (so they floor into
the same cells), features within 2e-6 (Box-Muller through libm vs the device's logf/sinf/cosf)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _programs(n_scenes, seed, force=None):
    from nerf_downstream_amd.co3d_3d.src.data import transforms as T

    random.seed(seed), np.random.seed(seed)
    comp = T.Compose([T.RandomRotation(upright_axis="y"), T.RandomAffine(upright_axis="y"),
                      T.CoordinateDropout(application_ratio=0.9), T.RandomHorizontalFlip(upright_axis="y"),
                      T.CoordinateUniformTranslation(max_translation=0.2), T.CoordinateJitter(), T.RandomScale(scale_ratio=0.4),
                      T.RandomFeatureJitter(start_ind=4, feature_dim=27)])
    rows, streams = [], []
    for _ in range(n_scenes):
        stages = comp.draw()
        if force == "no_dropout":
            stages = [s for s in stages if s[0] != "dropout"]
        rows.append(T.compile_program(stages))
        streams.append(int(np.random.randint(0, 2 ** 32, dtype=np.uint64)))
    return np.stack(rows), np.array(streams, np.uint32)


def _batch(ns, seed, C):
    rng = np.random.default_rng(seed)
    coords, feats = [], []
    for b, n in enumerate(ns):
        xyz = np.stack(np.unravel_index(np.sort(rng.choice(128 ** 3, n, replace=False)), (128,) * 3), 1)
        coords.append(np.concatenate([np.full((n, 1), b), xyz], 1))
        feats.append(rng.normal(size=(n, C)))
    return np.concatenate(coords).astype(np.int32), np.concatenate(feats).astype(np.float32)


@pytest.mark.parametrize("ns,features,as_int,force", [
    ((5000, 4321, 7000), ("density", "sh"), True, None),
    ((3000, 0, 257, 1), ("sh",), False, None),          # an empty scene, a one-voxel scene, block-straddling sizes
    ((2500, 2500), ("sh", "density", "ones"), True, "no_dropout"),
])
def test_augment_batch_matches_oracle(ns, features, as_int, force):
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.co3d_3d.src.data import transforms as T
    from oracle.augment import augment_batch as oracle_augment

    width = {"density": 1, "sh": 27, "ones": 1}
    C = sum(width[f] for f in features)
    raw = T.raw_columns(features)
    for seed in range(4):
        coords, feats = _batch(ns, seed, C)
        params, streams = _programs(len(ns), 10 + seed, force)
        offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int32)
        key = 0x9E3779B97F4A7C15 ^ (seed * 0x1000193)
        want_c, want_f = oracle_augment(coords.astype(np.float32), feats, offs, params, streams, key, raw)
        dc = torch.from_numpy(coords if as_int else coords.astype(np.float32)).cuda()
        got_c, got_f = ME.utils.augment_batch(dc, torch.from_numpy(feats).cuda(), torch.from_numpy(offs),
                                              torch.from_numpy(params), torch.from_numpy(streams.view(np.int32).copy()), key, raw)
        assert got_c.shape == want_c.shape and got_f.shape == want_f.shape
        if force == "no_dropout":
            assert got_c.shape[0] == sum(ns)
        assert np.array_equal(got_c.cpu().numpy(), want_c), "coordinates must agree bit for bit"
        np.testing.assert_allclose(got_f.cpu().numpy(), want_f, atol=2e-6, rtol=0)


def test_augmented_batch_trains():
    """Loader -> collate -> process_input (augment on the prepare stream) -> TensorField.sparse(): voxels that
    fall into one cell after the transform are averaged (reference a3), and a training step runs on it."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.co3d_3d.src.data import transforms as T
    from nerf_downstream_amd.co3d_3d.src.data.synthetic import SparseVoxelDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle.augment import augment_batch as oracle_augment

    random.seed(0), np.random.seed(0)
    ds = SparseVoxelDataset("train", num_samples=8, num_classes=4, grid=48,
                            train_transformations=["RandomRotation", "RandomAffine", "CoordinateDropout", "RandomHorizontalFlip",
                                                   "CoordinateUniformTranslation", "CoordinateJitter", "RandomScale",
                                                   "RandomFeatureJitter"])
    torch.manual_seed(0)
    net = get_model("ResNet14", 28, 4).cuda().train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01)
    for step in range(3):
        host = collate_mink([ds[4 * (step % 2) + j] for j in range(4)])
        batch = {k: (v.cuda() if torch.is_tensor(v) and k != "aug_params" else v) for k, v in host.items()}
        field = net.process_input(batch)
        x = field.sparse()
        want_c, want_f = oracle_augment(host["coordinates"].numpy(), host["features"].numpy(), host["scene_offsets"].numpy(),
                                        host["aug_params"].numpy(), host["aug_streams"].numpy().view(np.uint32),
                                        host["aug_seed"], T.raw_columns(["density", "sh"]))
        cells = np.floor(want_c).astype(np.int64)
        uniq, inv = np.unique(cells, axis=0, return_inverse=True)
        assert x.F.shape[0] == len(uniq) < len(cells)  # the transform makes voxels collide
        got = {tuple(r): f for r, f in zip(x.C.cpu().numpy().tolist(), x.F.detach().cpu().numpy())}
        sums = np.zeros((len(uniq), 28), np.float64)
        np.add.at(sums, inv.reshape(-1), want_f.astype(np.float64))
        mean = sums / np.bincount(inv.reshape(-1), minlength=len(uniq))[:, None]
        for j in np.random.default_rng(step).integers(0, len(uniq), 200):
            np.testing.assert_allclose(got[tuple(uniq[j].tolist())], mean[j], atol=1e-4)
        loss = torch.nn.functional.cross_entropy(net(field), batch["labels"].long())
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert torch.isfinite(loss)


def test_deferred_augmented_batch_equals_immediate():
    """Two-phase prepare of an augmented batch with dropout: process_input(defer=True) never blocks (the
    survivor count travels through pinned memory), finish_input builds the same field as the blocking path."""
    from nerf_downstream_amd.co3d_3d.src.data.synthetic import SparseVoxelDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.co3d_3d.src.models.mink.base_model import _PendingField

    random.seed(1), np.random.seed(1)
    ds = SparseVoxelDataset("train", num_samples=4, num_classes=4, grid=48,
                            train_transformations=["RandomRotation", "CoordinateDropout", "RandomHorizontalFlip", "RandomScale"])
    ds.transformations.transforms[1].application_ratio = 1.0  # every scene drops voxels
    host = collate_mink([ds[j] for j in range(4)])
    batch = {k: (v.cuda() if torch.is_tensor(v) and k != "aug_params" else v) for k, v in host.items()}
    torch.manual_seed(0)
    net = get_model("ResNet14", 28, 4).cuda().eval()
    with torch.no_grad():
        want = net(net.process_input(batch))
        pend = net.process_input(batch, defer=True)
        assert isinstance(pend, _PendingField)
        got = net(net.finish_input(pend))
        again = net.process_input(batch, defer=True)
        got2 = net(again)  # forward on the pending object itself also works (sparse() finishes it)
    assert torch.equal(want, got) and torch.equal(want, got2)
