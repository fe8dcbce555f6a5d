"""CPU: the train.py surface (argparse flags, gin files, dataset schema, collate, checkpoints)
end to end.  BASELINE config #0 ("2-class synthetic plenoxel voxels, batch=2 on the CPU path")
runs the SAME training loop with the oracle's mini-ME injected -- the product itself has no CPU
path and must refuse to run without a GPU."""
import os

import numpy as np
import pytest
import torch

from nerf_downstream_amd import gin_lite as gin

CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_downstream_amd", "co3d_3d", "configs")


@pytest.fixture(autouse=True)
def _clean_gin():
    gin.clear_config()
    yield
    gin.clear_config()


def test_gin_subset_parses_reference_style_files(tmp_path):
    gin.parse_config_files_and_bindings([f"{CFG}/co3d_cls.gin", f"{CFG}/resnet34.gin"], ["train.gpus=8", "train.lr = 0.05"])
    assert gin.query_parameter("get_model.name") == "ResNet34"  # later file wins
    assert gin.query_parameter("train.lr") == 0.05  # bindings win over files
    assert gin.query_parameter("train.loggers") == ["csv", "neptune"]
    assert gin.query_parameter("Co3DDatasetBase.features") == ["sh"]
    with pytest.raises(gin.GinError):
        gin.query_parameter("train.nope")
    ref = "/root/reference/co3d_3d/configs"
    if os.path.isdir(ref):  # every config file of the reference parses (not available on the GPU box)
        for f in sorted(os.listdir(ref)):
            gin.parse_config_file(os.path.join(ref, f))
        assert "get_model.name" in gin.query_parameter("logged.hyper_params")


def test_cli_flags_match_reference():
    from nerf_downstream_amd.co3d_3d.train import build_parser

    a = build_parser().parse_args(["--ginc", "a.gin", "--ginc", "b.gin", "--ginb", "train.lr=1", "--gpus", "4", "--seed", "3",
                                   "--resume", "--run_name", "r", "--run_name_postfix", "p", "--save_path", "s", "--debug"])
    assert a.ginc == ["a.gin", "b.gin"] and a.ginb == ["train.lr=1"] and a.gpus == 4 and a.seed == 3 and a.resume and a.debug


def test_dataset_schema_and_collate():
    from nerf_downstream_amd.co3d_3d.src.data.synthetic import SparseVoxelDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink

    ds = SparseVoxelDataset(phase="train", num_samples=4, num_classes=51, grid=32, features=["density", "sh"])
    s = ds[1]
    assert set(s) == {"coordinates", "features", "xyzs", "labels"}  # reference co3d.py:231-242
    assert s["coordinates"].dtype == torch.float32 and s["coordinates"].shape[1] == 3
    assert s["features"].shape == (s["coordinates"].shape[0], 28) and s["labels"].shape == (1,)
    assert torch.equal(s["coordinates"], s["coordinates"].floor())  # integer voxel indices as floats
    assert torch.equal(ds[1]["features"], s["features"])  # deterministic
    b = collate_mink([ds[0], ds[1]])
    n0 = ds[0]["coordinates"].shape[0]
    assert b["coordinates"].dtype == torch.float32 and b["coordinates"].shape[1] == 4
    assert torch.all(b["coordinates"][:n0, 0] == 0) and torch.all(b["coordinates"][n0:, 0] == 1)
    assert b["labels"].dtype == torch.int64 and b["labels"].tolist() == [0, 1]
    assert SparseVoxelDataset(features=["sh"], grid=32)[0]["features"].shape[1] == 27


def test_co3d_npz_loader(tmp_path, monkeypatch):
    from nerf_downstream_amd.co3d_3d.src.data.co3d import CLASSES, Co3DDataset

    rng = np.random.default_rng(0)
    links = np.sort(rng.choice(128 ** 3, 500, replace=False)).astype(np.int32)
    scene = tmp_path / "data" / "plenoxel_co3d_sceneA"
    scene.mkdir(parents=True)
    sh = rng.integers(0, 255, (500, 27)).astype(np.uint8)
    np.savez(scene / "data.npz", links=links, density=rng.random((500, 1)).astype(np.float32), sh=sh,
             sh_min=np.float32(-2.0), sh_scale=np.float32(0.01), reso=[[128] * 3, [256] * 3])
    (tmp_path / "filelist").mkdir()
    (tmp_path / "filelist" / "train.txt").write_text("cup sceneA extra\n")
    monkeypatch.chdir(tmp_path)
    ds = Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=["density", "sh"])
    s = ds[0]
    c = s["coordinates"].long()
    assert torch.equal(c[:, 0] * 128 * 128 + c[:, 1] * 128 + c[:, 2], torch.from_numpy(links).long())
    assert s["labels"][0] == CLASSES.index("cup") and s["features"].shape == (500, 28)
    assert torch.allclose(s["features"][:, 1:], torch.from_numpy(sh.astype(np.float32) * 0.01 - 2.0))


def _write_scenes(tmp_path, n_scenes=2, per_channel=False):
    rng = np.random.default_rng(1)
    (tmp_path / "filelist").mkdir()
    lines, raw = [], []
    for j in range(n_scenes):
        n = 300 + 57 * j
        links = np.sort(rng.choice(128 ** 3, n, replace=False)).astype(np.int32)
        sh = rng.integers(0, 256, (n, 27)).astype(np.uint8)
        dens = rng.random((n, 1)).astype(np.float32)
        scale = (rng.random(27).astype(np.float32) * 0.02 + 0.001) if per_channel else np.float32(0.0123 + j)
        mn = (rng.standard_normal(27).astype(np.float32)) if per_channel else np.float32(-1.5 - j)
        scene = tmp_path / "data" / f"plenoxel_co3d_s{j}"
        scene.mkdir(parents=True)
        np.savez(scene / "data.npz", links=links, density=dens, sh=sh, sh_min=mn, sh_scale=scale, reso=[[128] * 3, [256] * 3])
        lines.append(f"cup s{j}")
        raw.append({"links": links, "density": dens, "sh_q": sh, "sh_scale": scale, "sh_min": mn})
    (tmp_path / "filelist" / "train.txt").write_text("\n".join(lines) + "\n")
    return raw


@pytest.mark.parametrize("per_channel", [False, True])
def test_compact_co3d_batch_and_decode_oracle(tmp_path, monkeypatch, per_channel):
    """Compact (on-disk form) samples + collate, and the decode oracle against the CPU loader: the
    decoded batch must be exactly what the ordinary loader + collate produce."""
    from nerf_downstream_amd.co3d_3d.src.data.co3d import Co3DDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink
    from oracle.decode import decode_batch

    raw = _write_scenes(tmp_path, per_channel=per_channel)
    monkeypatch.chdir(tmp_path)
    feats = ["sh", "density"]  # order matters: columns follow the list (reference co3d.py:226-229)
    plain = collate_mink([Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=feats)[i] for i in range(2)])
    ds = Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=feats, compact=True)
    batch = collate_mink([ds[0], ds[1]])
    assert batch["links"].dtype == torch.int32 and batch["sh_q"].dtype == torch.uint8 and batch["sh_scale"].shape == (2, 27)
    assert batch["scene_offsets"].tolist() == [0, 300, 657] and batch["feature_names"] == ("sh", "density")
    coords, f = decode_batch(raw, features=feats)
    assert np.array_equal(coords, plain["coordinates"].numpy().astype(np.int32))
    assert np.array_equal(f, plain["features"].numpy())  # bit-exact: same float32 multiply-then-add
    # the "xyzs" feature (configs/feature_coord.gin; a per-scene reduction): the decode oracle against the CPU loader's
    # torch expression -- float32 on both sides, within a rounding of the division
    fx = ["xyzs", "density"]
    plain_x = collate_mink([Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=fx)[i] for i in range(2)])
    dsx = Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=fx, compact=True)
    assert collate_mink([dsx[0], dsx[1]])["feature_names"] == ("xyzs", "density")
    _, f_x = decode_batch(raw, features=fx)
    assert f_x.shape == plain_x["features"].shape and np.abs(f_x - plain_x["features"].numpy()).max() < 2e-7
    assert np.abs(f_x[:, :3]).max() <= 1.0 + 1e-6


def test_model_parameter_names_and_counts():
    """State-dict layout of the reference ResNetBase and its parameter counts."""
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    m = get_model("ResNet14", 28, 51, ME=OME)
    keys = list(m.state_dict())
    for k in ["conv1.kernel", "bn1.bn.weight", "bn1.bn.running_var", "bn1.bn.num_batches_tracked", "layer1.0.conv1.kernel",
              "layer1.0.norm2.bn.bias", "layer3.0.downsample.0.kernel", "layer4.0.downsample.1.bn.running_mean",
              "final.kernel", "final.bias"]:
        assert k in keys, k
    assert m.conv1.kernel.shape == (27, 28, 64) and m.layer2[0].downsample[0].kernel.shape == (1, 64, 128)
    assert m.final.kernel.shape == (512, 51) and m.final.bias.shape == (1, 51)
    assert sum(p.numel() for p in m.parameters()) == 14_412_339
    m34 = get_model("ResNet34", 28, 51, ME=OME)
    assert sum(p.numel() for p in m34.parameters()) == 63_526_451
    assert len(m34.layer3) == 6
    # Bottleneck family (reference resnet_block.py:76-132, resnet.py:195-202): 1x1x1 reduce, 3x3x3, 1x1x1 expand x4
    m50 = get_model("ResNet50", 28, 51, ME=OME)
    keys = list(m50.state_dict())
    for k in ["layer1.0.conv1.kernel", "layer1.0.conv3.kernel", "layer1.0.norm3.bn.weight", "layer1.0.downsample.0.kernel",
              "layer3.5.conv2.kernel", "layer4.2.norm3.bn.running_mean", "final.kernel"]:
        assert k in keys, k
    assert m50.layer1[0].conv1.kernel.shape == (64, 64) and m50.layer1[0].conv2.kernel.shape == (27, 64, 64)
    assert m50.layer1[0].conv3.kernel.shape == (64, 256) and m50.layer2[0].downsample[0].kernel.shape == (1, 256, 512)
    assert m50.layer2[0].conv2.stride == 2 and m50.layer2[0].conv1.stride == 1 and m50.final.kernel.shape == (2048, 51)
    want, inp = 27 * 28 * 64 + 2 * 64 + 2048 * 51 + 51, 64
    for planes, blocks in zip((64, 128, 256, 512), (3, 4, 6, 3)):
        for j in range(blocks):
            want += inp * planes + 27 * planes * planes + planes * 4 * planes + 2 * (planes + planes + 4 * planes)
            if j == 0:
                want += inp * 4 * planes + 2 * 4 * planes
            inp = 4 * planes
    assert sum(p.numel() for p in m50.parameters()) == want
    assert [len(getattr(get_model("ResNet101", 28, 51, ME=OME), f"layer{i}")) for i in (1, 2, 3, 4)] == [3, 4, 23, 3]


def test_res16unet_names_and_shapes():
    """State-dict layout of the reference Res16UNet (res16unet.py:60-352) and the gin-configured base class."""
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    m = get_model("Res16UNet14A", 28, 20, ME=OME)
    keys = list(m.state_dict())
    for k in ["conv0p1s1.0.kernel", "conv0p1s1.1.bn.weight", "conv0p1s1.3.kernel", "conv0p1s1.4.bn.running_mean", "conv1p1s2.0.kernel",
              "block1.0.conv1.kernel", "block2.0.downsample.0.kernel", "conv4p8s2.1.bn.bias", "convtr4p16s2.0.kernel",
              "block5.0.downsample.1.bn.weight", "convtr7p2s2.1.bn.num_batches_tracked", "block8.0.norm2.bn.bias", "final.kernel",
              "final.bias"]:
        assert k in keys, k
    assert m.conv1p1s2[0].kernel.shape == (8, 32, 32) and m.conv1p1s2[0].stride == 2
    assert m.convtr4p16s2[0].kernel.shape == (8, 256, 128) and m.block5[0].conv1.kernel.shape == (27, 128 + 128, 128)
    assert m.block8[0].conv1.kernel.shape == (27, 96 + 32, 96) and m.final.kernel.shape == (96, 20) and m.final.bias.shape == (1, 20)
    assert m.block2[0].downsample[0].kernel.shape == (32, 64)  # 1x1x1 stride 1: plain matrix (use_mm)
    gin.clear_config()
    gin.parse_config_file(f"{CFG}/res16unet.gin")
    try:
        u = get_model(in_channel=28, out_channel=20, ME=OME)
        assert type(u).__name__ == "Res16UNet" and u.LAYERS == (1, 1, 2, 2, 2, 2, 1, 1) and len(u.block3) == 2
        assert u.PLANES == (32, 48, 64, 96, 96, 96, 64, 64) and u.convtr5p8s2[0].kernel.shape == (8, 96, 96)
    finally:
        gin.clear_config()
    assert len(get_model("Res16UNet34C", 28, 20, ME=OME).block4) == 6


def test_product_train_refuses_cpu(tmp_path):
    from nerf_downstream_amd.co3d_3d.train import train

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    gin.parse_config_files_and_bindings([f"{CFG}/co3d_cls.gin", f"{CFG}/synthetic_2class_cpu.gin"], ["train.gpus=1", "train.max_steps=1"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        train(save_path=str(tmp_path), resume_training=False, run_name=None, run_name_postfix=None)


def test_baseline_config0_cpu_plumbing(tmp_path):
    """co3d_3d/train.py, Mink-ResNet14, 2-class synthetic plenoxels, batch 2, CPU oracle backend."""
    from nerf_downstream_amd.co3d_3d.train import load_checkpoint, train
    from oracle import me_cpu as OME

    gin.parse_config_files_and_bindings(
        [f"{CFG}/co3d_cls.gin", f"{CFG}/resnet14.gin", f"{CFG}/synthetic_2class_cpu.gin"],
        ["train.gpus=0", "train.max_steps=4", "train.val_every_n_steps=4", "train.log_every_n_steps=1",
         "SparseVoxelDataset.grid=24", "train.lr=0.01"],
    )
    res = train(save_path=str(tmp_path), resume_training=False, run_name="t", run_name_postfix=None, ME=OME)
    assert res["global_step"] == 4
    logged = [h for h in res["history"] if "train/loss" in h]
    assert len(logged) == 4 and all(np.isfinite(h["train/loss"]) for h in logged)
    val = [h for h in res["history"] if "val/acc1" in h]
    assert len(val) == 1 and 0.0 <= val[0]["val/acc1"] <= 100.0
    ckpt = tmp_path / "t" / "last.ckpt"
    assert ckpt.exists() and (tmp_path / "t" / "metrics.csv").exists()
    sd = torch.load(ckpt, weights_only=False)
    assert all(k.startswith("model.") for k in sd["state_dict"]) and sd["global_step"] == 4
    assert sd["batch_in_epoch"] == 4 and sd["epoch"] == 0  # position inside the epoch: a resumed run skips those batches
    import csv

    rows_before = list(csv.DictReader(open(tmp_path / "t" / "metrics.csv")))
    assert len(rows_before) == 5  # 4 training rows + 1 validation row
    # resume continues from the stored step, keeps the metrics history and does not replay consumed samples
    gin.bind_parameter("train.max_steps", 6)
    res2 = train(save_path=str(tmp_path), resume_training=True, run_name="t", run_name_postfix=None, ME=OME)
    assert res2["global_step"] == 6
    rows_after = list(csv.DictReader(open(tmp_path / "t" / "metrics.csv")))
    assert len(rows_after) == 5 + 3 and [r["global_step"] for r in rows_after[:5]] == [r["global_step"] for r in rows_before]
    sd2 = torch.load(ckpt, weights_only=False)  # the 4-batch epoch 0 was skipped as already consumed: 2 batches into epoch 1
    assert (sd2["epoch"], sd2["batch_in_epoch"]) == (1, 2)


def test_segmentation_cpu_plumbing(tmp_path):
    """co3d_3d/train.py with train.training_module = SegmentationTraining: Res16UNet on per-voxel labels, CPU oracle
    backend; and the confusion-matrix metrics against a hand-computed case (reference utils fast_hist / per_class_iu)."""
    from nerf_downstream_amd.co3d_3d.src.modules.segmentation_training import confusion, iou_metrics
    from nerf_downstream_amd.co3d_3d.train import train
    from oracle import me_cpu as OME

    pred = torch.tensor([0, 0, 1, 1, 2, 2, 0, 1])
    label = torch.tensor([0, 1, 1, 1, 2, 0, 255, -1])  # the last two are ignored (outside 0..n-1)
    h = confusion(pred, label, 3)
    assert h.tolist() == [[1, 0, 1], [1, 2, 0], [0, 0, 1]]
    miou, macc, oa = iou_metrics(h)
    assert abs(miou - 100 * (1 / 3 + 2 / 3 + 1 / 2) / 3) < 1e-6 and abs(macc - 100 * (1 / 2 + 2 / 3 + 1) / 3) < 1e-6
    assert abs(oa - 100 * 4 / 6) < 1e-6
    assert iou_metrics(confusion(pred[:4], label[:4], 5))[0] == pytest.approx(100 * (1 / 2 + 2 / 3) / 2)  # absent classes do not count

    gin.parse_config_files_and_bindings(
        [f"{CFG}/co3d_cls.gin", f"{CFG}/synthetic_seg.gin", f"{CFG}/res16unet.gin"],
        ["train.gpus=0", "train.max_steps=3", "train.val_every_n_steps=3", "train.log_every_n_steps=1", "train.batch_size=2",
         "train.val_batch_size=2", "SparseVoxelSegDataset.grid=16", "SparseVoxelSegDataset.num_samples=8", "train.lr=0.01"],
    )
    res = train(save_path=str(tmp_path), resume_training=False, run_name="s", run_name_postfix=None, ME=OME)
    assert res["global_step"] == 3
    logged = [x for x in res["history"] if "train/loss" in x]
    assert len(logged) == 3 and all(np.isfinite(x["train/loss"]) and 0 <= x["train/mIoU"] <= 100 for x in logged)
    assert 0 < logged[0]["train/ignore_ratio"] < 20
    val = [x for x in res["history"] if "val/mIoU" in x]
    assert len(val) == 1 and 0.0 <= val[0]["val/mIoU"] <= 100.0 and np.isfinite(val[0]["val/loss"])
    assert (tmp_path / "s" / "best.ckpt").exists()
    with pytest.raises(ValueError, match="training_module"):
        gin.bind_parameter("train.training_module", "PruningTraining")
        train(save_path=str(tmp_path), resume_training=False, run_name="s", run_name_postfix=None, ME=OME)


def test_gin_configurable_injection_rules():
    @gin.configurable
    def f(a, b=2, c=3):
        return a, b, c

    @gin.configurable
    class K:
        def __init__(self, x=1, y=2):
            self.x, self.y = x, y

    gin.parse_config("f.b = 20\nf.c = [1,\n  2]\nK.y = 'z'")
    assert f(1) == (1, 20, [1, 2])       # bound parameters fill what the caller leaves out
    assert f(1, 5, c=7) == (1, 5, 7)     # explicit arguments always win
    assert (K().x, K().y, K(y=9).y) == (1, "z", 9)
    gin.bind_parameter("f.nope", 1)
    with pytest.raises(gin.GinError):
        f(1)                             # binding a parameter the configurable does not have
    with pytest.raises(gin.GinError):
        gin.parse_config("f.b = undefined_name")


def test_coordinate_plan_compilation():
    """Pure host logic of the prepare-ahead pipeline: request trace -> de-duplicated plan."""
    from nerf_downstream_amd.minkowski.coords import CoordinateManager

    trace = [("ktable", 1, 1, 3, 1, False), ("stride", 1, 2), ("ktable", 1, 2, 2, 1, False), ("stride", 2, 2),
             ("ktable", 2, 4, 3, 1, False), ("ktable", 4, 4, 3, 1, False), ("boff", 4),
             ("ktable", 4, 4, 3, 1, False), ("ktable", 2, 4, 3, 1, True), ("perm", 2, 128), ("stride", 2, 2)]
    plan = CoordinateManager.compile_plan(trace)
    assert plan.count(("stride", 2, 2)) == 1 and len(plan) == 8
    assert ("ktable", 2, 4, 3, 1, True) in plan and ("ktable", 2, 4, 3, 1, False) not in plan  # built transposed once
    assert plan.index(("stride", 1, 2)) < plan.index(("ktable", 1, 2, 2, 1, False))            # first-use order kept
    assert CoordinateManager.plan_stride_chain(plan) == (2, 2)
    assert CoordinateManager.plan_stride_chain(None) == ()


def test_co3d_last_ckpt_variant(tmp_path, monkeypatch):
    """The second on-disk format of the reference loader (co3d.py:133-162): a Plenoxel training checkpoint `last.ckpt`
    on a 256^3 grid, used when a scene has no pre-processed data.npz -- ordinary and compact (GPU-decoded) samples."""
    from nerf_downstream_amd.co3d_3d.src.data.co3d import Co3DDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink
    from oracle.decode import decode_batch

    rng = np.random.default_rng(5)
    n = 400
    links = np.sort(rng.choice(256 ** 3, n, replace=False)).astype(np.int64)
    sh = rng.integers(0, 256, (n, 27)).astype(np.uint8)
    dens = rng.random((n, 1)).astype(np.float32)
    scene = tmp_path / "data" / "plenoxel_co3d_ck0"
    scene.mkdir(parents=True)
    torch.save({"state_dict": {"model.links_idx": torch.from_numpy(links), "model.density_data": torch.from_numpy(dens),
                               "model.sh_data": torch.from_numpy(sh)},
                "model.sh_data_min": torch.tensor(-1.25), "model.sh_data_scale": torch.tensor(0.0123), "reso_idx": 1}, scene / "last.ckpt")
    (tmp_path / "filelist").mkdir()
    (tmp_path / "filelist" / "train.txt").write_text("kite ck0\n")
    monkeypatch.chdir(tmp_path)
    ds = Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=["density", "sh"])
    s = ds[0]
    c = s["coordinates"].long()
    assert int(c.max()) > 127  # a 256^3 grid
    assert torch.equal(c[:, 0] * 256 * 256 + c[:, 1] * 256 + c[:, 2], torch.from_numpy(links))
    assert torch.allclose(s["features"][:, 1:], torch.from_numpy(sh.astype(np.float32) * np.float32(0.0123) + np.float32(-1.25)))
    assert ds.sample_lengths() is None  # only data.npz headers can be read without loading the scene
    cds = Co3DDataset(phase="train", data_root=str(tmp_path / "data"), features=["density", "sh"], compact=True)
    batch = collate_mink([cds[0]])
    assert batch["reso"] == (256, 256, 256) and batch["links"].dtype == torch.int32
    coords, feats = decode_batch([{"links": links, "density": dens, "sh_q": sh, "sh_scale": np.float32(0.0123), "sh_min": np.float32(-1.25)}],
                                 features=["density", "sh"], reso=(256, 256, 256))
    plain = collate_mink([s])
    assert np.array_equal(coords, plain["coordinates"].numpy().astype(np.int32)) and np.array_equal(feats, plain["features"].numpy())


def _write_scannet_tree(root, n_scenes=3):
    import pickle

    rng = np.random.default_rng(11)
    data_root = root / "perfception-scannet"
    (root / "split").mkdir(parents=True)
    names, scales, raw = [], {}, {}
    for j in range(n_scenes):
        name = f"scene{j:04d}_00"
        reso = np.array([64, 48, 56])
        n = 900 + 100 * j
        links = np.sort(rng.choice(int(reso.prod()), n, replace=False)).astype(np.int64)
        d = {"links": links, "density": rng.random((n, 1)).astype(np.float32) * 5, "sh": rng.integers(0, 256, (n, 27)).astype(np.uint8),
             "sh_scale": np.float32(0.01), "sh_min": np.float32(-1.0), "reso": reso,
             "labels": rng.choice([0, 1, 2, 5, 13, 16, 39, 40], n).astype(np.int64), "dists": (rng.random(n) * 0.1).astype(np.float32)}
        scene = data_root / f"plenoxel_torch_{name}"
        scene.mkdir(parents=True)
        np.savez(scene / "data.npz", **d)
        names.append(name)
        scales[name] = 1.0 + 0.25 * j
        raw[name] = d
    for f in ("scannet_256_train.txt", "scannet_256_val.txt"):
        (root / "split" / f).write_text("# comment\n" + "\n".join(names) + "\n")
    with open(root / "split" / "scene_scales.data", "wb") as f:
        pickle.dump(scales, f)
    return data_root, names, scales, raw


def test_plenoxel_scannet_dataset_and_segmentation_run(tmp_path):
    """PeRFception-ScanNet loader (reference scannet.py:450-660): void / ignore handling, stride sub-sampling, metric
    coordinates, label mapping -- checked against a direct evaluation of the reference formulas -- then two training
    steps of Res16UNet on it through train.py (CPU oracle backend; float coordinates are floored and averaged)."""
    from nerf_downstream_amd.co3d_3d.src.data.scannet import VALID_CLASS_IDS, PlenoxelScannetDataset
    from nerf_downstream_amd.co3d_3d.train import train
    from oracle import me_cpu as OME

    data_root, names, scales, raw = _write_scannet_tree(tmp_path)
    ds = PlenoxelScannetDataset("train", data_root=str(data_root), features=["density", "sh"], ignore_label=-255, valid_thres=0.05,
                                ignore_thres=0.08)
    assert len(ds) == 3 and ds.NUM_CLASSES == 20
    s = ds[1]
    d = raw[names[1]]
    keep = d["dists"] < 0.08
    links, reso = d["links"][keep], d["reso"]
    grid = np.stack([links // (reso[1] * reso[2]), links % (reso[1] * reso[2]) // reso[2], links % reso[2]], 1).astype(np.float32)
    sel = (grid % 2 == 0).all(1)
    want_xyz = (grid[sel] / reso * 2 - 1.0) / scales[names[1]] / 0.02
    assert np.allclose(s["coordinates"].numpy(), want_xyz, atol=1e-4) and s["coordinates"].dtype == torch.float32
    dens = d["density"][keep].reshape(-1)
    dens = dens / (np.abs(dens).max() + 1e-5)  # more than one feature: density is max-normalised (reference :602-603)
    assert np.allclose(s["features"][:, 0].numpy(), dens[sel], atol=1e-6)
    assert np.allclose(s["features"][:, 1:].numpy(), (d["sh"][keep][sel].astype(np.float32) * np.float32(0.01) - 1.0), atol=1e-6)
    lab = d["labels"][keep].copy()
    lab[d["dists"][keep] > 0.05] = -255  # void
    lab = lab[sel]
    want = np.array([VALID_CLASS_IDS.index(v) if v in VALID_CLASS_IDS else -255 for v in lab])
    assert np.array_equal(s["labels"], want) and s["labels"].dtype == np.int64
    assert set(np.unique(s["labels"])) <= set(range(20)) | {-255} and (s["labels"] == -255).any() and (s["labels"] >= 0).any()
    with pytest.raises(NotImplementedError):
        PlenoxelScannetDataset("train", data_root=str(data_root), train_transformations=["ElasticDistortion"])

    gin.parse_config_files_and_bindings(
        [f"{CFG}/scannet_plenoxel.gin", f"{CFG}/res16unet.gin"],
        ["train.gpus=0", "train.max_steps=2", "train.val_every_n_steps=2", "train.log_every_n_steps=1", "train.batch_size=2",
         "train.val_batch_size=1", "train.train_num_workers=0", "train.val_num_workers=0", "train.lr=0.01",
         f"PlenoxelScannetDataset.data_root='{data_root}'", "get_model.name='Res16UNet14A'"])
    res = train(save_path=str(tmp_path / "run"), resume_training=False, run_name="s", run_name_postfix=None, ME=OME)
    logged = [h for h in res["history"] if "train/loss" in h]
    assert res["global_step"] == 2 and len(logged) == 2 and all(np.isfinite(h["train/loss"]) for h in logged)
    assert any("val/mIoU" in h for h in res["history"])


def test_density_based_sample_recipe(tmp_path, monkeypatch):
    """`DensityBasedSample` (reference transforms.py:655-682; bound by configs/co3d_aug3.gin:13-15): the reference's own
    config file parses, and a recipe that lists the class keeps exactly the voxels above the scene's density percentile
    (np.percentile semantics: the value is in percent, the comparison strict) -- in the decoded and in the compact
    sample form alike.  After CoordinateDropout it is refused (the percentile would be taken over a random subset)."""
    from nerf_downstream_amd import gin_lite as gin
    from nerf_downstream_amd.co3d_3d.src.data import transforms as T
    from nerf_downstream_amd.co3d_3d.src.data.co3d import Co3DDataset

    raw = _write_scenes(tmp_path)
    monkeypatch.chdir(tmp_path)
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gin.clear_config()
    try:
        gin.parse_config_files_and_bindings([os.path.join(ROOT, "nerf_downstream_amd", "co3d_3d", "configs", "co3d_aug3.gin")],
                                            ["DensityBasedSample.percentile = 40.0"])
        assert T.DensityBasedSample().percentile == 40.0 and T.DensityBasedSample().density_dim == 3
        kw = dict(phase="train", data_root=str(tmp_path / "data"), features=["density", "sh"],
                  train_transformations=["DensityBasedSample", "RandomScale"])
        plain, compact = Co3DDataset(**kw)[1], Co3DDataset(compact=True, **kw)[1]
    finally:
        gin.clear_config()
    d = raw[1]["density"].reshape(-1)
    keep = d > np.percentile(d, 40.0)
    assert 0 < keep.sum() < len(d) and plain["coordinates"].shape[0] == keep.sum() == compact["links"].shape[0]
    assert np.array_equal(compact["links"].numpy(), raw[1]["links"][keep])
    assert np.array_equal(plain["features"][:, 0].numpy(), d[keep]) and plain["aug_params"].shape == (44,)
    with pytest.raises(NotImplementedError):
        T.Compose([T.CoordinateDropout(), T.DensityBasedSample()])


def test_single_rank_resume_replays_the_epoch_permutation():
    """Single-rank shuffling is a function of (seed, epoch) (EpochSampler): a run resumed inside epoch e, b batches in,
    sees exactly the remaining indices of epoch e's permutation -- no scene twice, none skipped -- and the skipped
    samples are dropped by index, not loaded."""
    from nerf_downstream_amd.co3d_3d.src.data.data_module import EpochSampler

    s = EpochSampler(23, seed=5)
    s.set_epoch(3)
    full = list(s)
    assert sorted(full) == list(range(23)) and list(s) == full  # same epoch, same order
    s.set_epoch(4)
    assert list(s) != full
    r = EpochSampler(23, seed=5)  # the resumed process
    r.set_epoch(3)
    r.skip(2 * 4)  # two batches of four were trained on
    assert list(r) == full[8:]
    assert list(r) == full  # the skip applies to one iteration only
