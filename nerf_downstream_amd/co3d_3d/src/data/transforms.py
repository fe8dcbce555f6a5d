"""Scene augmentations (counterpart of the reference's co3d_3d/src/data/transforms.py; the gin names and
constructor arguments of the classes used by configs/co3d_aug3.gin:2-23 are kept).

The reference transforms every scene on the CPU inside the DataLoader workers.  Here a transform only
DRAWS its per-scene randomness (`draw()` appends stages to a list); `compile_program` folds the stage
list of one scene into the MINK_AUG_* parameter row of include/mink_hip.h and the whole batch is
transformed by `mink_augment_scenes` on the GPU when it reaches the model
(`MinkowskiBaseModel.process_input`).  Per-voxel randomness (dropout coins, coordinate jitter, feature
noise) comes from a Philox stream on the device, keyed by the batch seed and the scene's stream id.

Stage kinds: ("linear", M 3x3) c <- c @ M | ("translate", t) | ("flip", axes) c[ax] <- max(c[ax]) - c[ax] |
("dropout", ratio) | ("jitter", amplitude) c <- c + amplitude * (u - 0.5) |
("feature_jitter", std, start, dim) raw columns [start, start+dim) += (normal - 0.5) * std."""
import random

import numpy as np

from nerf_downstream_amd import gin_lite as gin

AUG = dict(A=0, a=9, FLIP=12, B=15, b=24, BJ=27, JITTER=36, DROPOUT=37, FEAT_STD=38, FEAT_START=39, FEAT_DIM=40,
           FLIP_ALL=41, PARAMS=44)  # include/mink_hip.h MINK_AUG_*
_AXIS = {"x": 0, "y": 1, "z": 2}
RAW_COLUMNS = {"xyzs": [0, 1, 2], "density": [3], "sh": list(range(4, 31)), "ones": [-1]}  # reference co3d.py:205-214


def raw_columns(feature_names):
    """raw-layout column of every selected feature column (RandomFeatureJitter indexes the raw layout)."""
    return [c for f in feature_names for c in RAW_COLUMNS[f]]


def rotation_matrix(axis, theta):
    """Rotation by theta about `axis` (Rodrigues); equals the reference's expm(cross(eye(3), axis/|axis|*theta))
    (transforms.py:334-336)."""
    k = np.asarray(axis, np.float64)
    k = k / np.linalg.norm(k)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(theta) * K + (1 - np.cos(theta)) * (K @ K)


class _Gated:
    application_ratio = 1.0

    def applies(self):
        return random.random() < self.application_ratio

    def __call__(self, coords, feats, labels):
        raise RuntimeError(f"{type(self).__name__} runs on the GPU: collect draw() into a program (see Compose.draw) "
                           "and hand the batch to MinkowskiBaseModel.process_input")


@gin.configurable()
class RandomRotation(_Gated):
    def __init__(self, upright_axis="z", axis_std=0.01, application_ratio=0.9):
        self.upright_axis, self.axis_std, self.application_ratio = _AXIS[upright_axis.lower()], axis_std, application_ratio

    def draw(self, stages):
        if self.applies():
            axis = self.axis_std * np.random.randn(3)
            axis[self.upright_axis] += 1
            stages.append(("linear", rotation_matrix(axis, random.random() * 2 * np.pi)))


@gin.configurable()
class RandomAffine(_Gated):
    def __init__(self, upright_axis="z", axis_std=0.1, scale_range=0.2, affine_range=0.1, application_ratio=0.9):
        self.upright_axis, self.axis_std = _AXIS[upright_axis.lower()], axis_std
        self.scale_range, self.affine_range, self.application_ratio = scale_range, affine_range, application_ratio

    def draw(self, stages):
        if self.applies():
            axis = self.axis_std * np.random.randn(3)
            axis[self.upright_axis] += 1
            angle = 2 * (random.random() - 0.5) * np.pi
            shear = np.diag(2 * (np.random.rand(3) - 0.5) * self.scale_range + 1)
            shear = shear + 2 * (np.random.rand(3, 3) - 0.5) * self.affine_range
            stages.append(("linear", rotation_matrix(axis, angle) @ shear))


@gin.configurable()
class RandomScale(_Gated):
    def __init__(self, scale_ratio=0.1, application_ratio=0.9):
        self.scale_ratio, self.application_ratio = scale_ratio, application_ratio

    def draw(self, stages):
        if self.applies():
            stages.append(("linear", np.eye(3) * np.random.uniform(1 - self.scale_ratio, 1 + self.scale_ratio)))


@gin.configurable()
class RandomTranslation(_Gated):
    def __init__(self, max_translation=3, application_ratio=0.9):
        self.max_translation, self.application_ratio = max_translation, application_ratio

    def draw(self, stages):
        if self.applies():
            stages.append(("translate", 2 * (np.random.rand(3) - 0.5) * self.max_translation))


@gin.configurable()
class CoordinateUniformTranslation(_Gated):
    def __init__(self, max_translation=0.2):
        self.max_translation = max_translation

    def draw(self, stages):
        if self.max_translation > 0:
            stages.append(("translate", np.random.uniform(-self.max_translation, self.max_translation, size=3)))


@gin.configurable()
class RandomHorizontalFlip(_Gated):
    def __init__(self, upright_axis="z", application_ratio=0.9):
        self.horz_axes = sorted(set(range(3)) - {_AXIS[upright_axis.lower()]})
        self.application_ratio = application_ratio

    def draw(self, stages):
        if self.applies():  # one gate, then BOTH horizontal axes are mirrored (reference :444-449)
            stages.append(("flip", tuple(self.horz_axes)))


@gin.configurable()
class CoordinateDropout(_Gated):
    def __init__(self, dropout_ratio=0.2, application_ratio=0.2):
        self.dropout_ratio, self.application_ratio = dropout_ratio, application_ratio

    def draw(self, stages):
        if self.applies():
            stages.append(("dropout", float(self.dropout_ratio)))


@gin.configurable()
class CoordinateJitter(_Gated):
    def __init__(self, jitter_std=0.5, application_ratio=0.7):
        self.jitter_std, self.application_ratio = jitter_std, application_ratio

    def draw(self, stages):
        if self.applies():
            stages.append(("jitter", 2.0 * self.jitter_std))


@gin.configurable()
class RandomFeatureJitter(_Gated):
    def __init__(self, std=0.01, application_ratio=0.9, start_ind=4, feature_dim=27):
        self.std, self.application_ratio, self.start_ind, self.feature_dim = std, application_ratio, start_ind, feature_dim

    def draw(self, stages):
        if self.applies():
            stages.append(("feature_jitter", float(self.std), int(self.start_ind), int(self.feature_dim)))


@gin.configurable()
class DensityBasedSample:
    """Keep the voxels whose density exceeds the scene's `percentile`-th percentile (reference transforms.py:655-682:
    `np.percentile(density, percentile)`, i.e. a value in PERCENT -- the 0.95 bound by configs/co3d_aug3.gin:13 keeps
    99 % of a scene -- and a strict `>`).  A deterministic row filter of one scene: it needs the scene's own density
    column, so it is applied where the sample is loaded (DataLoader worker, on the raw sample in either its decoded or
    its compact on-disk form), not in the per-batch GPU program.  It commutes with every coordinate transform and with
    the SH jitter, hence it may stand anywhere in a recipe BEFORE CoordinateDropout (after it the percentile would be
    taken over a random subset: not supported)."""

    prefilter = True

    def __init__(self, percentile=0.95, density_dim=3):
        assert density_dim > 0, f"density_dim should be larger than 0, but got {density_dim}"
        self.percentile, self.density_dim = percentile, density_dim
        if density_dim != RAW_COLUMNS["density"][0]:
            raise NotImplementedError("DensityBasedSample filters on the density column (raw column 3, co3d.py:205-214)")

    def mask(self, density):
        d = np.asarray(density, dtype=np.float32).reshape(-1)
        return d > np.percentile(d, self.percentile)

    def draw(self, stages):  # nothing for the GPU program
        return None


def compile_program(stages):
    """Fold one scene's drawn stage list into a MINK_AUG_* parameter row (float32 [PARAMS]).  Everything
    before the flip becomes (A, a), everything after it (B, b); BJ carries the linear stages that follow
    the jitter.  Composition is done in float64; the device applies the folded form in fp32."""
    has_flip = any(s[0] == "flip" for s in stages)
    pre = has_flip
    A, a, B, b, BJ = np.eye(3), np.zeros(3), np.eye(3), np.zeros(3), None
    P = np.zeros(AUG["PARAMS"], np.float64)
    seen = set()

    def once(kind):
        if kind in seen:
            raise NotImplementedError(f"more than one {kind} stage per scene")
        seen.add(kind)

    for s in stages:
        kind = s[0]
        if kind == "linear":
            L = np.asarray(s[1], np.float64).reshape(3, 3)
            if pre:
                A, a = A @ L, a @ L
            else:
                B, b = B @ L, b @ L
                BJ = None if BJ is None else BJ @ L
        elif kind == "translate":
            t = np.asarray(s[1], np.float64).reshape(3)
            if pre:
                a = a + t
            else:
                b = b + t
        elif kind == "flip":
            once(kind)
            for ax in s[1]:
                P[AUG["FLIP"] + ax] = 1
            P[AUG["FLIP_ALL"]] = 0 if "dropout" in seen else 1
            pre = False
        elif kind == "dropout":
            once(kind)
            if not 0 <= s[1] < 1:
                raise ValueError(f"dropout ratio {s[1]}")
            P[AUG["DROPOUT"]] = s[1]
        elif kind == "jitter":
            once(kind)
            if pre:
                raise NotImplementedError("CoordinateJitter listed before RandomHorizontalFlip")
            BJ, P[AUG["JITTER"]] = np.eye(3), s[1]
        elif kind == "feature_jitter":
            once(kind)
            P[AUG["FEAT_STD"]], P[AUG["FEAT_START"]], P[AUG["FEAT_DIM"]] = s[1], s[2], s[3]
        else:
            raise ValueError(f"unknown augmentation stage {kind!r}")
    P[AUG["A"]:AUG["A"] + 9], P[AUG["a"]:AUG["a"] + 3] = A.reshape(-1), a
    P[AUG["B"]:AUG["B"] + 9], P[AUG["b"]:AUG["b"] + 3] = B.reshape(-1), b
    P[AUG["BJ"]:AUG["BJ"] + 9] = (np.eye(3) if BJ is None else BJ).reshape(-1)
    return P.astype(np.float32)


class Compose:
    """The reference's Compose applies the transforms one after the other (transforms.py:710-720); this one
    lets each transform draw, in the same order, and returns the scene's stage list."""

    def __init__(self, transforms):
        self.transforms = list(transforms)
        self.prefilters = [t for t in self.transforms if getattr(t, "prefilter", False)]
        seen_dropout = False
        for t in self.transforms:
            seen_dropout = seen_dropout or isinstance(t, CoordinateDropout)
            if seen_dropout and getattr(t, "prefilter", False):
                raise NotImplementedError(f"{type(t).__name__} after CoordinateDropout: the filter would see a random subset")

    def row_mask(self, density):
        """AND of the recipe's per-scene row filters (None without any): applied by the dataset to the raw sample."""
        m = None
        for t in self.prefilters:
            k = t.mask(density)
            m = k if m is None else (m & k)
        return m

    def draw(self):
        stages = []
        for t in self.transforms:
            t.draw(stages)
        return stages

    def sample(self):
        """-> (params float32 [PARAMS], stream id) for one scene."""
        return compile_program(self.draw()), int(np.random.randint(0, 2 ** 32, dtype=np.uint64))

    def __repr__(self):
        return f"Compose({[type(t).__name__ for t in self.transforms]})"
