"""CPU restatement of the PeRFception-CO3D `data.npz` decode of the reference loader -- TEST
INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference (it ships no fixture
for its loader); pinned here against the product's own CPU loader functions and a hand-computed case.

Follows co3d_3d/src/data/co3d.py:160-166 (`sh.astype(float32) * sh_scale + sh_min`), :196-203
(links -> x, y, z with x = links // (ry*rz), y = links % (ry*rz) // rz, z = links % rz) and :222-229
(the selected feature columns are concatenated in the order of the `features` list)."""
import numpy as np


def decode_batch(scenes, features=("density", "sh"), reso=(128, 128, 128)):
    """scenes: list of dicts {links int32 [N], density f32 [N] or [N,1], sh_q uint8 [N,27], sh_scale, sh_min}
    -> coords int32 [sumN, 4] (batch, x, y, z), feats float32 [sumN, C]."""
    coords, feats = [], []
    yz = reso[1] * reso[2]
    for b, s in enumerate(scenes):
        links = np.asarray(s["links"]).astype(np.int64)
        xyz = np.stack([links // yz, links % yz // reso[2], links % reso[2]], 1)
        coords.append(np.concatenate([np.full((len(links), 1), b, np.int64), xyz], 1).astype(np.int32))
        sh = np.asarray(s["sh_q"]).astype(np.float32) * np.float32(1) * np.asarray(s["sh_scale"], np.float32) \
            + np.asarray(s["sh_min"], np.float32)
        # "xyzs" (co3d.py:209-214): each point minus the mean of ITS OWN three coordinates (the reference reduces over dim=1),
        # divided by the largest norm of the scene; float32, every operation rounded separately, in this order
        f = xyz.astype(np.float32)
        m = ((f[:, 0] + f[:, 1]) + f[:, 2]) / np.float32(3)
        d = f - m[:, None]
        nrm = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        xyzs = (d / nrm.max()).astype(np.float32) if len(links) else d
        cols = {"density": np.asarray(s["density"], np.float32).reshape(-1, 1), "sh": sh.astype(np.float32),
                "ones": np.ones((len(links), 1), np.float32), "xyzs": xyzs}
        feats.append(np.concatenate([cols[f] for f in features], 1).astype(np.float32))
    return np.concatenate(coords), np.concatenate(feats)
