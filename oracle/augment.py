"""CPU restatement of the batch augmentation -- TEST INFRASTRUCTURE (see
oracle/__init__.py).  PARITY UNPINNED by the reference: its transforms draw from numpy's global
generator and ship no fixture, so what is pinned here is (a) the per-scene stage algebra against a
direct stage-by-stage evaluation of the reference's formulas (`stagewise`, following
co3d_3d/src/data/transforms.py:34-41, :255-265, :276-281, :288-294, :351-358, :367-373, :416-427,
:444-450) and (b) Philox4x32-10 against the published known-answer vectors of Random123.

`augment_batch` evaluates the canonical per-scene form of include/mink_hip.h (MINK_AUG_*) in numpy
float32 with the same operation order as the HIP kernel, so coordinates agree bit for bit."""
import numpy as np

A, a, FLIP, B, b, BJ, JITTER, DROPOUT, FEAT_STD, FEAT_START, FEAT_DIM, FLIP_ALL, PARAMS = 0, 9, 12, 15, 24, 27, 36, 37, 38, 39, 40, 41, 44
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11): ten rounds, key bumped
    between rounds.  Arguments broadcast; returns four uint32 arrays."""
    c = [np.asarray(v, np.uint64) & _U32 for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c[0], np.uint64(_M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & _U32, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & _U32]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def u01(w):
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def _vec_mat(v, M):
    """row vectors [n,3] (f32) times row-major 3x3 (f32): ((v0*M0j + v1*M1j) + v2*M2j), each op rounded."""
    M = M.reshape(3, 3)
    out = np.empty_like(v)
    for j in range(3):
        out[:, j] = (v[:, 0] * M[0, j] + v[:, 1] * M[1, j]) + v[:, 2] * M[2, j]
    return out


def augment_batch(coords, feats, scene_offsets, params, streams, seed, raw_cols):
    """coords f32 [n,4] (batch,x,y,z), feats f32 [n,C], params f32 [S,PARAMS], streams uint32 [S]
    -> (coords', feats') of the surviving voxels, in order."""
    coords, feats = np.asarray(coords, np.float32), np.asarray(feats, np.float32)
    params = np.asarray(params, np.float32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    oc, of = [], []
    for s in range(len(scene_offsets) - 1):
        lo, hi = int(scene_offsets[s]), int(scene_offsets[s + 1])
        P, c, f = params[s], coords[lo:hi, 1:], feats[lo:hi].copy()
        vox = np.arange(hi - lo, dtype=np.uint32)
        r = philox4x32_10(vox, 0, int(streams[s]), 0, k0, k1)
        keep = u01(r[0]) >= P[DROPOUT]
        p = _vec_mat(c, P[A:A + 9]) + P[a:a + 3]
        q = p.copy()
        for j in range(3):
            if P[FLIP + j] != 0:
                pool = p[:, j] if P[FLIP_ALL] != 0 else p[keep, j]
                if len(pool):
                    q[:, j] = pool.max() - p[:, j]
        o = _vec_mat(q, P[B:B + 9]) + P[b:b + 3]
        if P[JITTER] != 0:
            jit = np.stack([P[JITTER] * (u01(r[1 + j]) - np.float32(0.5)) for j in range(3)], 1).astype(np.float32)
            o = o + _vec_mat(jit, P[BJ:BJ + 9])
        if P[FEAT_STD] != 0:
            for col, raw in enumerate(raw_cols):
                j = raw - int(P[FEAT_START])
                if raw < 0 or j < 0 or j >= int(P[FEAT_DIM]):
                    continue
                g = philox4x32_10(vox, 1 + j // 4, int(streams[s]), 0, k0, k1)
                wa, wb = (g[2], g[3]) if j & 2 else (g[0], g[1])
                u1 = ((wa >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
                ang = 2 * np.pi * u01(wb).astype(np.float64)
                z = np.sqrt(-2 * np.log(u1)) * (np.sin(ang) if j & 1 else np.cos(ang))
                f[:, col] += ((z - 0.5) * float(P[FEAT_STD])).astype(np.float32)
        oc.append(np.concatenate([coords[lo:hi, :1], o.astype(np.float32)], 1)[keep])
        of.append(f[keep])
    return np.concatenate(oc), np.concatenate(of)


def stagewise(coords, stages, keep=None, jitter_u=None):
    """Direct float64 evaluation of a drawn stage list on one scene's [n,3] coordinates, one stage
    after the other exactly as the reference's Compose does (transforms.py:710-720).  `keep` /
    `jitter_u` are the per-voxel draws (mask, uniforms [n,3]) taken from the same Philox stream."""
    c = np.asarray(coords, np.float64).copy()
    alive = np.ones(len(c), bool)
    for st in stages:
        kind = st[0]
        if kind == "linear":
            c = c @ np.asarray(st[1], np.float64)
        elif kind == "translate":
            c = c + np.asarray(st[1], np.float64)
        elif kind == "dropout":
            alive &= keep
        elif kind == "flip":
            for ax in st[1]:
                c[:, ax] = c[alive, ax].max() - c[:, ax]
        elif kind == "jitter":
            c = c + st[1] * (np.asarray(jitter_u, np.float64) - 0.5)
        else:
            raise ValueError(kind)
    return c[alive]
