"""Pins the gather/GEMM/scatter restatement (oracle/me_cpu.py) against the independent
dense conv3d identity (oracle/dense_ref.py) -- forward and backward."""
import numpy as np
import pytest
import torch

from helpers import batch_scenes


def _field(seeds, cin, negative=False):
    from oracle import me_cpu as ME

    coords, feats = batch_scenes(seeds, grid=16, cin=cin, negative=negative)
    return ME, ME.TensorField(coordinates=coords, features=feats)


@pytest.mark.parametrize("ksize,stride,negative", [(3, 1, False), (3, 2, False), (3, 2, True), (1, 2, True), (3, 1, True)])
def test_conv_matches_dense(oracle_maps, ksize, stride, negative):
    from oracle import dense_ref

    torch.manual_seed(0)
    ME, tf = _field([1, 2], 5, negative)
    x = tf.sparse()
    # move to tensor stride 2 first so stride-2 convs are tested off the finest grid too
    pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)
    x2 = pool(x)
    for inp in (x, x2):
        ts = inp.tensor_stride[0]
        conv = ME.MinkowskiConvolution(5, 7, kernel_size=ksize, stride=stride, dimension=3)
        F_in = inp.F.detach().clone().requires_grad_(True)
        out = conv(ME.SparseTensor(F_in, inp.coordinate_map_key, inp._manager))
        m = inp._manager
        cin = m.coords[ts]
        cout = m.coords[ts * stride]
        F_ref = inp.F.detach().clone().requires_grad_(True)
        k_ref = conv.kernel.detach().clone().requires_grad_(True)
        ref = dense_ref.conv(cin, F_ref, k_ref, ksize, stride, ts, cout)
        assert out.F.shape == ref.shape
        assert torch.allclose(out.F, ref, atol=1e-5, rtol=1e-5)
        g = torch.randn_like(ref)
        out.F.backward(g)
        ref.backward(g)
        assert torch.allclose(F_in.grad, F_ref.grad, atol=1e-5, rtol=1e-5)
        assert torch.allclose(conv.kernel.grad, k_ref.grad, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("negative", [False, True])
def test_even_kernel_and_transposed_conv_match_dense(oracle_maps, negative):
    """The two extra operators of the segmentation family (reference res16unet.py:97-108,196-206):
    convolution k=2 s=2 (even kernel: offsets {0,1}, x fastest) and its transposed counterpart that up-samples
    onto the encoder's coordinate map; plus cat and slice."""
    from oracle import dense_ref

    torch.manual_seed(0)
    ME, tf = _field([5, 6], 5, negative)
    x = tf.sparse()
    m = x._manager
    down = ME.MinkowskiConvolution(5, 6, kernel_size=2, stride=2, dimension=3)
    F_in = x.F.detach().clone().requires_grad_(True)
    y = down(ME.SparseTensor(F_in, x.coordinate_map_key, m))
    assert y.tensor_stride[0] == 2 and down.kernel.shape == (8, 5, 6)
    F_ref, k_ref = x.F.detach().clone().requires_grad_(True), down.kernel.detach().clone().requires_grad_(True)
    ref = dense_ref.conv(m.coords[1], F_ref, k_ref, 2, 2, 1, m.coords[2])
    assert torch.allclose(y.F, ref, atol=1e-5, rtol=1e-5)
    g = torch.randn_like(ref)
    y.F.backward(g), ref.backward(g)
    assert torch.allclose(F_in.grad, F_ref.grad, atol=1e-5) and torch.allclose(down.kernel.grad, k_ref.grad, atol=1e-4)

    up = ME.MinkowskiConvolutionTranspose(6, 4, kernel_size=2, stride=2, dimension=3)
    Y_in = y.F.detach().clone().requires_grad_(True)
    z = up(ME.SparseTensor(Y_in, y.coordinate_map_key, m))
    assert z.tensor_stride[0] == 1 and z.F.shape == (x.F.shape[0], 4) and z.coordinate_map_key == x.coordinate_map_key
    Y_ref, u_ref = y.F.detach().clone().requires_grad_(True), up.kernel.detach().clone().requires_grad_(True)
    zref = dense_ref.conv_transpose(m.coords[2], Y_ref, u_ref, 2, 2, 2, m.coords[1])
    assert torch.allclose(z.F, zref, atol=1e-5, rtol=1e-5)
    g = torch.randn_like(zref)
    z.F.backward(g), zref.backward(g)
    assert torch.allclose(Y_in.grad, Y_ref.grad, atol=1e-5) and torch.allclose(up.kernel.grad, u_ref.grad, atol=1e-4)
    # every fine voxel has exactly one parent: one pair per output row
    assert (m.kernel_table_transposed(x.coordinate_map_key, y.coordinate_map_key, 2) >= 0).sum(1).tolist() == [1] * x.F.shape[0]

    c = ME.cat(z, x)
    assert c.F.shape == (x.F.shape[0], 9) and torch.equal(c.F[:, 4:], x.F) and c.coordinate_map_key == x.coordinate_map_key
    s = c.slice(tf)
    assert s.F.shape == (tf.F.shape[0], 9)
    inv = torch.from_numpy(m.field_inverse.astype("int64"))
    assert torch.equal(s.F, c.F[inv]) and torch.equal(torch.floor(tf.C).int(), torch.from_numpy(m.coords[1])[inv])


def test_sum_pool_matches_dense(oracle_maps):
    from oracle import dense_ref

    ME, tf = _field([3, 4], 4, negative=True)
    x = tf.sparse()
    pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)
    F_in = x.F.detach().clone().requires_grad_(True)
    y = pool(ME.SparseTensor(F_in, x.coordinate_map_key, x._manager))
    m = x._manager
    F_ref = x.F.detach().clone().requires_grad_(True)
    ref = dense_ref.sum_pool(m.coords[1], F_ref, 1, m.coords[2])
    assert torch.allclose(y.F, ref, atol=1e-5)
    g = torch.randn_like(ref)
    y.F.backward(g)
    ref.backward(g)
    assert torch.allclose(F_in.grad, F_ref.grad, atol=1e-6)
    # sum (not average): total mass is conserved
    assert torch.allclose(y.F.sum(0), x.F.sum(0), atol=1e-3)


def test_global_avg_pool_and_field_average(oracle_maps):
    from oracle import me_cpu as ME

    coords = torch.tensor([[0, 0.2, 0.7, 1.1], [0, 0.9, 0.1, 1.9], [1, -0.5, 0.0, 0.0], [0, 5.0, 5.0, 5.0], [1, -0.1, 0.3, 0.9]])
    feats = torch.arange(10.0).reshape(5, 2)
    tf = ME.TensorField(coordinates=coords, features=feats)
    x = tf.sparse()
    # rows 0,1 collapse to (0,0,0,1); rows 2,4 collapse to (1,-1,0,0)
    assert x.C.tolist() == [[0, 0, 0, 1], [1, -1, 0, 0], [0, 5, 5, 5]]
    assert torch.allclose(x.F, torch.tensor([[1.0, 2.0], [6.0, 7.0], [6.0, 7.0]]))
    y = ME.MinkowskiGlobalAvgPooling()(x)
    assert torch.allclose(y.F, torch.tensor([[3.5, 4.5], [6.0, 7.0]]))
    assert y.C.tolist() == [[0, 0, 0, 0], [1, 0, 0, 0]]


def test_sparse_collate(oracle_maps):
    from oracle import me_cpu as ME

    c = [np.array([[1, 2, 3], [4, 5, 6]], np.float32), torch.tensor([[7.0, 8.0, 9.0]])]
    f = [np.ones((2, 3), np.float32), torch.zeros(1, 3)]
    bc, bf = ME.utils.sparse_collate(c, f, dtype=torch.float32)
    assert bc.dtype == torch.float32 and bc.tolist() == [[0, 1, 2, 3], [0, 4, 5, 6], [1, 7, 8, 9]]
    assert bf.shape == (3, 3)
