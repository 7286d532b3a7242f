"""Hand-derived known answers for the RoIAlign restatement (oracle/roi_align_ref.c).
torchvision is absent from the image and the reference holds no vector for this op:
PARITY UNPINNED -- these cases follow from the published algorithm (aligned=True, sampling_ratio=-1)."""
import numpy as np
import torch


def test_full_frame_box_is_identity(oracle):
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, 5, 14, 14, generator=g)
    rois = torch.tensor([[0.0, 0.0, 224.0, 224.0], [0.0, 0.0, 224.0, 224.0]])
    out = oracle.roi_align(feat, rois, torch.tensor([0, 1], dtype=torch.int32), (14, 14), 14 / 224)
    # x1s=-0.5, bin=1, grid=ceil(14/14)=1, sample = -0.5 + p + 0.5 = p  -> exact resample at integer cells
    assert torch.equal(out, feat)
    grid, nbr = oracle.roi_align_indices(rois, 14, 14, (14, 14), 14 / 224)
    assert (grid == 1).all()
    yy, xx = np.meshgrid(np.arange(14), np.arange(14), indexing="ij")
    assert (nbr[0, :, :, 0] == yy).all() and (nbr[0, :, :, 1] == xx).all()
    # last row/col clamps high neighbour to H-1
    assert nbr[0, 13, 0, 2] == 13 and nbr[0, 0, 13, 3] == 13 and nbr[0, 0, 0, 2] == 1


def test_zero_box_gives_zeros_and_empty_grid(oracle):
    feat = torch.ones(1, 3, 14, 14)
    rois = torch.zeros(1, 4)
    out = oracle.roi_align(feat, rois, torch.zeros(1, dtype=torch.int32), (14, 14), 14 / 224)
    assert float(out.abs().max()) == 0.0
    grid, nbr = oracle.roi_align_indices(rois, 14, 14, (14, 14), 14 / 224)
    assert (grid == 0).all() and (nbr == -1).all()


def test_single_cell_box_bilinear(oracle):
    # box covering exactly feature cell (y=3, x=5): pixels [80,96)x[48,64) at stride 16
    feat = torch.arange(14 * 14, dtype=torch.float32).reshape(1, 1, 14, 14)   # f[y,x] = 14y + x (linear)
    rois = torch.tensor([[80.0, 48.0, 96.0, 64.0]])
    out = oracle.roi_align(feat, rois, torch.zeros(1, dtype=torch.int32), (2, 2), 1 / 16)
    # x1s=4.5, roi_w=1, bin=.5, grid=1 -> samples x = 4.75, 5.25 ; y = 2.75, 3.25 ; f is linear so
    # bilinear interpolation is exact
    exp = torch.tensor([[14 * 2.75 + 4.75, 14 * 2.75 + 5.25], [14 * 3.25 + 4.75, 14 * 3.25 + 5.25]])
    assert torch.allclose(out[0, 0], exp, atol=1e-5)


def test_adaptive_grid_and_backward_adjoint(oracle):
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(2, 3, 9, 9, generator=g, requires_grad=True)
    rois = torch.tensor([[3.0, 5.0, 70.0, 60.0], [-4.0, -4.0, 20.0, 30.0], [10.0, 10.0, 150.0, 150.0]])
    img = torch.tensor([0, 1, 1], dtype=torch.int32)
    grid, _ = oracle.roi_align_indices(rois, 9, 9, (3, 3), 1 / 8)
    assert grid.tolist() == [[3, 3], [2, 1], [6, 6]]       # ceil(roi_h/3), ceil(roi_w/3) in feature cells
    out = oracle.roi_align(feat, rois, img, (3, 3), 1 / 8)
    ct = torch.randn(out.shape, generator=g)
    (out * ct).sum().backward()
    # <A f, c> == <f, A^T c> for a linear operator; also check linearity of the forward
    lhs = float((out.detach() * ct).sum())
    rhs = float((feat.detach() * feat.grad).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
