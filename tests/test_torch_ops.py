"""``torch.library`` registration of the HIP ops (BASELINE north_star: "PyTorch-ROCm custom ops over a thin C ABI")."""
import pytest
import torch


def test_ops_registered_and_gpu_only():
    import celldetection_amd.torch_ops as t
    for name in ('nms', 'fouriers2contours', 'local_refinement', 'remove_border_contours', 'box_votes'):
        assert hasattr(torch.ops.cpn_hip, name), name
    assert 'iou_threshold' in str(torch.ops.cpn_hip.nms.default._schema)
    with pytest.raises(NotImplementedError):  # no CPU kernel: the product path never falls back
        torch.ops.cpn_hip.nms(torch.rand(4, 4), torch.rand(4), .5)
    try:
        import torchvision  # noqa: F401
        has_tv = True
    except ImportError:
        has_tv = False
    if not has_tv:
        assert t.install_torchvision_nms() in (True, False)
        assert 'iou_threshold' in str(torch.ops.torchvision.nms.default._schema)


@pytest.mark.gpu
def test_torch_ops_equal_python_ops():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import celldetection_amd.torch_ops as t
    from celldetection_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(500, 2, generator=g) * 80
    boxes = torch.cat((xy, xy + torch.rand(500, 2, generator=g) * 20), 1).to(dev)
    scores = torch.rand(500, generator=g).to(dev)
    exp = ops.nms(boxes, scores, .3)
    assert torch.equal(torch.ops.cpn_hip.nms(boxes, scores, .3), exp)
    try:
        import torchvision  # noqa: F401
    except ImportError:
        t.install_torchvision_nms()
        assert torch.equal(torch.ops.torchvision.nms(boxes, scores, .3), exp)  # the reference's call form
    f, loc = torch.randn(9, 5, 4, generator=g).to(dev), (torch.rand(9, 2, generator=g) * 50).to(dev)
    assert torch.equal(torch.ops.cpn_hip.fouriers2contours(f, loc, 32), ops.fouriers2contours(f, loc, 32)[0])
    votes = torch.ops.cpn_hip.box_votes(boxes, .3)
    keep, v = ops.filter_by_box_voting(boxes, .3, 1.2, return_votes=True)
    assert torch.equal(votes[votes >= 1.2], v)
    con = (torch.rand(20, 16, 2, generator=g) * 60).to(dev)
    a = torch.ops.cpn_hip.remove_border_contours(con, 48, 64, 4., 15, -3., 5.)
    b = ops.remove_border_contours(con, (48, 64), 4, offsets=torch.tensor([-3., 5.]))
    assert torch.equal(a, b)
