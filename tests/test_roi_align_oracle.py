"""Pins oracle/roi_align_oracle.py: (1) the reference's own fixture (common/lib/roi_pooling/debug.py:10-11)
with the answer its compiled CPU kernel gives; (2) backward == adjoint of forward."""
import numpy as np
import pytest

from oracle import roi_align_oracle as R


def test_forward_known_answer_debug_fixture():
    feature = np.arange(81 * 2 * 3, dtype=np.float32).reshape(2, 3, 9, 9)
    rois = np.array([[0, 0, 0, 9, 9], [1, 0, 0, 9, 9], [1, 0, 0, 7, 7]], dtype=np.float32)
    out = R.roi_align_forward(feature, rois, 1.0, 3, 3, 1)
    np.testing.assert_allclose(out[0, 0], [[15, 18, 21], [42, 45, 48], [69, 72, 75]], rtol=0, atol=1e-5)
    # channel c of image b is the same ramp shifted by (b*3+c)*81
    np.testing.assert_allclose(out[1, 2], out[0, 0] + (1 * 3 + 2) * 81, atol=1e-4)
    assert out.shape == (3, 3, 3, 3)


def test_backward_is_adjoint_of_forward():
    rng = np.random.RandomState(0)
    x = rng.randn(2, 4, 11, 13)
    rois = np.array([[0, 3.1, 2.2, 90.5, 70.0], [1, -20.0, -8.0, 50.0, 40.0], [1, 100.0, 60.0, 400.0, 300.0],
                     [0, 10.0, 10.0, 10.5, 10.2]], dtype=np.float32)
    for sr in (1, 2, 0):
        y = R.roi_align_forward(x, rois, 1.0 / 8, 5, 4, sr)
        dy = rng.randn(*y.shape)
        dx = R.roi_align_backward(dy, rois, 1.0 / 8, 5, 4, 2, 4, 11, 13, sr)
        assert abs((dy * y).sum() - (dx * x).sum()) < 1e-8 * max(1.0, abs((dy * y).sum()))


@pytest.mark.parametrize("sr", [0, 1, 2, 3])
def test_forward_matches_reference_binary(sr):
    """oracle/roi_align_oracle.py against the REFERENCE's own compiled CPU kernel (oracle/_ref, built by oracle/build_ref.sh from
    /root/reference/common/lib/roi_pooling/cpu/ROIAlign_cpu.cpp) on random features and RoIs incl. degenerate / out-of-image ones."""
    from oracle import ref_roi_align as REF
    if not REF.available():
        pytest.skip("oracle/_ref/libroi_align_ref.so not built (needs /root/reference; run oracle/build_ref.sh)")
    rng = np.random.RandomState(100 + sr)
    B, C, H, W = 2, 5, 11, 17
    feat = rng.randn(B, C, H, W).astype(np.float32)
    rois = []
    for k in range(12):
        x1, y1 = rng.uniform(-40, 16 * W), rng.uniform(-40, 16 * H)
        w, h = rng.uniform(0, 200), rng.uniform(0, 150)
        rois.append([k % B, x1, y1, x1 + w, y1 + h])
    rois += [[0, 0, 0, 16 * W - 1, 16 * H - 1], [1, 50.0, 40.0, 50.0, 40.0], [1, 300.0, 200.0, 290.0, 190.0], [0, -100.0, -100.0, -50.0, -60.0]]
    rois = np.array(rois, dtype=np.float32)
    want = REF.roi_align_forward(feat, rois, 1.0 / 16, 7, 5, sr)
    got = R.roi_align_forward(feat, rois, 1.0 / 16, 7, 5, sr)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6), float(np.abs(got - want).max())
