"""Frame front-end: the oracle restatement against the third-party pieces this
image does have (numpy.dot, Pillow's BILINEAR resize -- what scipy.misc.imresize called), against SciPy's documented
bytescale examples, and against the golden planes; then the host C implementation against the oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import frame_frontend as ff  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "frontend.npz")


def golden_frames():
    g = np.load(GOLDEN)
    return {k[4:]: (g[k], g["plane_" + k[4:]]) for k in g.files if k.startswith("rgb_")}


def sample_images(rng, h, w):
    yield rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    yield (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)
    img = np.zeros((h, w), np.uint8)
    img[h // 3: h // 2, w // 4: w // 2] = 255
    img[::7] = 77
    yield img


@pytest.mark.parametrize("shape", [(210, 160, 84, 84), (250, 160, 84, 84), (84, 84, 84, 84), (100, 60, 84, 84),
                                   (40, 50, 84, 84), (210, 160, 42, 42), (211, 157, 84, 84), (210, 84, 84, 84)])
def test_resample_restatement_equals_pillow(shape):
    Image = pytest.importorskip("PIL.Image")
    h, w, oh, ow = shape
    rng = np.random.default_rng(h * 1000 + w)
    for img in sample_images(rng, h, w):
        want = np.asarray(Image.fromarray(img, "L").resize((ow, oh), resample=Image.BILINEAR))
        assert np.array_equal(ff.pil_bilinear_u8(img, oh, ow), want)


def test_gray_product_equals_numpy_dot_on_frames():
    rng = np.random.default_rng(7)
    for shape in [(210, 160, 3), (250, 160, 3), (33, 17, 3), (210, 160, 4)]:
        rgb = rng.integers(0, 256, size=shape, dtype=np.uint8)
        assert np.array_equal(ff.rgb2gray(rgb), np.dot(rgb[..., :3], [0.299, 0.587, 0.114]))      # Environment.py:54


def test_bytescale_documented_examples():
    """The examples of scipy.misc.bytescale's docstring (SciPy <= 1.2)."""
    img = np.array([[91.06794177, 3.39058326, 84.4221549], [73.88003259, 80.91433048, 4.88878881],
                    [51.53875334, 34.45808177, 27.5873488]])
    assert ff.bytescale(img).tolist() == [[255, 0, 236], [205, 225, 4], [140, 90, 70]]
    assert ff.bytescale(img, high=200, low=100).tolist() == [[200, 100, 192], [180, 188, 102], [155, 135, 128]]
    assert ff.bytescale(np.full((3, 3), 5.0)).tolist() == [[0] * 3] * 3          # max == min: scale 255 / 1
    u8 = np.arange(6, dtype=np.uint8).reshape(2, 3)
    assert ff.bytescale(u8) is u8


def test_oracle_reproduces_golden_planes():
    for name, (rgb, plane) in golden_frames().items():
        got = ff.preprocess_u8(rgb)
        assert got.dtype == np.uint8 and got.shape == (84, 84)
        assert np.array_equal(got, plane), name
        f = ff.preprocess(rgb)
        assert f.dtype == np.float32 and f.min() >= -1.0 and f.max() <= 0.9921875
        assert np.array_equal(f, plane.astype(np.float32) / 128.0 - 1.0)                              # Environment.py:60


def test_frame_queue_is_the_reference_fifo():
    q = ff.FrameQueue()
    planes = [np.full((84, 84), i, np.uint8) for i in range(6)]
    for i, pl in enumerate(planes):
        q.push(pl)
        s = q.state_u8()
        if i < 3:
            assert s is None                                   # Environment.py:64-65
        else:
            assert s.shape == (84, 84, 4) and s[0, 0].tolist() == [i - 3, i - 2, i - 1, i]     # oldest first
    q.clear()
    q.push(planes[0])
    assert q.state_u8() is None


def test_host_preprocess_equals_oracle():
    import ga3c_amd  # noqa: F401
    import _native as nat
    lib = nat.host_lib()
    rng = np.random.default_rng(3)
    cases = list(golden_frames().values())
    cases += [(rng.integers(0, 256, size=s, dtype=np.uint8), None) for s in [(210, 160, 3), (210, 160, 4), (96, 96, 3),
                                                                              (84, 84, 3), (60, 200, 3)]]
    for rgb, plane in cases:
        rgb = np.ascontiguousarray(rgb)
        out = np.zeros((84, 84), np.uint8)
        h, w, c = rgb.shape
        nat.check_host(lib.ga3c_frame_preprocess(nat.ptr(rgb, nat.u8p), h, w, c, 84, 84, nat.ptr(out, nat.u8p)))
        want = plane if plane is not None else ff.preprocess_u8(rgb)
        assert np.array_equal(out, want)
    assert lib.ga3c_frame_preprocess(nat.ptr(rgb, nat.u8p), 10, 10, 2, 84, 84, nat.ptr(out, nat.u8p)) < 0
