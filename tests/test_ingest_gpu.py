"""Raw-frame ingest on the device: nalo_undist_set + nalo_frame_upload_raw = PhotometricUndistorter::processFrame
(util/Undistort.cpp:214-251) + the remap of Undistort::undistort (:435-530) + the INTER_NEAREST mask / colour resizes of undistort_mask
(:385-433) fused in front of makeImages, against the CPU oracle (orc_undist.c) followed by the oracle's makeImages. Every product and the bilinear sum
are fp32 operations in the reference's order (the kernel is built without FMA contraction): level-0 irradiance and the whole pyramid are BIT-EXACT."""
import numpy as np
import pytest

import orc
from nalo_slam_amd import binding

pytestmark = pytest.mark.gpu


def radial_remap(w, h, wo, ho, k1=-0.12):
    """a plausible rectification table: radial distortion around the image centre; pixels whose four taps do not fit the original image are -1"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    nx, ny = (x - w / 2) / (0.6 * w), (y - h / 2) / (0.6 * w)
    r2 = nx * nx + ny * ny
    rx = ((nx * (1 + k1 * r2)) * 0.6 * w + wo / 2).astype(np.float32)
    ry = ((ny * (1 + k1 * r2)) * 0.6 * w + ho / 2).astype(np.float32)
    bad = ~((rx >= 0) & (ry >= 0) & (rx.astype(int) + 1 < wo) & (ry.astype(int) + 1 < ho))
    rx[bad] = -1; ry[bad] = -1
    rx[0, :7] = -1; ry[0, :7] = -1                       # a few explicit outside entries
    return rx, ry


@pytest.mark.parametrize("dtype,photometric,remap", [(np.uint8, 2, True), (np.uint8, 1, True), (np.uint8, 0, True), (np.uint16, 2, True), (np.uint8, 2, False)],
                         ids=["u8_full", "u8_response_only", "u8_none", "u16_full", "passthrough"])
def test_raw_ingest_bit_exact(dtype, photometric, remap):
    rng = np.random.RandomState(11)
    w, h = 640, 480
    wo, ho = (672, 512) if remap else (w, h)
    depth = 256 if dtype == np.uint8 else 65536
    raw = rng.randint(0, depth, (ho, wo)).astype(dtype)
    G = np.cumsum(rng.rand(depth) + 0.05).astype(np.float32)
    G = (255.0 * (G - G[0]) / (G[-1] - G[0])).astype(np.float32)
    yy, xx = np.mgrid[0:ho, 0:wo]
    vmap = (1.0 - 0.4 * (((xx - wo / 2) / wo) ** 2 + ((yy - ho / 2) / ho) ** 2)).astype(np.float32)
    vinv = (np.float32(1.0) / vmap).astype(np.float32)
    rx, ry = radial_remap(w, h, wo, ho) if remap else (None, None)
    K = (0.52 * w, 0.52 * w, (w - 1) / 2.0, (h - 1) / 2.0)
    c = binding.Context(w, h, K, n_slots=2)
    c.undist_set(wo, ho, G, vinv if photometric == 2 else None, photometric, rx, ry)
    for exposure in (0.02, 0.0):                         # exposure <= 0 switches the photometric part off for that frame (factor * raw)
        c.frame_upload_raw(0, raw, exposure=exposure, factor=0.5)
        ph = photometric if exposure > 0 else 0
        img = orc.undistort(raw, G, vinv if ph == 2 else None, ph, 0.5, rx, ry, w, h)
        dI_o, ab_o = orc.make_images(img, c.levels)
        L = orc.lib()
        for lvl in range(c.levels):
            o0, o1 = L.orc_pyr_offset(w, h, lvl), L.orc_pyr_offset(w, h, lvl + 1)
            dI_g, ab_g = c.frame_download(0, lvl)
            assert np.array_equal(dI_g, dI_o[o0:o1]), (exposure, lvl)
            assert np.array_equal(ab_g, ab_o[o0:o1])
    if remap:
        assert (img == 0).sum() >= 7 and (img != 0).mean() > 0.9
    c.close()


def test_ingest_rejects_bad_tables():
    w, h = 320, 240
    c = binding.Context(w, h, (160.0, 160.0, 159.5, 119.5), n_slots=1)
    G = np.linspace(0, 255, 256).astype(np.float32)
    rx = np.full((h, w), 10.0, np.float32); ry = np.full((h, w), 10.0, np.float32)
    rx[5, 5] = 400.0                                     # taps outside a 330-wide original
    with pytest.raises(binding.NaloError):
        c.undist_set(330, 250, G, None, 1, rx, ry)
    with pytest.raises(binding.NaloError):
        c.undist_set(330, 250, G, None, 1, None, None)   # passthrough with a different size
    with pytest.raises(binding.NaloError):
        c.frame_upload_raw(0, np.zeros((250, 330), np.uint8))   # tables not set
    with pytest.raises(binding.NaloError):
        c.undist_set(330, 250, G[:100], None, 1, np.full((h, w), 3.0, np.float32), np.full((h, w), 3.0, np.float32))
    c.close()


def test_ingest_mask_and_colour_feed_the_dense_map():
    """resizeMask / resizeColor on the device: a frame ingested with original-size mask + colour produces the same makeMap output as a frame uploaded with the
    host-resized (oracle) mask and colour"""
    rng = np.random.RandomState(2)
    w, h, wo, ho = 640, 480, 660, 496
    raw = rng.randint(0, 256, (ho, wo)).astype(np.uint8)
    rx, ry = radial_remap(w, h, wo, ho, k1=-0.05)
    mask_o = np.zeros((ho, wo), np.uint8); mask_o[300:470, 80:600] = 7; mask_o[100:200, 50:300] = 3
    bgr_o = rng.randint(0, 256, (ho, wo, 3)).astype(np.uint8)
    K = (0.52 * w, 0.52 * w, (w - 1) / 2.0, (h - 1) / 2.0)
    G = np.linspace(0, 255, 256).astype(np.float32)
    c = binding.Context(w, h, K, n_slots=2)
    c.undist_set(wo, ho, G, None, 1, rx, ry)
    c.frame_upload_raw(0, raw, mask_org=mask_o, bgr_org=bgr_o)
    img = orc.undistort(raw, G, None, 1, 1.0, rx, ry, w, h)
    c.frame_upload(1, img, mask=orc.resize_nearest_u8(mask_o, w, h).astype(np.float32), bgr=orc.resize_nearest_u8(bgr_o, w, h))
    plane = np.array([0.0, -1.0, 0.05, 1.6], np.float32)
    T = np.eye(4)[:3]
    a = c.dense_make_map(0, plane, 7.0, T, cap=200000)
    b = c.dense_make_map(1, plane, 7.0, T, cap=200000)
    assert a["n"] == b["n"] and a["n"] > 1000
    for k in ("u", "v", "idepth", "color", "bgr", "rect"):
        assert np.array_equal(a[k], b[k]), k
    c.close()


def test_raw_ingest_async_equals_sync():
    """nalo_frame_upload_raw_async (copy stream + event, several frames in flight, each slot with its own raw buffer) produces the pyramids of the synchronous
    entry point, whatever the order the frames are waited for"""
    rng = np.random.RandomState(5)
    w, h, wo, ho = 640, 480, 672, 512
    G = np.cumsum(rng.rand(256) + 0.05).astype(np.float32); G = (255.0 * (G - G[0]) / (G[-1] - G[0])).astype(np.float32)
    rx, ry = radial_remap(w, h, wo, ho)
    c = binding.Context(w, h, (0.52 * w, 0.52 * w, (w - 1) / 2.0, (h - 1) / 2.0), n_slots=4)
    c.undist_set(wo, ho, G, None, 1, rx, ry)
    raws = [rng.randint(0, 256, (ho, wo)).astype(np.uint8) for _ in range(3)]
    ref = []
    for r in raws:
        c.frame_upload_raw(3, r, exposure=0.01)
        ref.append([c.frame_download(3, l)[0].copy() for l in range(c.levels)])
    pinned = []
    for k, r in enumerate(raws):
        a = c.pinned_array((ho, wo), np.uint8); a[:] = r; pinned.append(a)
    for rep in range(2):                                 # twice: the second round overwrites slots that kernels may still be reading
        for k in range(3):
            c.frame_upload_raw_async(k, pinned[k], exposure=0.01)
        for k in (2, 0, 1):
            c.frame_wait(k)
        c.sync()
        for k in range(3):
            for l in range(c.levels):
                assert np.array_equal(c.frame_download(k, l)[0], ref[k][l]), (rep, k, l)
    c.close()


def test_camera_file_to_ingest_end_to_end(tmp_path):
    """camera.txt -> nalo_io_make_rectification -> nalo_undist_set -> nalo_frame_upload_raw, on a landscape sensor with an explicit output K whose rectified
    view reaches below the original image (the reference's `iy < wOrg-1` slip, util/Undistort.cpp:980, used to produce a table the product's own
    ingest rejected). The product's table must be accepted as it is, pixels outside the original image are 0, the rest is bit-exact against the oracle."""
    import ctypes as C
    from test_io_cpu import CameraFile, _rectify
    lib = C.CDLL(binding.lib_path())
    (tmp_path / "below.txt").write_text("Pinhole 400 400 319.5 239.5 0\n640 480\n0.5 0.5 0.5 0.1 0\n640 480\n")
    cf = CameraFile(); assert lib.nalo_io_read_camera(str(tmp_path / "below.txt").encode(), C.byref(cf)) == 0
    rc, K, rx, ry, pt = _rectify(lib, cf)
    assert rc == 0 and pt == 0
    w, h, wo, ho = cf.w, cf.h, cf.w_org, cf.h_org
    rng = np.random.RandomState(3)
    raw = rng.randint(0, 256, (ho, wo)).astype(np.uint8)
    G = np.linspace(0, 255, 256).astype(np.float32)
    c = binding.Context(w, h, tuple(K), n_slots=1)
    c.undist_set(wo, ho, G, None, 1, rx, ry)                                  # raised "remap entry outside the original image" before the fix
    c.frame_upload_raw(0, raw, exposure=1.0, factor=1.0)
    img = orc.undistort(raw, G, None, 1, 1.0, rx, ry, w, h)
    dI_o, _ = orc.make_images(img, c.levels)
    dI_g, _ = c.frame_download(0, 0)
    assert np.array_equal(dI_g, dI_o[:w * h])
    outside = (rx < 0).reshape(-1)
    assert outside.sum() > 1000 and (dI_g[outside, 0] == 0).all() and (dI_g[~outside, 0] != 0).mean() > 0.98
    c.close()
