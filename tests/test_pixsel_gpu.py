"""GPU parity of the candidate-pixel selection (PixelSelector::select / makeMaps / makeMaps_lidar + FusedWithMask) through the
C-ABI vs the scalar CPU oracle on the same frame, the same randomPattern and the same rand() draws. Integer / index work: exact equality of the
status maps and counts. The frame carries a patch of vertical stripes (dy == 0 exactly): cells there whose drawn direction is (0,1) see a zero
projected gradient, the case in which the count -> scan -> select scheme has to repeat with corrected flags."""
import numpy as np
import pytest

import orc
from nalo_slam_amd import binding, synth

pytestmark = pytest.mark.gpu


def striped(img):
    out = img.copy()
    h, w = out.shape
    xs = np.arange(w // 3, w // 3 + 240)
    out[h // 4:h // 4 + 120, xs] = (100 + 60 * ((xs // 3) % 2)).astype(np.float32)[None, :]
    return out


@pytest.fixture(scope="module", params=[(640, 480), (616, 376)])
def frame(request):
    w, h = request.param
    win = synth.make_window(w=w, h=h, W=2, P=50, seed=21, n_extra=0)
    img = striped(win.images[1])
    mask = np.zeros((h, w), np.float32)
    rng = np.random.RandomState(5)
    mask[h // 3:, :] = rng.randint(0, 200, size=(h - h // 3, w)).astype(np.float32)          # 0 = no lidar return
    mask[:, : w // 5] = 0
    c = binding.Context(w, h, win.K, n_slots=2)
    c.frame_upload(0, img, mask=mask)
    rp, draws = orc.pixsel_libc_tables(w * h)
    c.pixsel_set_random(rp, draws)
    dI, ab = orc.make_images(img, 3)
    o1, o2 = w * h, w * h + (w // 2) * (h // 2)
    imgs = (dI[:o1], ab[:o1], ab[o1:o2], ab[o2:])
    yield c, w, h, imgs, rp, draws, mask
    c.close()


@pytest.mark.parametrize("pot", [1, 2, 3, 5, 7, 9, 16])
def test_select_exact(frame, pot):
    c, w, h, imgs, rp, draws, mask = frame
    _, sm = c.pixsel_make_hists(0)
    _, sm_o = orc.pixsel_make_hists(imgs[1], w, h)
    assert np.array_equal(sm, sm_o)
    for thf in (1.0, 2.0):
        got, n = c.pixsel_select(0, pot, thf)
        ref, n_o = orc.pixsel_select(*imgs, w, h, sm_o, rp, pot, thf)
        assert np.array_equal(n, n_o) and np.array_equal(got, ref)
        assert n[0] == (got == 1).sum() and n[1] == (got == 2).sum() and n[2] == (got == 4).sum() and n[0] > 50
        idx, st = c.pixsel_get_selected()
        nz = np.flatnonzero(got)
        assert np.array_equal(idx, nz) and np.array_equal(st, got.reshape(-1)[nz].astype(np.uint8))


def test_zero_projection_cells_are_exercised(frame):
    """the stripe patch really produces cells that pass the threshold but select nothing (the speculative count is wrong there)"""
    c, w, h, imgs, rp, draws, mask = frame
    _, sm = orc.pixsel_make_hists(imgs[1], w, h)
    ref, n = orc.pixsel_select(*imgs, w, h, sm, rp, 3, 1.0)
    ag0 = imgs[1].reshape(h, w)
    th = np.repeat(np.repeat(sm.reshape(h // 32, w // 32), 32, 0), 32, 1)
    cells = 0
    for y in range(h // 4 + 6, h // 4 + 110, 3):
        for x in range(w // 3 + 6, w // 3 + 230, 3):
            if y + 3 <= th.shape[0] and x + 3 <= th.shape[1] and (ag0[y:y + 3, x:x + 3] > th[y:y + 3, x:x + 3]).any() and not (ref[y:y + 3, x:x + 3] == 1).any():
                cells += 1
    assert cells > 0


@pytest.mark.parametrize("density,potential", [(1500.0, 3), (600.0, 3), (20000.0, 3), (400.0, 1)])
def test_make_maps_exact(frame, density, potential):
    c, w, h, imgs, rp, draws, mask = frame
    got, num, pot = c.pixsel_make_maps(0, density, potential, recursionsLeft=1, thFactor=1.0)
    ref, num_o, pot_o = orc.pixsel_make_maps(*imgs, w, h, rp, density, potential, 1, 1.0)
    assert (num, pot) == (num_o, pot_o) and np.array_equal(got, ref)
    assert num == (got != 0).sum()
    got2, num2, pot2 = c.pixsel_make_maps(0, density, potential, recursionsLeft=0, thFactor=2.0)      # the initializer's call (CoarseInitializer.cpp:811 uses thFactor 2)
    ref2, num2_o, pot2_o = orc.pixsel_make_maps(*imgs, w, h, rp, density, potential, 0, 2.0)
    assert (num2, pot2) == (num2_o, pot2_o) and np.array_equal(got2, ref2)


def test_make_maps_lidar_exact(frame):
    c, w, h, imgs, rp, draws, mask = frame
    got, num = c.pixsel_make_maps_lidar(0, 3, 1.0)
    ref, num_o = orc.pixsel_make_maps_lidar(*imgs, w, h, rp, mask, draws, 3, 1.0)
    assert num == num_o and np.array_equal(got, ref)
    assert num == (got == 1).sum() + (got == 2).sum()
    _, sm = orc.pixsel_make_hists(imgs[1], w, h)
    plain, _ = orc.pixsel_select(*imgs, w, h, sm, rp, 3, 1.0)
    assert ((plain == 0) & (got == 1)).sum() > 0 and ((plain == 1) & (got == 2)).sum() > 0 and ((plain == 2) & (got == 1)).sum() >= 0


def test_errors(frame):
    c, w, h, *_ = frame
    with pytest.raises(RuntimeError):
        c.pixsel_select(1, 3)                                   # empty slot
    c.frame_upload(1, np.full((h, w), 50, np.float32))
    with pytest.raises(RuntimeError):
        c.pixsel_select(1, 3)                                   # makeHists has not run on this frame
    with pytest.raises(RuntimeError):
        c.pixsel_make_maps_lidar(1, 3)                          # no mask
    m, num, pot = c.pixsel_make_maps(1, 1500.0, 3)              # flat frame: nothing to select
    assert num == 0 and not m.any()
