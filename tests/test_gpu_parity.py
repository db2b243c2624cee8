"""GPU parity: the HIP path (through the C ABI, include/orb_hip.h) against the
CPU oracle on the same seeded inputs.  Integer, byte and index results must be
bit-exact; the orientation angle is also compared bit-exactly (the north-star
tolerance for it is 1e-4 degrees, asserted as well so a regression reports
which of the two bars it crossed).
"""
import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, splitmix64, synth_descriptors, synth_queries, synth_scene

pytestmark = pytest.mark.gpu

PRESETS = {"fork": 0x1F, "canonical": 0}


@pytest.fixture(scope="module")
def hip():
    from orb_slam2_amd import binding

    binding.lib()  # fails loudly if the HIP library is missing
    return binding


def _ctx(hip, preset, nfeatures=1000, nlevels=8, w=640, h=480, batch=1):
    return hip.Context(nfeatures, 1.2, nlevels, 20, 7, w, h, batch, 0, PRESETS[preset])


def assert_kps_equal(got, ref, what=""):
    assert len(got) == len(ref), "%s: %d keypoints, oracle %d" % (what, len(got), len(ref))
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        np.testing.assert_array_equal(got[f], ref[f], err_msg="%s field %s" % (what, f))
    d = np.abs(got["angle"].astype(np.float64) - ref["angle"].astype(np.float64))
    d = np.minimum(d, 360 - d)
    assert (d <= 1e-4).all(), "%s: angle differs by up to %g degrees" % (what, d.max())
    np.testing.assert_array_equal(got["angle"], ref["angle"], err_msg="%s angle not bit-exact" % what)


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_tables(hip, oracle, preset):
    for nf in (1000, 2000, 4000):
        c = _ctx(hip, preset, nf)
        e = oracle.Extractor(nf, 1.2, 8, 20, 7, PRESETS[preset])
        tg, to = c.tables(), e.tables()
        for k in tg:
            np.testing.assert_array_equal(tg[k], to[k], err_msg=k)
        c.close()


@pytest.mark.parametrize("area", [True, False])
def test_resize(hip, oracle, area):
    c = _ctx(hip, "fork")
    for (w, h, dw, dh, seed) in [(640, 480, 533, 400, 1), (533, 400, 444, 333, 2), (179, 134, 149, 112, 3),
                                 (1241, 376, 1034, 313, 4), (97, 61, 80, 50, 5), (64, 64, 40, 33, 6)]:
        img = synth_scene(w, h, seed)
        np.testing.assert_array_equal(c.resize(img, dw, dh, area), oracle.resize(img, dw, dh, area),
                                      err_msg="%dx%d->%dx%d area=%s" % (w, h, dw, dh, area))
    # strided input
    big = synth_scene(700, 300, 9)
    view = big[10:250, 20:620]
    np.testing.assert_array_equal(c.resize(np.ascontiguousarray(view), 500, 200, area), oracle.resize(view, 500, 200, area))
    c.close()


@pytest.mark.parametrize("legacy", [False, True])
def test_gaussian_blur(hip, oracle, legacy):
    c = hip.Context(1000, 1.2, 8, 20, 7, 640, 480, 1, 0, 0x1F | (hip.ORB_Q_GAUSS_LEGACY if legacy else 0))
    for (w, h, seed) in [(640, 480, 1), (179, 134, 2), (65, 17, 3), (7, 7, 4), (5, 3, 5), (1, 9, 6), (130, 1, 7)]:
        img = synth_scene(w, h, seed)
        np.testing.assert_array_equal(c.gaussian_blur(img), oracle.gaussian(img, legacy), err_msg="%dx%d" % (w, h))
    sat = np.full((40, 40), 255, np.uint8)
    np.testing.assert_array_equal(c.gaussian_blur(sat), oracle.gaussian(sat, legacy))
    c.close()


CASES = [  # (w, h, nfeatures, nlevels, config id)
    (640, 480, 1000, 8, 1),
    (1241, 376, 2000, 8, 2),
    (752, 480, 2000, 8, 3),
    (160, 120, 300, 4, 1),
    (333, 517, 500, 6, 1),
]


@pytest.mark.parametrize("preset", ["fork", "canonical"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d" % (c[0], c[1]))
def test_extract_stages_and_end_to_end(hip, oracle, preset, case):
    w, h, nf, nl, cfg = case
    c = _ctx(hip, preset, nf, nl, w, h)
    e = oracle.Extractor(nf, 1.2, nl, 20, 7, PRESETS[preset])
    for f in range(2):
        img = synth_scene(w, h, frame_seed(cfg, f))
        kg, dg = c.extract(img)
        ko, do = e.extract(img)
        for l in range(nl):
            np.testing.assert_array_equal(c.pyramid_level(0, l), e.level(l), err_msg="pyramid level %d" % l)
        for l in range(nl):
            xs, ys, sc = c.fast_keys(0, l)
            ref = e.level_keys(l, after_quadtree=False)
            assert len(xs) == len(ref), "level %d: %d FAST keys, oracle %d" % (l, len(xs), len(ref))
            np.testing.assert_array_equal(xs, ref["x"].astype(np.int32), err_msg="FAST x level %d" % l)
            np.testing.assert_array_equal(ys, ref["y"].astype(np.int32), err_msg="FAST y level %d" % l)
            np.testing.assert_array_equal(sc, ref["response"].astype(np.int32), err_msg="FAST score level %d" % l)
        assert_kps_equal(kg, ko, "frame %d" % f)
        np.testing.assert_array_equal(dg, do, err_msg="descriptors frame %d" % f)
    c.close()


def _special_images(w, h):
    rng = splitmix64(0xFA57, w * h)
    noise = (rng & np.uint64(255)).astype(np.uint8).reshape(h, w)                 # corner-dense everywhere
    dots = np.full((h, w), 60, np.uint8)
    dots[::4, ::4] = 230                                                           # isolated maxima on a lattice
    flat = np.full((h, w), 128, np.uint8)                                          # nothing at all
    faint = (120 + (rng >> np.uint64(8)) % np.uint64(12)).astype(np.uint8).reshape(h, w)  # only minThFAST finds corners
    half = noise.copy()
    half[:, w // 2:] = 90                                                          # dense and empty cells side by side
    return {"noise": noise, "dots": dots, "flat": flat, "faint": faint, "half": half}


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_extract_special_images(hip, oracle, preset):
    """Images that drive the FAST kernel through its rare paths: cells with more corners than its
    corner list holds, cells with none, cells where only the minThFAST pass finds corners."""
    w, h = 320, 240
    imgs = _special_images(w, h)
    dense = oracle.fast(imgs["noise"], 7, nms=False)
    assert len(dense) > 0.3 * w * h, "the noise image is meant to be corner-dense (%d corners)" % len(dense)
    c = _ctx(hip, preset, 1000, 4, w, h)
    e = oracle.Extractor(1000, 1.2, 4, 20, 7, PRESETS[preset])
    for name, img in imgs.items():
        kg, dg = c.extract(img)
        ko, do = e.extract(img)
        for l in range(4):
            xs, ys, sc = c.fast_keys(0, l)
            ref = e.level_keys(l, after_quadtree=False)
            assert len(xs) == len(ref), "%s level %d: %d FAST keys, oracle %d" % (name, l, len(xs), len(ref))
            np.testing.assert_array_equal(xs, ref["x"].astype(np.int32), err_msg="%s FAST x level %d" % (name, l))
            np.testing.assert_array_equal(ys, ref["y"].astype(np.int32), err_msg="%s FAST y level %d" % (name, l))
            np.testing.assert_array_equal(sc, ref["response"].astype(np.int32), err_msg="%s FAST score level %d" % (name, l))
        assert_kps_equal(kg, ko, name)
        np.testing.assert_array_equal(dg, do, err_msg="descriptors " + name)
    c.close()


def test_extract_batch_equals_single(hip, oracle):
    n = 5
    c = _ctx(hip, "fork", batch=n)
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
    imgs = [synth_scene(640, 480, frame_seed(1, 10 + f)) for f in range(n)]
    res = c.extract_batch(imgs)
    for f in range(n):
        ko, do = e.extract(imgs[f])
        assert_kps_equal(res[f][0], ko, "batch frame %d" % f)
        np.testing.assert_array_equal(res[f][1], do)
        np.testing.assert_array_equal(c.pyramid_level(f, 3), e.level(3))
    # a smaller batch and a smaller frame on the same context
    res2 = c.extract_batch(imgs[:2])
    for f in range(2):
        np.testing.assert_array_equal(res2[f][1], res[f][1])
    small = synth_scene(320, 240, 77)
    ks, ds = c.extract(small)
    ko, do = e.extract(small)
    assert_kps_equal(ks, ko, "small frame on a large context")
    np.testing.assert_array_equal(ds, do)
    c.close()


@pytest.mark.parametrize("preset", ["fork", "canonical"])
@pytest.mark.parametrize("scale", [1.1, 1.2, 1.25, 1.3, 1.41, 1.5])
def test_batch_pyramid_at_other_scale_factors(hip, oracle, preset, scale):
    """Batches of eight frames or more build the pyramid with one launch per level, in one of two shapes: one
    wavefront per workgroup on 64 x 8 tiles (level ratios up to about 1.3, 4-byte-aligned rows) or four
    wavefronts on 64 x 32 tiles (larger ratios).  Every level of every frame against the oracle, and the
    extraction's outputs with it; widths that are and are not multiples of four."""
    for (w, h, nl) in ((640, 480, 5), (467, 355, 4)):
        n = 9
        c = hip.Context(800, scale, nl, 20, 7, w, h, n, 0, PRESETS[preset])
        e = oracle.Extractor(800, scale, nl, 20, 7, PRESETS[preset])
        imgs = [synth_scene(w, h, frame_seed(2, 40 + f)) for f in range(n)]
        res = c.extract_batch(imgs)
        for f in (0, 4, 8):
            ko, do = e.extract(imgs[f])
            for l in range(nl):
                np.testing.assert_array_equal(c.pyramid_level(f, l), e.level(l),
                                              err_msg="scale %g %dx%d frame %d level %d" % (scale, w, h, f, l))
            assert_kps_equal(res[f][0], ko, "scale %g frame %d" % (scale, f))
            np.testing.assert_array_equal(res[f][1], do)
        c.close()


def test_extract_batch_threaded_staging(hip, oracle):
    """Batches of eight frames or more are staged by several host threads, each handing slices to the DMA
    engine; strided rows take the row-by-row copy.  Every frame must come out as from a single call."""
    n, w, h = 19, 320, 240
    c = _ctx(hip, "fork", 500, 6, w, h, batch=n)
    e = oracle.Extractor(500, 1.2, 6, 20, 7, PRESETS["fork"])
    wide = np.stack([synth_scene(w + 24, h, frame_seed(2, 50 + f)) for f in range(n)])
    imgs = [wide[f][:, 7 : 7 + w] for f in range(n)]  # views: contiguous copies are made by the binding
    res = c.extract_batch(imgs)
    for f in range(n):
        ko, do = e.extract(np.ascontiguousarray(imgs[f]))
        assert_kps_equal(res[f][0], ko, "frame %d" % f)
        np.testing.assert_array_equal(res[f][1], do)
    c.close()


@pytest.mark.parametrize("parts,threads", [("3", "3"), ("2", "16"), ("1", "2")])
def test_extract_batch_slices_arrive_in_group_order(hip, oracle, monkeypatch, parts, threads):
    """A host batch is copied in slices of a few frames on one stream, the later frame groups' slices first, and
    those groups start on an event while the first group's frames are still being copied: several slices per
    group, more threads than slices, a single group.  Every frame must equal its one-frame call."""
    monkeypatch.setenv("ORBHIP_PARTS", parts)
    monkeypatch.setenv("ORBHIP_STAGE_THREADS", threads)
    n, w, h = 26, 640, 480
    c = _ctx(hip, "fork", batch=n)
    imgs = [synth_scene(w, h, frame_seed(5, f)) for f in range(n)]
    for rep in range(2):  # the second call reuses staging that still holds the first call's frames
        order = imgs if rep == 0 else imgs[::-1]
        res = c.extract_batch(order)
        for f in (0, 7, 8, 9, 15, 16, 24, 25):
            k1, d1 = c.extract(order[f])
            assert_kps_equal(res[f][0], k1, "frame %d" % f)
            np.testing.assert_array_equal(res[f][1], d1)
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
    ko, do = e.extract(imgs[0])
    assert_kps_equal(res[n - 1][0], ko, "last frame of the reversed batch")
    np.testing.assert_array_equal(res[n - 1][1], do)
    c.close()


def test_extract_with_plain_division(hip, oracle, monkeypatch):
    """The kernels divide by workgroup-uniform numbers through multipliers the host prepares, and fall back to
    ordinary division where a multiplier would not be exact; ORBHIP_PLAIN_DIVISION forces the fallback."""
    monkeypatch.setenv("ORBHIP_PLAIN_DIVISION", "1")
    n, w, h = 11, 400, 300
    c = _ctx(hip, "fork", 700, 7, w, h, batch=n)
    e = oracle.Extractor(700, 1.2, 7, 20, 7, PRESETS["fork"])
    imgs = [synth_scene(w, h, frame_seed(2, 80 + f)) for f in range(n)]
    res = c.extract_batch(imgs)
    for f in range(n):
        ko, do = e.extract(imgs[f])
        assert_kps_equal(res[f][0], ko, "frame %d" % f)
        np.testing.assert_array_equal(res[f][1], do)
    kg, dg = c.extract(imgs[3])
    ko, do = e.extract(imgs[3])
    assert_kps_equal(kg, ko, "single frame")
    np.testing.assert_array_equal(dg, do)
    c.close()


def test_extract_after_sparse_frames(hip, oracle):
    """The quadtree kernel sizes its LDS block from the key counts of earlier batches; a frame with
    far more keys than any before must come out the same (its levels run from global memory)."""
    w, h = 640, 480
    c = _ctx(hip, "fork")
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
    sparse = np.full((h, w), 100, np.uint8)
    sparse[100:140, 200:260] = 200  # one bright rectangle: a handful of corners
    rich = synth_scene(w, h, frame_seed(1, 40))
    for i, img in enumerate([sparse, sparse, sparse, rich, rich, sparse]):
        kg, dg = c.extract(img)
        ko, do = e.extract(img)
        assert_kps_equal(kg, ko, "frame %d" % i)
        np.testing.assert_array_equal(dg, do, err_msg="descriptors frame %d" % i)
    assert len(e.extract(sparse)[0]) < 50 and len(e.extract(rich)[0]) > 900
    c.close()


def test_repeated_batches_are_identical(hip):
    """The same batch over and over (frame groups on two streams, LDS-resident stages, the asynchronous
    key-count read-back that resizes the quadtree's LDS block): every repetition must be bit-identical."""
    import torch

    n, w, h = 40, 640, 480
    c = _ctx(hip, "fork", batch=n)
    frames = np.stack([synth_scene(w, h, frame_seed(1, 60 + f % 10)) for f in range(n)])
    d_imgs = torch.from_numpy(frames).cuda()
    cap = c.cap
    outs = [torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda"), torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda"),
            torch.zeros(n, dtype=torch.int32, device="cuda")]
    ref = None
    for step in range(300):
        for t in outs[:2]:
            t.fill_(0xAB)
        c.extract_batch_device(d_imgs, *outs)
        if step % 20 == 0 or step == 299:
            torch.cuda.synchronize()
            c.check_last_batch()
            mask = torch.arange(cap, device="cuda")[None, :] < outs[2][:, None]
            cur = [outs[0] * mask[:, :, None], outs[1] * mask[:, :, None], outs[2].clone()]
            if ref is None:
                ref = cur
            else:
                for a, b in zip(cur, ref):
                    assert torch.equal(a, b), "step %d differs from step 0" % step
    c.close()


def test_extract_device_entry_point(hip, oracle):
    import torch

    n, w, h = 3, 640, 480
    c = _ctx(hip, "fork", batch=n)
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
    imgs = np.stack([synth_scene(w, h, frame_seed(1, 20 + f)) for f in range(n)])
    # padded rows: stride 704
    buf = torch.zeros((n, h, 704), dtype=torch.uint8, device="cuda")
    buf[:, :, :w] = torch.from_numpy(imgs).cuda()
    d_imgs = buf[:, :, :w]
    cap = c.cap
    d_kps = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    c.extract_batch_device(d_imgs, d_kps, d_desc, d_cnt)
    torch.cuda.synchronize()
    c.check_last_batch()
    cnt = d_cnt.cpu().numpy()
    for f in range(n):
        ko, do = e.extract(imgs[f])
        kg = d_kps[f, : cnt[f]].cpu().numpy().view(hip.KP_DTYPE).reshape(-1)
        assert_kps_equal(kg, ko, "device frame %d" % f)
        np.testing.assert_array_equal(d_desc[f, : cnt[f]].cpu().numpy(), do)
    c.close()


def test_extract_unaligned_rows(hip, oracle):
    """Frames whose rows, first pixel or frame-to-frame distance are not multiples of four bytes (the library copies
    them into aligned rows once; the stages read whole dwords): an odd width stored without padding, a view one
    byte into a padded buffer, frames an odd number of bytes apart.  Every frame equals the oracle's."""
    import torch

    n, w, h = 3, 333, 251
    c = hip.Context(500, 1.2, 6, 20, 7, w, h, n, 0, PRESETS["fork"])
    e = oracle.Extractor(500, 1.2, 6, 20, 7, PRESETS["fork"])
    imgs = np.stack([synth_scene(w, h, frame_seed(6, f)) for f in range(n)])
    want = [e.extract(imgs[f]) for f in range(n)]
    cap = c.cap
    views = []
    views.append(torch.from_numpy(imgs).cuda())                       # stride 333
    buf = torch.zeros((n, h, 340), dtype=torch.uint8, device="cuda")   # aligned stride, first pixel at byte 1
    buf[:, :, 1 : w + 1] = torch.from_numpy(imgs).cuda()
    views.append(buf[:, :, 1 : w + 1])
    flat = torch.zeros(n * (h * 336 + 3), dtype=torch.uint8, device="cuda")  # aligned rows, frames 84339 bytes apart
    odd = torch.as_strided(flat, (n, h, w), (h * 336 + 3, 336, 1))
    odd.copy_(torch.from_numpy(imgs).cuda())
    views.append(odd)
    for vi, d_imgs in enumerate(views):
        d_kps = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
        d_desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
        c.extract_batch_device(d_imgs, d_kps, d_desc, d_cnt)
        torch.cuda.synchronize()
        c.check_last_batch()
        cnt = d_cnt.cpu().numpy()
        for f in range(n):
            ko, do = want[f]
            kg = d_kps[f, : cnt[f]].cpu().numpy().view(hip.KP_DTYPE).reshape(-1)
            assert_kps_equal(kg, ko, "view %d frame %d" % (vi, f))
            np.testing.assert_array_equal(d_desc[f, : cnt[f]].cpu().numpy(), do)
    assert len(want[0][0]) > 100
    c.close()


def test_extract_width_off_four_in_aligned_rows(hip, oracle):
    """Aligned rows (stride 640, frames 640 x 480 bytes apart) that hold 637, 638 or 639 pixels: the last dword of a
    row straddles the image width, the bytes beyond it belong to no frame (here: 255, which would show).  A batch of
    nine takes the batch kernels (whose staging patches such dwords byte by byte), one frame the few-frames path."""
    import torch

    n, h = 9, 480
    for w in (637, 638, 639):
        c = hip.Context(1000, 1.2, 8, 20, 7, w, h, n, 0, PRESETS["fork"])
        e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
        imgs = np.stack([synth_scene(w, h, frame_seed(9, 3 * w + f)) for f in range(n)])
        buf = torch.full((n, h, 640), 255, dtype=torch.uint8, device="cuda")
        buf[:, :, :w] = torch.from_numpy(imgs).cuda()
        d_imgs = buf[:, :, :w]
        cap = c.cap
        for nb in (n, 1):
            d_kps = torch.zeros((nb, cap, 28), dtype=torch.uint8, device="cuda")
            d_desc = torch.zeros((nb, cap, 32), dtype=torch.uint8, device="cuda")
            d_cnt = torch.zeros(nb, dtype=torch.int32, device="cuda")
            c.extract_batch_device(d_imgs[:nb], d_kps, d_desc, d_cnt)
            torch.cuda.synchronize()
            c.check_last_batch()
            cnt = d_cnt.cpu().numpy()
            for f in ((0, n // 2, n - 1) if nb == n else (0,)):
                ko, do = e.extract(imgs[f])
                kg = d_kps[f, : cnt[f]].cpu().numpy().view(hip.KP_DTYPE).reshape(-1)
                assert_kps_equal(kg, ko, "width %d, batch %d, frame %d" % (w, nb, f))
                np.testing.assert_array_equal(d_desc[f, : cnt[f]].cpu().numpy(), do)
                np.testing.assert_array_equal(c.pyramid_level(f, 1), e.level(1), err_msg="level 1, width %d" % w)
        c.close()


def test_empty_and_invalid_inputs(hip):
    c = _ctx(hip, "fork")
    k, d = c.extract(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and len(d) == 0
    flat = np.full((480, 640), 90, np.uint8)  # no corners at all
    k, d = c.extract(flat)
    assert len(k) == 0
    with pytest.raises(hip.OrbError):  # coarsest level smaller than one FAST cell
        c.extract(np.zeros((100, 100), np.uint8))
    with pytest.raises(hip.OrbError):  # larger than the context was sized for
        c.extract(np.zeros((481, 640), np.uint8))
    with pytest.raises(hip.OrbError):
        hip.Context(1000, 1.2, 8, 20, 0)  # minThFAST 0 is not covered by the single-score-map argument
    c.close()


def _pair(oracle, w, h, nf, seed, preset):
    e = oracle.Extractor(nf, 1.2, 8, 20, 7, PRESETS[preset])
    a = synth_scene(w, h, seed)
    b = synth_scene(w, h, seed, shift=(3, 2), noise_seed=seed + 1)
    k1, d1 = e.extract(a)
    k2, d2 = e.extract(b)
    return k1, d1, k2, d2


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_search_for_initialization(hip, oracle, preset):
    w, h = 752, 480
    c = _ctx(hip, preset, 2000, 8, w, h)
    for s in range(3):
        k1, d1, k2, d2 = _pair(oracle, w, h, 2000, frame_seed(3, s), preset)
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)  # Tracking.cc:302-304
        for (window, ratio, ori) in [(100, 0.9, True), (30, 0.9, True), (100, 0.6, False), (400, 0.95, True)]:
            ng, mg, pg = c.match_window(k1, d1, k2, d2, w, h, prev, window, ratio, ori)
            no, mo, po = oracle.search_for_initialization(k1, d1, k2, d2, w, h, prev, window, ratio, ori, PRESETS[preset])
            np.testing.assert_array_equal(mg, mo, err_msg="matches12 window=%d" % window)
            assert ng == no
            np.testing.assert_array_equal(pg, po)
        assert no > 50  # the pair really matches
    # edge cases: empty frames, query positions far outside the image
    empty_k = np.zeros(0, hip.KP_DTYPE)
    empty_d = np.zeros((0, 32), np.uint8)
    n, m, _ = c.match_window(k1, d1, empty_k, empty_d, w, h, prev)
    assert n == 0 and (m == -1).all()
    far = prev + 5000
    ng, mg, _ = c.match_window(k1, d1, k2, d2, w, h, far)
    no, mo, _ = oracle.search_for_initialization(k1, d1, k2, d2, w, h, far, 100, 0.9, True, PRESETS[preset])
    assert ng == no == 0
    c.close()


def _random_frame(hip, n, w, h, seed):
    """n level-0 keypoints at random integer positions with random angles and descriptors."""
    r = splitmix64(seed, n * 3)
    k = np.zeros(n, hip.KP_DTYPE)
    k["x"] = (r[:n] % np.uint64(w)).astype(np.float32)
    k["y"] = (r[n : 2 * n] % np.uint64(h)).astype(np.float32)
    k["angle"] = (r[2 * n :] % np.uint64(360)).astype(np.float32)
    k["size"], k["octave"], k["class_id"] = 31, 0, -1
    return k, synth_descriptors(n, seed + 1)


def test_matcher_wide_windows_use_the_candidate_pool(hip, oracle):
    """Windows that hold more candidates than a query's fixed list segment (256) put the rest into the shared
    pool; a call that overruns the pool is made again with a larger one (host entry point) or reports
    ORB_ERR_CAPACITY through orb_check_last_batch and succeeds once the pool has grown (asynchronous entry)."""
    import torch

    w, h = 752, 480
    k2, d2 = _random_frame(hip, 4000, w, h, 31)
    # (queries, window): overflow within the initial pool; then 3000 x ~3700 entries, beyond its 4 Mi
    for n1, window in ((300, 150), (700, 2000), (3000, 2000)):
        c = _ctx(hip, "fork", 2000, 8, w, h)
        k1, d1 = _random_frame(hip, n1, w, h, 77 + n1)
        m = len(d1[::3])  # a third of the queries have a near-duplicate close by, at the same orientation
        d1[::3] = d2[:m] ^ np.uint8(1)
        k1["angle"][::3] = k2["angle"][:m]
        k1["x"][::3] = np.clip(k2["x"][:m] + 3, 0, w - 1)
        k1["y"][::3] = np.clip(k2["y"][:m] - 2, 0, h - 1)
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
        ng, mg, pg = c.match_window(k1, d1, k2, d2, w, h, prev, window, 0.9, True)
        no, mo, po = oracle.search_for_initialization(k1, d1, k2, d2, w, h, prev, window, 0.9, True, PRESETS["fork"])
        assert ng == no and no > n1 // 6
        np.testing.assert_array_equal(mg, mo)
        np.testing.assert_array_equal(pg, po)
        if n1 == 3000:
            # asynchronous entry point on a fresh context: the first call overruns the pool and says so ...
            c2 = _ctx(hip, "fork", 2000, 8, w, h)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            d_k1, d_d1, d_k2, d_d2 = t(k1.view(np.uint8).reshape(1, n1, 28)), t(d1[None]), t(k2.view(np.uint8).reshape(1, 4000, 28)), t(d2[None])
            d_n1, d_n2 = t(np.array([n1], np.int32)), t(np.array([4000], np.int32))
            d_m12 = torch.zeros((1, n1), dtype=torch.int32, device="cuda")
            d_nm = torch.zeros(1, dtype=torch.int32, device="cuda")
            d_prev = t(prev[None])
            c2.match_window_batch_device(d_k1, d_d1, d_n1, d_k2, d_d2, d_n2, w, h, d_prev, d_m12, d_nm, window, 0.9, True)
            # ... the pair whose lists were cut carries a match count of -1, and a host-pointer call made on the
            # context before the report is collected neither takes the report for its own nor loses it
            torch.cuda.synchronize()
            assert int(d_nm.item()) == -1
            ks, ds = k1[:300].copy(), d1[:300].copy()
            ps = prev[:300].copy()
            ns, ms, _ = c2.match_window(ks, ds, k2, d2, w, h, ps, 150, 0.9, True)
            nso, mso, _ = oracle.search_for_initialization(ks, ds, k2, d2, w, h, ps, 150, 0.9, True, PRESETS["fork"])
            assert ns == nso
            np.testing.assert_array_equal(ms, mso)
            with pytest.raises(hip.OrbError, match="candidate pool"):
                c2.check_last_batch()
            # ... the next one runs with the pool that call asked for
            d_prev = t(prev[None])
            c2.match_window_batch_device(d_k1, d_d1, d_n1, d_k2, d_d2, d_n2, w, h, d_prev, d_m12, d_nm, window, 0.9, True)
            c2.check_last_batch()
            assert int(d_nm.item()) == no
            np.testing.assert_array_equal(d_m12[0].cpu().numpy(), mo)
            c2.close()
        c.close()


def test_window_search_under_contention(hip, oracle):
    """SearchForInitialization when many queries want the same keypoints (src/ORBmatcher.cc:63-64, 82-89): groups of
    queries that are near-copies of ONE keypoint of the other frame at growing or shrinking distances (every
    acceptance in a group hides the keypoint from the next query, or is stolen by it), exact ties, and chains in which
    a query's second best is taken by the query before it.  The kernel resolves 64 queries at a time and has to stop
    wherever an earlier query of the run disturbs a later one; the result must be the walk in order."""
    w, h = 752, 480
    rng = np.random.default_rng(11)
    for case in range(6):
        n2 = 1500
        k2, d2 = _random_frame(hip, n2, w, h, 900 + case)
        n1 = 1200
        k1, d1 = _random_frame(hip, n1, w, h, 950 + case)
        group = (2, 3, 8, 70, 5, 16)[case]          # queries per target keypoint (70: more than one run of 64)
        order = rng.permutation(n1)
        targets = rng.permutation(n2)
        for g in range(n1 // group):
            tgt = targets[g % n2]
            members = order[g * group : (g + 1) * group] if case % 2 == 0 else np.sort(order[g * group : (g + 1) * group])
            for r, q in enumerate(members):
                d1[q] = d2[tgt]
                # distance to the target: ascending, descending or equal along the group
                nflip = (r % 12, (group - r) % 12, 3, r % 7, 2, (5 * r) % 11)[case]
                bits = rng.choice(256, nflip, replace=False)
                for b in bits:
                    d1[q, b >> 3] ^= np.uint8(1 << (b & 7))
                k1["x"][q] = np.clip(k2["x"][tgt] + rng.integers(-6, 7), 0, w - 1)
                k1["y"][q] = np.clip(k2["y"][tgt] + rng.integers(-6, 7), 0, h - 1)
                k1["angle"][q] = k2["angle"][tgt]
        # a second keypoint next to every third target with a close descriptor: second bests that get taken
        for g in range(0, min(n2, 300), 3):
            a, b = targets[g], targets[(g + 1) % n2]
            d2[b] = d2[a]
            d2[b, 5] ^= np.uint8(0x0F)
            k2["x"][b], k2["y"][b] = np.clip(k2["x"][a] + 2, 0, w - 1), k2["y"][a]
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
        c = _ctx(hip, "fork", 2000, 8, w, h)
        most = 0
        for (window, ratio, ori) in ((40, 0.9, True), (100, 0.99, False), (15, 0.7, True)):
            ng, mg, pg = c.match_window(k1, d1, k2, d2, w, h, prev, window, ratio, ori)
            no, mo, po = oracle.search_for_initialization(k1, d1, k2, d2, w, h, prev, window, ratio, ori, PRESETS["fork"])
            assert ng == no, "case %d window %d" % (case, window)
            np.testing.assert_array_equal(mg, mo, err_msg="case %d window %d" % (case, window))
            np.testing.assert_array_equal(pg, po)
            most = max(most, no)
        assert most > 10, "case %d" % case
        c.close()


def test_out_of_range_angles_do_not_leave_the_histogram(hip, oracle):
    """The reference asserts that a rotation bin lies in [0, 30), which keypoint angles in [0, 360) guarantee.
    Angles a caller made up outside that range land in an extra bin no maximum can pick: those matches are
    removed by the rotation check, every other query keeps a valid result."""
    w, h = 752, 480
    c = _ctx(hip, "fork", 2000, 8, w, h)
    k1, d1, k2, d2 = _pair(oracle, w, h, 2000, frame_seed(3, 7), "fork")
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    n_ok, m_ok, _ = c.match_window(k1, d1, k2, d2, w, h, prev, 100, 0.9, True)
    bad = np.flatnonzero(m_ok >= 0)[::5]
    k1b = k1.copy()
    k1b["angle"][bad[0::3]] = 1.0e4
    k1b["angle"][bad[1::3]] = -1.0e4
    k1b["angle"][bad[2::3]] = 3.0e9
    n, m, _ = c.match_window(k1b, d1, k2, d2, w, h, prev, 100, 0.9, True)
    assert (m[bad] == -1).all() and ((m >= -1) & (m < len(k2))).all()
    assert n == int((m >= 0).sum()) and 0 < n < n_ok
    c.close()


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_search_by_projection(hip, oracle, preset):
    w, h = 640, 480
    c = _ctx(hip, preset, 1000, 8, w, h)
    k1, d1, k2, d2 = _pair(oracle, w, h, 1000, frame_seed(1, 50), preset)
    nq = len(k1)
    r = splitmix64(4242, nq * 3)
    jitter = ((r[: 2 * nq] % np.uint64(9)).astype(np.float32) - 4).reshape(nq, 2)
    u = k1["x"] + 3 + jitter[:, 0]
    v = k1["y"] + 2 + jitter[:, 1]
    u[::37] = -5.0  # projections outside the image are skipped
    scale = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS[preset]).tables()["scale"]
    for th in (15, 7, 30):
        radius = th * scale[k1["octave"]]
        lmin = k1["octave"] - 1
        lmax = k1["octave"] + 1
        qobs = (r[2 * nq :] % np.uint64(5) != 0).astype(np.uint8)
        occ = (splitmix64(7, len(k2)) % np.uint64(11) == 0).astype(np.uint8)
        for ori in (True, False):
            ng, mg = c.match_projected(u, v, radius, lmin, lmax, k1["angle"], d1, qobs, k2, d2, w, h, occ, 100, ori)
            no, mo = oracle.search_by_projection(u, v, radius, lmin, lmax, k1["angle"], d1, qobs, k2, d2, w, h, occ, 100, ori, PRESETS[preset])
            np.testing.assert_array_equal(mg, mo, err_msg="th=%d ori=%s" % (th, ori))
            assert ng == no
    assert no > 50
    c.close()


def test_projected_searches_under_contention(hip, oracle):
    """SearchByProjection(Frame&, const Frame&, ..) and the local-map search when groups of queries want the same
    keypoint (src/ORBmatcher.cc:345-347, 364-366): the first map point WITH observations that takes a keypoint
    occupies it for every later query, map points without observations take it without occupying it, and the last
    taker's map point stays.  The kernel resolves 64 queries at a time; the result must be the walk in order."""
    w, h = 640, 480
    rng = np.random.default_rng(5)
    scale = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"]).tables()["scale"]
    for case in range(4):
        n2, nq = 1200, 1100
        k2, d2 = _random_frame(hip, n2, w, h, 700 + case)
        k2["octave"] = rng.integers(0, 3, n2)
        kq, dq = _random_frame(hip, nq, w, h, 750 + case)
        group = (3, 9, 70, 5)[case]
        targets = rng.permutation(n2)
        u = kq["x"].copy()
        v = kq["y"].copy()
        lvl = rng.integers(0, 3, nq).astype(np.int32)
        for g in range(nq // group):
            tgt = targets[g % n2]
            for r in range(group):
                q = g * group + r
                dq[q] = d2[tgt]
                for b in rng.choice(256, int(rng.integers(0, 9)), replace=False):
                    dq[q, b >> 3] ^= np.uint8(1 << (b & 7))
                u[q] = np.clip(k2["x"][tgt] + rng.integers(-3, 4), 0, w - 1)
                v[q] = np.clip(k2["y"][tgt] + rng.integers(-3, 4), 0, h - 1)
                lvl[q] = k2["octave"][tgt]
                kq["angle"][q] = k2["angle"][tgt]
            # a near-twin of the target next to it: the group's second choice
            tw = targets[(g + n2 // 2) % n2]
            d2[tw] = d2[tgt]
            d2[tw, 7] ^= np.uint8(0x33)
            k2["x"][tw], k2["y"][tw], k2["octave"][tw] = np.clip(k2["x"][tgt] + 1, 0, w - 1), k2["y"][tgt], k2["octave"][tgt]
        qobs = (rng.integers(0, 3, nq) != 0).astype(np.uint8)  # a third of the map points have no observations
        occ = (rng.integers(0, 13, n2) == 0).astype(np.uint8)
        radius = (np.float32(15) * scale[lvl]).astype(np.float32)
        c = _ctx(hip, "fork", 1000, 8, w, h)
        for ori in (True, False):
            ng, mg = c.match_projected(u, v, radius, lvl - 1, lvl + 1, kq["angle"], dq, qobs, k2, d2, w, h, occ, 100, ori)
            no, mo = oracle.search_by_projection(u, v, radius, lvl - 1, lvl + 1, kq["angle"], dq, qobs, k2, d2, w, h, occ, 100, ori, PRESETS["fork"])
            assert ng == no, "case %d ori %s" % (case, ori)
            np.testing.assert_array_equal(mg, mo, err_msg="case %d ori %s" % (case, ori))
        assert no > 20
        for ratio in (0.8, 0.99):
            ng, mg = c.match_local_map(u, v, radius, lvl - 1, lvl, dq, qobs, k2, d2, w, h, occ, 100, ratio)
            no, mo = oracle.search_local_map(u, v, radius, lvl - 1, lvl, dq, qobs, k2, d2, w, h, occ, 100, ratio, PRESETS["fork"])
            assert ng == no, "local map, case %d ratio %s" % (case, ratio)
            np.testing.assert_array_equal(mg, mo)
        c.close()


def test_hamming_host(hip, oracle):
    d = synth_descriptors(200, 5)
    for i in range(199):
        assert hip.hamming(d[i], d[i + 1]) == oracle.hamming(d[i], d[i + 1], bithack=True)


def test_brute_force_top2(hip, oracle):
    db = synth_descriptors(20011, 11)
    q = synth_queries(db, 1500, 12)
    # exact duplicates of a query's best row further down the database (ties for best)
    db[15000:15040] = db[(splitmix64(5, 40) % np.uint64(5000)).astype(np.int64)]
    q[200:240] = db[15000:15040]
    bf = hip.BruteForce(0)
    bf.set_database(db)
    ig, bg, sg = bf.query_top2(q)
    io, bo, so = oracle.bf_top2(q, db)
    np.testing.assert_array_equal(ig, io)
    np.testing.assert_array_equal(bg, bo)
    np.testing.assert_array_equal(sg, so)
    assert (bg[200:240] == 0).all() and (sg[200:240] == 0).all()
    # ragged sizes: fewer rows than one group, one row, empty
    for nd in (0, 1, 2, 3, 5, 257):
        bf.set_database(db[:nd])
        ig, bg, sg = bf.query_top2(q[:70])
        io, bo, so = oracle.bf_top2(q[:70], db[:nd])
        np.testing.assert_array_equal(ig, io)
        np.testing.assert_array_equal(bg, bo)
        np.testing.assert_array_equal(sg, so)
    bf.close()


def test_brute_force_random_shapes(hip, oracle):
    """Row and query counts around the scan's block sizes (32 rows and 32 queries per matrix instruction, 64 rows per
    tile, 512 queries per workgroup), descriptors of extreme weight (the distance is formed as popcount(query) minus a
    dot product), many ties, a row numbering that does not start at 0."""
    rng = np.random.default_rng(77)
    bf = hip.BruteForce(0)
    for nd, nq in ((31, 33), (32, 32), (33, 31), (63, 511), (64, 512), (65, 513), (127, 1), (4099, 1025), (70001, 97)):
        db = rng.integers(0, 256, (nd, 32), dtype=np.uint8)
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        db[rng.integers(0, nd)] = 0      # no bit set
        db[rng.integers(0, nd)] = 255    # every bit set
        q[rng.integers(0, nq)] = 0
        q[rng.integers(0, nq)] = 255
        for k in range(min(nq, 40)):     # exact copies of rows, some of them twice in the database
            r = int(rng.integers(0, nd))
            q[k] = db[r]
            if k % 3 == 0:
                db[int(rng.integers(0, nd))] = db[r]
        sparse = rng.integers(0, nd, max(1, nd // 7))
        db[sparse] &= rng.integers(0, 256, (len(sparse), 32), dtype=np.uint8) & 0x11  # light rows
        base = int(rng.integers(0, 1 << 20))
        bf.set_database(db, base)
        ig, bg, sg = bf.query_top2(q)
        io, bo, so = oracle.bf_top2(q, db)
        io = np.where(io >= 0, io + base, io)
        np.testing.assert_array_equal(ig, io, err_msg="%d rows, %d queries" % (nd, nq))
        np.testing.assert_array_equal(bg, bo)
        np.testing.assert_array_equal(sg, so)
    bf.close()


def test_brute_force_chunks_share_thresholds(hip, oracle):
    """The chunks of a query run side by side and raise each other's thresholds: a database cut into many chunks
    (512 queries over 400 000 rows: 48 chunks of 130 tiles, so every workgroup reads the published thresholds several
    times), with each query's best row copied into other chunks (the lowest row number has to win whichever chunk
    finds its copy first), near copies before and after it (second best equal to or one above the best, in another
    chunk), and queries without any close row.  Three runs: what a chunk sees of the others depends on timing, the
    result must not."""
    rng = np.random.default_rng(4242)
    nd, nq = 400000, 512
    db = rng.integers(0, 256, (nd, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    for k in range(384):
        home = int(rng.integers(0, nd))
        q[k] = db[home]
        if k % 4 != 3:  # exact copies elsewhere: ties for the best
            for _ in range(1 + k % 3):
                db[int(rng.integers(0, nd))] = db[home]
        if k % 2 == 0:  # near copies: one to three bits off
            for _ in range(1 + k % 2):
                r = int(rng.integers(0, nd))
                db[r] = db[home]
                for _ in range(int(rng.integers(1, 4))):
                    db[r, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    io, bo, so = oracle.bf_top2(q, db)
    bf = hip.BruteForce(0)
    bf.set_database(db)
    for run in range(3):
        ig, bg, sg = bf.query_top2(q)
        np.testing.assert_array_equal(ig, io, err_msg="run %d" % run)
        np.testing.assert_array_equal(bg, bo)
        np.testing.assert_array_equal(sg, so)
    assert (bo[:384] == 0).sum() > 370 and (so[:384] <= 3).sum() > 200  # (a planted row now and then overwrites another)
    bf.close()


def test_brute_force_sharded_merge(hip, oracle):
    """Row shards on one device, packed per-shard results merged by the merge
    kernel: equals the scan of the whole database (the all-gather only moves
    the packed words)."""
    import torch

    db = synth_descriptors(30000, 21)
    q = synth_queries(db, 999, 22)
    db[29000:29030] = db[100:130]  # duplicates across shards
    q[:30] = db[100:130]
    io, bo, so = oracle.bf_top2(q, db)
    d_q = torch.from_numpy(q).cuda()
    for R in (1, 2, 3, 8):
        bounds = [r * len(db) // R for r in range(R + 1)]
        gathered = torch.zeros((R, len(q)), dtype=torch.int64, device="cuda")
        shards = []
        for r in range(R):
            bf = hip.BruteForce(0)
            d_db = torch.from_numpy(db[bounds[r] : bounds[r + 1]]).cuda()
            bf.set_database_device(d_db, bounds[r])
            bf.query_packed_device(d_q, gathered[r])
            shards.append(bf)
        d_idx = torch.zeros(len(q), dtype=torch.int32, device="cuda")
        d_b = torch.zeros(len(q), dtype=torch.int16, device="cuda")
        d_s = torch.zeros(len(q), dtype=torch.int16, device="cuda")
        hip.merge_packed_device(0, gathered, R, len(q), d_idx, d_b, d_s)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_idx.cpu().numpy(), io, err_msg="R=%d" % R)
        np.testing.assert_array_equal(d_b.cpu().numpy().view(np.uint16), bo)
        np.testing.assert_array_equal(d_s.cpu().numpy().view(np.uint16), so)
        for bf in shards:
            bf.close()


TUM1_K = (517.306408, 516.469215, 318.643040, 255.313989)
TUM1_D = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


def test_undistort_and_image_bounds(hip, oracle):
    c = _ctx(hip, "fork")
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
    k, _ = e.extract(synth_scene(640, 480, frame_seed(1, 3)))
    extra = np.zeros(6, oracle.KP_DTYPE)  # corners, principal point, a point outside the image
    extra["x"] = [0, 640, 0, 640, TUM1_K[2], -50]
    extra["y"] = [0, 0, 480, 480, TUM1_K[3], 700]
    k = np.concatenate([k, extra])
    for D in (TUM1_D, (0, 0, 0, 0, 0), (-0.28, 0.07, 0.0002, 0.00002, 0.0), (1e-9, 0, 0, 0, 0)):
        got, ref = c.undistort_keypoints(k, TUM1_K, D), oracle.undistort_keypoints(k, TUM1_K, D)
        assert (got == ref).all(), "undistorted keypoints differ for D=%s" % (D,)
        np.testing.assert_array_equal(c.compute_image_bounds(640, 480, TUM1_K, D), oracle.compute_image_bounds(640, 480, TUM1_K, D))
    import torch

    d_in = torch.from_numpy(k.view(np.uint8).reshape(-1, 28).copy()).cuda()
    d_out = torch.zeros_like(d_in)
    c.undistort_keypoints_device(d_in, d_out, len(k), TUM1_K, TUM1_D)
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(oracle.KP_DTYPE).reshape(-1) == oracle.undistort_keypoints(k, TUM1_K, TUM1_D)).all()
    c.set_image_bounds(None)
    c.close()


def test_extraction_and_undistortion_in_one_call(hip, oracle):
    """orb_extract_undistorted = ExtractORB + UndistortKeyPoints of the Frame constructor (src/Frame.cc:67-74,
    include/Frame.h:121): keypoints and descriptors as orb_extract gives them, undistorted keypoints as the oracle
    undistorts those; without distortion (k1 == 0) the undistorted keypoints are the keypoints; a frame without
    keypoints gives none."""
    c = _ctx(hip, "fork")
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, PRESETS["fork"])
    for seed, (w, h) in ((5, (640, 480)), (6, (637, 471))):
        img = np.ascontiguousarray(synth_scene(640, 480, frame_seed(1, seed))[:h, :w])
        ko, do = e.extract(img)
        for D in (TUM1_D, (0, 0, 0, 0, 0), (-0.28, 0.07, 0.0002, 0.00002, 0.0)):
            k, un, d = c.extract_undistorted(img, TUM1_K, D)
            assert len(k) == len(ko) and (k == ko).all() and (d == do).all()
            assert (un == oracle.undistort_keypoints(ko, TUM1_K, D)).all(), "D=%s" % (D,)
    k, un, d = c.extract_undistorted(np.full((480, 640), 90, np.uint8), TUM1_K, TUM1_D)
    assert len(k) == 0 and len(un) == 0 and len(d) == 0
    k2, d2 = c.extract(img)  # the context goes on as before
    assert (k2 == ko).all() and (d2 == do).all()
    c.close()


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_matchers_with_undistorted_bounds(hip, oracle, preset):
    """Both windowed searches on undistorted keypoints with the grid laid over the undistorted image bounds."""
    q = PRESETS[preset]
    w, h = 640, 480
    c = _ctx(hip, preset)
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, q)
    a = synth_scene(w, h, frame_seed(1, 21))
    b = synth_scene(w, h, frame_seed(1, 21), shift=(4, -3), noise_seed=99)
    k1, d1 = e.extract(a)
    k2, d2 = e.extract(b)
    k1u, k2u = oracle.undistort_keypoints(k1, TUM1_K, TUM1_D), oracle.undistort_keypoints(k2, TUM1_K, TUM1_D)
    bounds = c.compute_image_bounds(w, h, TUM1_K, TUM1_D)
    oracle.set_image_bounds(bounds)
    try:
        prev0 = np.stack([k1u["x"], k1u["y"]], axis=1).astype(np.float32)
        no, mo, po = oracle.search_for_initialization(k1u, d1, k2u, d2, w, h, prev0, 100, 0.9, True, q)
        ng, mg, pg = c.match_window(k1u, d1, k2u, d2, w, h, prev0, 100, 0.9, True)
        assert ng == no and no > 20
        np.testing.assert_array_equal(mg, mo)
        np.testing.assert_array_equal(pg, po)
        # projected search: queries = frame-1 keypoints near the bounds as well as inside
        n = len(k1u)
        u = k1u["x"] + np.float32(2.5)
        v = k1u["y"] - np.float32(1.5)
        u[:5] = [bounds[0] - 1, bounds[1] + 1, bounds[0], bounds[1], bounds[0] + 0.5]
        rad = np.full(n, 15.0, np.float32) * (np.float32(1.2) ** k1u["octave"]).astype(np.float32)
        lvl = k1u["octave"].astype(np.int32)
        no, m2o = oracle.search_by_projection(u, v, rad, lvl - 1, lvl + 1, k1u["angle"], d1, None, k2u, d2, w, h, None, 100, True, q)
        ng, m2g = c.match_projected(u, v, rad, lvl - 1, lvl + 1, k1u["angle"], d1, None, k2u, d2, w, h, None, 100, True)
        assert ng == no and no > 20
        np.testing.assert_array_equal(m2g, m2o)
    finally:
        oracle.set_image_bounds(None)
        c.set_image_bounds(None)
    c.close()


def test_distinctive_descriptors(hip, oracle):
    """MapPoint::ComputeDistinctiveDescriptors batched: ragged lists incl. empty, 1, 2, 64, 65, > LDS capacity,
    near-duplicate descriptors (many ties), both the fork's self-comparison and upstream behaviour."""
    c = _ctx(hip, "fork")
    sizes = [0, 1, 2, 3, 4, 7, 16, 63, 64, 65, 130, 600, 0, 5, 33]
    base = synth_descriptors(64, 0xD15C)
    lists = []
    for k, n in enumerate(sizes):
        r = splitmix64(77 + k, max(n, 1) * 4).reshape(-1, 4)
        rows = []
        for i in range(n):
            v = base[int(r[i, 0] % np.uint64(8))].copy()            # a few clusters -> many equal medians
            nflip = int(r[i, 1] % np.uint64(40))
            bits = splitmix64(int(r[i, 2] & np.uint64(0xFFFFFFFF)), max(nflip, 1)) % np.uint64(256)
            for b in bits[:nflip]:
                v[int(b) // 8] ^= np.uint8(1 << (int(b) % 8))
            rows.append(v)
        lists.append(np.stack(rows) if rows else np.zeros((0, 32), np.uint8))
    desc = np.concatenate(lists)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    for self_distance in (False, True):
        best, med = c.distinctive_descriptors(desc, offsets, self_distance)
        for p, rows in enumerate(lists):
            ri, rm = oracle.distinctive_descriptor(rows, self_distance)
            assert best[p] == ri, "point %d (n=%d, self=%s): index %d, oracle %d" % (p, len(rows), self_distance, best[p], ri)
            if ri >= 0:
                assert med[p] == rm, "point %d: median %d, oracle %d" % (p, med[p], rm)
    # device entry point
    import torch

    d_desc = torch.from_numpy(desc).cuda()
    d_off = torch.from_numpy(offsets).cuda()
    d_best = torch.zeros(len(sizes), dtype=torch.int32, device="cuda")
    d_med = torch.zeros(len(sizes), dtype=torch.int32, device="cuda")
    c.distinctive_descriptors_device(d_desc, d_off, d_best, d_med)
    torch.cuda.synchronize()
    best, _ = c.distinctive_descriptors(desc, offsets, False)
    np.testing.assert_array_equal(d_best.cpu().numpy(), best)
    c.close()


def test_extract_4k_frame(hip, oracle):
    """Config-4 shape: one 3840x2160 frame, 4000 features (about 130k FAST keys on level 0)."""
    w, h, nf = 3840, 2160, 4000
    img = synth_scene(w, h, frame_seed(4, 0))
    c = _ctx(hip, "fork", nf, 8, w, h)
    e = oracle.Extractor(nf, 1.2, 8, 20, 7, PRESETS["fork"])
    kg, dg = c.extract(img)
    ko, do = e.extract(img)
    for l in (1, 7):
        np.testing.assert_array_equal(c.pyramid_level(0, l), e.level(l), err_msg="pyramid level %d" % l)
    xs, ys, sc = c.fast_keys(0, 0)
    ref = e.level_keys(0, after_quadtree=False)
    assert len(xs) == len(ref) and len(ref) > 50000
    np.testing.assert_array_equal(xs, ref["x"].astype(np.int32))
    assert_kps_equal(kg, ko, "4K frame")
    np.testing.assert_array_equal(dg, do)
    c.close()
