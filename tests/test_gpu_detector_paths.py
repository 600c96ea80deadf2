"""GPU: detector code paths that the default configuration does not take on every call.

* the standalone score + NMS kernels (unaligned widths take them always; OKVFE_NO_FUSED_NMS forces
  them for aligned widths) must give the same candidates as the fused score+NMS kernel -- both
  against the oracle;
* the fix-up of the fused kernel: plateaus that cross strip borders (x = 247|248, 495|496) and tile
  borders (y = 29|30, ...), where the raster-scan rule of the reference needs the neighbour strip;
* the greedy selection when the candidate list is longer than the LDS record chunk (several refills)
  and when max_keypoints cuts the greedy short inside a round;
* per-stage profiling mask of okvfe_profile_enable.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the A/B knobs exist in the LAB build only (make -C okvis2_amd/csrc lab); the product library reads
# no environment variable (tests/test_capi_host.py::test_product_library_reads_no_environment)
LAB_LIB = os.path.join(ROOT, "okvis2_amd", "libokvfe_lab.so")


def _lab_environ(**knobs):
    assert os.path.exists(LAB_LIB), "libokvfe_lab.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    return dict(os.environ, OKVFE_LIB=LAB_LIB, **knobs)


def _plateau_image(w, h, seed, bx, by):
    """Random blocks of bx x by equal pixels: equal Harris scores on horizontally adjacent pixels,
    with block edges placed on purpose at the strip and tile borders of the fused kernel."""
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 256, size=((h + by - 1) // by, (w + bx - 1) // bx), dtype=np.uint8)
    return np.ascontiguousarray(np.kron(small, np.ones((by, bx), dtype=np.uint8))[:h, :w])


@pytest.mark.parametrize("w,h,bx,by", [(752, 480, 2, 1), (752, 480, 4, 3), (752, 480, 8, 5),
                                       (1024, 120, 2, 2), (256, 64, 2, 1)])
def test_fused_nms_plateaus_across_borders(oracle, w, h, bx, by):
    img = _plateau_image(w, h, 5 + bx, bx, by)
    # radius 12 does not fit the occupancy grid in LDS (legacy greedy kernel), radius 38 does
    for thr, radius, maxk in ((1, 12.0, 4000), (150, 38.0, 700)):
        fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1 << 16)
        ref = oracle.detect(img, radius, 0, thr, maxk)
        assert len(ref) > 20
        G.assert_keypoints_equal(fe.detect(img), ref)
        fe.set_keep_score_map(True)   # (round 4: the default writes no map where the selection can do without)
        G.assert_keypoints_equal(fe.detect(img), ref)


def test_select_many_candidates_and_limit(oracle):
    """> 4 record chunks of candidates (noise image, low threshold) and a keypoint limit that is
    reached in the middle of a round."""
    w, h = 752, 480
    img = synth.noise_image(w, h, 77)
    for maxk, radius in ((700, 38.0), (37, 38.0), (5, 20.0), (1, 38.0)):
        fe = capi.Frontend(w, h, radius, 0, 40, maxk, max_candidates=1 << 16)
        ref = oracle.detect(img, radius, 0, 40, maxk)
        assert 0 < len(ref) <= maxk
        G.assert_keypoints_equal(fe.detect(img), ref)


def test_selection_orders_thousands_of_tied_scores(oracle):
    """The array-bin selection orders its candidates itself (round 4): log buckets of the score, chunks
    of whole buckets, survivors rank-sorted in LDS.  An exact checkerboard gives a few distinct scores
    shared by thousands of maxima: one bucket far above the 1024 keys a round holds, which is then split
    by key range, i.e. by (y, x) -- the order the reference's sort gives tied scores.  Also with a cap
    that falls inside such a run, and with a stripe of noise that adds ordinary buckets around it."""
    w, h = 752, 480
    yy, xx = np.mgrid[0:h, 0:w]
    board = np.where(((xx // 9) + (yy // 9)) % 2 == 0, 40, 215).astype(np.uint8)
    mixed = board.copy()
    mixed[200:280] = synth.noise_image(w, 80, 5)
    for img in (board, mixed):
        for maxk, radius in ((700, 38.0), (150, 38.0), (700, 17.0)):
            fe = capi.Frontend(w, h, radius, 0, 100, maxk, max_candidates=1 << 15)
            ref = oracle.detect(img, radius, 0, 100, maxk)
            assert len(ref) > 50
            G.assert_keypoints_equal(fe.detect(img), ref)      # map-free: fix-up on the records (bitmap path)
            fe.set_keep_score_map(True)
            G.assert_keypoints_equal(fe.detect(img), ref)      # the same through the score map
            fe.set_keep_score_map(False)
            G.assert_keypoints_equal(fe.detect(img), ref)
    # the tie run really is longer than one round of the kernel
    cand = oracle.nms(oracle.harris_score(board), 100)
    _, counts = np.unique(cand["score"], return_counts=True)
    assert counts.max() > 1024, counts.max()


@pytest.mark.parametrize("w,h,radius,thr,maxk", [(752, 480, 38.0, 40, 700), (640, 480, 10.0, 30, 800),
                                                 (256, 64, 12.0, 20, 500), (1024, 128, 25.0, 30, 700)])
def test_keypoints_on_the_image_rim_without_a_score_map(oracle, w, h, radius, thr, maxk):
    """Map-free detection recomputes the nine sub-pixel scores of a kept keypoint from 7 x 7 pixels
    (harris_scores_3x3: three aligned dwords per row, the dword before a row / past its end is not read,
    gradient products on the image rim are zero).  Structure two pixels inside the rim puts a quarter of the
    kept keypoints on columns 2 / 3 / w - 4 / w - 3 and the same rows -- the array-bin, the linked-list
    (radius 10) and the packed-strip (1024 px) forms, against the oracle and against the score-map form."""
    fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1 << 16)
    on_rim = 0
    for seed in (100, 101):
        img = synth.noise_image(w, h, seed).copy()
        img[2:4, :] = np.where((np.arange(w) // 3) % 2 == 0, 250, 5)
        img[-4:-2, :] = img[2:4, :]
        img[:, 2:4] = np.where((np.arange(h) // 3) % 2 == 0, 250, 5)[:, None]
        img[:, -4:-2] = img[:, 2:4]
        ref = oracle.detect(img, radius, 0, thr, maxk)
        G.assert_keypoints_equal(fe.detect(img), ref)
        assert fe.device_outputs().scores is None
        fe.set_keep_score_map(True)
        G.assert_keypoints_equal(fe.detect(img), ref)
        assert fe.device_outputs().scores is not None
        fe.set_keep_score_map(False)
        xs, ys = np.floor(ref["x"] + 0.5), np.floor(ref["y"] + 0.5)
        on_rim += int(((xs <= 3) | (xs >= w - 4) | (ys <= 3) | (ys >= h - 4)).sum())
    assert on_rim > 50


def test_few_or_no_candidates(oracle):
    """The self-ordering selection on candidate sets around its first chunk sizes (0, a handful, 63 .. 129 boxes
    = 100 .. 155 kept corners), and a batch that mixes empty and ordinary images."""
    w, h = 752, 480
    fe = capi.Frontend(w, h, 38.0, 0, 150, 700, max_candidates=1 << 15)

    def check(img):
        ref = oracle.detect(img, 38.0, 0, 150, 700)
        G.assert_keypoints_equal(fe.detect(img), ref)
        return len(ref)

    assert check(np.full((h, w), 128, np.uint8)) == 0
    rng = np.random.default_rng(3)
    kept = []
    for n in (1, 3, 63, 64, 65, 127, 129):
        img = np.full((h, w), 20, np.uint8)
        for _ in range(n):
            x, y, s = int(rng.integers(10, w - 30)), int(rng.integers(10, h - 30)), int(rng.integers(6, 14))
            img[y:y + s, x:x + s] = int(rng.integers(120, 255))
        kept.append(check(img))
    assert kept[0] >= 1 and kept[-1] > 100
    cfg = synth.euroc_config()
    fb = G.make_frontend(cfg, max_batch=4)
    for ci, cam in enumerate(cfg.cams):
        fb.set_camera(ci, cam)
    imgs = np.stack([np.full((h, w), 128, np.uint8), G.image_for(cfg, 5), np.zeros((h, w), np.uint8), G.image_for(cfg, 6)])
    d = torch.from_numpy(imgs).cuda()
    cams = np.array([0, 1, 0, 1], np.int32)
    fb.detect_describe_batch_device(d.data_ptr(), 4, cams, np.tile(np.array([0, 1, 0], np.float32), (4, 1)), None)
    torch.cuda.synchronize()
    for i in range(4):
        cam = cfg.cams[cams[i]]
        rays, jac = oracle.awareness_maps(cam)
        rk, rd = oracle.detect_describe(imgs[i], cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                        oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.0, 1.0, 0.0))
        k, dd, _, _ = fb.download(i)
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(dd, rd) and (len(k) == 0) == (i % 2 == 0)


def test_sort_network_sizes_around_its_limits(oracle):
    """Candidate counts around the limits of the register-blocked sort: one thread's 16 keys, one
    LDS pass, just below / above 4096 and 8192 keys (above 8192 the two-stride LDS network or the
    HBM workspace network takes over); ties in the score (binary noise) order by (y, x)."""
    w = 752
    rng = np.random.default_rng(99)
    base = (rng.integers(0, 2, (480, w)) * 255).astype(np.uint8)
    counts = {}
    for h in range(16, 481, 4):
        counts[h] = len(oracle.nms(oracle.harris_score(np.ascontiguousarray(base[:h])), 1))
    picks = []
    for lim in (4096, 8192, 16384):
        picks.append(max((h for h in counts if counts[h] <= lim), key=lambda h: counts[h]))
        picks.append(min((h for h in counts if counts[h] > lim), key=lambda h: counts[h]))
    assert any(4096 < counts[h] <= 8192 for h in picks) and any(8192 < counts[h] <= 16384 for h in picks)
    assert any(counts[h] > 16384 for h in picks)
    cases = [(h, 1) for h in sorted(set(picks))]
    # a handful of candidates (up to one thread's 16 keys, and one more): thresholds from the scores
    img64 = np.ascontiguousarray(base[:64])
    sc = np.sort(oracle.nms(oracle.harris_score(img64), 1)["score"])[::-1]
    for want in (1, 5, 16, 17, 40):
        thr = int(sc[want - 1])
        if thr > int(sc[want]):  # no tie across the cut
            cases.append((64, thr))
    assert len(cases) >= 8
    for h, thr in cases:
        img = np.ascontiguousarray(base[:h])
        fe = capi.Frontend(w, h, 9.0, 0, thr, 4000, max_candidates=1 << 15)
        ref = oracle.detect(img, 9.0, 0, thr, 4000)
        assert len(ref) > 0
        G.assert_keypoints_equal(fe.detect(img), ref)


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
w, h = 752, 480
for kind, thr in (("corners", 150), ("noise", 800), ("plateau", 1)):
    if kind == "plateau":
        rng = np.random.default_rng(3)
        img = np.ascontiguousarray(np.kron(rng.integers(0, 256, (h, w // 2), dtype=np.uint8),
                                           np.ones((1, 2), np.uint8)))
    else:
        img = synth.noise_image(w, h, 9) if kind == "noise" else synth.corners_image(w, h, 9)
    fe = capi.Frontend(w, h, 38.0, 0, thr, 700, max_candidates=1 << 16)
    ref = O.detect(img, 38.0, 0, thr, 700)
    G.assert_keypoints_equal(fe.detect(img), ref)
print("UNFUSED-OK")
"""


def test_standalone_score_and_nms_kernels(oracle):
    env = _lab_environ(OKVFE_NO_FUSED_NMS="1")
    out = subprocess.run([sys.executable, "-c", _CHILD, ROOT], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "UNFUSED-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


_LEGACY_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
w, h = 640, 480
for seed, radius, maxk in ((3, 10.0, 1000), (4, 6.0, 4000)):
    img = synth.corners_image(w, h, seed)
    fe = capi.Frontend(w, h, radius, 0, 5, maxk, max_candidates=1 << 16)
    G.assert_keypoints_equal(fe.detect(img), O.detect(img, radius, 0, 5, maxk))
print("LEGACY-OK")
"""


def test_legacy_select_kernel_for_grids_outside_lds(oracle):
    """Occupancy grids that do not fit in LDS normally take select_greedy_kernel<false> (grid in
    HBM; covered by the small-radius cases above); OKVFE_LEGACY_SELECT keeps the older
    one-accept-per-round kernel reachable, which is also the path for > 65536 candidates."""
    env = _lab_environ(OKVFE_LEGACY_SELECT="1")
    out = subprocess.run([sys.executable, "-c", _LEGACY_CHILD, ROOT], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "LEGACY-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


_KNOB_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
for (w, h, radius, thr, maxk) in ((752, 480, 38.0, 150, 700), (1024, 128, 20.0, 60, 500), (512, 256, 12.0, 40, 800)):
    for seed in (1, 2):
        img = synth.corners_image(w, h, seed) if seed == 1 else synth.noise_image(w, h, seed)
        fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1 << 15)
        ref = O.detect(img, radius, 0, thr, maxk)
        assert len(ref) > 10
        G.assert_keypoints_equal(fe.detect(img), ref)
print("KNOB-OK")
"""


@pytest.mark.parametrize("knob", ["OKVFE_LEGACY_SORT", "OKVFE_K1_NOPACK", "OKVFE_SELECT_OCC_HBM",
                                  "OKVFE_LEGACY_SELECT", "OKVFE_LAZY_BINCAP", "OKVFE_K1_TH=61", "OKVFE_K1_TH=25",
                                  "OKVFE_SELECT_PRESORTED", "OKVFE_LAZY_ROUNDCAP=64", "OKVFE_LAZY_ROUNDCAP=7",
                                  "OKVFE_KEEP_SCORE_MAP", "OKVFE_K1_GENERIC_NOMAP"])
def test_ab_knobs_keep_their_paths_exact(oracle, knob):
    """The A/B switches the profiling notes refer to (read once per process, hence a child process
    each) select older or alternative kernels: two-stride LDS sort, unpacked last strips, occupancy
    grid in HBM, one-accept-per-round selection, the array-bin selection on keys sorted by a launch
    of their own (round 4: it orders its candidates itself), and that kernel with rounds of 64 / 7
    keys, which sends every chunk through the key-range split; map-free detection through the map-writing
    instantiation of the score kernel (its run-time branch) instead of the one compiled without the store.
    Each must stay bit-exact."""
    env = _lab_environ()
    k, _, v = knob.partition("=")
    env[k] = v or "1"
    out = subprocess.run([sys.executable, "-c", _KNOB_CHILD, ROOT], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "KNOB-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


_KNOB_DESC_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
cfg = synth.euroc_config()
fe = G.make_frontend(cfg)
cam = cfg.cams[0]
fe.set_camera(0, cam)
rays, jac = O.awareness_maps(cam)
for seed, grav in ((4, (0.0, 1.0, 0.0)), (5, (0.3, 0.9, -0.2))):
    img = G.image_for(cfg, seed)
    rk, rd = O.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                               O.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), grav)
    kps, desc, bp, bpv = fe.detect_describe(img, cam=0, gravity=grav)
    G.assert_keypoints_equal(kps, rk)
    assert np.array_equal(desc, rd) and len(kps) > 50
print("KNOB-OK")
"""


@pytest.mark.parametrize("knob", ["OKVFE_PARAM_MEMCPY", "OKVFE_NO_FUSED_SETUP", "OKVFE_DESC_WAVES=5", "OKVFE_DESC_GENERIC"])
def test_ab_knobs_of_the_describe_path(oracle, knob):
    """Parameter upload through the DMA engine instead of the copy kernel, the extractor's setup as
    its own launch instead of inside the selection kernel, the 5-wave describe instantiation."""
    env = _lab_environ()
    k, _, v = knob.partition("=")
    env[k] = v or "1"
    out = subprocess.run([sys.executable, "-c", _KNOB_DESC_CHILD, ROOT], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "KNOB-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


_RING_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
cfg = synth.euroc_config()
fe = G.make_frontend(cfg, max_batch=2)
cam = cfg.cams[0]
fe.set_camera(0, cam)
rays, jac = O.awareness_maps(cam)
img = G.image_for(cfg, 21)
d_img = torch.from_numpy(np.stack([img, img])).cuda()
ids = np.zeros(2, dtype=np.int32)
s = torch.cuda.current_stream().cuda_stream
refs = {}
# 20 calls = 2.5 laps of the 8-slot parameter ring, a different gravity pair each time, no
# synchronisation in between; every call's descriptors must be those of ITS parameters
gs = [np.array([[np.sin(0.3 * i), np.cos(0.3 * i), 0.1 * (i % 3)], [0.0, 1.0, 0.02 * i]], dtype=np.float32) for i in range(20)]
outs = []
for i, g in enumerate(gs):
    fe.detect_describe_batch_device(d_img.data_ptr(), 2, ids, g, s)
    torch.cuda.synchronize() if i % 7 == 6 else None
    if i in (0, 9, 19):
        torch.cuda.synchronize()
        outs.append((i, [fe.download(j) for j in range(2)]))
for i, res in outs:
    for j in range(2):
        rk, rd = O.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                   O.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), tuple(float(v) for v in gs[i][j]))
        G.assert_keypoints_equal(res[j][0], rk)
        assert np.array_equal(res[j][1], rd), (i, j)
print("RING-OK")
"""


@pytest.mark.parametrize("knob", ["", "OKVFE_TEST_SKIP_RING_RELEASE"])
def test_parameter_ring_laps(oracle, knob):
    """The per-call parameter blocks travel through an 8-slot pinned ring: slots are reused after the
    event behind their last reader -- or, for a call that never released its slot (forced here by the
    test knob), after a wait on that call's stream."""
    env = _lab_environ() if knob else dict(os.environ)
    if knob:
        env[knob] = "1"
    out = subprocess.run([sys.executable, "-c", _RING_CHILD, ROOT], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "RING-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_profile_stage_mask():
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg, max_batch=2)
    for ci, cam in enumerate(cfg.cams):
        fe.set_camera(ci, cam)
    L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 4)
    d_img = torch.from_numpy(np.stack([L, R])).cuda()
    cam_ids = np.array([0, 1], dtype=np.int32)
    grav = np.tile(np.array([0.0, 1.0, 0.0], dtype=np.float32), (2, 1))
    s = torch.cuda.current_stream().cuda_stream

    def run(n):
        for _ in range(n):
            fe.detect_describe_batch_device(d_img.data_ptr(), 2, cam_ids, grav, s)

    fe.profile_enable(True, stages=("harris", "describe"))
    run(3)
    p = fe.profile_read()
    assert p["harris"][1] == 3 and p["describe"][1] == 3 and p["harris"][0] > 0.0
    assert all(p[k][1] == 0 for k in ("nms", "sort", "select", "compact", "match"))
    fe.profile_enable(True)
    run(2)
    p = fe.profile_read()
    assert all(p[k][1] == 2 for k in ("harris", "nms", "sort", "select", "describe", "compact"))
    fe.profile_enable(False)
    run(1)
    assert all(v[1] == 0 for v in fe.profile_read().values())


@pytest.mark.parametrize("w,h", [(64, 64), (252, 65), (256, 90), (260, 121), (500, 64), (1000, 64),
                                 (248, 91), (496, 119), (68, 200), (1280, 67)])
def test_detect_ragged_sizes(oracle, w, h):
    """Aligned widths around the 248-pixel strip width and heights around the 30-row tile of the
    fused score+NMS kernel (partial last strip / tile, single tile, tile + 1 row ...)."""
    rng = np.random.default_rng(w * 1000 + h)
    for kind in ("noise", "blocks"):
        if kind == "noise":
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            img = _plateau_image(w, h, w + h, 3, 2)
        for thr, radius, maxk in ((1, 6.0, 4000), (500, 10.0, 300)):
            fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1 << 16)
            ref = oracle.detect(img, radius, 0, thr, maxk)
            G.assert_keypoints_equal(fe.detect(img), ref)


@pytest.mark.parametrize("n_images,mode", [(11, "aware"), (19, "upright"), (8, "gradient")])
def test_batch_of_odd_size_mixed_images(oracle, n_images, mode):
    """Batch sizes that are not multiples of 8 (the XCD-aware block -> image maps of the score,
    describe and NMS kernels have a tail path) with different image kinds in one batch, one of
    them without a single corner."""
    cfg = synth.mono640_config()
    cam = cfg.cams[0]
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                       rotation_invariant=(mode == "gradient"), max_batch=n_images, num_cameras=1)
    imgs = []
    for i in range(n_images):
        if i == 4:
            imgs.append(np.full((cfg.h, cfg.w), 90, np.uint8))  # flat: no candidates at all
        elif i % 3 == 0:
            imgs.append(synth.noise_image(cfg.w, cfg.h, 50 + i))
        else:
            imgs.append(synth.corners_image(cfg.w, cfg.h, 50 + i))
    imgs = np.stack(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    cam_ids, grav = None, None
    omode = oracle.MODE_GRADIENT if mode == "gradient" else oracle.MODE_UPRIGHT
    rays = jac = None
    if mode == "aware":
        fe.set_camera(0, cam)
        cam_ids = np.zeros(n_images, dtype=np.int32)
        grav = np.tile(np.array([0.05, 0.99, -0.1], dtype=np.float32), (n_images, 1))
        rays, jac = oracle.awareness_maps(cam)
        omode = oracle.MODE_CAMERA_AWARE
    fe.detect_describe_batch_device(d_img.data_ptr(), n_images, cam_ids, grav,
                                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    total = 0
    for i in range(n_images):
        k, d = oracle.detect_describe(imgs[i], cfg.uniformity_radius, 0, cfg.abs_threshold,
                                      cfg.max_kpts, omode, rays, jac, np.float32(cam.fu),
                                      (0.05, 0.99, -0.1))
        g = fe.download(i)
        G.assert_keypoints_equal(g[0], k)
        assert np.array_equal(g[1], d)
        total += len(k)
    assert total > 100 * (n_images // 2)
    assert len(fe.download(4)[0]) == 0


@pytest.mark.parametrize("w,h,n_images", [(1024, 150, 11), (260, 130, 23), (512, 512, 17), (1280, 100, 6)])
def test_packed_last_strips_batches(oracle, w, h, n_images):
    """Rows whose last 62-dword strip is narrow (1024 px: 7 dwords, 260 px: 2, 512 px: 3, 1280 px: 9):
    the fused score+NMS kernel lets the last strips of 8 / 21 / 16 / 6 consecutive images share one
    wave.  Batches that end inside a group, images of different kinds next to each other (plateaus
    set the fix-up flags of a whole wave row), one flat image; every image against the oracle."""
    fe = capi.Frontend(w, h, 12.0, 0, 60, 500, rotation_invariant=False, max_batch=n_images,
                       max_candidates=1 << 15)
    imgs = []
    for i in range(n_images):
        if i == 2:
            imgs.append(np.full((h, w), 17, np.uint8))
        elif i % 3 == 0:
            imgs.append(_plateau_image(w, h, 300 + i, 2 + i % 2, 1 + i % 3))
        elif i % 3 == 1:
            imgs.append(synth.noise_image(w, h, 300 + i))
        else:
            imgs.append(synth.corners_image(w, h, 300 + i))
    imgs = np.stack(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    fe.detect_describe_batch_device(d_img.data_ptr(), n_images, None, None,
                                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fe.check_capacity(n_images)  # raises if a candidate list overflowed
    total = 0
    for i in range(n_images):
        k, d = oracle.detect_describe(imgs[i], 12.0, 0, 60, 500, oracle.MODE_UPRIGHT, None, None,
                                      np.float32(1.0), (0.0, 1.0, 0.0))
        g = fe.download(i)
        G.assert_keypoints_equal(g[0], k)
        assert np.array_equal(g[1], d)
        total += len(k)
    assert total > 20 * n_images
    assert len(fe.download(2)[0]) == 0


def test_split_batch_api_equals_combined_call():
    cfg = synth.euroc_config()
    n = 6
    fe = G.make_frontend(cfg, max_batch=n)
    for ci, cam in enumerate(cfg.cams):
        fe.set_camera(ci, cam)
    imgs = np.stack([im for i in range(n // 2) for im in synth.stereo_pair(cfg.w, cfg.h, 60 + i)[:2]])
    d_img = torch.from_numpy(imgs).cuda()
    cam_ids = np.array([0, 1] * (n // 2), dtype=np.int32)
    grav = np.tile(np.array([0.0, 1.0, 0.0], dtype=np.float32), (n, 1))
    s = torch.cuda.current_stream().cuda_stream
    fe.detect_describe_batch_device(d_img.data_ptr(), n, cam_ids, grav, s)
    ref = [fe.download(i) for i in range(n)]
    fe.detect_batch_device(d_img.data_ptr(), n, s)
    fe.describe_batch_device(d_img.data_ptr(), n, cam_ids, grav, s)
    for i in range(n):
        got = fe.download(i)
        for a, b in zip(got, ref[i]):
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
        assert len(got[0]) > 50
    with pytest.raises(capi.OkvfeError):  # describe must follow a detect of the same size
        fe.describe_batch_device(d_img.data_ptr(), n - 2, cam_ids[:n - 2], grav[:n - 2], s)


_TOKEN_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import torch
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
cfg = synth.mono640_config()
capi.set_heavy_kernel_chaining(int(sys.argv[2]))
lanes = []
for l in range(3):
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                       rotation_invariant=False, max_batch=4, num_cameras=1)
    imgs = np.stack([synth.corners_image(cfg.w, cfg.h, 10 * l + i) for i in range(4)])
    lanes.append((fe, imgs, torch.from_numpy(imgs).cuda(), torch.cuda.Stream()))
torch.cuda.synchronize()
for rep in range(5):
    if int(sys.argv[2]) == 2:   # stage-major: all detects, then all describes
        for fe, _, d, st in lanes: fe.detect_batch_device(d.data_ptr(), 4, st.cuda_stream)
        for fe, _, d, st in lanes: fe.describe_batch_device(d.data_ptr(), 4, None, None, st.cuda_stream)
    else:
        for fe, _, d, st in lanes: fe.detect_describe_batch_device(d.data_ptr(), 4, None, None, st.cuda_stream)
torch.cuda.synchronize()
for fe, imgs, _, _ in lanes:
    for i in range(4):
        k, dsc = O.detect_describe(imgs[i], cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                   O.MODE_UPRIGHT)
        g = fe.download(i)
        G.assert_keypoints_equal(g[0], k)
        assert np.array_equal(g[1], dsc)
print("TOKEN-OK")
"""


@pytest.mark.parametrize("mode", [1, 2])
def test_score_token_across_contexts(oracle, mode):
    """okvfe_set_heavy_kernel_chaining chains the heavy kernels of three contexts on three streams
    through events: same results, no deadlock, in both enqueue orders."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", _TOKEN_CHILD, ROOT, str(mode)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TOKEN-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_candidate_overflow_is_deterministic_and_reported(oracle):
    """An NMS candidate list that overflows max_candidates leaves THAT image without keypoints
    (which maxima an overflowing list drops depends on atomic order) and okvfe_check_capacity names
    it; the other images of the batch are untouched.  Every path that reads counts honours it."""
    w, h = 752, 480
    busy = synth.noise_image(w, h, 3)            # tens of thousands of maxima at threshold 5
    calm = synth.corners_image(w, h, 4)
    fe = capi.Frontend(w, h, 38.0, 0, 150, 700, max_batch=3, max_candidates=8000)
    n_busy = len(oracle.nms(oracle.harris_score(busy), 150))
    n_calm = len(oracle.nms(oracle.harris_score(calm), 150))
    assert n_busy > 8000 > n_calm
    d_img = torch.from_numpy(np.stack([calm, busy, calm])).cuda()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), 3, None, None, st)
    st.synchronize()
    with pytest.raises(capi.OkvfeError) as e:
        fe.check_capacity(3)
    assert e.value.status == capi.ERR_CAPACITY and "image 1" in str(e.value)
    ref = oracle.detect_describe(calm, 38.0, 0, 150, 700, oracle.MODE_GRADIENT)
    for i in (0, 2):
        k, d, _, _ = fe.download(i)
        G.assert_keypoints_equal(k, ref[0])
        assert np.array_equal(d, ref[1])
    with pytest.raises(capi.OkvfeError) as e:
        fe.download(1)
    assert e.value.status == capi.ERR_CAPACITY
    # the gather blocks (what the cross-camera matchers read) carry an EMPTY image 1, not a truncated one
    nb = fe.gather_block_bytes()
    blocks = torch.zeros((3, nb), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fe.pack_gather_blocks_device(0, 3, blocks.data_ptr(), st)
    st.synchronize()
    host_counts = blocks[:, :4].cpu().numpy().copy().view(np.int32)[:, 0]
    assert host_counts[1] == 0 and host_counts[0] == len(ref[0]) == host_counts[2]
    fe.check_capacity(1)  # image 0 alone is fine


def test_host_fed_batches_equal_device_fed(oracle):
    """okvfe_detect_describe_batch_host (pinned host images, library-side double-buffered copy) gives
    the same rows as the device-resident call, over several calls that alternate the two buffers
    and with different images per call."""
    cfg = synth.euroc_config()
    B = 4
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                       num_cameras=2, max_batch=B)
    for ci in range(2):
        fe.set_camera(ci, cfg.cams[ci])
    st = torch.cuda.Stream()
    cam_ids = np.array([0, 1] * (B // 2), dtype=np.int32)
    batches = []
    for c in range(3):
        imgs = np.stack([synth.corners_image(cfg.w, cfg.h, 900 + 10 * c + i) for i in range(B)])
        grav = np.array([[0.01 * (c + i), 1.0, -0.02 * i] for i in range(B)], dtype=np.float32)
        batches.append((imgs, grav, torch.from_numpy(imgs).pin_memory()))
    got = []
    for imgs, grav, pinned in batches:  # enqueue all three without waiting: copies overlap kernels
        fe.detect_describe_batch_host(pinned.data_ptr(), B, cam_ids, grav, st)
        st.synchronize()  # results live in the context until the next call
        got.append([fe.download(i) for i in range(B)])
    for (imgs, grav, _), rows in zip(batches, got):
        d_img = torch.from_numpy(imgs).cuda()
        st.wait_stream(torch.cuda.current_stream())
        fe.detect_describe_batch_device(d_img.data_ptr(), B, cam_ids, grav, st)
        st.synchronize()
        for i in range(B):
            want = fe.download(i)
            for a, b in zip(rows[i], want):
                assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
            assert len(want[0]) > 100
    # back-to-back host-fed calls without a host wait in between: the last result must be right
    for imgs, grav, pinned in batches:
        fe.detect_describe_batch_host(pinned.data_ptr(), B, cam_ids, grav, st)
    st.synchronize()
    for i in range(B):
        for a, b in zip(fe.download(i), got[-1][i]):
            assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


def test_byte_mover_diagnostic(oracle):
    """okvfe_harris_byte_mover_device moves the fused kernel's bytes (score maps receive pixel
    bytes, no candidates); a detection afterwards recomputes everything."""
    import ctypes as C
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg, max_batch=2)
    imgs = np.stack([G.image_for(cfg, 31), G.image_for(cfg, 32)])
    d_img = torch.from_numpy(imgs).cuda()
    fe.harris_byte_mover_device(d_img.data_ptr(), 2)
    torch.cuda.synchronize()
    out = fe.device_outputs()
    host = np.empty((2, cfg.h, out.score_pitch), dtype=np.int32)
    st = capi.lib().okvfe_copy_to_host(C.c_void_p(host.ctypes.data), C.c_void_p(out.scores),
                                       C.c_size_t(host.nbytes), None)
    assert st == capi.OK
    inner = host[:, 1:-1, :]  # rows 0 and h-1 are the zero rim rows
    assert inner.min() >= 0 and inner.max() <= 255 and inner.max() > 0  # every stored int is a pixel byte
    for i in range(2):
        got = fe.detect(imgs[i])
        ref = oracle.detect(imgs[i], cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
        G.assert_keypoints_equal(got, ref)
    with pytest.raises(capi.OkvfeError):
        G.make_frontend(cfg, score_type=capi.SCORE_AGAST_9_16).harris_byte_mover_device(d_img.data_ptr(), 1)


def test_unaligned_image_pointer_takes_the_dense_fallback(oracle):
    """a batch whose device pointer is not dword aligned is refused by the fused
    score+NMS kernel at RUN time; the unfused pair then writes a dense map into the buffer whose
    layout was chosen slotted at creation.  Selection / sub-pixel / okvfe_get_device_outputs must
    follow the layout that was actually written -- and switch back on the next aligned call."""
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg, max_batch=2)
    fe.set_camera(0, cfg.cams[0])
    fe.set_camera(1, cfg.cams[1])
    imgs = np.stack([G.image_for(cfg, 61), G.image_for(cfg, 62)])
    P = cfg.w * cfg.h
    buf = torch.zeros(2 * P + 8, dtype=torch.uint8, device="cuda")
    grav = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (2, 1))
    cams = np.array([0, 1], np.int32)
    stream = torch.cuda.current_stream().cuda_stream
    want = []
    for ci in range(2):
        rays, jac = oracle.awareness_maps(cfg.cams[ci])
        want.append(oracle.detect_describe(imgs[ci], cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                           oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cfg.cams[ci].fu),
                                           (0.0, 1.0, 0.0)))
    # round 4: aligned single-scale Harris calls write no score map unless asked to; the unfused fallback
    # always does (its NMS kernel reads it)
    for off in (0, 1, 4):
        buf[off:off + 2 * P] = torch.from_numpy(imgs.reshape(-1)).cuda()
        fe.detect_describe_batch_device(buf.data_ptr() + off, 2, cams, grav, stream)
        torch.cuda.synchronize()
        assert (fe.device_outputs().scores is None) == (off % 4 == 0)
        for ci in range(2):
            k, d, _, _ = fe.download(ci)
            G.assert_keypoints_equal(k, want[ci][0])
            assert np.array_equal(d, want[ci][1])
    fe.set_keep_score_map(True)
    for off, strips_expected in ((1, 0), (0, None), (3, 0), (4, None)):
        buf[off:off + 2 * P] = torch.from_numpy(imgs.reshape(-1)).cuda()
        fe.detect_describe_batch_device(buf.data_ptr() + off, 2, cams, grav, stream)
        torch.cuda.synchronize()
        out = fe.device_outputs()
        if strips_expected is not None:
            assert out.score_strips == strips_expected and out.score_pitch == cfg.w
        else:
            assert out.score_strips >= 1 and out.score_pitch > cfg.w
        for ci in range(2):
            k, d, _, _ = fe.download(ci)
            G.assert_keypoints_equal(k, want[ci][0])
            assert np.array_equal(d, want[ci][1])
        # the score map read through the reported layout equals the oracle's
        sc = oracle.harris_score(imgs[1])
        pitch = out.score_pitch
        import ctypes
        host = np.empty(2 * cfg.h * pitch, dtype=np.int32)
        st = capi.lib().okvfe_copy_to_host(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(out.scores),
                                           ctypes.c_size_t(host.nbytes), None)
        assert st == 0
        m = host.reshape(2, cfg.h, pitch)[1]
        cols = np.array([capi.lib().okvfe_score_column(fe._h, int(x)) for x in range(cfg.w)])
        inner = np.s_[3:cfg.h - 3, 3:cfg.w - 3]
        assert np.array_equal(m[:, cols][inner], sc[inner])
