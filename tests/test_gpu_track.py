"""ellc_track_frame (main.cpp:330 + :368 + :499-502 as one device sequence) against the same stages called one by one."""
import os
import subprocess
import numpy as np
import pytest
from egomotion_with_local_loop_closures_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, L = 320, 240, 4


def make_ctx(ellc, pair, diag=False, **kw):
    fx, fy, cx, cy = pair["intrinsics"]
    ctx = ellc.Context(ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2, **kw), diag=diag)
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"])
    st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
    ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
    ctx.frame_upload(0, pair["cur_image"])
    return ctx


@pytest.mark.parametrize("arith", ["exact", "fast"])
@pytest.mark.parametrize("case", [(21, 0.02, 0.05), (22, 0.03, 0.08)])   # 16 iterations; 29: the state-driven schedule needs its continuation
def test_track_frame_equals_the_separate_calls(ellc, oracle, arith, case):
    """... and, on the first frame, the oracle running the same sequence (r05: comparing the fused call with the separate calls alone
    is a self-comparison)."""
    seed, rot, trans = case
    pair = synth.make_pair(W, H, seed=seed, rot=rot, trans=trans)
    kw = dict(arith=ellc.ARITH_FAST) if arith == "fast" else {}
    a = make_ctx(ellc, pair, **kw)
    b = make_ctx(ellc, pair, **kw)
    for rep in range(2):   # twice: the second frame starts from the state the first left
        # one by one
        pose, iters, wgt = a.align([0], [0], save_weights=True)
        seeds = a.depth_seeds()
        pwo = ellc.concatenate_relative_pose(pose[0], np.zeros(6, np.float32))
        a.depth_observe(0, pwo); a.depth_fill_holes(); a.depth_regularize(False); a.depth_update_depth_image()
        # fused
        p2, i2, w2, s2 = b.track_frame(0, save_weights=True)
        assert np.array_equal(p2, pose[0]) and np.array_equal(i2, iters[0]) and w2 == wgt[0] and s2 == seeds
        sa, sb = a.depth_get_state(), b.depth_get_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (rep, k)
        for l in range(L):
            da, va = a.keyframe_depth_level(0, l); db, vb = b.keyframe_depth_level(0, l)
            assert np.array_equal(da, db) and np.array_equal(va, vb)
            assert np.array_equal(a.keyframe_weights(0, l)[0], b.keyframe_weights(0, l)[0])
        if rep == 0:   # against the oracle: the alignment to the metric's tolerance, the depth cycle from the returned pose bit for bit
            from test_gpu_depth import assert_state_equal
            st0 = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
            _, _, _, dm, pose_ref, iters_ref = oracle_track_cycle(oracle, pair, st0, p2, W, H)
            assert float(np.linalg.norm(p2 - pose_ref)) <= 1e-5
            assert np.abs(np.asarray(i2) - np.asarray(iters_ref)).max() <= (0 if arith == "exact" else 1), (i2, iters_ref)
            assert_state_equal(sb, dm.get_state(), "track_frame vs oracle")
    if case[0] == 22:
        assert int(iters.sum()) > 20 or rep   # (the continuation path was exercised on the first frame)
    a.close(); b.close()


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_tracking_call_abandoned_half_way_equals_the_call_that_is_not(ellc, arith):
    """The tracking call's resident launch carries the staging and the count of the valid hypotheses (r06) and, behind it, the depth
    stages read what its block 0 left: a launch abandoned in round r (ellc_debug_persist_delay(0, -r)) and finished by the launch
    path must leave every tracked frame and the whole depth map as the undisturbed call does."""
    pair = synth.make_pair(W, H, seed=21, rot=0.02, trans=0.05)
    kw = dict(arith=ellc.ARITH_FAST) if arith == "fast" else {}
    a = make_ctx(ellc, pair, diag=True, **kw)
    ref = [a.track_frame(0, save_weights=True) for _ in range(3)]
    sa = a.depth_get_state()
    wa = [a.keyframe_weights(0, l) for l in range(L)]   # (saved once per frame: by the selection launch's riders, or by the continuation)
    assert a.debug_persist_counters()[:2] == (3, 0)
    a.close()
    for r in (1, 3, 8, 12):
        b = make_ctx(ellc, pair, diag=True, **kw)
        b.debug_persist_delay(0, -r)
        got = [b.track_frame(0, save_weights=True) for _ in range(3)]
        sb = b.depth_get_state()
        wb = [b.keyframe_weights(0, l) for l in range(L)]
        launches, abandoned, _ = b.debug_persist_counters()
        b.close()
        for l in range(L):
            assert wa[l][1] == wb[l][1] and np.array_equal(wa[l][0], wb[l][0]), (r, l)
        assert launches == 3 and abandoned >= 1, (r, launches, abandoned)
        for x, y in zip(ref, got):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3], r
        for k in sa:
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (r, k)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_eager_lists_change_no_bit_and_every_writer_invalidates_them(ellc, arith):
    """r06: in a tracking context the depth map's export builds the next alignment's compact lists right behind itself (the alignment
    then starts without staging's compaction). Same kernel, same order, same chunks: against a context with that switched off
    (ellc_debug_set_eager_lists(0): every alignment builds its lists itself, as up to r05) not a bit may differ — over tracked
    frames, through the fused call and the separate calls, across a keyframe switch, and after every other writer of the slot's
    planes (the lists built behind the export are then stale and must not be used)."""
    pair = synth.make_pair(W, H, seed=21, rot=0.02, trans=0.05)
    pair2 = synth.make_pair(W, H, seed=23, rot=0.015, trans=0.04)
    kw = dict(arith=ellc.ARITH_FAST) if arith == "fast" else {}
    a = make_ctx(ellc, pair, diag=True, **kw)
    b = make_ctx(ellc, pair, diag=True, **kw)
    b.debug_set_eager_lists(False)
    # (r06, on top: a call whose lists are there has no staging launch either — the resident launch builds the state records and
    # takes the seeds count along; `f` keeps the staging kernel: a third way to the same bits)
    f = make_ctx(ellc, pair, diag=True, **kw)
    f.debug_set_fold_staging(False)
    for ctx in (a, b, f):
        ctx.frame_upload(1, pair2["cur_image"])   # a second frame: calls of two alignments (one keyframe, two frames) take the folded path too

    def same(what):
        ra, rb, rf = a.align([0], [0]), b.align([0], [0]), f.align([0], [0])
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)), what
        assert all(np.array_equal(x, y) for x, y in zip(ra, rf)), (what, "staging kept")
        ra, rb, rf = a.align([0, 0], [0, 1]), b.align([0, 0], [0, 1]), f.align([0, 0], [0, 1])
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)), (what, "two alignments")
        assert all(np.array_equal(x, y) for x, y in zip(ra, rf)), (what, "two alignments, staging kept")
        sa, sb, sf = a.depth_get_state(), b.depth_get_state(), f.depth_get_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (what, k)
            assert np.array_equal(sa[k], sf[k], equal_nan=True), (what, k, "staging kept")
        for slot in (0, 1):   # the saved weights: added by blocks of the selection launch (a, b) or by their own launch (f)
            for l in range(L):
                wa, na = a.keyframe_weights(slot, l)
                for other, tag in ((b, "eager off"), (f, "staging and the weights' launch kept")):
                    wo, no = other.keyframe_weights(slot, l)
                    assert na == no and np.array_equal(wa, wo), (what, slot, l, tag)

    for rep in range(3):                                    # tracked frames: from the second on, a's alignment finds its lists built
        ra, rb, rf = a.track_frame(0, save_weights=True), b.track_frame(0, save_weights=True), f.track_frame(0, save_weights=True)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and ra[2] == rb[2] and ra[3] == rb[3], rep
        assert np.array_equal(ra[0], rf[0]) and np.array_equal(ra[1], rf[1]) and ra[2] == rf[2] and ra[3] == rf[3], (rep, "staging kept")
    launches, abandoned, _ = a.debug_persist_counters()
    assert launches == 3 and abandoned == 0
    same("after three tracked frames")
    for ctx in (a, b, f):                                   # the separate calls (what ellc_main's unfused loop does)
        pose, _, _ = ctx.align([0], [0], save_weights=True)
        pwo = ellc.concatenate_relative_pose(pose[0], np.zeros(6, np.float32))
        ctx.depth_observe(0, pwo); ctx.depth_fill_holes(); ctx.depth_regularize(False); ctx.depth_update_depth_image()
    same("after the separate calls")
    writers = [
        ("keyframe_set_depth_level", lambda c: c.keyframe_set_depth_level(0, 1, *[x.copy() for x in c.keyframe_depth_level(0, 1)])),
        ("keyframe_set_depth", lambda c: c.keyframe_set_depth(0, pair2["depth0"], pair2["var0"])),
        ("update_depth_image again", lambda c: c.depth_update_depth_image()),
        ("keyframe_upload + depth", lambda c: (c.keyframe_upload(0, pair2["kf_image"]), c.keyframe_set_depth(0, pair2["depth0"], pair2["var0"]))),
        ("copy_slot", lambda c: (c.keyframe_upload(1, pair["kf_image"]), c.keyframe_set_depth(1, pair["depth0"], pair["var0"]), c.copy_slot(1, 0, 1, 1))),
        ("single-step API", lambda c: c.gn_iterate(0, 0, 1, np.zeros(6, np.float32))),
        ("export, then tracked frame", lambda c: (c.depth_update_depth_image(), c.track_frame(0, save_weights=True))),
    ]
    for name, fn in writers:
        fn(a); fn(b); fn(f)
        same(name)
    # a keyframe switch: propagate + regularise + export into slot 1, the next frames are aligned against it
    for ctx in (a, b, f):
        ctx.keyframe_upload(1, pair["cur_image"])
        ctx.depth_create_keyframe(1, pair["xi_true"])
        ctx.frame_upload(1, pair["kf_image"])
    for rep in range(2):
        ra, rb, rf = a.align([1], [1], save_weights=True), b.align([1], [1], save_weights=True), f.align([1], [1], save_weights=True)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)), ("new keyframe", rep)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rf)), ("new keyframe", rep, "staging kept")
        ra, rb, rf = a.track_frame(1), b.track_frame(1), f.track_frame(1)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and ra[3] == rb[3], ("new keyframe, tracked", rep)
        assert np.array_equal(ra[0], rf[0]) and np.array_equal(ra[1], rf[1]) and ra[3] == rf[3], ("new keyframe, tracked", rep, "staging kept")
    a.close(); b.close(); f.close()


def oracle_track_cycle(O, pair, st, pose_gpu, Wd, Hd, save_weights=True):
    """The reference's tracked-frame sequence on the CPU oracle (main.cpp:330 alignment with saved weights; :391 / :499-502 update of the
    depth map with the new frame: observeDepthRowParallel, doRegularization = fill holes + regularise, updateDepthImage). The depth
    stages take the pose the GPU returned: the alignments agree to the stated tolerance (1e-5), and from one and the same pose the
    depth stages must agree bit for bit."""
    from helpers import oracle_problem
    cfg, kf, cur, dm = oracle_problem(O, Wd, Hd, L, pair, early_exit=1)
    dm.set_state(st)
    pose_ref, iters_ref, _ = O.align(kf, cur, dm.depth_pyr(), save_weights=save_weights)
    pwo = np.asarray(O.concat_relative(np.asarray(pose_gpu, np.float32), np.zeros(6, np.float32)), np.float32)
    cur.set_pose(origin=pwo, world=pwo)
    dm.set_current(cur)
    dm.observe(); dm.fill_holes(); dm.regularize(False); dm.update_depth_image()
    return cfg, kf, cur, dm, pose_ref, iters_ref


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_c1_tracked_frame_640x480_early_exit_against_the_oracle(ellc, oracle, arith):
    """BASELINE configs[1] as it is worded — one keyframe against one frame, 640x480, 4 levels, Gauss-Newton TO CONVERGENCE (the
    reference's early exit on, ImageFunc.cpp:251-252) plus the depth cycle — through ONE ellc_track_frame call, against the oracle
    running the same sequence: iteration counts, pose <= 1e-5, every field of every pixel of the depth map, the exported depth /
    variance pyramid, the saved weights; then createKeyFrame onto the tracked frame (main.cpp:404-436)."""
    from test_gpu_depth import assert_state_equal
    from helpers import bits_equal
    Wd, Hd = 640, 480
    pair = synth.make_pair(Wd, Hd, seed=77, rot=0.006, trans=0.03)
    st = synth.make_depth_state(Wd, Hd, 9, pair["kf_image"], pair["idepth_true"])
    fx, fy, cx, cy = pair["intrinsics"]
    kw = dict(arith=ellc.ARITH_FAST) if arith == "fast" else {}
    ctx = ellc.Context(ellc.default_config(Wd, Hd, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2, **kw))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"])
    ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
    ctx.frame_upload(0, pair["cur_image"])
    pose, iters, wgt, seeds = ctx.track_frame(0, save_weights=True)
    cfg, kf, cur, dm, pose_ref, iters_ref = oracle_track_cycle(oracle, pair, st, pose, Wd, Hd)
    # the alignment: to convergence, the oracle's iteration counts (the tolerance mode's termination test sits at the scale of its
    # tolerance: a count may differ by one there), the pose within the metric's bar
    assert int(np.sum(iters_ref)) < 32, "the scene must converge before the caps: early exit is what this test is about"
    if arith == "exact":
        assert list(iters) == list(iters_ref), (iters, iters_ref)
    else:
        assert np.abs(np.asarray(iters) - np.asarray(iters_ref)).max() <= 1, (iters, iters_ref)
    assert float(np.linalg.norm(pose - pose_ref)) <= 1e-5
    # the depth cycle from the same pose: bit for bit
    assert_state_equal(ctx.depth_get_state(), dm.get_state(), "tracked frame: observe + fill + regularise + border")
    for l in range(L):
        d_ref, v_ref = dm.pyr_level(l)
        d, v = ctx.keyframe_depth_level(0, l)
        if l == 0:
            d_ref = np.where(d_ref < 0, 0, d_ref)   # arrays hold -1, the Mat (what the tracker reads) holds 0
        assert bits_equal(d, d_ref) and bits_equal(v, v_ref), l
    # saved weights of the last executed iteration of every level (PixelWisePyramid.cpp:544-549): same pixels, values to the tolerance
    # the poses agree to (exact mode: the per-pixel weights are bit-identical for identical poses; the poses differ in their last bits)
    if list(iters) == list(iters_ref):
        for l in range(L):
            w_ref, n_ref = kf.weights(l)
            w_gpu, n_gpu = ctx.keyframe_weights(0, l)
            assert n_gpu == n_ref
            assert np.array_equal(w_gpu > 0, w_ref > 0) or np.mean((w_gpu > 0) != (w_ref > 0)) < 1e-4, l
            assert np.allclose(w_gpu, w_ref, rtol=2e-3, atol=1e-6), (l, float(np.abs(w_gpu - w_ref).max()))
    # createKeyFrame onto the tracked frame (propagate, regularise x2, fill, rescale, export); the rescale's sum order differs (5e-5)
    newkf = oracle.Frame(cfg, pair["cur_image"], 5)
    pwo = np.asarray(oracle.concat_relative(pose, np.zeros(6, np.float32)), np.float32)
    newkf.set_pose(origin=pwo)
    dm.create_keyframe(newkf)
    ctx.keyframe_from_frame(1, 0)
    ctx.depth_create_keyframe(1, pwo)
    ref, got = dm.get_state(), ctx.depth_get_state()
    assert np.array_equal(got["valid"], ref["valid"]) and np.array_equal(got["blacklisted"], ref["blacklisted"])
    m = ref["valid"] != 0
    for fld in ("invDepth", "invDepthSmoothed", "variance", "varianceSmoothed"):
        assert np.allclose(got[fld][m], ref[fld][m], rtol=2e-4), fld
    for l in range(L):
        d_ref, v_ref = dm.pyr_level(l)
        d, v = ctx.keyframe_depth_level(1, l)
        if l == 0:
            d_ref = np.where(d_ref < 0, 0, d_ref)
        assert np.allclose(d, d_ref, rtol=2e-4) and np.allclose(v, v_ref, rtol=4e-4), l
    ctx.close()


def test_track_frame_errors(ellc):
    pair = synth.make_pair(W, H, seed=3)
    fx, fy, cx, cy = pair["intrinsics"]
    ctx = ellc.Context(ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=2))
    with pytest.raises(ellc.EllcError):
        ctx.track_frame(0)           # no depth map yet
    ctx.close()


def test_driver_fused_and_unfused_write_the_same_files(tmp_path):
    """ellc_main tracks through ellc_track_frame (--fused, the default); with --no-fused every stage is its own call: identical files."""
    from test_gpu_driver import make_sequence, W as DW, H as DH, N
    frames, _ = make_sequence()
    raw = tmp_path / "frames.raw"
    raw.write_bytes(b"".join(np.ascontiguousarray(f, np.uint8).tobytes() for f in frames))
    exe = os.path.join(ROOT, "egomotion_with_local_loop_closures_amd", "csrc", "ellc_main")
    outs = []
    for name, extra in (("fused", ["--fused"]), ("unfused", ["--no-fused"])):
        d = tmp_path / name; d.mkdir()
        r = subprocess.run([exe, str(raw), str(DW), str(DH), str(N), str(d), "LC"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode()
        outs.append(((d / "poses_orig.txt").read_text(), (d / "matchframes.txt").read_text()))
    assert outs[0] == outs[1] and len(outs[0][0].strip().split("\n")) == N - 1
