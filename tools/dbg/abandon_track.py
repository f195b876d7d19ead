import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from egomotion_with_local_loop_closures_amd import api as ellc, synth
import test_gpu_track as T
pair = synth.make_pair(T.W, T.H, seed=21, rot=0.02, trans=0.05)
for sw in (True, False):
    for eager in (1, 0):
        for r in (0, 1):
            b = T.make_ctx(ellc, pair, diag=True)
            b.debug_set_eager_lists(bool(eager))
            if r: b.debug_persist_delay(0, -r)
            out = [b.track_frame(0, save_weights=sw) for _ in range(2)]
            print("save_weights", sw, "eager", eager, "abandon round", r, "counters", b.debug_persist_counters())
            for o in out: print("   pose", np.round(o[0], 6), "iters", o[1], "wgt %.4f seeds %.3f" % (o[2], o[3]))
            b.close()
# the five-call form (align + separate depth calls) with the abandon hook
for r in (0, 1):
    a = T.make_ctx(ellc, pair, diag=True)
    if r: a.debug_persist_delay(0, -r)
    pose, iters, wgt = a.align([0], [0], save_weights=True)
    print("five-call align, abandon round", r, np.round(pose[0], 6), iters[0], a.debug_persist_counters())
    a.close()
