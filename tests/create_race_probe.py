"""Child process of tests/test_gpu_stage1.py::test_create_waits_for_its_own_fills (GPU box): a thread keeps the NULL stream busy with short spins
(at most two queued) while a context is created and two frames are staged; then the stream drains and the pass runs.  Prints one line per
attempt and `bad: <n>`.  The library before round 5's fix (se3d_create_ex waits for its null-stream fills) prints bad: 4 of 5."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smartedgesensor3dhumanpose_amd import rig as R, scene as S  # noqa: E402
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D, rig_tensors  # noqa: E402

rig = R.extr_rig_16cam()
rig.set_tensors(*rig_tensors(rig))
frames, _ = S.generate_sequence(rig, S.SceneConfig(n_persons=6, n_frames=2, seed=11))
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
SPIN = 12_000_000   # ~6 ms on torch's default (= the null) stream
bad = 0
for attempt in range(5):
    staged = threading.Event()

    def spinner():
        t_end, evs = time.time() + 3.0, []
        while not staged.is_set() and time.time() < t_end:
            if len(evs) >= 2:
                while not evs[0].query() and time.time() < t_end:
                    time.sleep(0.0002)
                evs.pop(0)
            torch.cuda._sleep(SPIN)
            e = torch.cuda.Event()
            e.record()
            evs.append(e)
    th = threading.Thread(target=spinner)
    th.start()
    time.sleep(0.05)
    t0 = time.time()
    node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=2, max_det_per_cam=8, max_hyp=32)   # its fills queue behind the spins
    t1 = time.time()
    node.stage(frames)                                                                        # upload on the context's own stream
    staged.set()
    th.join()
    torch.cuda.synchronize()                                                                  # the null stream drains: late fills would land now
    node.run_stage1()
    res = node.fetch_stage1(len(frames))
    node.close()
    ok = all(g.flags == 0 and g.n_persons >= 5 for g in res)
    bad += not ok
    print(attempt, "create %.3f s" % (t1 - t0), [(g.flags, g.n_persons) for g in res], "OK" if ok else "BAD", flush=True)
print("bad:", bad)
