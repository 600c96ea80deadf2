"""CPU: the C++ schedule of the cross-camera gather (okvfe::cameraOwner / pairSchedule in
okvis2_amd/host/okvfe_cross_camera.hpp) run as TWO concurrent processes, one per rank of a 2-rank
world, against okvis2_amd.multigpu (the Python class drives the same C entry points): every camera
has exactly one owner, every FoV-overlapping pair exactly one matcher rank, slots agree.  The
Hilti 2022 rig's real overlap relation (Frontend.cpp:1990-2000 visits exactly these 9 pairs)."""
import os
import subprocess

import pytest

from okvis2_amd import multigpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tests", "cpp", "cross_camera_cli")
HILTI_PAIRS = [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4)]


def _run_ranks(n, world, pairs):
    bits = "".join("1" if (i, j) in pairs or (j, i) in pairs else "0" for i in range(n) for j in range(n))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "okvis2_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    procs = [subprocess.Popen([CLI, "schedule", str(n), str(world), str(r), bits], env=env,
                              stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    res = []
    for o in outs:
        cams, prs, slots = {}, [], None
        for line in o.splitlines():
            t = line.split()
            if t[0] == "cam":
                cams[int(t[1])] = int(t[3])
            elif t[0] == "pair":
                prs.append((int(t[1]), int(t[2])))
            elif t[0] == "slots":
                slots = int(t[1])
        res.append((cams, prs, slots))
    return res


@pytest.mark.parametrize("n,world,pairs", [(5, 2, HILTI_PAIRS), (5, 1, HILTI_PAIRS), (2, 2, [(0, 1)]),
                                           (5, 4, HILTI_PAIRS), (3, 2, [(0, 2)])])
def test_cpp_schedule_matches_python_and_partitions_the_work(n, world, pairs):
    if not os.path.exists(CLI):
        pytest.skip("tests/cpp/cross_camera_cli not built (run __graft_entry__.build())")
    res = _run_ranks(n, world, set(pairs))
    want = multigpu.pair_schedule(n, lambda i, j: (i, j) in pairs, world)
    seen_cams, seen_pairs = {}, []
    for r, (cams, prs, slots) in enumerate(res):
        assert slots == (n + world - 1) // world
        for c, s in cams.items():
            assert multigpu.camera_owner(c, world) == r and s == c // world
            assert c not in seen_cams
            seen_cams[c] = r
        assert prs == [(i, j) for (i, j, o) in want if o == r]
        seen_pairs += prs
    assert sorted(seen_cams) == list(range(n))
    assert sorted(seen_pairs) == sorted(pairs)
