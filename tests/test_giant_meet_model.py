"""CPU model of td_giant_pieces' first grid barrier (gp_grid_meet, tokendagger_amd/csrc/td_kernels.hip): every workgroup of a launch
must leave it with the same answer although each gives up on its own clock — a workgroup that carried on alone would wait at the next
barrier for workgroups that have gone.  tests/twin/meet_model.cpp restates the protocol (fetch-add on arrival; give up by compare-and-swap
of a DEAD bit against a value that has not reached the target) with host threads and runs arrival patterns in which workgroups come a
millisecond after the others' patience has run out, or right at its edge.  The GPU side: tests/test_gpu_giant_coop.py."""
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_every_workgroup_leaves_the_first_barrier_with_the_same_answer():
    src = ROOT / "tests" / "twin" / "meet_model.cpp"
    out = ROOT / "tests" / "twin" / "_build" / "meet_model"
    out.parent.mkdir(parents=True, exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", str(src), "-o", str(out)])
    for nblk in (2, 3, 4):
        r = subprocess.run([str(out), "600", str(nblk)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        f = dict(zip(r.stdout.split()[0::2], map(int, r.stdout.split()[1::2])))
        assert f["disagreements"] == 0
        assert f["all_true"] + f["all_false"] == f["trials"]
        if len(os.sched_getaffinity(0)) >= 6:  # (with fewer cores than spinning threads nobody ever finds everybody there in time)
            assert f["all_true"] > 10 and f["all_false"] > 10, r.stdout   # both outcomes occur
        # (wrong_true / wrong_false in the output count trials whose outcome is not the one the arrival times suggest: the host's
        # scheduler stalls threads for longer than the model's patience now and then — the agreement is what the protocol owes)
