"""Host-side logic of progressivecodec_amd.pipeline that needs no GPU: the hardware-queue request made at import time and the interval
fold behind bench.py's in-schedule roofline figure."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fold_intervals_union_sum_window_and_concurrency():
    from progressivecodec_amd.pipeline import fold_intervals
    assert fold_intervals([], []) == {"busy_ms": 0.0, "sum_ms": 0.0, "window_ms": 0.0, "mean_in_flight": 0.0}
    f = fold_intervals([0, 1, 5, 5.5], [2, 3, 6, 5.7])                      # [0,3) and [5,6): busy 4 of a window of 6
    assert (f["busy_ms"], f["window_ms"]) == (4.0, 6.0) and abs(f["sum_ms"] - 5.2) < 1e-12 and abs(f["mean_in_flight"] - 1.3) < 1e-12
    g = fold_intervals([10, 10, 10, 10], [11, 11, 11, 11])                   # four kernels side by side
    assert g["busy_ms"] == 1.0 and g["sum_ms"] == 4.0 and g["mean_in_flight"] == 4.0
    h = fold_intervals([3, 0, 1], [4, 0.5, 2])                               # unsorted input, disjoint
    assert h["busy_ms"] == 2.5 and h["window_ms"] == 4.0
    n = fold_intervals([0, 0.2], [1, 0.4])                                   # nested
    assert n["busy_ms"] == 1.0 and abs(n["sum_ms"] - 1.2) < 1e-12


def _import_env(extra):
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env.update(extra)
    code = ("import sys, os; sys.path.insert(0, %r); import progressivecodec_amd; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), 'torch' in sys.modules)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout.split()


def test_package_import_requests_hardware_queues_without_touching_the_gpu_runtime():
    """`import progressivecodec_amd` puts GPU_MAX_HW_QUEUES=16 into the environment when the variable is unset (the HIP runtime reads it when
    it initialises: the overlapped schedule keeps ~20 streams busy), leaves a caller's own setting alone, and imports neither torch nor HIP --
    so the N > 1 launcher's parent may import the package (bench.launch_ranks does)."""
    assert _import_env({}) == ["16", "False"]
    assert _import_env({"GPU_MAX_HW_QUEUES": "4"}) == ["4", "False"]
    from progressivecodec_amd.pipeline import request_hw_queues
    old = os.environ.pop("GPU_MAX_HW_QUEUES", None)
    try:
        assert request_hw_queues(12) is True and os.environ["GPU_MAX_HW_QUEUES"] == "12"        # (HIP not started in the CPU test process)
        assert request_hw_queues(16) is True and os.environ["GPU_MAX_HW_QUEUES"] == "12"        # an existing setting is kept
        os.environ["GPU_MAX_HW_QUEUES"] = "not-a-number"
        assert request_hw_queues() is False
    finally:
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
        if old is not None:
            os.environ["GPU_MAX_HW_QUEUES"] = old
