"""executors.pin_process_to_compact_cpus: the host-side helper bench.py calls before torch / HIP start their threads."""
import os

import pytest

from multiview_stitcher_amd import executors


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="Linux only")
def test_pin_process_to_compact_cpus(monkeypatch):
    before = sorted(os.sched_getaffinity(0))
    try:
        monkeypatch.setenv("MVS_PIN_CPUS", "0")
        assert executors.pin_process_to_compact_cpus() == before                  # switched off: nothing changes
        monkeypatch.delenv("MVS_PIN_CPUS")
        got = executors.pin_process_to_compact_cpus(slot=0, n_cpus=len(before))    # a mask narrower than two blocks is left alone
        assert got == before
        if len(before) >= 16:
            got = executors.pin_process_to_compact_cpus(slot=1, n_cpus=8)
            assert got == before[8:16]
            os.sched_setaffinity(0, before)
        monkeypatch.setenv("MVS_PIN_CPUS", f"{before[0]}-{before[0]}")        # ("0" alone means: off)
        assert executors.pin_process_to_compact_cpus() == [before[0]]
    finally:
        os.sched_setaffinity(0, before)
    q = executors.cpu_quota_cores()
    assert q is None or q > 0
