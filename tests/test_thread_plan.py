"""Per-rank host thread plan of an N-rank bench run (ml2048_amd.shard.host_thread_plan; bench.py applies it and
reports it per rank in config.per_rank): what 8 ranks on one node get at 32 / 64 / 128 / 256 usable CPUs, and that
"auto" never silently selects the schedule drawn on the GPU.  No GPU needed."""
import pytest

from ml2048_amd.shard import host_thread_plan


@pytest.mark.parametrize("cpus,threads", [(32, 1), (64, 4), (128, 9), (256, 9)])
def test_plan_for_eight_ranks(cpus, threads):
    plan = host_thread_plan(cpus, 8)
    assert plan == {"threads": threads, "schedule": "host-async", "cpus_per_rank": cpus // 8}
    # workers + stitcher + Python + runtime helpers fit the rank's share wherever there is room for them at all
    if cpus // 8 >= 5:
        assert 8 * (plan["threads"] + 4) <= cpus


def test_plan_single_rank_box_and_starved_ranks():
    assert host_thread_plan(16, 1) == {"threads": 9, "schedule": "host-async", "cpus_per_rank": 16}   # the 1-GPU pool box
    assert host_thread_plan(8, 1)["threads"] == 4
    assert host_thread_plan(16, 8) == {"threads": 1, "schedule": "host", "cpus_per_rank": 2}          # no room for producers
    assert host_thread_plan(1, 8)["schedule"] == "host"


@pytest.mark.parametrize("req", ["host", "host-async", "device"])
def test_explicit_schedule_is_kept(req):
    for cpus in (8, 64, 256):
        assert host_thread_plan(cpus, 8, req)["schedule"] == req
    assert all(host_thread_plan(c, w)["schedule"] != "device" for c in (1, 4, 16, 64, 256) for w in (1, 2, 4, 8))
