"""The host logic of the replay exchange between its two collectives (agz_gather_plan: the function
comm_allgather_records calls on the gathered {records, bytes} pairs, exported so that a host carrying the bytes
itself uses the same checks and the same chunk stride), and the loud-failure path of the run-time RCCL binding.
CPU only: neither needs an engine.  Caller served: /root/reference/src/train.jl:56-66 across ranks."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import alphago_jl_amd as ag

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = 32          # sizeof(agz_game_header)


def plan(pairs):
    L = ag.load()
    cnt = np.ascontiguousarray(np.asarray(pairs, np.int64).reshape(-1))
    stride, total = C.c_int64(-7), C.c_int64(-7)
    st = L.agz_gather_plan(cnt.ctypes.data_as(C.POINTER(C.c_int64)), len(pairs), C.byref(stride), C.byref(total))
    return st, stride.value, total.value, (L.agz_last_error(None) or b"").decode()


def test_unequal_ranks_pad_to_the_largest():
    st, stride, total, _ = plan([(3, 3 * HDR + 1000), (0, 0), (7, 4096), (1, HDR)])
    assert st == 0 and total == 11
    assert stride == 4096 and stride % 256 == 0
    st, stride, total, _ = plan([(2, 4104), (5, 640)])          # 4104 -> next multiple of 256
    assert (st, stride, total) == (0, 4352, 7)


def test_all_ranks_empty_is_a_valid_exchange_of_nothing():
    st, stride, total, _ = plan([(0, 0)] * 8)
    assert (st, stride, total) == (0, 0, 0)


def test_one_rank_world():
    st, stride, total, _ = plan([(4, 4 * HDR + 8)])
    assert (st, stride, total) == (0, 256, 4)


@pytest.mark.parametrize("bad,who", [
    ([(1, HDR), (2, HDR + 4)], 1),             # bytes not a multiple of 8
    ([(1, HDR), (-3, 64)], 1),                 # negative record count
    ([(1, -8), (1, HDR)], 0),                  # negative byte count
    ([(0, 64), (1, HDR)], 0),                  # bytes without records
    ([(1, HDR), (2, 0)], 1),                   # records without bytes
    ([(1, HDR), (1, HDR), (3, 2 * HDR)], 2),   # fewer bytes than three bare headers
])
def test_inconsistent_announcements_name_the_rank(bad, who):
    st, _, _, msg = plan(bad)
    assert st == ag._lib.RCCL_ERROR and f"rank {who} " in msg, (st, msg)


def test_a_rank_that_failed_before_the_exchange_fails_everybody():
    """the sentinel {-1, status}: the failing rank still joins the count collective, every rank gets the same error
    out of the plan and nobody enters the payload collective"""
    st, _, _, msg = plan([(5, 4096), (-1, ag._lib.HIP_ERROR), (2, 512)])
    assert st == ag._lib.RCCL_ERROR and "rank 1 failed before the exchange" in msg and str(ag._lib.HIP_ERROR) in msg


def test_argument_checks():
    L = ag.load()
    stride = C.c_int64()
    assert L.agz_gather_plan(None, 2, C.byref(stride), None) == ag._lib.BAD_ARGUMENT
    cnt = (C.c_int64 * 2)(0, 0)
    assert L.agz_gather_plan(cnt, 0, C.byref(stride), None) == ag._lib.BAD_ARGUMENT
    assert L.agz_gather_plan(cnt, 1, None, None) == ag._lib.BAD_ARGUMENT
    assert L.agz_gather_plan(cnt, 1, C.byref(stride), None) == 0          # total_records_out may be NULL


def test_missing_rccl_is_a_status_not_a_crash():
    """the not-found path called dlerror() twice and built a std::string from NULL (SIGSEGV).  Force it
    with the soname override in a fresh process (the binding is resolved once per process)."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import ctypes as C, alphago_jl_amd as ag\n"
        "L = ag.load(); idb = (C.c_uint8 * 128)()\n"
        "st = L.agz_comm_unique_id(idb); msg = (L.agz_last_error(None) or b'').decode()\n"
        "st2 = L.agz_comm_unique_id(idb)\n"
        "print(st, st2, msg)\n" % ROOT)
    env = dict(os.environ, AGZ_RCCL_SONAME="libagz_no_such_rccl.so.9")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    st, st2, msg = r.stdout.strip().split(" ", 2)
    assert int(st) == int(st2) == ag._lib.RCCL_ERROR
    assert "libagz_no_such_rccl.so.9 not found" in msg and "cannot open shared object file" in msg
