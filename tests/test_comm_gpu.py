"""The one collective of the multi-GPU path (lps_comm_*: ncclBroadcast of the packed SNP table, RCCL).  The test box has ONE GPU, so the
communicator has one rank: this checks that librccl is found and bound, that a communicator comes up through both entry points and that a
broadcast moves the bytes it is given; the multi-rank run is the driver's 8-GPU bench (bench.py --gpus N)."""
import ctypes as C

import numpy as np
import pytest

from lps import hip

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_and_broadcast():
    L = hip.load()
    uid = (C.c_uint8 * 128)()
    assert L.lps_comm_unique_id(uid) == 0, L.lps_comm_last_error()
    comm = L.lps_comm_create(0, 1, 0, uid)
    assert comm, L.lps_comm_last_error()
    assert L.lps_comm_size(comm) == 1 and L.lps_comm_rank(comm) == 0
    buf = np.arange(1 << 20, dtype=np.uint8)
    want = buf.copy()
    ms = C.c_double(-1)
    assert L.lps_comm_bcast(comm, buf.ctypes.data, buf.size, 0, C.byref(ms)) == 0, L.lps_comm_last_error()
    assert np.array_equal(buf, want) and ms.value >= 0
    assert L.lps_comm_bcast(comm, buf.ctypes.data, buf.size, 3, None) != 0          # root outside the communicator
    L.lps_comm_destroy(comm)


def test_create_all_one_device():
    L = hip.load()
    devs = (C.c_int * 1)(0)
    comms = (C.c_void_p * 1)()
    assert L.lps_comm_create_all(1, devs, comms) == 0, L.lps_comm_last_error()
    assert L.lps_comm_size(comms[0]) == 1
    L.lps_comm_destroy(comms[0])
