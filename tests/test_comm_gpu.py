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


def test_broadcast_table_stays_on_the_device_and_feeds_a_context():
    """lps_comm_bcast_to_device + lps_set_variants_device: the packed SNP table goes collective -> context without a host hop, and the phase
    result from that table equals the one from the host table (and the oracle's)."""
    import lps_oracle
    from lps import abi
    from lps.synth import Synth
    L = hip.load()
    s = Synth(seed=11, contig_len=400_000, n_snp=500, coverage=14.0, n_threads=2)
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt)
    R = abi.Reads.from_synth(s)
    P = abi.default_params()
    n = V.n
    packed = np.concatenate([np.ascontiguousarray(V.pos, np.int32).view(np.uint8), np.ascontiguousarray(V.ref0, np.uint8), np.ascontiguousarray(V.alt0, np.uint8)])
    uid = (C.c_uint8 * 128)()
    assert L.lps_comm_unique_id(uid) == 0
    comm = L.lps_comm_create(0, 1, 0, uid)
    assert comm, L.lps_comm_last_error()
    dptr = C.c_void_p(0); ms = C.c_double(-2)
    assert L.lps_comm_bcast_to_device(comm, packed.ctypes.data, packed.size, 0, C.byref(dptr), C.byref(ms)) == 0, L.lps_comm_last_error()
    assert dptr.value and ms.value >= 0
    base = int(dptr.value)
    with hip.Context(0, P) as ctx:
        want = ctx.phase(V, s.ref, R)
        # same reads, the table taken from the communicator's device buffer
        assert L.lps_begin_chromosome(ctx.h) == 0
        t = abi.VariantTable(); t.n = n; t.pos, t.ref0, t.alt0 = base, base + 4 * n, base + 5 * n
        assert L.lps_set_variants_device(ctx.h, C.byref(t)) == 0, L.lps_last_error(ctx.h)
        ref = np.ascontiguousarray(s.ref, dtype=np.uint8)
        assert L.lps_set_reference(ctx.h, ref.ctypes.data, ref.size) == 0
        assert L.lps_push_reads(ctx.h, C.byref(R.c)) == 0
        ctx.n_var = n; ctx.n_reads = R.n_reads
        got = ctx.run_phase(abi.PhaseOut(n))
        assert np.array_equal(got.phase_set, want.phase_set) and np.array_equal(got.gt, want.gt)
        # a table whose positions are not strictly increasing is refused by the device-side check
        bad = packed.copy(); bad[:8] = bad[8:16]
        assert L.lps_comm_bcast_to_device(comm, bad.ctypes.data, bad.size, 0, C.byref(dptr), None) == 0
        assert L.lps_begin_chromosome(ctx.h) == 0
        t.pos, t.ref0, t.alt0 = int(dptr.value), int(dptr.value) + 4 * n, int(dptr.value) + 5 * n
        assert L.lps_set_variants_device(ctx.h, C.byref(t)) != 0
        assert b"strictly increasing" in L.lps_last_error(ctx.h)
    ref_out, _ = lps_oracle.phase(P, V, s.ref, R)
    assert np.array_equal(want.phase_set, ref_out.phase_set)
    L.lps_comm_destroy(comm)
