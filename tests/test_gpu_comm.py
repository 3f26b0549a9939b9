"""The RCCL entry points on one rank of real librccl (SURVEY 8 e): gather with counts, the host-side timeout, the teardown paths.  Several
ranks: test_gpu_multirank.py (stand-in library), test_parallel_gloo.py (CPU).  (Folded from the round-3 file in round 6: the test is unchanged.)"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import REPO
from helpers import _api_modules, _records, default_cache, default_lists, make_tc, params_from_tc
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.params import affine_params_from_tc

pytestmark = pytest.mark.gpu


def test_gatherv_timeout_and_the_fixed_teardown_paths():
    """Round-3 entry points on a one-rank communicator: klt_gatherv_featbuf_async (a count per rank), klt_comm_set_timeout (a wait
    that cannot hang), klt_comm_destroy followed by a fence on a buffer that took part in a collective (ADVICE: the buffer kept an
    event of the destroyed communicator), a table regrown while its gather may still run (ADVICE: sync_all now waits for the side
    stream), klt_featbuf_upload_async and klt_affine_copy_async."""
    import ctypes as C
    from pyfeaturetrack_amd._abi import load_library
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    lib = load_library()
    c = Context(0)
    try:
        with pytest.raises(KltBackendError, match="klt_comm_init_rank"):
            c.gatherv_featbuf_async(0, 1, [4], 0)
        with pytest.raises(KltBackendError, match="klt_comm_init_rank"):
            c.comm_set_timeout(10.0)
        uid = (C.c_uint8 * 128)()
        assert lib.klt_comm_unique_id(uid) == 0
        c.comm_init(1, 0, bytes(uid))
        n = 2000
        fl = np.zeros(n, FEAT_DTYPE)
        fl["x"], fl["y"], fl["val"] = np.arange(n), -np.arange(n), np.arange(n) % 5 - 2
        rin, rout = c.host_records(n)
        rin[:] = fl
        assert lib.klt_featbuf_upload_async(c._h, 0, rin.ctypes.data, n) == 0
        c.gatherv_featbuf_async(0, 1, [n], 0)
        c.comm_set_timeout(5000.0)
        c.comm_wait()                                       # completes: no timeout
        assert np.array_equal(c.featbuf_download(1, n), fl)
        with pytest.raises(KltBackendError, match="count"):
            c.gatherv_featbuf_async(0, 1, [-1], 0)
        with pytest.raises(KltBackendError, match="gatherv"):
            c.gatherv_featbuf_async(0, 1, [n], 2)
        # a gather into a table that has to grow while an earlier, smaller gather into it may still be running
        for k in range(20):
            c.gather_featbuf_async(0, 2, 100 + 90 * k, root=0)
        assert np.array_equal(c.featbuf_download(2, 1810), fl[:1810])
        # destroy, then fence a buffer that took part: no stale event is waited on
        c.comm_destroy()
        c.comm_fence_featbuf(0)
        c.comm_fence_featbuf(1)
        c.featbuf_upload(0, fl)
        assert np.array_equal(c.featbuf_download(0, n), fl)
        # a second communicator on the same context works (buffers keep no event of the first)
        assert lib.klt_comm_unique_id(uid) == 0
        c.comm_init(1, 0, bytes(uid))
        c.allgather_featbuf_async(0, 3, n)
        assert np.array_equal(c.featbuf_download(3, n), fl)
        c.comm_destroy()
        # affine state snapshot
        from helpers import make_tc
        tc = make_tc(levels=2, ss=2)
        tc.affineConsistencyCheck = 2
        c.configure(tc)
        c.affine_alloc(0, 64)
        c.affine_copy(1, 0, 64, with_templates=True)
        a, b = c.affine_download(0, 64), c.affine_download(1, 64)
        assert np.array_equal(a, b) and np.all(a["Axx"] == 1) and np.all(a["valid"] == 0)
        with pytest.raises(KltBackendError):
            c.affine_copy(2, 7, 64)
    finally:
        c.close()
