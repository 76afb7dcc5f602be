"""Test infrastructure: N ranks of the multi-GPU count exchange emulated on the CPU.

The exchange of libcobs_gpu.so is a host-computed plan (comm.cpp: plan_exchange, exported as
cobs_gpu_exchange_plan) executed over RCCL.  Here the plans of all N ranks are played with memcpy:
sends / receives between the ranks' local count rows and staging buffers, then the strided
assembly copies.  Checked on the way: every send size equals the peer's receive size (a mismatch
would hang or corrupt a real ncclSend / ncclRecv pair), all ranks pick the same collective, the
assembly writes every byte of the owned rows exactly once."""
import ctypes as C

import numpy as np


def exchange_plan(lib, begins, counts, doc_off, total, nq, eb, mode, rank):
    from cobs_amd import _capi
    N, F = len(begins), len(begins[0])
    b = (C.c_uint64 * (N * F))(*[x for r in begins for x in r])
    c = (C.c_uint64 * (N * F))(*[x for r in counts for x in r])
    d = (C.c_uint64 * F)(*doc_off)
    xf = (_capi.Xfer * N)()
    cp = (_capi.Copy2D * (N * F))()
    ncp = C.c_size_t(N * F)
    out = (C.c_uint64 * 6)()
    _capi.check(lib.cobs_gpu_exchange_plan(b, c, d, N, F, total, nq, eb, mode, rank, xf, cp, C.byref(ncp), out))
    return list(xf), list(cp)[:ncp.value], list(out)


def emulate(lib, local, begins, counts, doc_off, total, nq, eb, mode):
    """local[r]: rank r's count rows as the scan leaves them ([nq][local slots], flat uint8 view);
    begins / counts: [rank][file] slot layouts.  -> per rank (q_begin, q_count, assembled rows as flat
    uint8): the rows of its queries in global document order."""
    N = len(local)
    plans = [exchange_plan(lib, begins, counts, doc_off, total, nq, eb, mode, r) for r in range(N)]
    assert len({p[2][4] for p in plans}) == 1                      # same collective everywhere
    res = []
    for i in range(N):
        xf, cps, out = plans[i]
        q0, qn, staging_bytes, global_bytes, use_ag, my_row = out
        staging = np.zeros(staging_bytes, dtype=np.uint8)
        for j in range(N):
            if use_ag:
                # ncclAllGather: every rank's nq * row bytes land rank after rank
                assert xf[j].recv_bytes == nq * my_row and plans[j][2][5] == my_row
                staging[xf[j].recv_offset: xf[j].recv_offset + xf[j].recv_bytes] = local[j][:nq * my_row]
                continue
            if j == i:
                assert xf[j].send_bytes == 0 and xf[j].recv_bytes == 0
                continue
            peer = plans[j][0][i]                                   # what j sends to i
            assert peer.send_bytes == xf[j].recv_bytes, (N, i, j)   # ncclSend / ncclRecv sizes agree
            staging[xf[j].recv_offset: xf[j].recv_offset + xf[j].recv_bytes] = \
                local[j][peer.send_offset: peer.send_offset + peer.send_bytes]
        got = np.full(global_bytes, 0xAB, dtype=np.uint8)
        written = np.zeros(global_bytes, dtype=np.uint8)
        for c in cps:
            src = local[i] if c.src_is_local else staging
            if c.width == c.src_pitch == c.dst_pitch:
                n = c.width * c.height
                got[c.dst_offset:c.dst_offset + n] = src[c.src_offset:c.src_offset + n]
                written[c.dst_offset:c.dst_offset + n] += 1
                continue
            sv = np.lib.stride_tricks.as_strided(src[c.src_offset:], (c.height, c.width), (c.src_pitch, 1))
            np.lib.stride_tricks.as_strided(got[c.dst_offset:], (c.height, c.width), (c.dst_pitch, 1))[:] = sv
            np.lib.stride_tricks.as_strided(written[c.dst_offset:], (c.height, c.width), (c.dst_pitch, 1))[:] += 1
        assert (written == 1).all(), (N, i, mode)                  # every byte exactly once
        res.append((q0, qn, got))
    return res
