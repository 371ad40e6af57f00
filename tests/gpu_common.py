"""helpers shared by the -m gpu tests"""
import ctypes as C

import numpy as np

import alphago_jl_amd as ag
import orc

L = orc.lib()


def pos_soa(positions):
    """oracle positions -> the SoA the C ABI takes (boards, deltas newest-first, ndeltas, to_play)"""
    N = positions[0].N
    P = N * N
    B = len(positions)
    boards = np.zeros((B, P), np.int8)
    deltas = np.zeros((B, 7, P), np.int8)
    nd = np.zeros(B, np.int32)
    tp = np.zeros(B, np.int8)
    for b, p in enumerate(positions):
        boards[b] = np.frombuffer(p.board, dtype=np.int8, count=P)
        nd[b] = p.ndeltas
        for k in range(p.ndeltas):
            deltas[b, k] = np.frombuffer(p.deltas[k], dtype=np.int8, count=P)
        tp[b] = p.to_play
    return boards, deltas, nd, tp


def copy_weights_from_oracle(engine, onet, tower):
    """push the oracle network's parameters through agz_net_set_weights"""
    layers = list(range(0, 1 + 2 * tower)) + [orc.L_VALUE_CONV, orc.L_POLICY_CONV]
    for l in layers:
        for kind in range(7):
            n = L.or_net_param_count(onet, l, kind)
            buf = np.zeros(n, np.float32)
            assert L.or_net_get(onet, l, kind, orc.fptr(buf), n) == 0
            engine.set_weights(l, kind, buf)
    for l in (orc.L_VALUE_FC1, orc.L_VALUE_FC2, orc.L_POLICY_FC):
        for kind in (0, 1):
            n = L.or_net_param_count(onet, l, kind)
            buf = np.zeros(n, np.float32)
            assert L.or_net_get(onet, l, kind, orc.fptr(buf), n) == 0
            engine.set_weights(l, kind, buf)


class GpuNetForOracle:
    """an or_net_fn whose network is the HIP forward (agz_net_forward) -- lets the oracle's tree
    search run on exactly the numbers the engine's own search sees"""

    def __init__(self, engine):
        self.engine = engine
        self.calls = 0

        def _fn(ctx, positions, B, pi, v):
            self.calls += 1
            plist = [positions[b].contents for b in range(B)]
            gpi, gv = engine.forward(*pos_soa(plist))
            A = engine.A
            C.memmove(pi, gpi.ctypes.data, 4 * B * A)
            C.memmove(v, gv.ctypes.data, 4 * B)

        self.cb = orc.NET_FN(_fn)
