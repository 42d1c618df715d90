"""CPU stand-ins for the local kernels of the four-step transform (test infrastructure; uses the oracle)."""
import numpy as np


class OracleOps:
    """CPU stand-ins (oracle) for the local kernels of the four-step transform, written from the specification in
    include/zkp_hip.h (zkp_ntt_fr_axis0_dev, zkp_ntt_fr_layout_dev): the data flow, the layouts, the twiddle indexing and the
    all-to-all exchanges of zkp_hip/dist.py are what these tests exercise."""

    def __init__(self, orc):
        self.orc = orc

    def _tw(self, log_n, inverse, exps):
        import bigmodel as M
        w = M.root_of_unity(log_n)
        if inverse:
            w = pow(w, -1, M.R)
        return self.orc.fr_from_ints([pow(w, int(e), M.R) for e in exps])

    def ntt_batch(self, t, log_len, batch, inverse):
        import torch
        a = t.numpy().view(np.uint64).reshape(batch, 1 << log_len, 4)
        out = np.stack([self.orc.ntt_fr(a[b], inverse=inverse) for b in range(batch)])
        return torch.from_numpy(out.view(np.int64)).reshape(t.shape)

    def axis0(self, src, dst, log_len, cols, inverse, tw_log_n, col0):
        import torch
        L = 1 << log_len
        a = src.numpy().view(np.uint64).reshape(L, cols, 4)
        out = np.empty_like(a)
        for b in range(cols):
            col = self.orc.ntt_fr(np.ascontiguousarray(a[:, b]), inverse=inverse)
            if tw_log_n:
                col = self.orc.fr_mul(col, self._tw(tw_log_n, inverse, [(col0 + b) * k for k in range(L)]))
            out[:, b] = col
        dst.reshape(-1).copy_(torch.from_numpy(out.view(np.int64)).reshape(-1))

    @staticmethod
    def _phys(layout, n, b, e):
        if layout is None:
            return b * n + e
        lo_bits, mid_bits, mid_stride, hi_stride, batch_stride = layout
        lo, rest = e & ((1 << lo_bits) - 1), e >> lo_bits
        return b * batch_stride + lo + (rest & ((1 << mid_bits) - 1)) * mid_stride + (rest >> mid_bits) * hi_stride

    def layout(self, src, dst, log_n, batch, inverse, in_layout=None, out_layout=None, tw_log_n=0, tw_row0=0):
        import torch
        n = 1 << log_n
        a = src.numpy().view(np.uint64).reshape(-1, 4)
        out = dst.numpy().view(np.uint64).reshape(-1, 4).copy()
        for b in range(batch):
            v = np.stack([a[self._phys(in_layout, n, b, e)] for e in range(n)])
            v = self.orc.ntt_fr(v, inverse=inverse)
            if tw_log_n:
                v = self.orc.fr_mul(v, self._tw(tw_log_n, inverse, [(tw_row0 + b) * k for k in range(n)]))
            for k in range(n):
                out[self._phys(out_layout, n, b, k)] = v[k]
        dst.reshape(-1).copy_(torch.from_numpy(out.view(np.int64)).reshape(-1))
