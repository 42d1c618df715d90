"""CPU stand-in for the HIP library, for a DRY RUN of bench.py's own rank logic without a GPU (tests/test_bench_dryrun.py).

Test infrastructure, never a product path: bench.py loads this module only when ZKP_BENCH_DRYRUN names it, marks the line it prints
`"dry_run": true` with a metric string that says so, and no number in that line is a measurement.  What the dry run exercises is
everything AROUND the kernels in an N-rank run -- shard ranges, seeds per chunk, the all-gather of the partial sums and the EC-add
combine, the trapdoor check over all ranks' limb sums, the four-step data flow with its all-to-alls, `one_gpu_reference` next to
rank 0's shard, the strong-scaling arithmetic of --total-log-n, the JSON merge, the emit-once logic -- at world 8, which no box of
this pool can run on GPUs (six processes per card).  The kernels are replaced by the oracle (oracle/) and by the CPU statements of
zkp_ntt_fr_axis0_dev / zkp_ntt_fr_layout_dev in tests/oracle_ops.py; host-side entries (zkp_g1_mul, zkp_g1_xyzz_sum) are the real
library's, which needs no device for them."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "model"), os.path.join(ROOT, "zkp-implementation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _fq_one_mont():
    m = (1 << 384) % P_MOD
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], dtype=np.uint64)


def install(torch):
    """Patch the few torch.cuda calls bench.py makes and return the stand-in `zkp` module object."""
    from oracle import oracle as orc
    from oracle_ops import OracleOps
    import zkp_hip as real
    orc.build()
    ops = OracleOps(orc)
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 8
    torch.cuda.set_device = lambda *_a, **_k: None
    torch.cuda.synchronize = lambda *_a, **_k: None
    torch.cuda.empty_cache = lambda *_a, **_k: None
    torch.cuda.get_device_properties = lambda *_a, **_k: types.SimpleNamespace(multi_processor_count=256)

    def h(t, cols):
        return np.ascontiguousarray(t.detach().cpu().numpy()).view(np.uint64).reshape(-1, cols)

    class G1Bases:
        def __init__(self, pts):
            self.pts, self.bits, self.planes = pts, 0, 0

        @classmethod
        def from_device(cls, xy_tensor, n, is_inf_tensor=None, stream=None):
            return cls(h(xy_tensor, 12)[:n].copy())

        def precompute(self, window_bits=0):
            n = len(self.pts)
            if window_bits == 0:  # the library's automatic widths (api.hip: auto_window_bits)
                window_bits = 0 if n < 64 else 22 if n >= 1 << 22 else 20 if n >= 1 << 19 else 16 if n > 1 << 13 else 14 if n > 1 << 11 else 12
            self.bits = window_bits
            self.planes = -(-256 // window_bits) if window_bits else 0
            return self

        def info(self):
            return self.bits, self.planes

        def shards(self):
            return [(0, 0, 0, len(self.pts))]

        def __len__(self):
            return len(self.pts)

        def close(self):
            self.pts = None

    z = types.SimpleNamespace()
    z.G1Bases = G1Bases
    z.ZkpError = real.ZkpError
    z.CIRCUIT_POLYS = real.CIRCUIT_POLYS
    z.NTT_NATURAL, z.NTT_K1SLAB, z.NTT_COLUMNS = real.NTT_NATURAL, real.NTT_K1SLAB, real.NTT_COLUMNS
    z.init = lambda *_a, **_k: None
    z.shutdown = lambda: None
    z.profile_reset = lambda: None
    z.profile_enable = lambda *_a, **_k: None
    z.profile_read = lambda _name: (0.0, 0)

    def no_clock(_name):
        raise real.ZkpError(real.ZKP_E_DEVICE, "dry run: no device, no clock stamps")
    z.profile_clock_read = no_clock
    z.probe_mad_rate = lambda *_a, **_k: (0.0, 0.0, 0.0)
    z.g1_mul = real.g1_mul            # host code of the real library
    z.g1_xyzz_sum = real.g1_xyzz_sum  # host code of the real library

    def g1_fixed_base_mul_dev(scalars_tensor, n, out_xy_tensor, out_inf_tensor=None, stream=None):
        xy, _inf = orc.g1_fixed_base_mul(h(scalars_tensor, 4)[:n])
        out_xy_tensor.reshape(-1)[: 12 * n].copy_(torch.from_numpy(np.ascontiguousarray(xy).view(np.int64).reshape(-1)))
    z.g1_fixed_base_mul_dev = g1_fixed_base_mul_dev

    def msm_g1_dev(bases, scalars_tensor, n, stream=None):
        out, inf = orc.msm_pippenger(bases.pts[:n], None, h(scalars_tensor, 4)[:n])
        return np.asarray(out, dtype=np.uint64), int(inf)
    z.msm_g1_dev = msm_g1_dev

    def msm_g1_partial_dev(bases, scalars_tensor, n, stream=None):
        xy, inf = msm_g1_dev(bases, scalars_tensor, n)
        part = np.zeros(24, dtype=np.uint64)   # (X, Y, ZZ, ZZZ): an affine point with ZZ = ZZZ = 1; ZZ = 0 is the identity
        if not inf:
            part[:12] = xy
            part[12:18] = _fq_one_mont()
            part[18:24] = _fq_one_mont()
        return part
    z.msm_g1_partial_dev = msm_g1_partial_dev

    def ntt_fr_dev(tensor, log_n, batch=1, inverse=False, coset=None, stream=None):
        assert coset is None
        res = ops.ntt_batch(tensor.reshape(-1)[: (4 << log_n) * batch].reshape(batch, 1 << log_n, 4), log_n, batch, inverse)
        tensor.reshape(-1)[: (4 << log_n) * batch].copy_(res.reshape(-1))
    z.ntt_fr_dev = ntt_fr_dev

    def NttLayout(lo_bits, mid_bits, mid_stride, hi_stride, batch_stride):
        return (lo_bits, mid_bits, mid_stride, hi_stride, batch_stride)
    z.NttLayout = NttLayout

    def ntt_fr_axis0_dev(t_in, t_out, log_len, cols, inverse=False, tw_log_n=0, tw_col0=0, stream=None):
        ops.axis0(t_in, t_out, log_len, cols, inverse, tw_log_n, tw_col0)
    z.ntt_fr_axis0_dev = ntt_fr_axis0_dev

    def ntt_fr_layout_dev(t_in, t_out, log_n, batch, inverse=False, in_layout=None, out_layout=None, tw_log_n=0, tw_row0=0, stream=None):
        ops.layout(t_in, t_out, log_n, batch, inverse, in_layout=in_layout, out_layout=out_layout, tw_log_n=tw_log_n, tw_row0=tw_row0)
    z.ntt_fr_layout_dev = ntt_fr_layout_dev
    return z
