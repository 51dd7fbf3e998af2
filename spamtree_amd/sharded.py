"""Multi-GPU: one process per GPU, all sharing ONE problem.

`ShardedSpamTreeMV` is `SpamTreeMV` with the three sharded phases composed from the library's local steps and an
all-reduce(sum) through `torch.distributed` in between (backend "nccl" = RCCL over xGMI on the stream the library
launches on; backend "gloo" stages through host memory and is what the single-GPU / CPU tests use).  Each exchanged
entry is contributed by exactly one rank and is zero elsewhere, so results are bit-identical to the single-GPU run
for any number of ranks.  Ownership (whole subtrees below a cut level; replicated top) is decided inside the
library (`st_shard_plan`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .model import SpamTreeMV, SpamTreeError, _dp, _f64


class _DevArray:
    """Zero-copy view of a library-owned device buffer for torch (`__cuda_array_interface__`)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def shard_plan(pb_struct, world, limited_tree=False):
    """owner[u] (rank, or -1 = replicated) and the cut level, from the library's pure-host planner."""
    lib = _lib.load()
    nb = int(pb_struct.n_blocks)
    owner = np.zeros(nb, dtype=np.int64)
    cut = C.c_int32()
    if limited_tree:
        opt = _lib.StOptions(0, 1, 0, int(world), 0, 2)
        rc = lib.st_shard_plan_opt(C.byref(pb_struct), C.byref(opt), int(world), owner.ctypes.data_as(_lib.c_ip), C.byref(cut))
    else:
        rc = lib.st_shard_plan(C.byref(pb_struct), int(world), owner.ctypes.data_as(_lib.c_ip), C.byref(cut))
    if rc != 0:
        raise SpamTreeError(f"st_shard_plan failed ({rc}): {lib.st_last_error(None).decode()}")
    return owner, cut.value


class ShardedSpamTreeMV(SpamTreeMV):
    def __init__(self, *args, dist=None, force_protocol=False, **kw):
        import torch
        self.torch = torch
        self.dist = dist
        self.force_protocol = bool(force_protocol)   # run the local/exchange/finish steps even with one rank (tests)
        self.use_allreduce_w = bool(kw.pop("allreduce_w", False))   # exchange w by all-reduce (sum with zeros) instead of all-gather
        rank = dist.get_rank() if dist is not None else 0
        world = dist.get_world_size() if dist is not None else 1
        self.backend = dist.get_backend() if dist is not None else None
        super().__init__(*args, rank=rank, world=world, **kw)
        self.device_index = kw.get("device", 0)
        if self.backend == "nccl":
            # kernels and collectives share torch's current stream: no host synchronisation between them
            self._check(self.lib.st_set_stream(self.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    # ---- one exchange: all-reduce(sum) of a library-owned device buffer
    def _allreduce(self, ptr, n):
        if self.dist is None or n == 0 or (self.world == 1 and not self.force_protocol):
            return
        torch = self.torch
        if self.backend == "nccl":
            t = torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", self.device_index))
            self.dist.all_reduce(t)
        else:
            self._check(self.lib.st_synchronize(self.h))
            t = torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", self.device_index))
            host = t.cpu()
            self.dist.all_reduce(host)
            t.copy_(host)
            torch.cuda.synchronize()

    # ---- all-gather of a library-owned receive buffer (world slices of `cnt` doubles; this rank's slice is filled)
    def _allgather(self, recv_ptr, cnt):
        if self.dist is None or (self.world == 1 and not self.force_protocol):
            return
        torch = self.torch
        t = torch.as_tensor(_DevArray(recv_ptr, self.world * cnt), device=torch.device("cuda", self.device_index))
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(t, t[self.rank * cnt: (self.rank + 1) * cnt].clone())
        else:
            self._check(self.lib.st_synchronize(self.h))
            host = t.cpu()
            parts = [torch.empty(cnt, dtype=host.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, host[self.rank * cnt: (self.rank + 1) * cnt].clone())
            t.copy_(torch.cat(parts))
            torch.cuda.synchronize()

    def _exchange_w(self):
        """Last step of phase B: every rank's owned rows of w (+ its failure word) by all-gather."""
        snd, rcv, cnt = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._check(self.lib.st_mg_gather_w_pack(self.h, C.byref(snd), C.byref(rcv), C.byref(cnt)))
        self._allgather(rcv.value, cnt.value)
        return self._check(self.lib.st_mg_gather_w_unpack(self.h))

    def _exchange_comps(self, slot):
        ptr, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.st_mg_pack_comps(self.h, slot, C.byref(ptr), C.byref(n)))
        self._allreduce(ptr.value, n.value)
        ll = C.c_double(0.0)
        rc = self._check(self.lib.st_mg_finish(self.h, C.byref(ll)))
        return rc, ll.value

    def get_loglik_comps_w(self, slot) -> bool:
        if self.world == 1 and not self.force_protocol:
            return super().get_loglik_comps_w(slot)
        th = _f64(self.theta[slot])
        self._check(self.lib.st_factor_local(self.h, slot, _dp(th), th.size))
        rc, ll = self._exchange_comps(slot)
        self.last_errtype = rc if rc > 0 else -1
        if rc > 0:
            return False
        self.loglik_w[slot] = ll
        return True

    def get_loglik_w(self, slot):
        if self.world == 1 and not self.force_protocol:
            return super().get_loglik_w(slot)
        self._check(self.lib.st_loglik_local(self.h, slot))
        _, ll = self._exchange_comps(slot)
        self.loglik_w[slot] = ll
        return ll

    def deal_with_w(self, z=None, seed=0, it=0):
        if self.world == 1 and not self.force_protocol:
            return super().deal_with_w(z, seed, it)
        if z is not None:
            z = _f64(z)
            self._check(self.lib.st_sample_w_local(self.h, _dp(z), 0, 0))
        else:
            self._check(self.lib.st_sample_w_local(self.h, None, int(seed), int(it)))
        ptr, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.st_mg_top_region(self.h, C.byref(ptr), C.byref(n)))
        self._allreduce(ptr.value, n.value)
        self._check(self.lib.st_sample_w_top(self.h))
        if self.use_allreduce_w:        # the all-reduce form of the same exchange (st_mg_pack_w / st_mg_unpack_w)
            self._check(self.lib.st_mg_pack_w(self.h, C.byref(ptr), C.byref(n)))
            self._allreduce(ptr.value, n.value)
            rc = self._check(self.lib.st_mg_unpack_w(self.h))
        else:
            rc = self._exchange_w()
        if rc > 0:
            raise SpamTreeError("Error at gibbs_sample_w")

    gibbs_sample_w = deal_with_w

    def deal_with_w_loglik(self, slot, z=None, seed=0, it=0):
        """Sweep + log-density with ONE exchange after the replicated top: phase C of a rank's own blocks needs w of its
        own subtrees and of the top only, so it runs before the other ranks' rows arrive (the order of the library's native
        RCCL path, st_sample_w_loglik).  Returns the log-density."""
        if self.world == 1 and not self.force_protocol:
            self.deal_with_w(z, seed, it)
            return self.get_loglik_w(slot)
        if z is not None:
            z = _f64(z)
            self._check(self.lib.st_sample_w_local(self.h, _dp(z), 0, 0))
        else:
            self._check(self.lib.st_sample_w_local(self.h, None, int(seed), int(it)))
        ptr, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.st_mg_top_region(self.h, C.byref(ptr), C.byref(n)))
        self._allreduce(ptr.value, n.value)
        self._check(self.lib.st_sample_w_top(self.h))
        snd, rcv, cnt, pc, nc = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        self._check(self.lib.st_mg_gather_w_pack(self.h, C.byref(snd), C.byref(rcv), C.byref(cnt)))
        self._check(self.lib.st_loglik_local(self.h, slot))
        self._check(self.lib.st_mg_pack_comps(self.h, slot, C.byref(pc), C.byref(nc)))
        self._allgather(rcv.value, cnt.value)
        self._allreduce(pc.value, nc.value)
        rc = self._check(self.lib.st_mg_gather_w_unpack(self.h))
        if rc > 0:
            raise SpamTreeError("Error at gibbs_sample_w")
        ll = C.c_double(0.0)
        self._check(self.lib.st_mg_finish(self.h, C.byref(ll)))
        self.loglik_w[slot] = ll.value
        return ll.value

    def shard_info(self):
        r, w, c = C.c_int32(), C.c_int32(), C.c_int32()
        ob, orow = C.c_int64(), C.c_int64()
        self._check(self.lib.st_shard_info(self.h, C.byref(r), C.byref(w), C.byref(c), C.byref(ob), C.byref(orow)))
        return dict(rank=r.value, world=w.value, cut_level=c.value, owned_blocks=ob.value, owned_rows=orow.value)
