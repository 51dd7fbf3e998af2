"""CPU, gloo, world_size 2 and 3: the N>1 path's host logic -- the library's ownership plan (pure host code, no GPU)
and the sum-with-zeros exchange protocol, with the oracle standing in for the compute."""
import socket

import pytest


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,case", [(2, dict(side=25, q=1, seed=4, missing=0.1)), (3, dict(side=40, q=1, seed=5)),
                                        (2, dict(side=14, q=3, seed=6)),
                                        (3, dict(side=40, q=1, seed=7, limited_tree=True))])   # make_edges_limited's single parents (round 3)
def test_plan_and_exchange_under_gloo(world, case, tmp_path):
    import numpy as np
    import torch.multiprocessing as mp
    from tests._sharded_worker import plan_worker
    mp.spawn(plan_worker, args=(world, free_port(), case, str(tmp_path)), nprocs=world, join=True)
    cuts = [int(np.load(tmp_path / f"ok_{r}.npy")[0]) for r in range(world)]
    assert len(set(cuts)) == 1
