"""CPU, world_size 2 over gloo: the multi-GPU sharding logic (stream partition, bin ranges, the single
all-gather before synthesis, max-over-ranks timing).  The collectives run on CPU tensors here and on
RCCL/xGMI on the GPU box; the partition arithmetic is identical."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, K, ret):
    import torch
    import torch.distributed as dist
    from distant_speech_recognition_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, T = 3, 17
        g = torch.Generator().manual_seed(1234)
        full = torch.view_as_complex(torch.randn((S, K, T, 2), generator=g))        # same on every rank
        k0, k1 = sharding.bin_range_for_rank(K, rank, world)
        got = sharding.allgather_bins(full[:, k0:k1].contiguous(), K)
        assert torch.equal(got, full)
        # the even form: every rank's bins already in place in the padded block, one in-place all-gather per stream
        Yp = torch.zeros((S, sharding.padded_bin_rows(K, world), T), dtype=full.dtype)
        Yp[:, k0:k1] = full[:, k0:k1]
        got2 = sharding.allgather_bins_inplace(Yp, K)
        assert got2.shape == full.shape and torch.equal(got2, full)
        assert sharding.padded_bin_rows(K, world) % world == 0 and 0 <= sharding.padded_bin_rows(K, world) - K < world
        # option (ii) of the analysis input: channel-sharded transforms regrouped by bin (uneven channel and bin shards)
        N = 5
        Xall = torch.view_as_complex(torch.randn((S, K, N, T, 2), generator=g))      # same on every rank
        c0, c1 = sharding.bin_range_for_rank(N, rank, world)
        Xb = sharding.exchange_channels_for_bins(Xall[:, :, c0:c1].contiguous(), K, N)
        assert Xb.shape == (S, k1 - k0, N, T) and torch.equal(Xb, Xall[:, k0:k1])
        # the frame partition of the fused static-weight path: whole 16-frame tiles per rank, one all-gather along the frame axis
        for Tq in (17, 40, 5):
            Yq = torch.view_as_complex(torch.randn((S, K, Tq, 2), generator=g))
            a0, a1 = sharding.frame_range_for_rank(Tq, rank, world)
            assert (a0 % 16 == 0 or a0 == Tq) and (a1 % 16 == 0 or a1 == Tq) and 0 <= a0 <= a1 <= Tq
            assert torch.equal(sharding.allgather_frames(Yq[:, :, a0:a1].contiguous(), Tq), Yq)
        t = sharding.max_over_ranks(1.0 + rank, torch.device("cpu"))
        assert t == float(world)
        mine = sharding.streams_for_rank(7, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        assert sorted(sum(gathered, [])) == list(range(7))
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_bin_allgather_with_empty_trailing_ranks_world4():
    """K = 5 bins over 4 ranks: ceil(5 / 4) = 2 per rank -> ranks hold 2, 2, 1, 0 bins; the all-gather pads and trims"""
    import torch.multiprocessing as mp
    world, port = 4, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 5, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1, 2: 1, 3: 1}


@pytest.mark.parametrize("K", [257, 10, 3])
def test_bin_allgather_and_stream_partition_world2(K):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_partition_arithmetic():
    from distant_speech_recognition_amd import sharding
    for K in (1, 129, 257, 1025):
        for world in (1, 2, 3, 8):
            rngs = [sharding.bin_range_for_rank(K, r, world) for r in range(world)]
            assert rngs[0][0] == 0 and rngs[-1][1] == K
            assert all(a[1] == b[0] for a, b in zip(rngs, rngs[1:]))
    assert sharding.streams_for_rank(128, 3, 8) == list(range(3, 128, 8))
