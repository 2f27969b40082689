"""GPU: the bin-sharded and stream-sharded paths with TWO ranks sharing the one GPU of the test box (gloo carries the
collectives -- RCCL refuses two ranks on one device; on the 8-GPU node the same code runs over RCCL/xGMI)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    from distant_speech_recognition_amd import engine as eng, sharding
    from tests.util import design_prototype, ula_positions, la_delays
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        N, M, T, S = 16, 256, 96, 2
        K, D = M // 2 + 1, M // 2
        afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
        sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
        L = (T - afb.processing_delay + afb.lookahead) * D
        g = torch.Generator(device=dev).manual_seed(11)                      # replicated input
        pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000).round_()
        mpos = ula_positions(N, 25.0)
        wq = torch.from_numpy(eng.weights_mainlobe(M, N, 16000.0, la_delays(mpos, 0.5))[:K].astype(np.complex64)).to(dev)
        Rd = eng.mvdr_diffuse_model(mpos, M, 16000.0, device=dev)
        eng.mvdr_diagonal_loading(Rd, 0.01)
        W_full, _ = eng.mvdr_weights(Rd, wq)
        k0, k1 = sharding.bin_range_for_rank(K, rank, world)
        W_local, _ = eng.mvdr_weights(Rd[k0:k1].contiguous(), wq[k0:k1].contiguous(), first_bin=k0)
        assert torch.equal(W_local, W_full[k0:k1])
        out, Y = sharding.pipeline_bin_sharded(afb, sfb, pcm, W_local, K, rank, world, synth_rank=0)
        ref_Y = eng.bf_apply(W_full, afb.analysis(pcm))
        assert torch.equal(Y, ref_Y)                                          # every rank holds the whole Y after the all-gather
        if rank == 0:
            assert torch.equal(out, sfb.synthesize(ref_Y))
        else:
            assert out is None
        # stream sharding: each rank runs its streams, no collective on the data path
        mine = sharding.streams_for_rank(S, rank, world)
        Ym = afb.analysis_beamform(pcm[mine].contiguous(), W_full)
        assert torch.equal(Ym, afb.analysis_beamform(pcm, W_full)[mine])
        t = sharding.max_over_ranks(1.0 + rank, torch.device("cpu"))
        assert t == float(world)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(dev):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_bench_two_ranks_contract(dev):
    """bench.py launched the way the driver launches it for N = 2 (torch.distributed.run, one JSON line from rank 0,
    whole-job value); both ranks share this box's GPU, so gloo carries the barrier / max-over-ranks"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BTK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--streams", "4", "--frames", "1024"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "frames/s" and d["cpu_baseline"] is None and "roofline" in d and d["config"]["parallelism"].endswith("x2")
    # whole-job aggregate: 2 ranks x 4 streams x 1024 frames per step
    assert abs(d["value"] - 2 * 4 * 1024 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
