"""GPU, two devices: the collectives of the bin-sharded path over REAL RCCL with world = 2, one rank per GPU -- the in-place
all-gather of the beamformed block (sharding.allgather_bins_inplace / all_gather_into_tensor), the grouped all-gather of uneven
shards, and option (ii)'s all_to_all_single with uneven splits.  Skipped on a box with one GPU (the gloo and one-GPU tests cover the
arithmetic there, tests/test_sharding_gloo.py, tests/test_gpu_sharded_2rank.py); on the 8-GPU node this is the pre-flight of
bench_bin_sharded.py: first contact needs no code change."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        for N, M, T, S in ((16, 256, 96, 1), (6, 64, 40, 3)):                  # K = 129 / 33: uneven bin shards; S = 1 takes the in-place form
            K, D = M // 2 + 1, M // 2
            m_ = 4 if M >= 256 else 2
            afb = eng.FilterBank(design_prototype(M, m_), M, m_, 1, 2)
            sfb = eng.FilterBank(design_prototype(M, m_, "g"), M, m_, 1, 2, synthesis=True)
            L = (T - afb.processing_delay + afb.lookahead) * D
            g = torch.Generator(device=dev).manual_seed(11)                    # the same PCM on both ranks
            pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000).round_()
            gw = torch.Generator(device=dev).manual_seed(5)
            W_full = (torch.randn((K, N), device=dev, generator=gw) + 1j * torch.randn((K, N), device=dev, generator=gw)).to(torch.complex64)
            k0, k1 = sharding.bin_range_for_rank(K, rank, world)
            W_local = W_full[k0:k1].contiguous()
            ref_Y = eng.bf_apply(W_full, afb.analysis(pcm))
            out, Y = sharding.pipeline_bin_sharded(afb, sfb, pcm, W_local, K, rank, world, synth_rank=0)
            assert torch.equal(Y, ref_Y), "option (i): all-gather over RCCL"
            if rank == 0:
                assert torch.equal(out, sfb.synthesize(ref_Y))
            c0, c1 = sharding.bin_range_for_rank(N, rank, world)
            _, Y2 = sharding.pipeline_bin_sharded(afb, sfb, pcm[:, c0:c1].contiguous(), W_local, K, rank, world, synth_rank=0,
                                                  analysis_input="channels")
            assert torch.equal(Y2, ref_Y), "option (ii): all_to_all_single with uneven splits over RCCL"
            _, Y3 = sharding.pipeline_frame_sharded(afb, sfb, pcm, W_full, rank, world, synth_rank=0)
            assert torch.equal(Y3, afb.analysis_beamform(pcm, W_full)), "frame partition of the fused kernel"
        t = sharding.max_over_ranks(1.0 + rank, dev)
        assert t == float(world)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bin_sharded_collectives_over_rccl_two_gpus():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


@pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("analysis_input", ["replicated", "channels", "frames"])
def test_bench_bin_sharded_two_gpus(analysis_input):
    """bench_bin_sharded.py the way it runs on the 8-GPU node (torch.distributed.run, backend nccl), with both analysis inputs"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench_bin_sharded.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--mics", "32", "--bins", "512", "--frames", "128", "--analysis-input", analysis_input]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["analysis_input"] == analysis_input and np.isfinite(d["pcm_checksum"])
