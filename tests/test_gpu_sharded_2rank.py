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
        # option (ii): every rank transforms only ITS channels, the exchange regroups the snapshots by bin -- same Y
        c0, c1 = sharding.bin_range_for_rank(N, rank, world)
        out2, Y2 = sharding.pipeline_bin_sharded(afb, sfb, pcm[:, c0:c1].contiguous(), W_local, K, rank, world, synth_rank=0,
                                                 analysis_input="channels")
        assert torch.equal(Y2, ref_Y)
        # static weights: the frame partition of the fused kernel (no snapshots, no bin shards), one all-gather along the frame axis
        out3, Y3 = sharding.pipeline_frame_sharded(afb, sfb, pcm, W_full, rank, world, synth_rank=0)
        assert torch.equal(Y3, afb.analysis_beamform(pcm, W_full))
        assert (out3 is None) == (rank != 0)
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
    # round 5: the line shows N ranks -- the world size the communicator itself reports and every rank's own rate (the job's value is
    # the units of all ranks over the SLOWEST rank's time, so it cannot exceed the sum of the ranks' own rates)
    assert d["config"]["rccl_world"] == 2 and d["config"]["dist_backend"] == "gloo"
    pr = d["config"]["per_rank_frames_per_s"]
    assert len(pr) == 2 and all(v > 0 for v in pr) and d["value"] <= sum(pr) * (1 + 1e-9)


def test_bench_rccl_world_of_one_runs_the_multi_gpu_stage(dev):
    """bench.py's multi-GPU code path on RCCL with a world of ONE rank (BTK_BENCH_FORCE_DIST=1): the barrier / max-over-ranks /
    all_gather of the headline AND stages.c5_frame_sharded (C5 by frame range: fused kernel, all-gather of Y along the frame
    axis, synthesis) -- the stage first meets several GPUs on the driver's node, so everything in it that one GPU can run runs here"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BTK_BENCH_FORCE_DIST="1")
    env.pop("BTK_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--streams", "4", "--frames", "1024", "--no-cpu"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["rccl_world"] == 1 and d["config"]["dist_backend"] == "nccl"
    c5 = d["stages"]["c5_frame_sharded"]
    assert "error" not in c5, c5
    assert c5["world"] == 1
    for T in (512, 4096):
        e = c5["frames_%d" % T]
        assert e["frames_per_rank"] == T and e["allgather_bytes"] == 8 * 1025 * T
        assert 0 < e["rank_kernel_ms"] <= e["block_ms"] * 1.05 and e["allgather_alone_ms"] >= 0 and e["frames_per_s"] > 0


@pytest.mark.parametrize("M,r,N", [(512, 1, 5), (256, 1, 3), (2048, 1, 2), (64, 0, 4), (128, 2, 3)])
def test_bin_range_analysis_equals_slice(dev, M, r, N):
    """btk_fb_analysis_bins (what a rank of a bin-sharded run launches): bit-identical to the bin slice of the whole
    transform in all three analysis kernels (M = 512 specialised, register-FFT, generic), incl. ranges that contain bin 0,
    the Nyquist bin, a single bin, and the empty range of a trailing rank."""
    import torch
    from distant_speech_recognition_amd import engine as eng, sharding
    from tests.util import design_prototype
    m = 4 if M >= 256 else 2
    K, D = M // 2 + 1, M >> r
    fb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    g = torch.Generator(device=dev).manual_seed(M + N)
    pcm = (torch.randn((2, N, 37 * D + 11), device=dev, generator=g) * 1000).round_()
    X = fb.analysis(pcm)
    for k0, k1 in ((0, K), (0, 1), (K - 1, K), (K // 3, K // 3 + 7), (K // 2, K), (5, 5)):
        Xs = fb.analysis(pcm, bins=(k0, k1))
        assert Xs.shape == (2, k1 - k0, N, X.shape[3])
        assert torch.equal(Xs, X[:, k0:k1])
    # the shards of an 8-rank run tile the transform; K = 33 leaves rank 7 empty (ceil(33 / 8) = 5 bins per rank)
    if M == 64:
        parts = [fb.analysis(pcm, bins=sharding.bin_range_for_rank(K, rk, 8)) for rk in range(8)]
        assert parts[7].shape[1] == 0
        assert torch.equal(torch.cat(parts, dim=1), X)
        W = torch.randn((K, N), dtype=torch.complex64, device=dev)
        Ys = [eng.bf_apply(W[a:b].contiguous(), p_) for p_, (a, b) in zip(parts, [sharding.bin_range_for_rank(K, rk, 8) for rk in range(8)])]
        assert Ys[7].shape == (2, 0, X.shape[3])
        assert torch.equal(torch.cat(Ys, dim=1), eng.bf_apply(W, X))


def test_c_abi_allgather_bins_rccl_world1(dev, tmp_path):
    """btk_allgather_bins through the C-ABI with a real RCCL communicator (one rank: RCCL refuses two ranks on this box's
    single GPU; the per-(owner, stream) broadcast group is the same code for any world).  Runs in a process WITHOUT torch so
    that the library binds /opt/rocm's librccl -- the one the communicator was created with."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, numpy as np, sys
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
rccl = C.CDLL("/opt/rocm/lib/librccl.so", mode=C.RTLD_GLOBAL)
lib = C.CDLL(%r)
class UID(C.Structure): _fields_ = [("b", C.c_char * 128)]
uid = UID(); assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
S, K, T = 3, 33, 40
rng = np.random.default_rng(0)
Yl = (rng.normal(size=(S, K, T)) + 1j * rng.normal(size=(S, K, T))).astype(np.complex64)
dl, dy = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(dl), Yl.nbytes); hip.hipMalloc(C.byref(dy), Yl.nbytes)
hip.hipMemcpy(dl, Yl.ctypes.data_as(C.c_void_p), C.c_size_t(Yl.nbytes), 1)
hip.hipMemset(dy, 0, C.c_size_t(Yl.nbytes))
lib.btk_allgather_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p]
rc = lib.btk_allgather_bins(comm, dl, dy, S, K, T, 0, 1, None)
lib.btk_last_error.restype = C.c_char_p
assert rc == 0, lib.btk_last_error()
hip.hipDeviceSynchronize()
out = np.zeros_like(Yl)
hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dy, C.c_size_t(Yl.nbytes), 2)
assert np.array_equal(out, Yl)
k0, k1 = C.c_int(), C.c_int()
lib.btk_bin_range(33, 7, 8, C.byref(k0), C.byref(k1)); assert (k0.value, k1.value) == (33, 33)
lib.btk_bin_range(33, 6, 8, C.byref(k0), C.byref(k1)); assert (k0.value, k1.value) == (30, 33)
# the even form: ONE in-place ncclAllGather per stream on the padded block (world 1: the rank's rows are the whole block)
assert lib.btk_bin_rows_padded(33, 8) == 40 and lib.btk_bin_rows_padded(1025, 8) == 1032 and lib.btk_bin_rows_padded(33, 1) == 33
lib.btk_allgather_bins_inplace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p]
for s_n in (1, S):
    hip.hipMemcpy(dy, Yl.ctypes.data_as(C.c_void_p), C.c_size_t(Yl.nbytes), 1)
    rc = lib.btk_allgather_bins_inplace(comm, dy, s_n, K, T, 0, 1, None)
    assert rc == 0, lib.btk_last_error()
    hip.hipDeviceSynchronize()
    hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dy, C.c_size_t(Yl.nbytes), 2)
    assert np.array_equal(out, Yl)
assert lib.btk_allgather_bins_inplace(comm, dy, 1, K, T, 3, 2, None) != 0           # rank outside the world: refused
rccl.ncclCommDestroy(comm)
print("ok")
''' % os.path.join(root, "distant_speech_recognition_amd", "csrc", "libbtkhip.so")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr
