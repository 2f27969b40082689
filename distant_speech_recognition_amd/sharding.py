"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL
over xGMI on ROCm, "gloo" in CPU tests).

The path shards two ways:
  * by utterance STREAM -- streams are independent graphs (unit_test/test_online_beamforming.py:80-88
    builds one per utterance): stream s runs on rank s mod G, no data-path collective at all;
  * by frequency BIN -- every stage between the analysis FFT and the synthesis FFT is independent per
    bin (`for fbinX` loops, beamformer.cc:1298, postfilter.cc:184, pybeamformer.py:674): rank g owns a
    contiguous bin range, and ONE all-gather of the beamformed block Y[K_g][T] precedes synthesis.
With STATIC weights (the superdirective array of C5) a third partition costs less than either: the fused analysis -> beamformer
kernel never writes the snapshots, so nothing per-bin exists to shard; rank g runs the fused kernel over a contiguous range of
FRAMES (all channels, all bins; the analysis window makes it read m M - D samples of halo from the replicated PCM) and the same
single all-gather -- along the frame axis -- assembles Y before synthesis (pipeline_frame_sharded).
"""


def streams_for_rank(num_streams, rank, world):
    """Indices of the utterance streams rank `rank` processes (round-robin: s mod world == rank)."""
    return list(range(rank, num_streams, world))


def bin_range_for_rank(K, rank, world):
    """Contiguous bin range [k0, k1) of rank `rank`: ceil(K/world) bins per rank, last ranks may be short/empty."""
    per = -(-K // world)
    k0 = min(rank * per, K)
    return k0, min(k0 + per, K)


def allgather_bins(Y_local, K, group=None):
    """All-gather the beamformed block before synthesis.  Y_local complex [S][K_g][T] on this rank
    (K_g = its bin range, possibly shorter on the last ranks) -> Y complex [S][K][T] on every rank.
    One collective per block; payload 8*K*T*S bytes in total (33.6 MB at C5 with T=4096), i.e. a few
    MB per xGMI link -- the reason a plain ring all-gather is enough here."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-K // world)
    S, Kg, T = Y_local.shape
    k0, k1 = bin_range_for_rank(K, rank, world)
    assert Kg == k1 - k0, "local block does not match this rank's bin range"
    pad = torch.zeros((S, per, T), dtype=Y_local.dtype, device=Y_local.device)
    pad[:, :Kg] = Y_local
    buf = torch.view_as_real(pad).contiguous()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    full = torch.cat([torch.view_as_complex(o) for o in out], dim=1)[:, :K]
    return full.contiguous()


def padded_bin_rows(K, world):
    """rows per stream of a buffer the in-place all-gather runs in: world * ceil(K / world) >= K (btk_bin_rows_padded)"""
    return world * (-(-K // world))


def allgather_bins_inplace(Y, K, group=None):
    """north_star's "single all-gather": Y complex [S][Kp][T], Kp = padded_bin_rows(K, world), holds this rank's beamformed bins at
    rows [rank * per, ...) of every stream (the apply kernel wrote them there); ONE in-place all-gather per stream -- a single
    collective for the one-stream large-array case -- completes it on every rank without a staging copy (the C-ABI form is
    btk_allgather_bins_inplace).  Returns the [S][K][T] view."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-K // world)
    S, Kp, T = Y.shape
    assert Kp == per * world and Y.is_contiguous(), "Y must be contiguous [S][%d][T]" % (per * world)
    Yr = torch.view_as_real(Y)                                            # [S][Kp][T][2]
    for s in range(S):
        out = Yr[s].view(world * per * T * 2)
        chunk = out[rank * per * T * 2: (rank + 1) * per * T * 2]
        if dist.get_backend(group) != "nccl":
            chunk = chunk.clone()                                           # gloo (CPU tests) does not promise in-place aliasing
        dist.all_gather_into_tensor(out, chunk, group=group)
    return Y[:, :K]


def exchange_channels_for_bins(X_chan, K, N, group=None):
    """Option (ii) of SURVEY 8(e) for the analysis side of a bin-sharded run: every rank has transformed ITS channels completely
    (X_chan complex [S][K][N_r][T], channel range bin_range_for_rank(N, rank, world)) and one all-to-all hands every rank its bin
    range of ALL channels: -> X_bins [S][K_r][N][T].  Per block 8 K N T S bytes cross the fabric in total, 7/8 of a rank's eighth over
    its seven xGMI links (C5: 16.8 MB per link, 0.11 ms at 153 GB/s) -- against option (i), where every rank transforms all N
    channels (0.40 ms at C5) and exchanges nothing.  RCCL: one all_to_all_single with uneven splits; other backends (CPU tests):
    an all-gather of the padded blocks."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    S, Kx, Nr, T = X_chan.shape
    c0, c1 = bin_range_for_rank(N, rank, world)
    assert Kx == K and Nr == c1 - c0, "X_chan must hold all K bins of this rank's channel range"
    k0, k1 = bin_range_for_rank(K, rank, world)
    Kr = k1 - k0
    chans = [bin_range_for_rank(N, q, world) for q in range(world)]
    bins = [bin_range_for_rank(K, q, world) for q in range(world)]
    X_bins = torch.empty((S, Kr, N, T), dtype=X_chan.dtype, device=X_chan.device)
    if dist.get_backend(group) == "nccl":
        send = torch.cat([torch.view_as_real(X_chan[:, a:b].contiguous()).reshape(-1) for a, b in bins])
        in_split = [S * (b - a) * Nr * T * 2 for a, b in bins]
        out_split = [S * Kr * (b - a) * T * 2 for a, b in chans]
        recv = torch.empty(sum(out_split), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        off = 0
        for (a, b), n in zip(chans, out_split):
            if n:
                X_bins[:, :, a:b] = torch.view_as_complex(recv[off: off + n].view(S, Kr, b - a, T, 2))
            off += n
    else:
        per = -(-N // world)
        pad = torch.zeros((S, K, per, T), dtype=X_chan.dtype, device=X_chan.device)
        pad[:, :, :Nr] = X_chan
        buf = torch.view_as_real(pad).contiguous()
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf, group=group)
        for (a, b), o in zip(chans, out):
            if b > a:
                X_bins[:, :, a:b] = torch.view_as_complex(o)[:, k0:k1, : b - a]
    return X_bins


def frame_range_for_rank(T, rank, world, tile=16):
    """Contiguous frame range [t0, t1) of rank `rank`: whole tiles of the fused kernel, ceil(T / world) rounded up to `tile` frames."""
    per = -(-(-(-T // world)) // tile) * tile
    t0 = min(rank * per, T)
    return t0, min(t0 + per, T)


def allgather_frames(Y_local, T, group=None, tile=16):
    """All-gather along the FRAME axis: Y_local complex [S][K][T_g] (this rank's frame range, possibly short or empty on the last
    ranks) -> Y complex [S][K][T] on every rank.  One collective per block, the same 8 K T S bytes as the bin form."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-(-(-T // world)) // tile) * tile
    S, K, Tg = Y_local.shape
    t0, t1 = frame_range_for_rank(T, rank, world, tile)
    assert Tg == t1 - t0, "local block does not match this rank's frame range"
    pad = torch.zeros((S, K, per), dtype=Y_local.dtype, device=Y_local.device)
    pad[:, :, :Tg] = Y_local
    buf = torch.view_as_real(pad).contiguous()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([torch.view_as_complex(o) for o in out], dim=2)[:, :, :T].contiguous()


def pipeline_frame_sharded(afb, sfb, pcm, W, rank, world, group=None, synth_rank=None, tile=16):
    """Static weights on `world` GPUs: every rank holds the SAME multichannel PCM and the whole weight set W complex64 [K][N], runs the
    fused analysis -> beamformer kernel (no snapshots in HBM) over its frame range, ONE all-gather along the frame axis assembles
    Y [S][K][T], the synthesis bank runs on `synth_rank` (every rank if None).  Returns (pcm_out or None, Y).  Per rank: 1 / world of
    the fused kernel's work plus the window halo -- at C5 (256 mics, 2048 bins, 512 frames) 0.29 ms / world against 0.40 ms per rank
    for option (i) of the bin partition, whose every rank transforms all channels."""
    T = afb.num_frames(pcm.shape[-1])
    t0, t1 = frame_range_for_rank(T, rank, world, tile)
    import torch
    if t1 > t0:
        Y_local = afb.analysis_beamform(pcm, W, t0=t0, tcount=t1 - t0).contiguous()
    else:
        Y_local = torch.zeros((pcm.shape[0], afb.K, 0), dtype=torch.complex64, device=pcm.device)
    Y = allgather_frames(Y_local, T, group, tile) if world > 1 else Y_local
    out = sfb.synthesize(Y) if (synth_rank is None or synth_rank == rank) else None
    return out, Y


def max_over_ranks(value, device, group=None):
    """MAX-reduce a host scalar (step time) over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def bf_apply_bin_sharded(W_local, X_local, K, group=None):
    """Beamform this rank's bin range on its GPU and all-gather the result.
    W_local complex64 [K_g][N], X_local complex64 [S][K_g][N][T] -> Y complex64 [S][K][T]."""
    from . import engine
    Y_local = engine.bf_apply(W_local, X_local)
    return allgather_bins(Y_local, K, group)


def pipeline_bin_sharded(afb, sfb, pcm, W_local, K, rank, world, group=None, synth_rank=None, analysis_input="replicated"):
    """BASELINE config C5 end to end on `world` GPUs: every rank analyses the SAME multichannel PCM (the analysis FFT
    yields all bins of a channel, so the PCM -- N*D*4 bytes per frame -- is what is replicated), stores only its bin range
    of the snapshots (btk_fb_analysis_bins), beamforms it, ONE all-gather assembles Y [S][K][T], and the synthesis bank runs on `synth_rank`
    (every rank if None).  afb / sfb: engine.FilterBank analysis / synthesis plans; W_local complex64 [K_g][N].
    analysis_input = "replicated" (option (i) of SURVEY 8(e), the default: no exchange before the beamformer) or "channels"
    (option (ii): pcm holds only this rank's channel range bin_range_for_rank(N, rank, world); the ranks transform disjoint
    channels and one all-to-all regroups the snapshots by bin -- measured / predicted in DESIGN.md section 6).
    Returns (pcm_out or None, Y)."""
    k0, k1 = bin_range_for_rank(K, rank, world)
    # every channel is transformed on every rank (the FFT yields all bins), but only this rank's bins are stored: the
    # snapshot write -- 8 K N bytes per frame, the dominant cost of the analysis bank -- shrinks by `world`, and no
    # full-size X nor a slice copy exists (an empty trailing shard launches nothing)
    if analysis_input == "channels" and world > 1:
        X_local = exchange_channels_for_bins(afb.analysis(pcm), K, W_local.shape[-1], group)
    else:
        X_local = afb.analysis(pcm, bins=(k0, k1))          # [S][K_g][N][T]
    from . import engine
    S, T = X_local.shape[0], X_local.shape[-1]
    if world > 1 and S == 1:
        # one stream (the large-array case this sharding exists for): the apply kernel writes this rank's bins in place into the
        # padded block and a SINGLE in-place all-gather completes it -- no staging buffer, no copy
        import torch
        Yp = torch.zeros((1, padded_bin_rows(K, world), T), dtype=torch.complex64, device=X_local.device)
        if k1 > k0:
            engine.bf_apply(W_local, X_local, out=Yp[:, k0:k1])
        Y = allgather_bins_inplace(Yp, K, group)
    elif world > 1:
        Y = bf_apply_bin_sharded(W_local, X_local, K, group)
    else:
        Y = engine.bf_apply(W_local, X_local)
    out = sfb.synthesize(Y) if (synth_rank is None or synth_rank == rank) else None
    return out, Y
