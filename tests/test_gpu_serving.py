"""GPU: the host-to-host serving pipeline (distant_speech_recognition_amd/serving.py): int16 PCM over PCIe, three HIP streams,
`depth` buffer sets -- must give bit for bit what the same chain gives on resident float PCM, for every batch including the
ragged last one, with the output narrowed like the reference scripts' numpy.array(buf, numpy.int16)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("int16_in,int16_out,depth,interleaved", [(True, False, 3, False), (True, True, 2, False), (False, False, 1, False), (True, True, 3, True)])
def test_pipeline_equals_resident_chain(dev, int16_in, int16_out, depth, interleaved):
    import torch
    from distant_speech_recognition_amd import engine as eng
    from distant_speech_recognition_amd.serving import BatchBeamformerPipeline
    from tests.util import design_prototype, ula_positions, la_delays
    N, M, m, r, S, B = 6, 512, 4, 1, 7, 3
    D = M >> r
    afb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    sfb = eng.FilterBank(design_prototype(M, m, "g"), M, m, r, 2, synthesis=True)
    L = 45 * D
    rng = np.random.default_rng(8)
    host = rng.integers(-20000, 20000, size=(S, N, L)).astype(np.int16)
    delays = la_delays(ula_positions(N), 0.6)
    wq = eng.weights_mainlobe(M, N, 16000.0, delays)
    W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
    pipe = BatchBeamformerPipeline(afb, sfb, W, N, L, streams_per_batch=B, depth=depth, int16_in=int16_in, int16_out=int16_out,
                                   interleaved=interleaved)
    src = torch.from_numpy(host if int16_in else host.astype(np.float32))
    if interleaved:                                                 # the frames of a multi-channel WAV as stored: [S][L][N]
        src = torch.from_numpy(np.ascontiguousarray(np.transpose(host, (0, 2, 1))))
    got = pipe.run(src)
    got2 = pipe.run(src)                                            # buffer sets and events are reusable
    # the resident chain on the same samples
    pcm = torch.from_numpy(host.astype(np.float32)).to(dev)
    Y = afb.analysis_beamform(pcm, W)
    ref = sfb.synthesize(Y).cpu().numpy()
    if int16_out:
        ref = ref.astype(np.int16)                                  # numpy.array(buf, numpy.int16): toward zero
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.array_equal(got, ref) and np.array_equal(got2, ref)
    assert np.abs(ref.astype(np.float64)).max() > 100.0


def test_pcm_format_kernels(dev):
    import ctypes
    import torch
    from distant_speech_recognition_amd import _lib
    rng = np.random.default_rng(1)
    for n in (0, 1, 7, 8, 2048, 2049 + 5):
        a = rng.integers(-32768, 32768, size=n).astype(np.int16)
        d = torch.from_numpy(a).to(dev)
        f = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().btk_pcm_i16_to_f32(d.data_ptr(), f.data_ptr(), n, None))
        assert np.array_equal(f.cpu().numpy(), a.astype(np.float32))
        x = (rng.standard_normal(n) * 9000).astype(np.float32)
        o = torch.empty(n, dtype=torch.int16, device=dev)
        _lib.check(_lib.lib().btk_pcm_f32_to_i16(torch.from_numpy(x).to(dev).data_ptr(), o.data_ptr(), n, None))
        # toward zero; beyond the int16 range a C cast through int wraps, which is what numpy.array(buf, numpy.int16) does on x86
        assert np.array_equal(o.cpu().numpy(), x.astype(np.int32).astype(np.int16))
    # an unaligned view takes the scalar path
    a = rng.integers(-32768, 32768, size=4099).astype(np.int16)
    d = torch.from_numpy(a).to(dev)[1:]
    f = torch.empty(4098, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().btk_pcm_i16_to_f32(d.data_ptr(), f.data_ptr(), 4098, None))
    assert np.array_equal(f.cpu().numpy(), a[1:].astype(np.float32))


def test_pcm_deinterleave_kernel(dev):
    """btk_pcm_i16_deinterleave: int16 frames [L][N] -> planar float32 [N][stride] for sizes around the 64 x 64 tile"""
    import torch
    from distant_speech_recognition_amd import _lib
    rng = np.random.default_rng(2)
    for L, N, pad in ((1, 1, 0), (63, 5, 3), (64, 64, 0), (130, 65, 7), (1000, 8, 0)):
        a = rng.integers(-32768, 32768, size=(L, N)).astype(np.int16)
        d = torch.from_numpy(a).to(dev)
        o = torch.full((N, L + pad), -7.0, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().btk_pcm_i16_deinterleave(d.data_ptr(), o.data_ptr(), L, N, L + pad, None))
        got = o.cpu().numpy()
        assert np.array_equal(got[:, :L], a.T.astype(np.float32))
        assert np.all(got[:, L:] == -7.0)                           # the padding of a row is not touched


def test_gather_rows_kernel(dev):
    """btk_gather_rows: rows in SEPARATE pinned host allocations -> one device block by one kernel that reads the host memory
    through a pinned table: ragged lengths (0, below one 16-byte word, partial last word, exactly the pitch), the bytes behind a
    row zeroed, many workgroups per row and one; device-resident rows work the same"""
    import ctypes
    import torch
    from distant_speech_recognition_amd import _lib
    rng = np.random.default_rng(5)
    for nrows, pitch, lens in ((1, 16, [2]), (3, 64, [0, 14, 64]), (5, 4096 + 16, [4096 + 16, 4096 + 2, 4000, 18, 16]),
                               (64, 1 << 20, None), (2048, 8192, None)):
        if lens is None:
            lens = [int(x) * 2 for x in rng.integers(0, pitch // 2 + 1, size=nrows)]
            lens[0] = pitch
        rows = [torch.from_numpy(rng.integers(0, 256, size=max(n, 2), dtype=np.uint8)).pin_memory() for n in lens]
        tab = torch.zeros((nrows, 2), dtype=torch.int64).pin_memory()
        for r, (t, n) in enumerate(zip(rows, lens)):
            tab[r, 0] = t.data_ptr()
            tab[r, 1] = n
        out = torch.full((nrows, pitch), 0xAB, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().btk_gather_rows(tab.data_ptr(), out.data_ptr(), nrows, pitch, None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for r, (t, n) in enumerate(zip(rows, lens)):
            assert np.array_equal(got[r, :n], t.numpy()[:n]), (nrows, r, n)
            assert not got[r, n:].any(), (nrows, r, n)
        # the same rows resident on the device, the table too
        drows = [t.to(dev) for t in rows]
        dtab = tab.clone()
        for r, t in enumerate(drows):
            dtab[r, 0] = t.data_ptr()
        dtab = dtab.to(dev)
        out2 = torch.full((nrows, pitch), 0xCD, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().btk_gather_rows(dtab.data_ptr(), out2.data_ptr(), nrows, pitch, None))
        assert torch.equal(out, out2)
    # argument checks: a pitch or a destination that is no multiple of 16
    out = torch.zeros(64, dtype=torch.uint8, device=dev)
    tab = torch.zeros((1, 2), dtype=torch.int64).pin_memory()
    assert _lib.lib().btk_gather_rows(tab.data_ptr(), out.data_ptr(), 1, 24, None) != 0
    assert _lib.lib().btk_gather_rows(tab.data_ptr(), out.data_ptr() + 8, 1, 16, None) != 0
    assert _lib.lib().btk_gather_rows(None, None, 0, 16, None) == 0
