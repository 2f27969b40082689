"""Block-level Python front-end of the C-ABI (include/btkhip.h).

PyTorch is used only for device memory and streams (tensor.data_ptr()); every computation is a
hand-written HIP kernel in csrc/.  Tensor layouts:

    pcm  float32   [S][N][L]       X  complex64 [S][K][N][T]
    W    complex64 [S|1][K][N]     Y  complex64 [S][K][T]      out float32 [S][B*D]
"""
import collections
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, name):
    if not t.is_cuda:
        raise ValueError("%s must live in HBM (cuda tensor); the engine has no CPU path" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _check(t, name, dtype, shape, rows=False):
    """The C-ABI sees raw pointers: dtype, residency, contiguity and shape are checked here.  shape: an int (number of
    dimensions) or a tuple whose None entries are free.  rows=True admits a [..., :T] view of a buffer with padded rows
    (padded_rows) and returns the row stride in elements, for the entry points whose T_stride the caller may choose."""
    if rows:
        ts = _row_stride(t, name)
    else:
        _need_cuda(t, name)
        ts = t.shape[-1] if t.dim() else 0
    if t.dtype != dtype:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "%s must be %s, got %s" % (name, dtype, t.dtype))
    if isinstance(shape, int):
        ok = t.dim() == shape
    else:
        ok = t.dim() == len(shape) and all(e is None or int(e) == int(g) for e, g in zip(shape, t.shape))
    if not ok:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "%s has shape %s, expected %s" % (name, tuple(t.shape), shape))
    return ts


def rows_like(X, shape, dtype=torch.complex64):
    """Tensor of `shape` [..., T] whose rows are spaced like the rows of X (the C-ABI's per-frame outputs share the snapshots'
    T_stride): contiguous for contiguous X, a [..., :T] view of a padded buffer for row-padded X."""
    ts = X.stride(-2) if X.dim() >= 2 else X.shape[-1]
    T = shape[-1]
    if ts == T:
        return torch.empty(shape, dtype=dtype, device=X.device)
    return torch.empty(tuple(shape[:-1]) + (ts,), dtype=dtype, device=X.device)[..., :T]


def _row_stride(t, name):
    """Row stride (in elements) of a cuda tensor [..., rows, T] whose rows are contiguous and evenly spaced -- a contiguous
    tensor or a [..., :T] view of a buffer with padded rows (the C-ABI takes T_stride >= T everywhere)."""
    if not t.is_cuda:
        raise ValueError("%s must live in HBM (cuda tensor); the engine has no CPU path" % name)
    if t.numel() == 0:                          # an empty bin shard: nothing is read or written, any stride will do
        return max(int(t.shape[-1]), 1) if t.dim() else 1
    if t.dim() < 2 or t.stride(-1) != 1 or t.stride(-2) < t.shape[-1]:
        raise ValueError("%s: rows must be contiguous" % name)
    for i in range(t.dim() - 3, -1, -1):
        if t.shape[i] > 1 and t.stride(i) != t.shape[i + 1] * t.stride(i + 1):
            raise ValueError("%s: rows must be evenly spaced" % name)
    return t.stride(-2)


def padded_rows(shape, dtype, device, row_bytes_quantum=4096, pad=48):
    """Tensor of `shape` whose last-axis rows are `pad` elements apart from being contiguous whenever a contiguous row
    would be a multiple of 4 KiB: power-of-two row strides put the 257 bin rows of a tile on the same HBM channels
    (profiles/pad_ab.py: fused kernel 1.78 -> 1.67 ms, apply 1.66 -> 1.40 ms with 16-48 frames of padding)."""
    T = shape[-1]
    esz = torch.empty((), dtype=dtype).element_size()
    if T == 0 or (T * esz) % row_bytes_quantum:
        return torch.empty(shape, dtype=dtype, device=device)
    buf = torch.empty(tuple(shape[:-1]) + (T + pad,), dtype=dtype, device=device)
    return buf[..., :T]


class FilterBank:
    """Plan of an oversampled modulated-DFT bank (OverSampledDFTFilterBank, modulated.cc:232-268)."""

    def __init__(self, prototype, M, m, r, delay_compensation_type=0, synthesis=False):
        proto = np.ascontiguousarray(prototype, np.float64)
        if proto.shape != (m * M,):
            raise _lib.BtkError(_lib.BTK_ERR_CONSISTENCY,
                                "Prototype sizes do not match (%d vs. %d)." % (proto.size, m * M))
        self.M, self.m, self.r = M, m, r
        self.R = 1 << r
        self.D = M // self.R
        self.K = M // 2 + 1
        self.synthesis = bool(synthesis)
        self._h = C.c_void_p()
        check(_lib.lib().btk_fb_create(C.byref(self._h), M, m, r, delay_compensation_type, int(synthesis), _np_ptr(proto)))
        self.processing_delay = _lib.lib().btk_fb_processing_delay(self._h)
        self.lookahead = _lib.lib().btk_fb_lookahead(self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().btk_fb_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- analysis
    def num_frames(self, nsamples):
        return _lib.lib().btk_fb_analysis_num_frames(self._h, nsamples)

    def analysis(self, pcm, nsamples=None, t0=0, tcount=None, out=None, bins=None, pad_rows=False):
        """pcm float32 [S][N][L] (cuda) -> X complex64 [S][K][N][T]; bins = (k0, k1): only that bin range is computed
        into X [S][k1-k0][N][T] (the bin shard of one rank, sharding.py).  pad_rows: allocate X with padded rows
        (padded_rows: power-of-two row pitches put a tile's rows on the same HBM channels) -- bf_apply and nlms_process
        accept such a view, the other consumers want contiguous snapshots; `out` may itself be a row-padded view.
        pcm may be int16 -- the samples as a WAV stores them (feature/feature.cc:265-269) -- where the geometry has the int16 form of
        the bank (analysis_i16(): btk_fb_analysis_i16, M = 512; the whole bin range): half the PCM bytes, the same bits out."""
        i16 = pcm.dtype == torch.int16
        _check(pcm, "pcm", torch.int16 if i16 else torch.float32, 3)
        if i16 and (bins is not None or not self.analysis_i16()):
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "no int16 analysis kernel for M=%d m=%d r=%d%s: widen with pcm_i16_to_f32 first"
                                % (self.M, self.m, self.r, " on a bin range" if bins is not None else ""))
        S, N, L = pcm.shape
        nsamples = L if nsamples is None else nsamples
        if tcount is None:
            tcount = self.num_frames(nsamples) - t0
        k0, k1 = (0, self.K) if bins is None else (int(bins[0]), int(bins[1]))
        if not (0 <= k0 <= k1 <= self.K):
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "bin range [%d, %d) outside [0, %d]" % (k0, k1, self.K))
        if out is None:
            out = (padded_rows((S, k1 - k0, N, tcount), torch.complex64, pcm.device) if pad_rows
                   else torch.empty((S, k1 - k0, N, tcount), dtype=torch.complex64, device=pcm.device))
        ts = _check(out, "X", torch.complex64, (S, k1 - k0, N, None), rows=True)
        if out.shape[3] < tcount:
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "X holds %d frames, %d requested" % (out.shape[3], tcount))
        if i16:
            check(_lib.lib().btk_fb_analysis_i16(self._h, _ptr(pcm), nsamples, L, S, N, _ptr(out), ts, t0, tcount, _stream()))
        elif bins is None:
            check(_lib.lib().btk_fb_analysis(self._h, _ptr(pcm), nsamples, L, S, N, _ptr(out), ts, t0, tcount, _stream()))
        else:
            check(_lib.lib().btk_fb_analysis_bins(self._h, _ptr(pcm), nsamples, L, S, N, _ptr(out), ts, t0, tcount,
                                                  k0, k1, _stream()))
        return out

    def analysis_polyphase(self, pcm, nsamples=None, t0=0, tcount=None):
        """Polyphase sums before the FFT: float32 [S*N][T][M] (parity/debug entry)."""
        _need_cuda(pcm, "pcm")
        S, N, L = pcm.shape
        nsamples = L if nsamples is None else nsamples
        if tcount is None:
            tcount = self.num_frames(nsamples) - t0
        P = torch.empty((S * N, tcount, self.M), dtype=torch.float32, device=pcm.device)
        check(_lib.lib().btk_fb_analysis_polyphase(self._h, _ptr(pcm), nsamples, L, S, N, _ptr(P), t0, tcount, _stream()))
        return P

    def analysis_beamform(self, pcm, W, nsamples=None, t0=0, tcount=None, out=None):
        """Fused analysis -> fixed-weight beamformer: pcm [S][N][L], W complex64 [S|1][K][N] -> Y [S][K][T]
        (the N x K snapshots are not written to HBM).  pcm float32, or int16 -- the samples as a WAV stores them
        (feature/feature.cc:265-269): widened inside the kernel, half the bytes, the same bits out (btk_fb_analysis_bf_i16;
        geometries with an int16 kernel: fused_i16())."""
        i16 = pcm.dtype == torch.int16
        _check(pcm, "pcm", torch.int16 if i16 else torch.float32, 3)
        if i16 and not self.fused_i16():
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "no int16 fused kernel for M=%d m=%d r=%d: widen with pcm_i16_to_f32 first" % (self.M, self.m, self.r))
        S, N, L = pcm.shape
        nsamples = L if nsamples is None else nsamples
        if tcount is None:
            tcount = self.num_frames(nsamples) - t0
        if W.dim() == 2:
            W = W.unsqueeze(0)
        _check(W, "W", torch.complex64, (None, self.K, N))
        if W.shape[0] not in (1, S):
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "weights %s do not match pcm %s" % (tuple(W.shape), tuple(pcm.shape)))
        per_stream = int(W.shape[0] == S and S > 1)
        if out is None:
            # (the staged fall-back of the other geometries needs contiguous rows)
            fused = self.m == 4 and ((self.M in (256, 512) and self.r <= 2) or (self.M in (1024, 2048) and self.r == 1))
            out = (padded_rows((S, self.K, tcount), torch.complex64, pcm.device) if fused
                   else torch.empty((S, self.K, tcount), dtype=torch.complex64, device=pcm.device))
        t_stride = _row_stride(out, "Y")          # out may be a [..., :T] view of a row-padded buffer
        if out.dtype != torch.complex64 or tuple(out.shape[:2]) != (S, self.K) or out.shape[2] < tcount:
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Y must be complex64 [%d][%d][>=%d], got %s %s"
                                % (S, self.K, tcount, out.dtype, tuple(out.shape)))
        nb = _lib.lib().btk_fb_analysis_bf_scratch_bytes(self._h, S, N, per_stream, tcount)
        # the weight-pair scratch is written by a kernel of THIS launch's stream: one buffer per (device, stream), so that a plan
        # shared by several streams (serving.BatchBeamformerPipeline next to a caller's own stream) never has two launches
        # re-packing weights into the same bytes
        # The buffers come from torch's stream-aware caching allocator and are RETURNED to it when evicted: it re-issues a block only
        # to the stream that allocated it, so a recycled raw stream handle cannot be handed bytes another stream is still writing.
        # At most eight (device, stream) pairs are kept, least recently used first out: transient torch.cuda.Stream objects do not
        # pile up one buffer each.
        key = (pcm.device.index, int(torch.cuda.current_stream().cuda_stream))
        cache = self.__dict__.setdefault("_bf_scratch", collections.OrderedDict())
        buf = cache.pop(key, None)
        if buf is None or buf.numel() < nb:
            buf = torch.empty(nb, dtype=torch.uint8, device=pcm.device)
        cache[key] = buf
        while len(cache) > 8:
            cache.popitem(last=False)
        fn = _lib.lib().btk_fb_analysis_bf_i16 if i16 else _lib.lib().btk_fb_analysis_bf
        check(fn(self._h, _ptr(pcm), nsamples, L, S, N, _ptr(W), per_stream, _ptr(out), t_stride, t0, tcount, _ptr(buf), buf.numel(), _stream()))
        return out

    def analysis_i16(self):
        """True when analysis() takes int16 samples for this geometry (btk_fb_analysis_i16_direct)."""
        return _lib.lib().btk_fb_analysis_i16_direct(self._h) == 1

    def fused_i16(self):
        """True when analysis_beamform takes int16 samples for this geometry (btk_fb_analysis_bf_i16_fused)."""
        return _lib.lib().btk_fb_analysis_bf_i16_fused(self._h) == 1

    # ---- synthesis
    def num_blocks(self, nframes):
        return _lib.lib().btk_fb_synthesis_num_blocks(self._h, nframes)

    def synthesize(self, Y, nframes=None, b0=0, bcount=None, out=None):
        """Y complex64 [S][K][T] (cuda) -> float32 [S][bcount*D]."""
        t_stride = _row_stride(Y, "Y")            # Y may be a [..., :T] view of a row-padded buffer
        S, K, T = Y.shape
        if K != self.K:
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Y has %d bins, plan has %d" % (K, self.K))
        nframes = T if nframes is None else nframes
        if bcount is None:
            bcount = self.num_blocks(nframes) - b0
        if out is None:
            out = torch.empty((S, bcount * self.D), dtype=torch.float32, device=Y.device)
        check(_lib.lib().btk_fb_synthesis(self._h, _ptr(Y), nframes, t_stride, S, _ptr(out), out.shape[1], b0, bcount, _stream()))
        return out


def bf_apply(W, X, out=None):
    """y_k[t] = w_k^H x_k[t].  W complex64 [S|1][K][N], X complex64 [S][K][N][T] -> Y [S][K][T].  X may be a row-padded view
    (analysis(pad_rows=True)); Y then shares its row stride (the C-ABI has one T_stride for both)."""
    ts = _check(X, "X", torch.complex64, 4, rows=True)
    S, K, N, T = X.shape
    if W.dim() == 2:
        W = W.unsqueeze(0)
    _check(W, "W", torch.complex64, 3)
    if W.shape[1:] != (K, N) or W.shape[0] not in (1, S):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "weights %s do not match X %s" % (tuple(W.shape), tuple(X.shape)))
    if out is None:
        out = rows_like(X, (S, K, T))
    if _check(out, "Y", torch.complex64, (S, K, T), rows=True) != ts and K * T:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Y rows are %d frames apart, X rows %d: they share T_stride" % (out.stride(-2), ts))
    if K == 0 or T == 0:                       # the empty bin shard of a trailing rank: nothing to launch
        return out
    check(_lib.lib().btk_bf_apply(_ptr(W), int(W.shape[0] == S and S > 1), _ptr(X), _ptr(out), S, K, N, ts, T, _stream()))
    return out


def bf_apply_all_bins(Wfull, X):
    """half_band_shift == true (SubbandDS/GSC::next, beamformer.cc:1113-1128, 1276-1285): every one of the M bins has its
    own weight vector, y_k = w_k^H x_k for k = 0..M-1, where the snapshots of the bins above M/2 are the conjugate mirrors
    the analysis bank of a real signal produces, x_{M-k} = conj(x_k).  Wfull complex64 [M][N], X [S][K][N][T] ->
    Y complex64 [S][M][T]:  y_{M-k} = conj((conj w_{M-k})^H x_k), i.e. two passes of the same apply kernel."""
    _check(X, "X", torch.complex64, 4)
    S, K, N, T = X.shape
    M = 2 * (K - 1)
    _check(Wfull, "W", torch.complex64, (M, N))
    Y = torch.empty((S, M, T), dtype=torch.complex64, device=X.device)
    lo = bf_apply(Wfull[:K].contiguous(), X)
    Y[:, :K] = lo
    W2 = torch.zeros((K, N), dtype=torch.complex64, device=X.device)
    W2[1:K - 1] = torch.conj(Wfull[K:]).flip(0)              # row k' <- conj(w_{M-k'}), k' = 1 .. M/2-1
    up = bf_apply(W2, X)
    Y[:, K:] = torch.conj(up[:, 1:K - 1]).flip(1)
    return Y


# ---------------------------------------------------------------------------- host-side weight design
def weights_mainlobe(M, N, samplerate, delays, half_band_shift=False):
    """BeamformerWeights::calcMainlobe -> wq complex128 [M][N] (half_band_shift: beamformer.cc:515-527)."""
    delays = np.ascontiguousarray(delays, np.float64)
    if delays.shape != (N,):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION,
                            "Number of delays does not match number of channels (%d vs. %d)." % (delays.size, N))
    wq = np.zeros((M, N), np.complex128)
    fn = _lib.lib().btk_weights_mainlobe_halfband if half_band_shift else _lib.lib().btk_weights_mainlobe
    check(fn(M, N, float(samplerate), _np_ptr(delays), _np_ptr(wq)))
    return wq


def weights_mainlobe_2(M, N, samplerate, delays_t, delays_i):
    """LCMV quiescent weights (target + one null): calcMainlobe2 -> wq complex128 [M][N]."""
    dt = np.ascontiguousarray(delays_t, np.float64)
    di = np.ascontiguousarray(delays_i, np.float64)
    if dt.shape != (N,) or di.shape != (N,):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "The number of delays does not match number of channels (%d)" % N)
    wq = np.zeros((M, N), np.complex128)
    check(_lib.lib().btk_weights_mainlobe_2(M, N, float(samplerate), _np_ptr(dt), _np_ptr(di), _np_ptr(wq)))
    return wq


def weights_mainlobe_n(M, N, samplerate, delays_t, delays_is, NC):
    """LCMV quiescent weights (target + NC-1 nulls): calcMainlobeN -> wq complex128 [M][N]; delays_is [NC-1][N]."""
    dt = np.ascontiguousarray(delays_t, np.float64)
    di = np.ascontiguousarray(delays_is, np.float64).reshape(-1, N)
    if dt.shape != (N,) or di.shape != (NC - 1, N):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "The number of delays does not match number of channels (%d)" % N)
    wq = np.zeros((M, N), np.complex128)
    check(_lib.lib().btk_weights_mainlobe_n(M, N, float(samplerate), _np_ptr(dt), _np_ptr(di), int(NC), _np_ptr(wq)))
    return wq


def weights_blocking_matrix(a, NC=1):
    a = np.ascontiguousarray(a, np.complex128)
    N = a.shape[0]
    if N - NC <= 0:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "The number of sensors %d > the number of constraints %d" % (N, NC))
    B = np.zeros((N, N - NC), np.complex128)
    check(_lib.lib().btk_weights_blocking_matrix(_np_ptr(a), N, NC, _np_ptr(B)))
    return B


def weights_sidelobe(B, wa):
    B = np.ascontiguousarray(B, np.complex128)
    wa = np.ascontiguousarray(wa, np.complex128)
    N, bs = B.shape
    wl = np.zeros(N, np.complex128)
    check(_lib.lib().btk_weights_sidelobe(_np_ptr(B), _np_ptr(wa), N, N - bs, _np_ptr(wl)))
    return wl


def weights_gsc_effective(wq, wl, M, normalize=False):
    """complex64 [K][N] numpy array ready for bf_apply (wq - wl, bin 0 = wq_0)."""
    wq = np.ascontiguousarray(wq, np.complex128)
    N = wq.shape[1]
    wlp = None
    if wl is not None:
        wl = np.ascontiguousarray(wl, np.complex128)
        wlp = _np_ptr(wl)
    out = np.zeros((M // 2 + 1, N), np.complex64)
    check(_lib.lib().btk_weights_gsc_effective(_np_ptr(wq), wlp, M, N, int(normalize), _np_ptr(out)))
    return out


# ---------------------------------------------------------------------------- adaptive canceller (NLMS)
NLMS_DEFAULTS = dict(beta=0.97, gamma=0.01, init_diagonal_load=1.0e6, regularization_param=1.0e-4,
                     energy_floor=90.0, sil_thresh=1.0e8, max_wa_l2norm=100.0, min_frames=128,
                     slowdown_after=4096)      # lib/pybeamformer.py:597-607 == unit_test/confs/gsclms.json


class NLMSState:
    """Device-resident state of S independent SubbandGSCLMSBeamformer recursions
    (reset_stats, lib/pybeamformer.py:745-758)."""

    def __init__(self, S, M, N, device, Nc=1, **kw):
        self.p = dict(NLMS_DEFAULTS)
        self.p.update(kw)
        self.S, self.M, self.N, self.K = S, M, N, M // 2 + 1
        self.Nc = int(Nc)
        self.cextra = None                     # Nc > 1: complex64 [K][Nc-1][N], see set_constraints
        self.u = torch.zeros((S, self.K, N), dtype=torch.complex64, device=device)
        self.sigma2 = torch.empty((S, self.K), dtype=torch.float32, device=device)
        self.stream_state = torch.empty((S, 4), dtype=torch.float64, device=device)
        self.reset_stats()
        self._ws = None
        self._ws_groups = {}                   # interleaved launches: a workspace per stream group
        self._streams = None                   # ... and their HIP streams

    def set_constraints(self, vs):
        """Nc > 1: derive the extra projector directions from the array manifold vs complex [K][N] (host)."""
        if self.Nc > 1:
            self.cextra = torch.from_numpy(nlms_constraint_vectors(vs, self.Nc)).to(self.u.device)

    def reset_stats(self):
        self.u.zero_()
        self.sigma2.fill_(self.p["init_diagonal_load"])
        self.stream_state[:, 0] = self.p["init_diagonal_load"]
        self.stream_state[:, 1] = self.p["gamma"]
        self.stream_state[:, 2] = 0
        self.stream_state[:, 3] = 0
        self.frames_done = 0                   # host copy of the streams' frame counter (every stream advances by T per call)

    def params_array(self):
        p = self.p
        return np.array([p["beta"], p["gamma"], p["regularization_param"], p["energy_floor"], p["sil_thresh"],
                         p["max_wa_l2norm"], p["min_frames"], p["slowdown_after"]], np.float32)

    def workspace(self, T):
        need = _lib.lib().btk_nlms_workspace_bytes(self.S, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.u.device)
        return self._ws

    def group_workspace(self, g, Sg, T):
        need = _lib.lib().btk_nlms_workspace_bytes(Sg, T)
        w = self._ws_groups.get(g)
        if w is None or w.numel() < need:
            w = self._ws_groups[g] = torch.empty(need, dtype=torch.uint8, device=self.u.device)
        return w

    def group_streams(self, G):
        return side_streams(self.u.device, G)


def nlms_process(vs, X, state, out=None, interleave=None):
    """Adaptive GSC over a block: vs complex64 [K][N] (cuda), X [S][K][N][T] -> Y [S][K][T]; state updated in place.
    X may be a row-padded view (analysis(pad_rows=True)); Y then shares its row stride.
    interleave: None = decide by the launch shape (_nlms_interleave_plan), (1, T) = one launch, (G, chunk) = G stream groups on G
    HIP streams in chunks of `chunk` frames (a multiple of 64)."""
    ts = _check(X, "X", torch.complex64, 4, rows=True)
    S, K, N, T = X.shape
    _check(vs, "vs", torch.complex64, (K, N))
    if (S, K, N) != (state.S, state.K, state.N):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "nlms_process: shapes do not match the state")
    if out is None:
        out = rows_like(X, (S, K, T))
    if _check(out, "Y", torch.complex64, (S, K, T), rows=True) != ts:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Y rows are %d frames apart, X rows %d: they share T_stride" % (out.stride(-2), ts))
    params = state.params_array()
    ws = state.workspace(T)
    Nc = getattr(state, "Nc", 1)
    if Nc > 1:
        cx = getattr(state, "cextra", None)
        if cx is None:
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "nlms_process: Nc = %d needs state.set_constraints(vs) first" % Nc)
        _check(cx, "cextra", torch.complex64, (K, Nc - 1, N))
    cxp = _ptr(state.cextra) if Nc > 1 else None
    plan = _nlms_interleave_plan(S, K, N, T, state) if interleave is None else tuple(interleave)
    G, chunk, stagger = plan[0], plan[1], (plan[2] if len(plan) > 2 else True)
    if G <= 1:
        check(_lib.lib().btk_nlms_process_nc(_np_ptr(params), _ptr(vs), cxp, Nc, _ptr(X), _ptr(out),
                                             S, state.M, N, ts, T, _ptr(state.u), _ptr(state.sigma2), _ptr(state.stream_state),
                                             _ptr(ws), _stream()))
    else:
        # G groups of streams on G HIP streams, frame chunk after frame chunk: every stream is its own recursion and the kernels
        # carry their state from launch to launch (chunks are multiples of 64 frames: the same bits), so the groups may drift
        # apart -- and they do, which is the point (see _nlms_interleave_plan)
        cur = torch.cuda.current_stream()
        streams = state.group_streams(G)
        e0 = torch.cuda.Event()
        e0.record(cur)
        bounds = [(g * S) // G for g in range(G + 1)]
        lib = _lib.lib()
        for st in streams:
            st.wait_event(e0)
        # group g's first chunk is (g + 1) / G of a chunk (in units of 64 frames): the groups' launch boundaries start out spread
        # over the chunk period instead of coinciding
        segs = []
        for g in range(G):
            first = max(64, (chunk * (g + 1) // G) // 64 * 64) if stagger else chunk
            a = 0
            while a < T:
                n = min(first if a == 0 else chunk, T - a)
                segs.append((a, g, n))
                a += n
        for a, g, n in sorted(segs):
            _nlms_group_launch(lib, params, vs, cxp, Nc, X, out, ts, state, bounds[g], bounds[g + 1], a, n, g, chunk, streams[g])
        for st in streams:
            e = torch.cuda.Event()
            e.record(st)
            cur.wait_event(e)
            for t in (X, out, vs, state.u, state.sigma2, state.stream_state):
                t.record_stream(st)
    state.frames_done = getattr(state, "frames_done", 0) + T
    return out


# The engine's side HIP streams, ONE set per device shared by everything that forks work (interleaved canceller launches,
# AdaptiveGSCChain): the runtime maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams
# that land on one queue run their kernels in turn -- a second pair of streams created after the first made the chain's bank and
# canceller share a queue (9.6 -> 10.7 ms in bench.py, 9.95 with 8 queues).
_SIDE_STREAMS = {}


def side_streams(device, n):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _nlms_group_launch(lib, params, vs, cxp, Nc, X, out, ts, state, s0, s1, a, n, g, chunk_cap, st):
    """frames [a, a + n) of the streams [s0, s1) on the HIP stream st (group g's own workspace, sized for chunk_cap frames)"""
    if s1 <= s0 or n <= 0:
        return
    wg = state.group_workspace(g, s1 - s0, max(chunk_cap, n))
    check(lib.btk_nlms_process_nc(_np_ptr(params), _ptr(vs), cxp, Nc, _ptr(X[s0:s1, :, :, a:a + n]), _ptr(out[s0:s1, :, a:a + n]),
                                  s1 - s0, state.M, X.shape[2], ts, n, _ptr(state.u[s0:s1]), _ptr(state.sigma2[s0:s1]),
                                  _ptr(state.stream_state[s0:s1]), _ptr(wg), C.c_void_p(st.cuda_stream)))


# Wavefronts the canceller kernel of a channel count keeps resident on an MI355X (csrc/nlms_kernels.hip: one single-wavefront
# workgroup per (stream, group of bins); 256 CUs x 4 SIMDs x wavefronts per SIMD at the kernel's register count) and bins per
# wavefront.  Every wavefront walks all frames of the launch, so a launch costs (residency rounds) x (one walk): 32 streams x 65
# bin groups = 2 080 workgroups at 64 channels are a round of 2 048 and a round of 32 that takes as long -- 5.0-5.3 ms where 31
# streams take 3.85 (profiles/r06_nlms_residency.txt).  Cut into frame chunks and a few independent groups of streams on their own
# HIP streams, the groups drift out of step, a group's stragglers run beside the next chunk of the others, and the chip stays
# full: 3.75 ms, bit-identical.  Left to themselves the groups sometimes stay in step (one run in five: 5.3-5.6 ms again), so group g's
# first chunk is shortened to (g + 1) / G of a chunk: two groups x 512 frames, staggered, 3.74-3.77 ms five times out of five
# (profiles/r06_nlms_interleave.txt).
_NLMS_RESIDENT = ((8, 8, 3072), (16, 4, 3072), (32, 4, 2048), (64, 4, 2048), (128, 2, 2048), (1 << 30, 1, 2048))


def _nlms_interleave_plan(S, K, N, T, state):
    """(groups, chunk_frames, stagger) for nlms_process: (1, T, False) = one launch"""
    if os.environ.get("BTK_NLMS_INTERLEAVE", "1") == "0" or S < 4 or T < 512:
        return 1, T, False
    if getattr(state, "frames_done", 0) % 64:
        return 1, T, False                     # chunk boundaries must be multiples of 64 of the streams' frame counter
    bpw, slots = next((b, r) for n, b, r in _NLMS_RESIDENT if N <= n)
    nwg = S * ((K + bpw - 1) // bpw)
    if nwg <= slots:
        return 1, T, False
    return 2, (512 if T >= 1024 else 256), True


class AdaptiveGSCChain:
    """The adaptive chain of lib/pybeamformer.py:659-762 over analysis banks, S streams at once:
    PCM -> analysis bank -> snapshots X [S][K][N][T] (HBM) -> NLMS sidelobe canceller -> Y -> synthesis bank -> PCM.

    The canceller is a recursion in t that keeps two wavefronts per SIMD busy at C0 and leaves most of the register file and of
    the HBM bandwidth idle; the analysis bank streams.  So the bank runs AHEAD on one HIP stream, frame chunk after frame chunk
    into the one snapshot buffer, and the canceller follows on a second stream as soon as a chunk's snapshots are complete
    (profiles/adaptive_overlap_ab.py: 10.4 -> 9.7 ms per 32 x 4096 frames at C0).  Chunks are multiples of 64 frames, so the
    output is bit-identical to the one-launch-per-kernel chain (state and scan chunks carry, csrc/nlms_kernels.hip)."""

    def __init__(self, afb, sfb, chunk_frames=512, groups=None):
        if chunk_frames < 64 or chunk_frames % 64:
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "chunk_frames must be a multiple of 64, got %r" % (chunk_frames,))
        self.afb, self.sfb, self.chunk = afb, sfb, int(chunk_frames)
        self.groups = groups                   # canceller stream groups per chunk: None = by the launch shape, 1 = the round-5 form
        self._sa = self._sb = None

    def __call__(self, pcm, vs, state, X, Y, out=None, nsamples=None):
        """pcm [S][N][L]; X [S][K][N][T] and Y [S][K][T] (row-padded views allowed, same row pitch); returns the PCM blocks."""
        T = X.shape[-1]
        if self._sa is None:
            self._sa, self._sb = side_streams(pcm.device, 2)
        if out is None:
            # allocated on the CALLER's stream, where it is consumed: a block taken inside the side stream's context would belong to
            # that stream's pool and could be handed on while the caller still reads it
            out = torch.empty((X.shape[0], self.sfb.num_blocks(T) * self.sfb.D), dtype=torch.float32, device=pcm.device)
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        self._sa.wait_event(ev0)
        self._sb.wait_event(ev0)
        S, K, N = X.shape[0], X.shape[1], X.shape[2]
        # round 6 (groups > 1, not the default): the canceller of a chunk as G independent groups of streams, each on its own HIP
        # stream behind the bank's chunk.  It is what makes the canceller ALONE 25 % faster at 32 streams (_nlms_interleave_plan);
        # in the chain the bank's launches already run in the canceller's residency gaps: 9.5 ms with 1 or 2 groups, 10.7-11.8
        # with 4 or 8 (profiles/r06_nlms_interleave.txt).
        G = 1 if self.groups is None else int(self.groups)    # (measured: no gain inside the chain, the bank's kernels already fill the gaps)
        if state.frames_done % 64:
            G = 1
        gstreams = side_streams(pcm.device, 2 + G)[2:] if G > 1 else []
        for st in gstreams:
            st.wait_event(ev0)
        if G > 1:
            ts = _check(X, "X", torch.complex64, 4, rows=True)
            params, lib = state.params_array(), _lib.lib()
            Nc = getattr(state, "Nc", 1)
            cxp = _ptr(state.cextra) if Nc > 1 else None
            bounds = [(g * S) // G for g in range(G + 1)]
        for a in range(0, T, self.chunk):
            n = min(self.chunk, T - a)
            with torch.cuda.stream(self._sa):
                self.afb.analysis(pcm, nsamples=nsamples, t0=a, tcount=n, out=X[..., a:a + n])
                e = torch.cuda.Event()
                e.record(self._sa)
            if G > 1:
                for g, st in enumerate(gstreams):
                    st.wait_event(e)
                    _nlms_group_launch(lib, params, vs, cxp, Nc, X, Y, ts, state, bounds[g], bounds[g + 1], a, n, g, self.chunk, st)
            else:
                self._sb.wait_event(e)
                with torch.cuda.stream(self._sb):
                    nlms_process(vs, X[..., a:a + n], state, out=Y[..., a:a + n], interleave=(1, n))
        if G > 1:
            state.frames_done += T
            for st in gstreams:
                e = torch.cuda.Event()
                e.record(st)
                self._sb.wait_event(e)
        with torch.cuda.stream(self._sb):
            out = self.sfb.synthesize(Y, out=out)
            e = torch.cuda.Event()
            e.record(self._sb)
        cur.wait_event(e)
        for t in (pcm, X, Y, out):                      # the caching allocator must not hand these blocks on before the side streams are done
            for st in [self._sa, self._sb] + list(gstreams):
                t.record_stream(st)
        for st in gstreams:
            for t in (vs, state.u, state.sigma2, state.stream_state):
                t.record_stream(st)
        return out


def constraint_vectors(vs, Nc):
    """The Nc - 1 extra orthonormal directions the Nc-constraint cancellers' projector loses per bin (include/btkhip.h):
    vs complex [K][N] (host) -> complex128 [K][Nc-1][N] (host)."""
    vs = np.ascontiguousarray(vs, np.complex128)
    K, N = vs.shape
    out = np.zeros((K, Nc - 1, N), np.complex128)
    for k in range(K):
        B = weights_blocking_matrix(vs[k], Nc)
        check(_lib.lib().btk_nlms_constraint_vectors(_np_ptr(vs[k]), _np_ptr(B), N, Nc, _np_ptr(out[k])))
    return out


def nlms_constraint_vectors(vs, Nc):
    """constraint_vectors in the NLMS kernel's element type: complex64 [K][Nc-1][N] (host)"""
    return constraint_vectors(vs, Nc).astype(np.complex64)


def nlms_u_to_wa(u, B):
    """wa^H (complex128 [N-Nc]) from the engine state u (complex [N]) and the bin's blocking matrix B [N][N-Nc]."""
    u = np.ascontiguousarray(u, np.complex128)
    B = np.ascontiguousarray(B, np.complex128)
    N = u.shape[0]
    Nc = N - B.shape[1]
    wa = np.zeros(N - Nc, np.complex128)
    check(_lib.lib().btk_nlms_u_to_wa_nc(_np_ptr(u), _np_ptr(B), N, Nc, _np_ptr(wa)))
    return wa


def nlms_wa_to_u(waH, B):
    waH = np.ascontiguousarray(waH, np.complex128)
    B = np.ascontiguousarray(B, np.complex128)
    N = B.shape[0]
    u = np.zeros(N, np.complex128)
    check(_lib.lib().btk_nlms_wa_to_u(_np_ptr(waH), _np_ptr(B), N, _np_ptr(u)))
    return u


# ---------------------------------------------------------------------------- RLS canceller
RLS_PY_DEFAULTS = dict(beta=0.97, gamma=0.04, mu=0.97, init_diagonal_load=1.0e6, regularization_param=1.0e-2,
                       sil_thresh=1.0e8, constraint_option=3, alpha2=10.0, max_wa_l2norm=100.0, min_frames=128)
RLS_CC_DEFAULTS = dict(mu=0.9, diagonal_weight=0.0, qctype=0, alpha=-1.0, normalize_weight=False, update=True)


class RLSState:
    """Device-resident state of S independent RLS sidelobe cancellers.
    mode 1: SubbandGSCRLSBeamformer (lib/pybeamformer.py:765-928); mode 0: SubbandGSCRLS (beamformer.cc:1447-1645).
    v complex128 [K][N] or [S][K][N] (cuda): vs resp. wq of bins 0..M/2."""

    def __init__(self, mode, S, M, N, v, Nc=1, **kw):
        if mode not in (0, 1):
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "RLSState: mode must be 0 or 1")
        self.mode = mode
        self.Nc = int(Nc)
        self.cx = None                         # Nc > 1: complex128 [K][Nc-1][N] on the device, the further blocked directions
        self.p = dict(RLS_PY_DEFAULTS if mode == 1 else RLS_CC_DEFAULTS)
        unknown = set(kw) - set(self.p)
        if unknown:
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "RLSState: unknown parameters %s" % sorted(unknown))
        self.p.update(kw)
        _need_cuda(v, "v")
        self.S, self.M, self.N, self.K = S, M, N, M // 2 + 1
        if v.dtype != torch.complex128 or v.shape[-2:] != (self.K, N) or v.dim() not in (2, 3):
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "RLSState: v must be complex128 [K][N] or [S][K][N]")
        self.v = v.contiguous()
        self.per_stream = 1 if v.dim() == 3 else 0
        dev = v.device
        self.P = torch.empty((S, self.K, N, N), dtype=torch.complex128, device=dev)
        self.w = torch.empty((S, self.K, N), dtype=torch.complex128, device=dev)
        self.stream_state = torch.zeros((S, 4), dtype=torch.float64, device=dev)
        self._ws = None
        if self.Nc > 1:
            if self.per_stream:
                raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "RLSState: Nc > 1 takes one v [K][N] for all streams")
            # conj(B) B^T = I - v v^H/|v|^2 - sum_j c_j c_j^H (mode 1); mode 0: B B^H, i.e. the complex conjugates (include/btkhip.h)
            cx = constraint_vectors(self.v.cpu().numpy(), self.Nc)
            self.cx = torch.from_numpy(np.conj(cx) if mode == 0 else cx).to(dev).contiguous()
        if mode == 1:
            self.reset_stats()

    def init_precision_matrix(self, p0):
        """P = p0 B B^H (mode 0) / p0 conj(B) B^T (mode 1), w = 0 (init_precision_matrix with p0 = 1/sigma2,
        beamformer.cc:1482-1494)"""
        check(_lib.lib().btk_rls_init_nc(self.mode, _ptr(self.v), self.per_stream, None if self.cx is None else _ptr(self.cx), self.Nc,
                                         float(p0), self.S, self.K, self.N, _ptr(self.P), _ptr(self.w), _stream()))

    def reset_stats(self):
        """pybeamformer.py:913-925"""
        self.init_precision_matrix(1.0 / self.p["init_diagonal_load"])
        self.stream_state.zero_()
        self.stream_state[:, 0] = self.p["init_diagonal_load"]

    def params_array(self):
        p = self.p
        if self.mode == 1:
            return np.array([p["beta"], p["gamma"], p["mu"], p["init_diagonal_load"], p["regularization_param"],
                             p["sil_thresh"], p["constraint_option"], p["alpha2"], p["max_wa_l2norm"], p["min_frames"]],
                            np.float64)
        return np.array([p["mu"], p["diagonal_weight"], p["qctype"], p["alpha"], float(bool(p["normalize_weight"])),
                         float(bool(p["update"]))], np.float64)

    def workspace(self, T):
        need = _lib.lib().btk_rls_workspace_bytes(self.S, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.P.device)
        return self._ws


def rls_process(X, state, out=None):
    """RLS canceller over a block: X complex64 [S][K][N][T] -> Y [S][K][T]; state updated in place."""
    _need_cuda(X, "X")
    S, K, N, T = X.shape
    if (S, K, N) != (state.S, state.K, state.N):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "rls_process: shapes do not match the state")
    if out is None:
        out = torch.empty((S, K, T), dtype=torch.complex64, device=X.device)
    params = state.params_array()
    ws = state.workspace(T)
    check(_lib.lib().btk_rls_process_nc(state.mode, _np_ptr(params), _ptr(state.v), state.per_stream,
                                        None if state.cx is None else _ptr(state.cx), state.Nc, _ptr(X), _ptr(out),
                                        S, state.M, N, T, T, _ptr(state.P), _ptr(state.w), _ptr(state.stream_state),
                                        _ptr(ws), _stream()))
    return out


def rls_state_to_reference(mode, P, w, B):
    """Host change of basis for one bin: engine (P [N][N], w [N]) -> reference (Pz [N-Nc][N-Nc], wa resp. waH [N-Nc])
    with the bin's blocking matrix B [N][N-Nc] (orthonormal columns)."""
    P = np.asarray(P, np.complex128)
    w = np.asarray(w, np.complex128)
    B = np.asarray(B, np.complex128)
    if mode == 1:      # P = conj(B) Pz B^T, u = waH B^T
        return B.T @ P @ np.conj(B), w @ np.conj(B)
    return np.conj(B.T) @ P @ B, np.conj(B.T) @ w      # P = B Pz B^H, wl = B wa


# ---------------------------------------------------------------------------- Zelinski post-filter
class ZelinskiState:
    """Summed CSD / PSD state of S post-filters (the part of BeamformerWeights::CSDs_/wp1_,
    beamformer.cc:874-887, that the Zelinski gain depends on)."""

    def __init__(self, S, K, device):
        self.phi = torch.zeros((S, K), dtype=torch.complex64, device=device)
        self.psi = torch.zeros((S, K), dtype=torch.float32, device=device)
        self.w_last = torch.zeros((S, K), dtype=torch.float32, device=device)
        self.frames_done = 0

    def reset_csd(self):
        """What alloc_bfweight_ does to the post-filter state when weights are recomputed
        (beamformer.cc:1082-1092): CSD history restarts, the frame counter keeps counting."""
        self.phi.zero_()
        self.psi.zero_()
        self.w_last.zero_()


def bf_apply_zelinski(W, D, X, state, alpha=0.6, type_=2, min_frames=0, out=None):
    """Beamform + Zelinski post-filter over a block (ZelinskiPostFilter over SubbandDS/GSC/MVDR).
    W, D complex64 [S|1][K][N]; X [S][K][N][T] -> Y [S][K][T] (post-filtered).  X may be a row-padded view
    (analysis(pad_rows=True)); Y and the per-frame statistics then share its row stride."""
    ts = _check(X, "X", torch.complex64, 4, rows=True)
    S, K, N, T = X.shape
    if W.dim() == 2:
        W, D = W.unsqueeze(0), D.unsqueeze(0)
    _check(W, "W", torch.complex64, (None, K, N)); _check(D, "D", torch.complex64, tuple(W.shape))
    if W.shape[0] not in (1, S):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "weights %s do not match X %s" % (tuple(W.shape), tuple(X.shape)))
    if out is None:
        out = rows_like(X, (S, K, T))
    if _check(out, "Y", torch.complex64, (S, K, T), rows=True) != ts:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Y rows are %d frames apart, X rows %d: they share T_stride" % (out.stride(-2), ts))
    Cc = rows_like(X, (S, K, T))
    Ee = rows_like(X, (S, K, T), dtype=torch.float32)
    L = _lib.lib()
    check(L.btk_bf_apply_stats(_ptr(W), _ptr(D), int(W.shape[0] == S and S > 1), _ptr(X), _ptr(out), _ptr(Cc), _ptr(Ee),
                               S, K, N, ts, T, _stream()))
    check(L.btk_zelinski_process(_ptr(out), _ptr(Cc), _ptr(Ee), S, K, N, ts, T, float(alpha), int(type_), int(min_frames),
                                 state.frames_done, _ptr(state.phi), _ptr(state.psi), _ptr(state.w_last), _stream()))
    state.frames_done += T
    return out


class CoherencePostFilterState:
    """State of S McCowan / Lefkimmiatis post-filters: the recursively averaged weighted CSD sums (see
    csrc/pf_kernels.hip) plus the per-bin pair-weight matrices built from the noise coherence matrix R_."""

    def __init__(self, S, K, N, device, lefkimmiatis=False):
        self.S, self.K, self.N = S, K, N
        self.lefkimmiatis = bool(lefkimmiatis)
        self.u = torch.zeros((S, K), dtype=torch.complex64, device=device)
        self.v = torch.zeros((S, K), dtype=torch.complex64, device=device)     # Lefkimmiatis only
        self.psi = torch.zeros((S, K), dtype=torch.float32, device=device)     # McCowan only
        self.w_last = torch.zeros((S, K), dtype=torch.float32, device=device)
        self.Cs = self.Cv = self.lam = None
        self.frames_done = 0

    def set_coherence(self, R, threshold=0.99):
        """R complex64 [K][N][N] (cuda): R_ after set_diffuse_noise_model / set_noise_spatial_spectral_matrix /
        diagonal loading (postfilter.cc:536-660); threshold = threshold_of_Rij_."""
        _need_cuda(R, "R")
        K, N = self.K, self.N
        if tuple(R.shape) != (K, N, N):
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "set_coherence: R must be [K][N][N]")
        self.Cs = torch.empty((K, N, N), dtype=torch.complex64, device=R.device)
        self.Cv = torch.empty((K, N, N), dtype=torch.complex64, device=R.device) if self.lefkimmiatis else None
        check(_lib.lib().btk_pf_coherence_coeffs(_ptr(R.contiguous()), float(np.float32(threshold)), K, N, _ptr(self.Cs),
                                                 None if self.Cv is None else _ptr(self.Cv), _stream()))

    def set_lambda(self, R, d, min_sv=1.0e-8, svd_rule=None):
        """Lambda_k = d^H pinv(R_k) d (calcLambda, postfilter.cc:982-995) for all bins; returns the number of bins
        that fell back to the identity (:975-977).  svd_rule as in mvdr_weights(): "linpack" (default) takes the identity
        exactly where the reference's pseudoinverse() returns false (all bins, bin 0 included: postfilter.cc:971)."""
        _need_cuda(R, "R"); _need_cuda(d, "d")
        K, N = self.K, self.N
        rule = svd_rule_default() if svd_rule is None else svd_rule
        if rule not in SVD_RULES:
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "svd_rule must be one of %s, got %r" % (SVD_RULES, rule))
        R, d = R.contiguous(), d.contiguous()
        self.lam = torch.empty((K,), dtype=torch.complex64, device=R.device)
        fb = torch.zeros(1, dtype=torch.int32, device=R.device)
        sb = _lib.lib().btk_mvdr_scratch_bytes(K, N)
        scratch = torch.empty((sb,), dtype=torch.uint8, device=R.device) if sb else None
        check(_lib.lib().btk_mvdr_lambda(_ptr(R), _ptr(d), _ptr(self.lam), K, N, float(min_sv),
                                         None if scratch is None else _ptr(scratch), _ptr(fb), _stream()))
        if rule == "linpack":
            counts = _linpack_rule(R, d, None, self.lam, K, N, 0, 0, 0, min_sv, None)
            return int(counts.sum().item())
        return int(fb.item())


def _pf_args(W, D, X, out):
    """Shape / dtype / row-stride checks shared by the coherence post-filters; returns (T_stride, S, K, N, T, W, D, out)."""
    ts = _check(X, "X", torch.complex64, 4, rows=True)
    S, K, N, T = X.shape
    if W.dim() == 2:
        W, D = W.unsqueeze(0), D.unsqueeze(0)
    _check(W, "W", torch.complex64, (None, K, N)); _check(D, "D", torch.complex64, tuple(W.shape))
    if W.shape[0] not in (1, S):
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "weights %s do not match X %s" % (tuple(W.shape), tuple(X.shape)))
    if out is None:
        out = rows_like(X, (S, K, T))
    if _check(out, "Y", torch.complex64, (S, K, T), rows=True) != ts:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Y rows are %d frames apart, X rows %d: they share T_stride" % (out.stride(-2), ts))
    return ts, S, K, N, T, W, D, out


def bf_apply_mccowan(W, D, X, state, alpha=0.6, type_=2, min_frames=0, out=None):
    """Beamform + McCowan post-filter over a block.  W, D complex64 [S|1][K][N]; X [S][K][N][T] -> Y [S][K][T].  X may be a
    row-padded view (analysis(pad_rows=True)); Y and the per-frame statistics then share its row stride."""
    ts, S, K, N, T, W, D, out = _pf_args(W, D, X, out)
    if state.Cs is None:
        raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "McCowanPostFilter:  construct/set a noise coherence matrix")
    U = rows_like(X, (S, K, T))
    Ee = rows_like(X, (S, K, T), dtype=torch.float32)
    L = _lib.lib()
    check(L.btk_bf_apply_stats2(_ptr(W), _ptr(D), int(W.shape[0] == S and S > 1), _ptr(X), _ptr(out), _ptr(state.Cs), None,
                                _ptr(U), None, _ptr(Ee), S, K, N, ts, T, _stream()))
    check(L.btk_zelinski_process(_ptr(out), _ptr(U), _ptr(Ee), S, K, N, ts, T, float(alpha), int(type_) & 3, int(min_frames),
                                 state.frames_done, _ptr(state.u), _ptr(state.psi), _ptr(state.w_last), _stream()))
    state.frames_done += T
    return out


def bf_apply_lefkimmiatis(W, D, X, state, fbin_x1=0, alpha=0.6, type_=2, min_frames=0, out=None):
    """Beamform + Lefkimmiatis post-filter over a block (D = array manifold); X may be a row-padded view as for McCowan."""
    ts, S, K, N, T, W, D, out = _pf_args(W, D, X, out)
    if state.Cs is None or state.Cv is None or state.lam is None:
        raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "LefkimmiatisPostFilter:  construct/set a noise coherence matrix")
    U = rows_like(X, (S, K, T))
    V = rows_like(X, (S, K, T))
    Ee = rows_like(X, (S, K, T), dtype=torch.float32)
    L = _lib.lib()
    check(L.btk_bf_apply_stats2(_ptr(W), _ptr(D), int(W.shape[0] == S and S > 1), _ptr(X), _ptr(out), _ptr(state.Cs),
                                _ptr(state.Cv), _ptr(U), _ptr(V), _ptr(Ee), S, K, N, ts, T, _stream()))
    check(L.btk_lefkimmiatis_process(_ptr(out), _ptr(U), _ptr(V), _ptr(state.lam), int(fbin_x1), S, K, N, ts, T, float(alpha),
                                     int(type_), int(min_frames), state.frames_done, _ptr(state.u), _ptr(state.v),
                                     _ptr(state.w_last), _stream()))
    state.frames_done += T
    return out


# ---------------------------------------------------------------------------- covariance accumulation
def frame_energy(X, M):
    _need_cuda(X, "X")
    S, K, N, T = X.shape
    e = torch.empty((S, T), dtype=torch.float32, device=X.device)
    check(_lib.lib().btk_frame_energy(_ptr(X), S, M, N, T, T, _ptr(e), T, _stream()))
    return e


def cov_frame_gate(energy, label, threshold, count=None):
    """w = (energy > threshold) * label; returns (w [S][T], count [S])."""
    S, T = energy.shape
    w = torch.empty_like(energy)
    if count is None:
        count = torch.zeros(S, dtype=torch.float32, device=energy.device)
    check(_lib.lib().btk_cov_frame_gate(_ptr(energy), None if label is None else _ptr(label), S, T, T, float(threshold),
                                        _ptr(w), _ptr(count), _stream()))
    return w, count


def cov_accumulate(X, R=None, tf_weights=None, frame_weights=None, use_mfma=True):
    """R[s][k] += sum_t tf[s][k][t] fw[s][t] x x^H.  X [S][K][N][T] -> R complex64 [S][K][N][N].  X may be a row-padded view
    (analysis(pad_rows=True)); the per-frame weights, if any, then need the same row stride (rows_like(X, ...))."""
    ts = _check(X, "X", torch.complex64, 4, rows=True)
    S, K, N, T = X.shape
    if R is None:
        R = torch.zeros((S, K, N, N), dtype=torch.complex64, device=X.device)
    _check(R, "R", torch.complex64, (S, K, N, N))
    for w, name, shape in ((tf_weights, "tf_weights", (S, K, T)), (frame_weights, "frame_weights", (S, T))):
        if w is not None and _check(w, name, torch.float32, shape, rows=True) != ts:
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "%s rows are %d frames apart, X rows %d: they share T_stride" % (name, w.stride(-2), ts))
    check(_lib.lib().btk_cov_accumulate(_ptr(X), None if tf_weights is None else _ptr(tf_weights),
                                        None if frame_weights is None else _ptr(frame_weights), _ptr(R),
                                        S, K, N, ts, T, int(use_mfma), _stream()))
    return R


def cov_finalize(R, count, gamma=0.0):
    S, K, N, _ = R.shape
    per_bin = int(count.dim() == 2)
    check(_lib.lib().btk_cov_finalize(_ptr(R), _ptr(count), per_bin, S, K, N, float(gamma), _stream()))
    return R


def cov_mask_count(tf_weights, frame_weights=None, count=None):
    """count[s][k] += sum_t trunc(tf[s][k][t]) fw[s][t]  (integer per-bin counters of accu_stats_from_tfmask)."""
    _need_cuda(tf_weights, "tf_weights")
    S, K, T = tf_weights.shape
    if count is None:
        count = torch.zeros((S, K), dtype=torch.float32, device=tf_weights.device)
    check(_lib.lib().btk_cov_mask_count(_ptr(tf_weights), None if frame_weights is None else _ptr(frame_weights), S, K, T, T,
                                        _ptr(count), _stream()))
    return count


def cov_trace_normalize(R):
    N = R.shape[-1]
    check(_lib.lib().btk_cov_trace_normalize(_ptr(R), int(np.prod(R.shape[:-2])), N, _stream()))
    return R


def _sos_weights(which, Rt, Rn, *args):
    _need_cuda(Rt, "Rt"); _need_cuda(Rn, "Rn")
    K, N, _ = Rt.shape
    if Rn.shape != Rt.shape or Rt.dtype != torch.complex64 or Rn.dtype != torch.complex64:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "covariance matrices must both be complex64 [K][N][N]")
    L = _lib.lib()
    W = torch.empty((K, N), dtype=torch.complex64, device=Rt.device)
    scratch = torch.empty(L.btk_sos_scratch_bytes(K, N), dtype=torch.uint8, device=Rt.device)
    fail = torch.zeros(1, dtype=torch.int32, device=Rt.device)
    Rt, Rn = Rt.contiguous(), Rn.contiguous()
    if which == "bmvdr":
        check(L.btk_bmvdr_weights(_ptr(Rt), _ptr(Rn), K, N, int(args[0]), float(args[1]), _ptr(W), _ptr(scratch), _ptr(fail),
                                  _stream()))
    else:
        check(L.btk_gev_weights(_ptr(Rt), _ptr(Rn), K, N, _ptr(W), _ptr(scratch), _ptr(fail), _stream()))
    return W, int(fail.item())


def bmvdr_weights(Rt, Rn, ref_micx=0, offset=0.0):
    """Blind MVDR: wqH [K][N] complex64 (cuda) and the number of bins whose noise covariance is not positive definite."""
    return _sos_weights("bmvdr", Rt, Rn, ref_micx, offset)


def gev_weights(Rt, Rn):
    """GEV: wqH [K][N] complex64 (cuda), equal to the reference's up to one global sign; and the failure count."""
    return _sos_weights("gev", Rt, Rn)


# ---------------------------------------------------------------------------- MVDR weight design
def mvdr_diffuse_model(mpos, M, samplerate, sspeed=343740.0, device=None):
    mp = torch.as_tensor(np.ascontiguousarray(mpos, np.float32)).to(device)
    N = mp.shape[0]
    if mp.shape[1] < 3:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "The microphone positions should be described in the three dimensions")
    R = torch.empty((M // 2 + 1, N, N), dtype=torch.complex64, device=mp.device)
    check(_lib.lib().btk_mvdr_diffuse_model(_ptr(mp), N, M, float(samplerate), float(sspeed), _ptr(R), _stream()))
    return R


def mvdr_diagonal_loading(R, weight):
    nb, N = R.shape[-3], R.shape[-1]
    nb = int(np.prod(R.shape[:-2]))
    check(_lib.lib().btk_mvdr_diagonal_loading(_ptr(R), nb, N, float(weight), _stream()))
    return R


SVD_RULES = ("linpack", "exact")


def svd_rule_default():
    """The rule that decides where pseudoinverse() 'fails' and the identity replaces inv(R_k) (beamformer.cc:253-270, 2379-2384):
    "linpack" (default) takes the decision of the reference's float32 csvdc, rounding for rounding -- INFO != 0 or a singular
    value under the threshold -- so a design matches the reference bin for bin (on BASELINE config C5 that is delay-and-sum on
    about half the spectrum); "exact" solves every positive definite bin and uses the identity only where a singular value is
    really below the threshold.  BTK_MVDR_SVD_RULE overrides the default."""
    rule = os.environ.get("BTK_MVDR_SVD_RULE", "linpack")
    if rule not in SVD_RULES:
        raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "BTK_MVDR_SVD_RULE must be one of %s, got %r" % (SVD_RULES, rule))
    return rule


def csvdc_values(A):
    """LINPACK's float32 csvdc with job = 0 (matrix/linpack_c.cc:9516) for a batch: A complex64 [K][n][p] (cuda) ->
    (s float32 [K][m], e float32 [K][m], info int32 [K]), m = min(n + 1, p); bit-identical to the reference's compiled routine."""
    _check(A, "A", torch.complex64, 3)
    K, n, p = A.shape
    m = min(n + 1, p)
    s = torch.zeros((K, m), dtype=torch.float32, device=A.device)
    e = torch.zeros((K, m), dtype=torch.float32, device=A.device)
    info = torch.zeros((K,), dtype=torch.int32, device=A.device)
    if K > 0:
        sb = _lib.lib().btk_csvdc_scratch_bytes(K, n, p)
        scratch = torch.empty((sb,), dtype=torch.uint8, device=A.device) if sb else None
        check(_lib.lib().btk_csvdc_values(_ptr(A), K, n, p, _ptr(s), _ptr(e), _ptr(info), None if scratch is None else _ptr(scratch),
                                          _stream()))
    return s, e, info


def _linpack_rule(R, wq, W, lam, KS, N, first_bin, kper, skip_dc, threshold, flags):
    """btk_mvdr_linpack_rule on the current stream; returns the device counters [INFO != 0, singular value < threshold]."""
    counts = torch.zeros(2, dtype=torch.int32, device=R.device)
    scratch = torch.empty((_lib.lib().btk_mvdr_linpack_rule_scratch_bytes(KS, N),), dtype=torch.uint8, device=R.device)
    check(_lib.lib().btk_mvdr_linpack_rule(_ptr(R), _ptr(wq), None if W is None else _ptr(W), None if lam is None else _ptr(lam),
                                           KS, N, int(first_bin), int(kper), int(skip_dc), float(threshold),
                                           None if flags is None else _ptr(flags), _ptr(counts), _ptr(scratch), _stream()))
    return counts


def mvdr_weights(R, wq, threshold=1.0e-8, first_bin=0, svd_rule=None):
    """R complex64 [K][N][N], wq complex64 [K][N] (cuda) -> (W [K][N], number of bins that ended with the identity).
    first_bin: global index of row 0 when R / wq are one rank's bin range (only global bin 0 gets the all-ones weight).
    svd_rule (svd_rule_default() when None): with "linpack" the bins for which the reference's pseudoinverse() returns false --
    its float32 csvdc does not converge, or leaves a singular value under the threshold -- take the identity exactly as in
    calc_mvdr_weights (beamformer.cc:253-270, 2379-2396); every other bin keeps the Cholesky solution.  Bins whose Cholesky
    factorisation stops (R_k not positive definite) and that the rule did not already decide are re-solved through the
    pseudo-inverse (btk_mvdr_pinv_fallback).  mvdr_weights.last_counts = (INFO != 0, converged but under the threshold,
    identity from the fall-back) of the last call.
    S streams at once: R [S][K][N][N], wq [S][K][N] -> W [S][K][N] (btk_mvdr_weights_streams: one launch, every stream's bin 0
    gets the all-ones weight; first_bin must be 0)."""
    rule = svd_rule_default() if svd_rule is None else svd_rule
    if rule not in SVD_RULES:
        raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "svd_rule must be one of %s, got %r" % (SVD_RULES, rule))
    batched = R.dim() == 4
    if batched:
        _check(R, "R", torch.complex64, 4); _check(wq, "wq", torch.complex64, (R.shape[0], R.shape[1], R.shape[2]))
        if first_bin != 0:
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "mvdr_weights: stacked streams are whole bin ranges (first_bin = 0)")
        S, K, N, N2 = R.shape
    else:
        _check(R, "R", torch.complex64, 3); _check(wq, "wq", torch.complex64, (R.shape[0], R.shape[1]))
        S = 1
        K, N, N2 = R.shape
    if N2 != N:
        raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "R must be [K][N][N], got %s" % (tuple(R.shape),))
    W = torch.empty(tuple(wq.shape), dtype=torch.complex64, device=R.device)
    fb = torch.zeros(1, dtype=torch.int32, device=R.device)
    KS = K * S
    flags = torch.zeros(max(KS, 1), dtype=torch.int32, device=R.device)
    nident = 0
    mvdr_weights.last_counts = (0, 0, 0)
    if KS > 0:
        sb = _lib.lib().btk_mvdr_scratch_bytes(KS, N)
        scratch = torch.empty((sb,), dtype=torch.uint8, device=R.device) if sb else None
        sp = None if scratch is None else _ptr(scratch)
        if batched:
            check(_lib.lib().btk_mvdr_weights_streams(_ptr(R), _ptr(wq), _ptr(W), S, K, N, float(threshold), sp, _ptr(fb), _ptr(flags), _stream()))
        else:
            check(_lib.lib().btk_mvdr_weights_flags(_ptr(R), _ptr(wq), _ptr(W), K, N, int(first_bin), float(threshold), sp, _ptr(fb), _ptr(flags),
                                                    _stream()))
        c0 = c1 = 0
        if rule == "linpack":
            counts = _linpack_rule(R, wq, W, None, KS, N, first_bin, K if batched else 0, 1, threshold, flags)
            c0, c1 = (int(v) for v in counts.tolist())
        ni = C.c_int(0)
        if int(fb.item()) > 0 and (rule != "linpack" or int(flags.sum().item()) > 0):
            check(_lib.lib().btk_mvdr_pinv_fallback(_ptr(R), _ptr(wq), _ptr(W), KS, N, int(first_bin), float(threshold),
                                                    _ptr(flags), C.byref(ni), _stream()))
        nident = c0 + c1 + ni.value
        mvdr_weights.last_counts = (c0, c1, ni.value)
    return W, nident


mvdr_weights.last_counts = (0, 0, 0)


def mvdr_divide_nondiagonal(R, mu):
    """divide_all_nondiagonal_elements (beamformer.h:357-362): R_xy /= 1 + mu for x != y, in place; R complex64 [K][N][N]."""
    _check(R, "R", torch.complex64, 3)
    K, N, _ = R.shape
    if K > 0:
        check(_lib.lib().btk_mvdr_divide_nondiagonal(_ptr(R), K, N, float(mu), _stream()))
    return R


def pinv(A, threshold=1.0e-8):
    """pseudoinverse() (beamformer.cc:232-289) of one host matrix: (invA complex128 [N][M], ok) with ok == the reference's
    return value (False when a singular value fell below the threshold)."""
    A = np.ascontiguousarray(A, np.complex128)
    M, N = A.shape
    out = np.zeros((N, M), np.complex128)
    nz = C.c_int(0)
    check(_lib.lib().btk_pinv(_np_ptr(A), M, N, float(threshold), _np_ptr(out), C.byref(nz)))
    return out, nz.value == 0


# ---------------------------------------------------------------------------- WPE dereverberation
def wpe_band(M, band_width, samplerate):
    """set_band_width_ (dereverberation.cc:361-369): (lower_bandWidthN_, upper_bandWidthN_)."""
    if band_width == 0.0:
        lower = M // 2
    else:
        if band_width > samplerate / 2.0:
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "Bandwidth is greater than the Nyquist rate.")
        lower = int((band_width / (samplerate / 2.0)) * (M // 2))
    return lower, M - lower


def wpe_estimate(X, M, lower_num=0, upper_num=32, iterations_num=2, load_db=-18.0, band_width=0.0,
                 diagonal_bias=1.0e-4, samplerate=16000.0, G=None):
    """MultiChannelWPEDereverberation::estimate_filter.  X complex64 [S][K][C][T] -> G complex64 [S][C][K][C*L]."""
    _need_cuda(X, "X")
    S, K, Cn, T = X.shape
    L = upper_num - lower_num + 1
    if G is None:
        G = torch.zeros((S, Cn, K, Cn * L), dtype=torch.complex64, device=X.device)
    lo, up = wpe_band(M, band_width, samplerate)
    ws = torch.empty(_lib.lib().btk_wpe_workspace_bytes(S, K, Cn, lower_num, upper_num, T), dtype=torch.uint8, device=X.device)
    fail = torch.zeros(1, dtype=torch.int32, device=X.device)
    check(_lib.lib().btk_wpe_estimate(_ptr(X), S, K, Cn, T, T, lower_num, upper_num, iterations_num, float(load_db),
                                      float(diagonal_bias), lo, up, _ptr(G), _ptr(ws), _ptr(fail), _stream()))
    if int(fail.item()) > 0:
        raise _lib.BtkError(_lib.BTK_ERR_NUMERIC, "MultiChannelWPEDereverberation: Cholesky decomposition failed (%d systems).\n"
                            "Some channels may be too similar. Try to increase 'diagonal_bias'" % int(fail.item()))
    return G


def wpe_apply(X, G, M, lower_num=0, upper_num=32, band_width=0.0, samplerate=16000.0, out=None):
    """calc_every_channel_output for every frame and channel: X [S][K][C][T] -> dereverberated [S][K][C][T]."""
    _need_cuda(X, "X"); _need_cuda(G, "G")
    S, K, Cn, T = X.shape
    if out is None:
        out = torch.empty_like(X)
    lo, up = wpe_band(M, band_width, samplerate)
    check(_lib.lib().btk_wpe_apply(_ptr(X), _ptr(G), _ptr(out), S, K, Cn, T, T, lower_num, upper_num, lo, up, _stream()))
    return out
