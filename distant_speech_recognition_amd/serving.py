"""Host-to-host serving of the static-weight chain: utterances arrive in host memory as 16-bit PCM (what SampleFeature reads,
feature/feature.cc:265-269), leave as beamformed PCM, and the GPU chain in between (fused analysis + SubbandDS/GSC/MVDR apply,
synthesis) runs 50x faster than PCIe can feed it -- so the pipeline is built around the link:

  * samples cross PCIe as int16 (half the bytes of the floats the kernels compute on) from pinned buffers and STAY int16 in HBM:
    the fused kernel reads them as they are and widens them in registers (`btk_fb_analysis_bf_i16`: no widening pass, no float
    copy of the PCM, the same bits out); geometries without an int16 kernel and interleaved frames are widened on the device first
    (`btk_pcm_i16_to_f32` / `btk_pcm_i16_deinterleave`); the single-channel output comes back as float32 or int16
    (`btk_pcm_f32_to_i16`, the reference scripts' `numpy.array(buf, numpy.int16)`);
  * three HIP streams -- upload, compute, download -- and `depth` device buffer sets: batch b+1 uploads while batch b is
    transformed and batch b-1 downloads; events order the hand-overs, the host only waits for finished downloads;
  * utterance streams are independent (unit_test/test_online_beamforming.py:80-88 builds one graph per utterance), so a batch is
    any `streams_per_batch` of them; with several GPUs each rank runs its own pipeline over its share (sharding.streams_for_rank).

PyTorch supplies pinned memory, streams and events; every computation is a C-ABI call on the compute stream.

Sharing: the plans keep one weight-pair scratch per (device, stream), so `afb` / `sfb` may also be used from the caller's own stream
while run() is in flight.  run() returns HOST results only: device-side state a caller wants to read afterwards (the buffer sets)
must be ordered behind the pipeline's streams by the caller (`torch.cuda.current_stream().wait_stream(pipe.s_cmp)`)."""
import numpy as np
import torch

from . import _lib, engine
from ._lib import check


def _ptr(t):
    return t.data_ptr()


class BatchBeamformerPipeline:
    """Fixed-weight beamforming of batches of utterances held in host memory.

    afb, sfb   engine.FilterBank plans (analysis, synthesis) of the same geometry
    W          complex64 [K][N] (or [1][K][N]) effective weights on the device (engine.weights_gsc_effective)
    N          channels per utterance, L samples per channel and utterance (shorter utterances are zero-padded by the caller)
    streams_per_batch   utterances transformed by one launch of the chain
    depth      device buffer sets in flight (>= 2 overlaps the three stages)
    int16_out  narrow the output on the device and download int16 (the reference scripts' WAV samples)
    interleaved  host PCM is [S][L][N] int16, the frames of a multi-channel WAV as stored; de-interleaved on the device"""

    def __init__(self, afb, sfb, W, N, L, streams_per_batch=8, depth=3, int16_in=True, int16_out=False, interleaved=False, device=None):
        self.afb, self.sfb, self.N, self.L, self.B, self.depth = afb, sfb, int(N), int(L), int(streams_per_batch), int(depth)
        self.dev = W.device if device is None else device
        self.W = W if W.dim() == 3 else W.unsqueeze(0)
        self.int16_in, self.int16_out = bool(int16_in), bool(int16_out)
        self.interleaved = bool(interleaved)               # host PCM as stored in a multi-channel WAV: [S][L][N] int16
        if self.interleaved and not self.int16_in:
            raise _lib.BtkError(_lib.BTK_ERR_PARAMETER, "interleaved input is int16 (the frames of a multi-channel WAV)")
        # planar int16 input goes to the fused kernel as it is when the geometry has an int16 kernel
        self.direct_i16 = self.int16_in and not self.interleaved and afb.fused_i16() and self.L % 2 == 0
        self.T = afb.num_frames(self.L)
        self.nblk = sfb.num_blocks(self.T)
        self.out_len = self.nblk * sfb.D
        K = afb.K
        self.s_up, self.s_cmp, self.s_down = (torch.cuda.Stream(device=self.dev) for _ in range(3))
        self.sets = []
        for _ in range(self.depth):
            st = {
                "raw": torch.empty((self.B, self.L, self.N) if self.interleaved else (self.B, self.N, self.L),
                                   dtype=torch.int16 if self.int16_in else torch.float32, device=self.dev),
                "pcm": torch.empty((self.B, self.N, self.L), dtype=torch.float32, device=self.dev) if (self.int16_in and not self.direct_i16) else None,
                "Y": engine.padded_rows((self.B, K, self.T), torch.complex64, self.dev),
                "out": torch.empty((self.B, self.out_len), dtype=torch.float32, device=self.dev),
                "out16": torch.empty((self.B, self.out_len), dtype=torch.int16, device=self.dev) if self.int16_out else None,
                "host_out": torch.empty((self.B, self.out_len), dtype=torch.int16 if self.int16_out else torch.float32).pin_memory(),
                "uploaded": torch.cuda.Event(), "computed": torch.cuda.Event(), "downloaded": torch.cuda.Event(), "free": torch.cuda.Event(),
                "batch": None,
            }
            self.sets.append(st)

    # -- the three stages of one batch ---------------------------------------------------------------------------------
    def _upload(self, st, host_batch):
        with torch.cuda.stream(self.s_up):
            self.s_up.wait_event(st["free"])                       # the set's previous compute has consumed `raw`
            st["raw"][: host_batch.shape[0]].copy_(host_batch, non_blocking=True)
            st["uploaded"].record(self.s_up)

    def _compute(self, st, nb):
        with torch.cuda.stream(self.s_cmp):
            self.s_cmp.wait_event(st["uploaded"])
            self.s_cmp.wait_event(st["downloaded"])                # the set's previous output has left `out`
            if self.interleaved:
                for u in range(nb):                                # [L][N] frames -> planar float channels, all channels per launch
                    check(_lib.lib().btk_pcm_i16_deinterleave(_ptr(st["raw"][u]), _ptr(st["pcm"][u]), self.L, self.N, self.L, self.s_cmp.cuda_stream))
            elif self.int16_in and not self.direct_i16:
                check(_lib.lib().btk_pcm_i16_to_f32(_ptr(st["raw"]), _ptr(st["pcm"]), nb * self.N * self.L, self.s_cmp.cuda_stream))
            widened = self.int16_in and not self.direct_i16
            if widened:
                pcm = st["pcm"][:nb]
                st["free"].record(self.s_cmp)                      # `raw` may be overwritten by the next upload once widened ...
            else:
                pcm = st["raw"][:nb]                               # float32, or int16 read by the fused kernel itself
            Y = st["Y"][:nb]
            self.afb.analysis_beamform(pcm, self.W, out=Y)
            if not widened:
                st["free"].record(self.s_cmp)                      # ... or, without widening, once the analysis has read it
            self.sfb.synthesize(Y, out=st["out"][:nb])
            if self.int16_out:
                check(_lib.lib().btk_pcm_f32_to_i16(_ptr(st["out"]), _ptr(st["out16"]), nb * self.out_len, self.s_cmp.cuda_stream))
            st["computed"].record(self.s_cmp)

    def _download(self, st, nb):
        with torch.cuda.stream(self.s_down):
            self.s_down.wait_event(st["computed"])
            src = st["out16"] if self.int16_out else st["out"]
            st["host_out"][:nb].copy_(src[:nb], non_blocking=True)
            st["downloaded"].record(self.s_down)

    # -- driver ------------------------------------------------------------------------------------------------------------
    def run(self, host_pcm, out=None):
        """host_pcm: pinned int16 (or float32) tensor / array [S][N][L] -> beamformed PCM [S][out_len] (numpy, float32 or
        int16).  Batches of `streams_per_batch` utterances flow through the three streams."""
        if not torch.is_tensor(host_pcm):
            host_pcm = torch.from_numpy(np.ascontiguousarray(host_pcm))
        want = torch.int16 if self.int16_in else torch.float32
        shape = (self.L, self.N) if self.interleaved else (self.N, self.L)
        if host_pcm.dtype != want or tuple(host_pcm.shape[1:]) != shape:
            raise _lib.BtkError(_lib.BTK_ERR_DIMENSION, "host PCM must be %s [S][%d][%d], got %s %s"
                                % (want, shape[0], shape[1], host_pcm.dtype, tuple(host_pcm.shape)))
        if not host_pcm.is_pinned():
            host_pcm = host_pcm.pin_memory()                       # pageable memory would serialise the copies
        S = host_pcm.shape[0]
        if out is None:
            out = np.empty((S, self.out_len), np.int16 if self.int16_out else np.float32)
        nbatches = (S + self.B - 1) // self.B
        cur = torch.cuda.current_stream(self.dev)
        for s in (self.s_up, self.s_cmp, self.s_down):
            s.wait_stream(cur)
        for st in self.sets:
            st["batch"] = None
        for b in range(nbatches):
            st = self.sets[b % self.depth]
            if st["batch"] is not None:                            # the set is being reused: its output must be on the host first
                self._collect(st, out)
            lo = b * self.B
            nb = min(self.B, S - lo)
            self._upload(st, host_pcm[lo: lo + nb])
            self._compute(st, nb)
            self._download(st, nb)
            st["batch"] = (lo, nb)
        for st in self.sets:
            if st["batch"] is not None:
                self._collect(st, out)
        return out

    def _collect(self, st, out):
        st["downloaded"].synchronize()
        lo, nb = st["batch"]
        out[lo: lo + nb] = st["host_out"][:nb].numpy()
        st["batch"] = None
