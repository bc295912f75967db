"""HIP-graph replay of the generator forward pass.

One frame of ``TriPlaneGenerator.synthesis`` is ~280 kernel launches of fixed shapes; issued eagerly from Python they
cost ~4 ms of host time per frame.  ``GraphedSynthesis`` captures the whole call once (``torch.cuda.CUDAGraph`` = hipGraph on
ROCm; the C-ABI launches go to torch's capturing stream, the mouth-fill side stream forks/joins inside the capture) and
replays it per frame after copying the frame's inputs into static buffers.  Every kernel of the frame still runs on
every replay -- nothing is cached across frames.
"""
import torch


class GraphedSynthesis:
    def __init__(self, generator, batch=1, neural_rendering_resolution=128, warmup=3, with_ray_dist=False, **synthesis_kwargs):
        self.g = generator
        self.nrr = neural_rendering_resolution
        self.kwargs = dict(noise_mode='const', evaluation=True)
        self.kwargs.update(synthesis_kwargs)
        dev = next(generator.parameters()).device
        num_ws = generator.backbone.num_ws
        self.ws = torch.zeros(batch, num_ws, generator.w_dim, device=dev)
        self.c = torch.zeros(batch, 25, device=dev)
        self.uv = torch.zeros(batch, 256, 256, 3, device=dev)
        self.jitter = torch.zeros(batch, self.nrr * self.nrr, 48, device=dev)
        # batch-global mean |ray origin| (renderer.py:311) as an input: a rank that renders one shard of a larger batch gets the
        # value of the WHOLE batch from its caller (frame_parallel.global_ray_dist) instead of the mean over its own frames
        self.ray_dist = torch.zeros(1, device=dev) if with_ray_dist else None
        self.graph = None
        self.out = None
        self._warmup = warmup
        self._scratch = {}

    def _call(self):
        return self.g.synthesis(self.ws, self.c, {'uvcoords_image': self.uv}, neural_rendering_resolution=self.nrr,
                                jitter=self.jitter, ray_dist=self.ray_dist, **self.kwargs)

    def capture(self):
        # valid camera / inputs must be in the static buffers before capture (the warm-up runs execute real kernels)
        from . import hipops
        with hipops.scratch_owner(self._scratch):     # own stream-K scratch: graphs may replay concurrently; freed with self
            return self._capture()

    def _capture(self):
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(self._warmup):
                self._call()
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._call()
        return self

    @torch.no_grad()
    def __call__(self, ws, c, uvcoords_image, jitter, ray_dist=None):
        if (ray_dist is None) != (self.ray_dist is None):
            raise ValueError('ray_dist must be passed exactly when the graph was built with with_ray_dist=True')
        from . import hipops
        pairs = [(ws if ws.shape == self.ws.shape else ws.expand_as(self.ws), self.ws), (c[:, -25:], self.c),
                 (uvcoords_image, self.uv), (jitter.reshape(self.jitter.shape), self.jitter)]
        if ray_dist is not None:
            pairs.append((ray_dist.reshape(1), self.ray_dist))
        hipops.stage_inputs(pairs)          # one launch for the frame's inputs (strided / broadcast sources: Tensor.copy_)
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out


class GraphedDrive:
    """hipGraph of the drive loop's call (eval_seq.py:212): ``synthesis_withTexture`` on FIXED identity features (ws, texture and
    static feature pyramids of the inversion result, expanded to `batch`), replayed per block of drive frames after staging that
    block's cameras / UV maps / jitter (and `ray_dist`: [1] batch-global or [batch] per frame, see frame_parallel)."""

    def __init__(self, generator, ws, texture_feats, static_feats, batch=1, neural_rendering_resolution=128, warmup=3, ray_dist_elems=0,
                 **synthesis_kwargs):
        self.g = generator
        self.nrr = neural_rendering_resolution
        self.kwargs = dict(noise_mode='const', evaluation=True)
        self.kwargs.update(synthesis_kwargs)
        dev = ws.device
        self.ws = ws.expand(batch, -1, -1).contiguous()
        self.tex = [t.expand(batch, -1, -1, -1).contiguous() for t in texture_feats]
        self.sta = [t.expand(batch, -1, -1, -1).contiguous() for t in static_feats]
        self.c = torch.zeros(batch, 25, device=dev)
        self.uv = torch.zeros(batch, 256, 256, 3, device=dev)
        self.jitter = torch.zeros(batch, self.nrr * self.nrr, 48, device=dev)
        assert ray_dist_elems in (0, 1, batch), ray_dist_elems
        self.ray_dist = torch.zeros(ray_dist_elems, device=dev) if ray_dist_elems else None
        self.graph = None
        self.out = None
        self._warmup = warmup
        self._scratch = {}

    def _call(self):
        return self.g.synthesis_withTexture(self.ws, self.tex, self.c, {'uvcoords_image': self.uv}, static_feats=self.sta,
                                            neural_rendering_resolution=self.nrr, jitter=self.jitter, ray_dist=self.ray_dist, **self.kwargs)

    capture = GraphedSynthesis.capture
    _capture = GraphedSynthesis._capture

    @torch.no_grad()
    def set_identity(self, ws, texture_feats, static_feats):
        """Another identity of the same network: its features go into the tensors the call was captured with (one copy per pyramid level,
        broadcast over the call's frames)."""
        self.ws.copy_(ws.expand_as(self.ws))
        for dst, src in zip(self.tex + self.sta, list(texture_feats) + list(static_feats)):
            dst.copy_(src.expand_as(dst))
        return self

    @torch.no_grad()
    def __call__(self, c, uvcoords_image, jitter, ray_dist=None):
        if (ray_dist is None) != (self.ray_dist is None):
            raise ValueError('ray_dist must be passed exactly when the graph was built with ray_dist_elems > 0')
        from . import hipops
        pairs = [(c[:, -25:], self.c), (uvcoords_image, self.uv), (jitter.reshape(self.jitter.shape), self.jitter)]
        if ray_dist is not None:
            pairs.append((ray_dist.reshape(self.ray_dist.shape), self.ray_dist))
        hipops.stage_inputs(pairs)
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out


class FramePipeline:
    """`depth` captured frames in flight: frame k is replayed on stream k % depth with its own static buffers, so the
    latency-bound parts of consecutive frames (low-resolution layers, renderer sampling phases) fill each other's idle
    compute units.  `submit` returns (out, event, previous): `out` is the slot's STATIC output dict -- valid once `event` has
    completed and until the slot is replayed again (`depth` submissions later), so consume or copy it before then; `previous`
    is the event of the frame that occupied the slot before.  `drain` makes the caller's stream wait for everything in flight."""

    def __init__(self, generator, depth=2, batch=1, neural_rendering_resolution=128, **synthesis_kwargs):
        self.slots = [GraphedSynthesis(generator, batch, neural_rendering_resolution, **synthesis_kwargs) for _ in range(depth)]
        dev = next(generator.parameters()).device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.pending = [None] * depth
        self.k = 0

    def capture(self, ws, c, uv, jitter):
        """Capture every slot (each capture runs real frames, so valid inputs are required)."""
        for slot in self.slots:
            slot(ws, c, uv, jitter)
        torch.cuda.synchronize()
        return self

    @torch.no_grad()
    def submit(self, ws, c, uvcoords_image, jitter):
        i = self.k % len(self.slots)
        self.k += 1
        slot, stream = self.slots[i], self.streams[i]
        main = torch.cuda.current_stream()
        done_prev = self.pending[i]
        stream.wait_stream(main)                       # inputs produced on the caller's stream are ready
        with torch.cuda.stream(stream):
            from . import hipops
            hipops.stage_inputs([(ws if ws.shape == slot.ws.shape else ws.expand_as(slot.ws), slot.ws), (c[:, -25:], slot.c),
                                 (uvcoords_image, slot.uv), (jitter.reshape(slot.jitter.shape), slot.jitter)])
            slot.graph.replay()
            ev = torch.cuda.Event()
            ev.record(stream)
        self.pending[i] = ev
        return slot.out, ev, done_prev

    def drain(self):
        main = torch.cuda.current_stream()
        for ev in self.pending:
            if ev is not None:
                main.wait_event(ev)
