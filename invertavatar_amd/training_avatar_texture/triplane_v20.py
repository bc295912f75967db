"""Next3D++ (v20) animatable tri-plane generator on the MI355X backend.

API mirror of the reference's training_avatar_texture/triplane_v20.py: ``TriPlaneGenerator`` with
``mapping`` (:64), ``synthesis`` (:89), ``synthesis_withTexture`` (:152), ``synthesis_withCondition`` (:246),
``rasterize`` (:317), ``sample`` / ``sample_mixed`` (:341,:373), ``forward`` (:404) and ``OSGDecoder`` (:415),
with the reference's parameter names, argument meaning and return-dict keys.

Pipeline of one frame:
  texture backbone + static backbone (StyleGAN2, fused MFMA convolutions)
  -> rasterize: UV lookup of the neural texture, AA resize, alpha blend over the static features, mouth fill
     on the GPU (``ia_fill_mouth``; the reference round-trips through cv2 on the host)
  -> face backbone conditioned on those maps -> 128^2 stitch pasted into plane 0 -> blended tri-planes
  -> fused importance renderer (``ia_render_rays``: one launch) -> 32-channel 128^2 feature image
  -> super-resolution head -> 512^2 RGB.

Differences from the reference that do not change results: the three copies of the "rasterize / face
backbone / blend / render / SR" tail (:116-150, :179-244, :277-315) are one method here; ``rasterize`` can
skip the two levels the face backbone never reads (networks_stylegan2_new.py:536-540 with end_layer 6);
the stratified-sampling noise can be injected (``jitter=``) for reproducible evaluation.
"""
import torch
import torch.nn.functional as F

from .. import _runtime, dnnlib, hipops
from ..torch_utils import persistence
from ..training.networks_stylegan2 import FullyConnectedLayer
from .networks_stylegan2_new import Generator as StyleGAN2Backbone_cond
from .volumetric_rendering.renderer import ImportanceRenderer, ImportanceRenderer_bsMotion, fill_mouth  # noqa: F401
from .volumetric_rendering.ray_sampler import RaySampler, RaySampler_zxc  # noqa: F401

BBOX_256 = [57, 185, 64, 192]   # face region of the frontal plane, in 256^2 pixels (triplane_v20.py:114)
N_COND_LEVELS_USED = 4          # cond_list entries the face backbone consumes
CL_COPIES_ON_TEXTURE_STREAM = True   # False: the rasteriser's stream makes them (ia_rasterize_level's own copy)
LAUNCH_ORDER = 'tmfs'           # capture order of the frame's four independent branches: m mouth fill + rays, f face-backbone head,
                                # t texture backbone, s static backbone (profiles/r04_ab_launch_order_sweep.txt: two clusters 9 % apart)

FRAME_STYLE_BATCH = True        # synthesis(): the styles of all three backbones + the SR head in one launch pair at the top of the frame

RENDER_WRITES_SR_INPUT = True   # the fused renderer also writes its features in the SR head's operand format (no ia_act_split between them)
SINGLE_STREAM = False           # True: no side streams (every launch of a frame in program order on the caller's stream); used by
                                # bench.py to time kernels without neighbours from other streams


class _FrameState:
    """Per-generator orchestration state of the device path (side streams, tensors handed between the stages of ONE frame).
    It lives outside the module (invertavatar_amd._runtime) so that copy.deepcopy / pickle / torch.save of a generator that has
    already rendered see parameters and buffers only."""

    def __init__(self):
        self.streams = {}
        self.side_rays = None
        self.tex_cl = None

    def stream(self, name, device):
        st = self.streams.get(name)
        if st is None or st.device != device:
            st = self.streams[name] = torch.cuda.Stream(device=device)
        return st


def _state(gen):
    holder = _runtime.state(gen)
    if not hasattr(holder, 'frame'):
        holder.frame = _FrameState()
    return holder.frame


class _LazyLevels:
    """cond_list whose entry k is produced (rasterised) the first time the face backbone reads it."""

    def __init__(self, n, make):
        self._n, self._make, self._have = n, make, {}

    def __len__(self):
        return self._n

    def __getitem__(self, k):
        if k not in self._have:
            self._have[k] = self._make(k)
        return self._have[k]



def _aa_resize(x, size):
    return F.interpolate(x, size=(size, size), mode='bilinear', antialias=True)


@persistence.persistent_class
class TriPlaneGenerator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, topology_path=None, sr_num_fp16_res=0,
                 mapping_kwargs={}, rendering_kwargs={}, sr_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.renderer = ImportanceRenderer_bsMotion()
        self.ray_sampler = RaySampler_zxc()
        common = dict(img_resolution=256, mapping_kwargs=mapping_kwargs, **synthesis_kwargs)
        self.texture_backbone = StyleGAN2Backbone_cond(z_dim, c_dim, w_dim, img_channels=32, **common)
        self.face_backbone = StyleGAN2Backbone_cond(z_dim, c_dim, w_dim, img_channels=32, **common)
        self.backbone = StyleGAN2Backbone_cond(z_dim, c_dim, w_dim, img_channels=32 * 3, mapping_ws=self.texture_backbone.num_ws,
                                               **common)
        self.superresolution = dnnlib.util.construct_class_by_name(
            class_name=rendering_kwargs['superresolution_module'], channels=32, img_resolution=img_resolution,
            sr_num_fp16_res=sr_num_fp16_res, sr_antialias=rendering_kwargs['sr_antialias'], **sr_kwargs)
        self.decoder = OSGDecoder(32, {'decoder_lr_mul': rendering_kwargs.get('decoder_lr_mul', 1), 'decoder_output_dim': 32})
        self.neural_rendering_resolution = 128
        self.rendering_kwargs = rendering_kwargs
        self.fill_mouth = True

    # ------------------------------------------------------------------ latent mapping
    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs['c_gen_conditioning_zero']:
            c = torch.zeros_like(c)
        c = c[:, :self.c_dim]   # drop expression labels
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0), truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    # ------------------------------------------------------------------ helpers
    def _rays(self, c, neural_rendering_resolution):
        cam = c[:, -25:]
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        origins, dirs = self.ray_sampler(cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3), neural_rendering_resolution)
        return origins, dirs, neural_rendering_resolution

    def _two_backbones(self, ws, update_emas, synthesis_kwargs, partial=False, order='ts', side_work=None):
        """The texture and the static backbone are independent: on the device they run on two streams of their own so that their
        latency-bound low-resolution layers (a handful of workgroups each at batch 1) overlap.

        partial=True (synthesis): the rasteriser consumes the taps level by level (32^2, 64^2, 128^2) and the face backbone
        consumes the rasterised levels block by block, so nothing waits for whole networks: both backbones record an event
        after every tap (`pending[2]`, `pending[3]`: tap count -> event) and _planes rasterises level k when the face
        backbone asks for it.  The 256^2 blocks are not consumed by the rasteriser at all; the caller joins the streams
        (`pending[0]` texture, `pending[1]` static) where the last taps are consumed (_planes: static, before the plane
        blend; synthesis: texture, at the end of the frame)."""
        def tex(**kw):
            return self.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=update_emas, **kw, **synthesis_kwargs)

        def sta(**kw):
            return self.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=update_emas, **kw, **synthesis_kwargs)
        if not (ws.is_cuda and not torch.is_grad_enabled()) or SINGLE_STREAM:
            for work in (side_work or {}).values():
                work()
            return tex(), sta(), None
        st = _state(self)
        main, t_stream, s_stream = torch.cuda.current_stream(ws.device), st.stream('texture', ws.device), st.stream('static', ws.device)
        tex_cl = [None] * N_COND_LEVELS_USED
        counts = tuple(range(2, N_COND_LEVELS_USED + 1))       # taps 0 and 1 (image and features at 32^2) appear together
        ev_tex, ev_sta = {n: torch.cuda.Event() for n in counts}, {n: torch.cuda.Event() for n in counts}

        def make_cl(feats):   # the rasteriser gathers from channels-last copies; make them on the texture stream
            for k, t in enumerate(feats[:N_COND_LEVELS_USED]):
                if tex_cl[k] is None and t.dtype == torch.float32:
                    tex_cl[k] = hipops.channels_last_copy(t)

        def tap_tex(feats):
            if CL_COPIES_ON_TEXTURE_STREAM:
                make_cl(feats)
            ev_tex[len(feats)].record(t_stream)
        if not partial:
            t_stream.wait_stream(main)
            with torch.cuda.stream(t_stream):
                texture_feats = tex()
                make_cl(texture_feats)
            static_feats = sta()
            main.wait_stream(t_stream)
            pending = None
        else:
            # a graph replay queues its nodes branch by branch in capture order, a few microseconds each, so the order the
            # branches are captured in decides when each starts: `order` names it (t texture, s static, others: side_work)
            made = {}
            for ch in order:
                if ch == 't':
                    t_stream.wait_stream(main)
                    with torch.cuda.stream(t_stream):
                        made['t'] = tex(_tap=(counts, tap_tex))
                elif ch == 's':
                    s_stream.wait_stream(main)
                    with torch.cuda.stream(s_stream):
                        made['s'] = sta(_tap=(counts, lambda feats: ev_sta[len(feats)].record(s_stream)))
                else:
                    side_work[ch]()
            texture_feats, static_feats = made['t'], made['s']
            pending = (t_stream, s_stream, ev_tex, ev_sta)
        for t in list(texture_feats) + [t for t in tex_cl if t is not None] + (list(static_feats) if partial else []):
            t.record_stream(main)
        st.tex_cl = (texture_feats, tex_cl)
        return texture_feats, static_feats, pending

    def _frame_styles(self, ws):
        """Every affine style vector and demodulation coefficient of the frame -- three backbones and the SR head, ~190 layers -- in
        ONE `ia_styles_demod` launch pair on the caller's stream before the branches fork (they were 2 launches at the head of each
        network's chain: 8 per frame, ~25 us in front of each backbone's first convolution).  Marks the networks so that their own
        batchers skip this `ws`; returns True when the SR head's styles were part of the batch."""
        if not (FRAME_STYLE_BATCH and ws.is_cuda and not torch.is_grad_enabled()):
            return False
        from ..training.networks_stylegan2 import _StyleBatcher
        st = _state(self)
        nets = [self.texture_backbone.synthesis, self.backbone.synthesis, self.face_backbone.synthesis]
        if any(n.num_ws > ws.shape[1] for n in nets):
            return False
        if getattr(st, 'style_batcher', None) is None:
            st.style_batcher = _StyleBatcher()
        blocks, offs, fixed = [], [], []
        for n in nets:
            b, o = n._style_blocks()
            blocks += b; offs += o; fixed += [None] * len(b)
        sr = self.superresolution.style_blocks() if hasattr(self.superresolution, 'style_blocks') else None
        if sr is not None:
            blocks += sr; offs += [0] * len(sr); fixed += [ws.shape[1] - 1] * len(sr)
        if not st.style_batcher.prepare(blocks, offs, ws.to(torch.float32), fixed_w=fixed):
            return False
        for n in nets:
            _runtime.state(n).styles_for = ws
        return sr is not None

    def _start_face_head(self, ws, update_emas, synthesis_kwargs, sr_styles_ready=False):
        """The 4^2..32^2 blocks of the face backbone depend only on ws; run them on a third stream under the other backbones."""
        if not (ws.is_cuda and not torch.is_grad_enabled()) or SINGLE_STREAM:
            return None
        main, side = torch.cuda.current_stream(ws.device), _state(self).stream('face_head', ws.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            x, img, first = self.face_backbone.synthesis.forward_head(ws, update_emas=update_emas, **synthesis_kwargs)
            if hasattr(self.superresolution, 'hoist_styles'):      # three small launches off the render -> SR chain
                self.superresolution.hoist_styles(ws, prepared=sr_styles_ready)
            done = torch.cuda.Event()
            done.record(side)
        x.record_stream(main)
        img.record_stream(main)
        return x, img, first, done

    def _start_mouth_fill(self, mesh_condition, rays=None):
        """The mouth-hole fill depends only on the UV mask, and its flood is a one-workgroup, latency-bound kernel:
        start it on a side stream at the top of the frame so that it runs underneath the backbone convolutions.
        `rays` = (c, neural_rendering_resolution, ray_dist): the camera rays and the batch mean of |ray origin| depend
        only on the cameras, so they are produced on the same side stream (_FrameState.side_rays)."""
        uv = mesh_condition['uvcoords_image']
        st = _state(self)
        st.side_rays = None
        if not (uv.is_cuda and not torch.is_grad_enabled()) or SINGLE_STREAM:
            return None
        main, side = torch.cuda.current_stream(uv.device), st.stream('mouth', uv.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            uv_c = uv.float().contiguous()
            alpha = uv_c[..., 2:].permute(0, 3, 1, 2).contiguous()
            full_alpha, mouth = fill_mouth(alpha, blur_mouth_edge=False)
            upper_c = self._upper_alpha(alpha, mouth).reshape(-1, uv.shape[1], uv.shape[2]).contiguous()
            extra = ()
            if rays is not None:
                c, nrr, ray_dist = rays
                origins, dirs, nrr = self._rays(c, nrr)
                if ray_dist is None:
                    ray_dist = torch.norm(origins, dim=-1).mean().reshape(1)     # (renderer.py:311), no host sync
                st.side_rays = (origins, dirs, nrr, ray_dist)
                extra = (origins, dirs, ray_dist)
            done = torch.cuda.Event()
            done.record(side)
        for t in (uv_c, alpha, full_alpha, mouth, upper_c) + tuple(t for t in extra if torch.is_tensor(t)):
            t.record_stream(main)
        return alpha, full_alpha, mouth, done, uv_c, upper_c

    @staticmethod
    def _upper_alpha(alpha, mouth):
        """Face alpha with the upper part of the mouth hole closed (reference :324-326)."""
        upper = mouth.clone()
        upper[:, :, :87] = 0
        return torch.clamp(alpha + upper, min=0, max=1)

    @staticmethod
    def _split_static(static_feats):
        """The static backbone emits 96 = 3 x 32 channels; levels 0 and 5 carry all three planes and the
        rasteriser only blends over plane 0 of them (:109-112, :180-181)."""
        b = static_feats[0].shape[0]
        plane = static_feats[-1].view(b, 3, 32, *static_feats[-1].shape[-2:])
        first = static_feats[0].view(b, 3, 32, *static_feats[0].shape[-2:])[:, 0]
        return [first] + list(static_feats[1:-1]) + [plane[:, 0]], plane

    def _blend_planes(self, stitch, full_alpha, static_plane):
        """Paste the 128^2-resized face stitch + alpha into the bbox of plane 0; planes 1-2 stay static (:119-128)."""
        if (stitch.is_cuda and stitch.dtype == torch.float32 and stitch.shape[1:] == (32, 256, 256) and not torch.is_grad_enabled()
                and static_plane.shape[1:] == (3, 32, 256, 256)):
            # fused AA-resize + paste + blend, emitted channels-last; returned as the NCHW-shaped view of that buffer
            b = stitch.shape[0]
            sta = static_plane.reshape(b, 96, 256, 256)
            sta = sta if sta.is_contiguous() else sta.contiguous()
            cl = hipops.blend_planes(stitch.contiguous(), full_alpha.reshape(b, 256, 256).contiguous(), sta, BBOX_256)
            return cl.permute(0, 1, 4, 2, 3)
        y0, y1, x0, x1 = BBOX_256
        canvas = torch.zeros_like(stitch)
        alpha = torch.zeros_like(full_alpha)
        canvas[:, :, y0:y1, x0:x1] = _aa_resize(stitch, 128)
        alpha[:, :, y0:y1, x0:x1] = _aa_resize(full_alpha, 128)
        planes = static_plane.clone()
        planes[:, 0] = canvas * alpha + static_plane[:, 0] * (1 - alpha)
        return planes

    def _planes(self, ws, texture_feats, static_feats, mesh_condition, update_emas, synthesis_kwargs, all_levels=False, mouth=None,
                face_head=None, pending=None):
        static_for_raster, static_plane = self._split_static(static_feats)
        assert len(static_for_raster) == len(texture_feats), (len(static_for_raster), len(texture_feats))
        if pending is not None and not all_levels:
            # The rasteriser runs on a stream of its own: level k is launched there as soon as the events of the taps it reads exist
            # (taps 0 and 1 are both 32^2 and come with the first event), under the face backbone's previous block; the face
            # backbone only waits for level k's event when it reads cond_list[k].
            main = torch.cuda.current_stream(ws.device)
            rs = _state(self).stream('raster', ws.device)
            rs.wait_stream(main)
            levels, done = [], []
            with torch.cuda.stream(rs):
                prep = self._raster_prep(mesh_condition['uvcoords_image'], mouth)      # (joins the mouth fill on this stream)
                for k in range(N_COND_LEVELS_USED):
                    need = max(2, k + 1)
                    rs.wait_event(pending[2][need])
                    rs.wait_event(pending[3][need])
                    levels.append(self._raster_level(k, texture_feats[k], static_for_raster[k], BBOX_256, prep))
                    ev = torch.cuda.Event()
                    ev.record(rs)
                    done.append(ev)
            full_alpha = prep['full_alpha']
            for t in levels:
                t.record_stream(main)

            def level(k):
                main.wait_event(done[k])
                return levels[k]
            cond = _LazyLevels(N_COND_LEVELS_USED, level)
        else:
            if pending is not None:
                for st in pending[:2]:
                    torch.cuda.current_stream(ws.device).wait_stream(st)
            cond, full_alpha, _ = self.rasterize(texture_feats, mesh_condition['uvcoords_image'], static_for_raster, BBOX_256,
                                                 levels=None if all_levels else N_COND_LEVELS_USED, _mouth=mouth)
        head = None
        if face_head is not None:
            torch.cuda.current_stream(ws.device).wait_event(face_head[3])
            head = face_head[:3]
        stitch = self.face_backbone.synthesis(ws, cond, return_list=False, update_emas=update_emas, _head=head, **synthesis_kwargs)
        if pending is not None:      # the static backbone's 256^2 taps (the planes) are consumed from here on
            torch.cuda.current_stream(ws.device).wait_stream(pending[1])
        return self._blend_planes(stitch, full_alpha, static_plane)

    def _sr_input_consumer(self, ws, nrr):
        """(layer, styles [B,32], planes) of the convolution that reads the rendered features in split format -- block0.conv0 of a two-block
        head fed at its own input resolution, with its styles parked by the frame-level style batch -- or (None, None, 0)."""
        from ..training import networks_stylegan2 as sg2
        sr = self.superresolution
        block0 = getattr(sr, 'block0', None)
        conv0 = getattr(block0, 'conv0', None)
        if (not isinstance(block0, sg2.SynthesisBlock) or conv0 is None or conv0._pre is None or getattr(sr, 'input_resolution', None) != nrr
                or conv0.in_channels != 32 or block0.architecture == 'resnet' or self.rendering_kwargs.get('superresolution_noise_mode') == 'random'):
            return None, None, 0
        half = block0._half_ops(ws.device)
        if not conv0._takes_split_input(nrr, self.rendering_kwargs.get('superresolution_noise_mode', 'const'), half):
            return None, None, 0
        styles = conv0._pre[0]
        if styles.shape != (ws.shape[0], 32) or styles.dtype != torch.float32 or not styles.is_contiguous():
            return None, None, 0
        return conv0, styles, (1 if half else 2)

    def _render(self, ws, planes, origins, dirs, nrr, evaluation, jitter, synthesis_kwargs, ray_dist=None, u_importance=None):
        if evaluation:
            assert synthesis_kwargs.get('noise_mode') == 'const', ('noise_mode' in synthesis_kwargs, synthesis_kwargs.get('noise_mode'))
        # The head's first convolution reads the features as fp16 planes multiplied by its styles (hipops.SplitAct): when those styles
        # exist already (computed at the top of the frame) the renderer writes that copy itself -- no ia_act_split launch between the
        # renderer and the head (r06; the bits of ia_act_split on the stored image)
        split_kw, consumer = {}, None
        if RENDER_WRITES_SR_INPUT and planes.is_cuda and not torch.is_grad_enabled():
            consumer, styles, n_planes = self._sr_input_consumer(ws, nrr)
            if consumer is not None:
                split_kw = dict(split_styles=styles, split_planes=n_planes)
        feats, depth, _ = self.renderer(planes, self.decoder, origins, dirs, self.rendering_kwargs, evaluation=evaluation, jitter=jitter,
                                        dist=ray_dist, **({} if u_importance is None else {'u_importance': u_importance}), **split_kw)
        n = ws.shape[0]
        feature_image = feats.permute(0, 2, 1).reshape(n, feats.shape[-1], nrr, nrr).contiguous()
        split_data = getattr(feats, 'split_data', None)
        if consumer is not None and split_data is not None:
            feature_image._ia_split = hipops.SplitAct(split_data.reshape(n, split_data.shape[1], 4, nrr, nrr, 8), feats.shape[-1], consumer)
        depth_image = depth.permute(0, 2, 1).reshape(n, 1, nrr, nrr)
        rgb = feature_image[:, :3]
        sr_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'noise_mode'}
        image = self.superresolution(rgb, feature_image, ws, noise_mode=self.rendering_kwargs['superresolution_noise_mode'], **sr_kwargs)
        return image, rgb, depth_image, feature_image

    # ------------------------------------------------------------------ public synthesis entry points
    def synthesis(self, ws, c, mesh_condition, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                  use_cached_backbone=False, return_featmap=False, evaluation=False, jitter=None, ray_dist=None, **synthesis_kwargs):
        side = {}
        sr_ready = self._frame_styles(ws)
        side_work = {'m': lambda: side.__setitem__('m', self._start_mouth_fill(mesh_condition, rays=(c, neural_rendering_resolution, ray_dist))),
                     'f': lambda: side.__setitem__('f', self._start_face_head(ws, update_emas, synthesis_kwargs, sr_ready))}
        texture_feats, static_feats, pending = self._two_backbones(ws, update_emas, synthesis_kwargs, partial=True,
                                                                   order=LAUNCH_ORDER, side_work=side_work)
        mouth, face_head = side['m'], side['f']
        side_rays = _state(self).side_rays
        if side_rays is not None:            # made on the side stream; joined with the mouth fill inside rasterize()
            origins, dirs, nrr, ray_dist = side_rays
        else:
            origins, dirs, nrr = self._rays(c, neural_rendering_resolution)
        planes = self._planes(ws, texture_feats, static_feats, mesh_condition, update_emas, synthesis_kwargs, mouth=mouth,
                              face_head=face_head, pending=pending)
        image, rgb, depth, feature_image = self._render(ws, planes, origins, dirs, nrr, evaluation, jitter, synthesis_kwargs, ray_dist)
        if pending is not None:      # the texture backbone's 256^2 block (not consumed by this frame's image) ends with the frame
            torch.cuda.current_stream(ws.device).wait_stream(pending[0])
        out = {'image': image, 'image_raw': rgb, 'image_depth': depth}
        if return_featmap:
            out.update(feature_image=feature_image, triplane=planes, texture=texture_feats)
        return out

    def synthesis_withTexture(self, ws, texture_feats, c, mesh_condition, static_feats=None, neural_rendering_resolution=None,
                              update_emas=False, cache_backbone=False, use_cached_backbone=False, evaluation=False, jitter=None,
                              ray_dist=None, u_importance=None, **synthesis_kwargs):
        # same orchestration as synthesis(): mouth fill + rays on the side stream, face-backbone head on its own stream
        # (`jitter`, `u_importance`: the renderer's two random draws handed in, see ImportanceRenderer_bsMotion.forward)
        # (`ray_dist`: see ImportanceRenderer_bsMotion.forward -- 1 value for the batch or one per frame)
        mouth = self._start_mouth_fill(mesh_condition, rays=(c, neural_rendering_resolution, ray_dist))
        face_head = self._start_face_head(ws, update_emas, synthesis_kwargs)
        side_rays = _state(self).side_rays
        if side_rays is not None:
            origins, dirs, nrr, ray_dist = side_rays
        else:
            origins, dirs, nrr = self._rays(c, neural_rendering_resolution)
        if static_feats is None:
            static_feats = self.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=update_emas, **synthesis_kwargs)
        planes = self._planes(ws, texture_feats, static_feats, mesh_condition, update_emas, synthesis_kwargs, mouth=mouth,
                              face_head=face_head)
        image, rgb, depth, feature_image = self._render(ws, planes, origins, dirs, nrr, evaluation, jitter, synthesis_kwargs, ray_dist, u_importance)
        return {'image': image, 'image_raw': rgb, 'image_depth': depth, 'feature_image': feature_image, 'triplane': planes}

    def synthesis_withCondition(self, ws, c, mesh_condition, gt_texture_feats=None, gt_static_feats=None, texture_feats_conditions=None,
                                static_feats_conditions=None, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                                use_cached_backbone=False, only_image=False, return_feats=False, jitter=None, **synthesis_kwargs):
        origins, dirs, nrr = self._rays(c, neural_rendering_resolution)
        texture_feats = gt_texture_feats
        if texture_feats is None:
            texture_feats = self.texture_backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=texture_feats_conditions,
                                                            update_emas=update_emas, **synthesis_kwargs)
        static_feats = gt_static_feats
        if static_feats is None:
            static_feats = self.backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=static_feats_conditions,
                                                   update_emas=update_emas, **synthesis_kwargs)
        planes = self._planes(ws, texture_feats, static_feats, mesh_condition, update_emas, synthesis_kwargs)
        evaluation = synthesis_kwargs.get('noise_mode') == 'const'
        image, rgb, depth, feature_image = self._render(ws, planes, origins, dirs, nrr, evaluation, jitter, synthesis_kwargs)
        if only_image:
            return {'image': image}
        out = {'image': image, 'image_raw': rgb, 'image_depth': depth, 'feature_image': feature_image, 'triplane': planes}
        if return_feats:
            out['static'] = static_feats
            out['texture'] = texture_feats
        return out

    def rasterize(self, texture_feats, uvcoords_image, static_feats, bbox_256, levels=None, _mouth=None):
        """UV-rasterise the neural texture pyramid and blend it over the static features.

        uvcoords_image [B,H,W,3] = (u, v, mask).  Returns (list of [B, C_k+1, res_k, res_k], full_alpha, mouth_masks);
        the extra channel is the face alpha with the upper part of the mouth hole closed (:324-326).
        `levels` limits how many pyramid levels are produced (None = all, as the reference)."""
        prep = self._raster_prep(uvcoords_image, _mouth)
        n = len(texture_feats) if levels is None else min(levels, len(texture_feats))
        out = [self._raster_level(k, tex, sta, bbox_256, prep) for k, (tex, sta) in enumerate(zip(texture_feats[:n], static_feats[:n]))]
        return out, prep['full_alpha'], prep['mouth']

    def _raster_prep(self, uvcoords_image, _mouth=None):
        """Per-frame inputs of the rasteriser: UV grid, face alpha, filled mouth (joined here if it was started on the side stream)."""
        uv = uvcoords_image if uvcoords_image.dtype == torch.float32 else uvcoords_image.float()
        grid, alpha = uv[..., :2], uv[..., 2:].permute(0, 3, 1, 2)
        fused = uv.is_cuda and uv.shape[1:3] == (256, 256) and not torch.is_grad_enabled()
        uv_c = upper_c = None
        if _mouth is not None:   # started on the side stream by synthesis(): join it here
            alpha, full_alpha, mouth, done, uv_c, upper_c = _mouth
            torch.cuda.current_stream(uv.device).wait_event(done)
            upper_alpha = upper_c.unsqueeze(1)
        else:
            full_alpha, mouth = fill_mouth(alpha.clone(), blur_mouth_edge=False)
            upper_alpha = self._upper_alpha(alpha, mouth)
            if fused:
                uv_c, upper_c = uv.contiguous(), upper_alpha.reshape(-1, 256, 256).contiguous()
        # channels-last copies made on the texture stream by _two_backbones (same list object => same frame)
        st = _state(self)
        cached, st.tex_cl = st.tex_cl, None
        return dict(grid=grid, alpha=alpha, full_alpha=full_alpha, mouth=mouth, upper_alpha=upper_alpha, uv_c=uv_c, upper_c=upper_c,
                    fused=fused, cached=cached)

    def _raster_level(self, k, tex, sta, bbox_256, prep):
        """One pyramid level: texture sampled through the UV map, AA-resized, blended over the static features' bbox crop."""
        res = tex.shape[2]
        y0, y1, x0, x1 = [round(v * res / 256) for v in bbox_256]
        if prep['fused'] and res in (32, 64, 128) and tex.dtype == torch.float32 and sta.dtype == torch.float32:
            cached = prep['cached']
            cl = cached[1][k] if (cached is not None and k < len(cached[0]) and cached[0][k] is tex and k < len(cached[1])) else None
            return hipops.rasterize_level(tex, prep['uv_c'], prep['upper_c'], sta, (y0, y1, x0, x1), res, tex_cl=cl)
        rend = _aa_resize(F.grid_sample(tex, prep['grid'], align_corners=False), res)
        a = _aa_resize(prep['alpha'], res)
        s = _aa_resize(sta[:, :, y0:y1, x0:x1], res)
        return torch.cat([rend * a + s * (1 - a), _aa_resize(prep['upper_alpha'], res)], dim=1)

    def visualize_mesh_condition(self, mesh_condition, to_imgs=False):
        uv = mesh_condition['uvcoords_image'].clone().permute(0, 3, 1, 2)
        full_alpha, _ = fill_mouth(uv[:, 2:].clone(), blur_mouth_edge=False)
        if not to_imgs:
            return uv
        uv[full_alpha.expand(-1, 3, -1, -1) == 0] = -1
        return [img for img in ((uv + 1) * 127.5).to(dtype=torch.uint8).cpu()]   # uint8 CHW tensors (no torchvision here)

    # ------------------------------------------------------------------ point queries (shape extraction)
    def _query(self, ws, coordinates, directions, mesh_condition, update_emas, synthesis_kwargs):
        texture_feats = self.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=update_emas, **synthesis_kwargs)
        static_feats = self.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=update_emas, **synthesis_kwargs)
        planes = self._planes(ws, texture_feats, static_feats, mesh_condition, update_emas, synthesis_kwargs)
        self.renderer.plane_axes = self.renderer.plane_axes.to(planes.device)
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def sample(self, coordinates, directions, z, c, mesh_condition, truncation_psi=1, truncation_cutoff=None, update_emas=False,
               **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self._query(ws, coordinates, directions, mesh_condition, update_emas, synthesis_kwargs)

    def sample_mixed(self, coordinates, directions, ws, mesh_condition, truncation_psi=1, truncation_cutoff=None, update_emas=False,
                     **synthesis_kwargs):
        return self._query(ws, coordinates, directions, mesh_condition, update_emas, synthesis_kwargs)

    def forward(self, z, c, v, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, v, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)


class OSGDecoder(torch.nn.Module):
    """Tiny MLP that decodes averaged tri-plane features into density + 32 colour channels (:415-438).
    Inside ``ia_render_rays`` its two layers run on MFMA; this module form serves CPU tensors and point queries."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = torch.nn.Sequential(
            FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=options['decoder_lr_mul']),
            torch.nn.Softplus(),
            FullyConnectedLayer(self.hidden_dim, 1 + options['decoder_output_dim'], lr_multiplier=options['decoder_lr_mul']))

    def forward(self, sampled_features, ray_directions, sampled_embeddings=None):
        x = sampled_features.mean(1)
        n, m, c = x.shape
        x = self.net(x.view(n * m, c)).view(n, m, -1)
        rgb = torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001   # MipNeRF sigmoid clamping
        return {'rgb': rgb, 'sigma': x[..., 0:1]}
