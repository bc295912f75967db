"""InvertAvatar inversion network (reference: encoder_inversion/models/uvnet.py:15-203).

``inversionNet`` = e4e W+ encoder + texture UNet (offsets on the neural-texture pyramid, input in UV space)
+ tri-plane UNet (CS-SFT conditions for the static backbone), wrapped around a frozen ``TriPlaneGenerator``.
``AR_eval_forward`` is the incremental few-shot step of eval_seq.py:173-190: render the current estimate for the T
source frames, feed the residual to both UNets, whose ConvGRU states ``r_list`` carry over to the next group."""
import contextlib

import torch
import torch.nn.functional as F
from torch import nn

from ... import dnnlib
from ...torch_utils import misc
from .e4e import Encoder4Editing
from .helpers import face_pool_to
from .unet_encoders import TriPlaneSFTfeat_Encoder, TriPlanefeat_Encoder


class unet_encoder(nn.Module):
    def __init__(self, encoding_texture=False, encoding_triplane=False):
        super().__init__()
        self.texture_unet = TriPlanefeat_Encoder(inp_ch=7, res=256, use_gru=True) if encoding_texture else None
        self.triplane_unet = TriPlaneSFTfeat_Encoder(inp_ch=6, res=256, use_gru=True) if encoding_triplane else None

    def forward(self, x):
        raise NotImplementedError


UNET_CHAINS_CONCURRENT = True      # device path of AR_eval_forward: texture chain on a side stream beside the tri-plane chain


def _add_offsets(feats, offsets):
    """Offsets apply to the first len(offsets) features; the rest pass through (uvnet.py:142-146)."""
    return [f + o for f, o in zip(feats, offsets)] + list(feats[len(offsets):])


class inversionNet(nn.Module):
    def __init__(self, G_kwargs=None, generator=None, encoding_texture=True, encoding_triplane=False):
        super().__init__()
        self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.generator = generator if generator is not None else \
            dnnlib.util.construct_class_by_name(**G_kwargs).train().requires_grad_(False)
        self.register_buffer('latent_avg', self.generator.backbone.mapping.w_avg.reshape(1, 512))
        self.n_styles = self.generator.texture_backbone.num_ws
        self.encoder = self.set_encoder(self.n_styles, inp_ch=3)
        self.unet_encoder = unet_encoder(encoding_texture=encoding_texture, encoding_triplane=encoding_triplane)
        self.register_buffer('black_uv_bg', -1 * torch.ones(1, 3, 256, 256, dtype=torch.float32))

    def set_encoder(self, n_styles, inp_ch):
        return Encoder4Editing(n_styles, inp_ch)

    def switch_grad(self, nerf_requires_grad=False):
        for i in range(self.encoder.middle_ind):
            for p in self.encoder.styles[i].parameters():
                p.requires_grad = nerf_requires_grad

    def print_parameter_numbers(self):
        for name, mod in (('encoder', self.encoder), ('triplane_unet', self.unet_encoder.triplane_unet),
                          ('texture_unet', self.unet_encoder.texture_unet), ('generator', self.generator)):
            print(f'{name} parmeters number is :    ', sum(p.numel() for p in mod.parameters()))

    def initialize_encoders(self, ir_se50_path, triplanenet_path=None):
        """Warm-start the three trunks from an IR-SE50 checkpoint (uvnet.py:73-101)."""
        if self.unet_encoder.texture_unet is not None:
            misc.copy_params_and_buffers(self.generator.texture_backbone, self.unet_encoder.texture_unet, require_all=False)
        ckpt = torch.load(ir_se50_path, map_location='cpu')
        self.encoder.load_state_dict(ckpt, strict=False)
        unet = self.unet_encoder.triplane_unet
        if unet is not None:
            if triplanenet_path is None:   # 6-channel input: reuse the RGB filters for the first three channels
                w = ckpt['input_layer.0.weight']
                wide = torch.randn(w.shape[0], 6, w.shape[2], w.shape[3], dtype=torch.float32)
                wide[:, :3] = w
                ckpt['input_layer.0.weight'] = wide
                unet.load_state_dict(ckpt, strict=False)
            else:
                sd = torch.load(triplanenet_path, map_location='cpu')['state_dict']
                prefix = 'triplanenet_encoder.'
                merged = unet.state_dict()
                merged.update({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
                unet.load_state_dict(merged, strict=False)

    def encode(self, x):
        if x.shape[-1] != 256:
            x = face_pool_to(self.face_pool, x)
        if type(self.encoder) is nn.ModuleList:
            codes = torch.cat([enc(x) for enc in self.encoder], dim=1)
        else:
            codes = self.encoder(x)
        return codes + self.latent_avg.repeat(codes.shape[0], 1, 1)

    def get_unet_uvinput(self, uv, delta_x):
        """Image-space residual pulled into UV space: [gt texture (3), residual (3), mask (1)] at 256^2 (uvnet.py:117-121)."""
        uv_gttex, uv_pverts = uv.split(3, dim=1)
        mask = uv_pverts[:, -1:]
        uv_delta = F.grid_sample(delta_x, uv_pverts.permute(0, 2, 3, 1)[..., :2], mode='bilinear', align_corners=False)
        return torch.cat([uv_gttex, uv_delta * mask + self.black_uv_bg * (1 - mask), mask], dim=1)

    def _backbones(self, ws, feat_conditions=None):
        g = self.generator
        tex = g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
        sta = g.backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=feat_conditions, update_emas=False,
                                   noise_mode='const')
        return tex, sta

    def forward(self, x, cam, v, e4e_results=None, visualize_input=False, return_feats=False):
        """One-shot inversion of the frames in x['image'] (uvnet.py:123-157)."""
        g = self.generator
        with torch.no_grad():
            if e4e_results is None:
                ws = self.encode(x['image'][:, :3])
                tex0, sta0 = self._backbones(ws)
            else:
                ws, tex0, sta0 = e4e_results['w'], e4e_results['texture'], e4e_results['static']
            y0 = g.synthesis_withTexture(ws, tex0, cam, v, static_feats=sta0, noise_mode='const')
            if y0['image'].shape[-1] != x['image'].shape[-1]:
                y0['image'] = F.interpolate(y0['image'], size=(256, 256), mode='bilinear', align_corners=False, antialias=True)
            delta_x = y0['image'] - x['image'][:, :3]
        assert x['uv'] is not None
        x_input = self.get_unet_uvinput(x['uv'], delta_x)
        texture_feats = _add_offsets(tex0, self.unet_encoder.texture_unet(x_input, return_list=True))
        sft = self.unet_encoder.triplane_unet(torch.cat([x['image'][:, :3], delta_x], dim=1))
        static_feats = g.backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=sft, update_emas=False,
                                            noise_mode='const')
        out = g.synthesis_withTexture(ws, texture_feats, cam, v, static_feats=static_feats, noise_mode='const')
        if return_feats:
            out['texture'], out['static'] = texture_feats, static_feats
        out['w'] = ws
        out['e4e_image'] = y0['image']
        if visualize_input:
            out['x_input'] = torch.clamp(x_input, min=-1, max=1)
        return out

    def trunks_in_eval_mode(self):
        """True when no BatchNorm of the two IR-SE50 trunks is in training mode -- what the script arranges (eval_seq.py:96-97:
        input_layer.eval() / body.eval() after .train()) and what makes a frame's trunk features independent of the other frames."""
        bn = torch.nn.modules.batchnorm._BatchNorm
        return not any(m.training for unet in (self.unet_encoder.texture_unet, self.unet_encoder.triplane_unet)
                       for part in (unet.input_layer, unet.body) for m in part.modules() if isinstance(m, bn))      # (both attributes: unet_encoders.py:162)

    def require_eval_trunks(self):
        if not self.trunks_in_eval_mode():
            raise RuntimeError('trunk_features runs the IR-SE50 trunks frame by frame: their BatchNorms must be in eval mode '
                               '(eval_seq.py:96-97 calls input_layer.eval() and body.eval()); in training mode the batch statistics '
                               'span the group and the running statistics would diverge per rank')

    @torch.no_grad()
    def trunk_features(self, image, uv, y0_image):
        """IR-SE50 trunk features of both UNets for source frames [T, ...] given their renders from the e4e features: the per-frame
        half of AR_eval_forward (eval-mode BatchNorm: no coupling between frames).  Returns {'texture': [5 tensors], 'triplane': [5]}."""
        self.require_eval_trunks()
        delta_x = y0_image - image[:, :3]
        tri_input = torch.cat([image[:, :3], delta_x], dim=-3)
        # two latency-bound launch chains (3.3 ms each on one frame, 3.6 ms on four: r05 stage profile) that read delta_x and nothing
        # of each other: on the device the texture trunk runs on the side stream AR_eval_forward uses for the texture chain
        fork = UNET_CHAINS_CONCURRENT and delta_x.is_cuda and not torch.is_grad_enabled()
        if fork:
            from ... import _runtime
            st = _runtime.state(self)
            if getattr(st, 'chain_stream', None) is None or st.chain_stream.device != delta_x.device:
                st.chain_stream = torch.cuda.Stream(device=delta_x.device)
            main, side = torch.cuda.current_stream(delta_x.device), st.chain_stream
            side.wait_stream(main)
        with (torch.cuda.stream(side) if fork else contextlib.nullcontext()):
            uv_input = self.get_unet_uvinput(uv, delta_x)
            tex = self.unet_encoder.texture_unet.forward_onlyEncoder(uv_input.unsqueeze(0))
        tri = self.unet_encoder.triplane_unet.forward_onlyEncoder(tri_input.unsqueeze(0))
        if fork:
            main.wait_stream(side)
            for t in tex:
                t.record_stream(main)
        return {'texture': tex, 'triplane': tri}

    @torch.no_grad()
    def AR_eval_forward(self, x, vid_c, vid_v, ws, r_list, e4e_results=None, return_fake=False, y0_image=None, parts=('texture', 'triplane'),
                        trunk_feats=None):
        """Incremental update from one group of T source frames (uvnet.py:160-203).
        x['image'] [T,3,512,512], x['uv'] [T,6,256,256]; r_list = [texture GRU states, tri-plane GRU states].
        `y0_image`: the group's render from the e4e features, when the caller already has it (inversion_parallel renders the source
        frames of all groups sharded over the ranks before the UNet chains run).  `parts`: which of the two independent UNet chains
        to run -- 'texture' (texture UNet -> texture feature offsets) and / or 'triplane' (tri-plane UNet -> conditioned static
        backbone); the features of a chain that is left out come back as the e4e features, its ConvGRU states untouched.
        `trunk_feats`: {'texture': [...], 'triplane': [...]}, the IR-SE50 trunk features of the group's T frames when the caller has
        them already (`trunk_features`; inversion_parallel computes them sharded by frame): the chains then run their decoders only."""
        g = self.generator
        T = vid_c.shape[0]
        if ws is None:
            ws = self.encode(x['image'][0:1])
        if e4e_results is None:
            texture_feats, static_feats = self._backbones(ws)
        else:
            texture_feats, static_feats = e4e_results['texture'], e4e_results['static']
        vid_ws = ws.expand(T, -1, -1)

        def over_frames(feats):
            return [f.expand(T, -1, -1, -1) for f in feats]

        y0 = {'image': y0_image} if y0_image is not None else g.synthesis_withTexture(
            vid_ws, over_frames(texture_feats), vid_c, vid_v, static_feats=over_frames(static_feats), noise_mode='const')
        delta_x = y0['image'] - x['image'][:, :3]
        uv_input = None
        # The two chains read delta_x and nothing of each other, and each is a long sequence of small launches (IR-SE50 trunk on four
        # frames + ConvGRU decoder: ~7.5 ms that leave most of the chip idle): on the device the texture chain runs on a side stream
        # beside the tri-plane chain (r04: the same split inversion_parallel makes between two ranks).
        fork = (UNET_CHAINS_CONCURRENT and len(parts) == 2 and delta_x.is_cuda and not torch.is_grad_enabled())
        if fork:
            from ... import _runtime
            st = _runtime.state(self)
            if getattr(st, 'chain_stream', None) is None or st.chain_stream.device != delta_x.device:
                st.chain_stream = torch.cuda.Stream(device=delta_x.device)
            main, side = torch.cuda.current_stream(delta_x.device), st.chain_stream
            side.wait_stream(main)
        if 'texture' in parts:
            with (torch.cuda.stream(side) if fork else contextlib.nullcontext()):
                if trunk_feats is not None:
                    offsets, r_list[0] = self.unet_encoder.texture_unet.forward_onlyDecoder(T, trunk_feats['texture'], r_list[0])
                else:
                    uv_input = self.get_unet_uvinput(x['uv'], delta_x)
                    offsets, r_list[0] = self.unet_encoder.texture_unet(uv_input.unsqueeze(0), r_list=r_list[0], return_list=True)
                texture_feats = _add_offsets(texture_feats, offsets)
        if 'triplane' in parts:
            if trunk_feats is not None:
                sft, r_list[1] = self.unet_encoder.triplane_unet.forward_onlyDecoder(T, trunk_feats['triplane'], r_list[1])
            else:
                tri_input = torch.cat([x['image'][:, :3], delta_x], dim=-3)
                sft, r_list[1] = self.unet_encoder.triplane_unet(tri_input.unsqueeze(0), r_list=r_list[1])
            static_feats = g.backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=sft, update_emas=False,
                                                noise_mode='const')
        if fork:
            main.wait_stream(side)
            for t in list(texture_feats) + list(r_list[0]) + ([uv_input] if uv_input is not None else []):
                t.record_stream(main)
        updated = {'w': ws, 'texture': texture_feats, 'static': static_feats}
        if not return_fake:
            return updated, r_list
        fake = g.synthesis_withTexture(vid_ws, over_frames(texture_feats), vid_c, vid_v, static_feats=over_frames(static_feats),
                                       noise_mode='const', evaluation=True)['image']
        return updated, {'e4e': y0['image'], 'image': fake, 'x_input': uv_input}, r_list
