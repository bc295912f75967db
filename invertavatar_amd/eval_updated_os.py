"""One-shot inversion + reenactment harness: counterpart of the reference's eval_updated_os.py:94-95,171-200 for the improved
one-shot encoders (encoder_inversion/models/uvnet_new.py).

Flow of the script, restated: the whole network in eval() mode (:94-95); ``ws = G.encode(source)``; texture / static features of
that identity (:171-174); ONE forward of the inversion network on the source frame with those e4e results (:176-178); the updated
static features replace only the LAST entry (the 256^2 tri-plane) of the e4e static list (:179); the drive loop is
``synthesis_withTexture(..., evaluation=True)`` per frame as in eval_seq.py (:198)."""
import torch


@torch.no_grad()
def one_shot_inversion(net, image, uv, cam, uvcoords):
    """image [1,3,512,512], uv [1,6,256,256], cam [1,25], uvcoords [1,256,256,3] -> (ws, {'w','texture','static'})."""
    g = net.generator
    ws = net.encode(image)
    tex = g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
    sta = g.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
    out = net({'image': image, 'uv': uv}, cam, {'uvcoords_image': uvcoords}, e4e_results={'w': ws, 'texture': tex, 'static': sta},
              return_feats=True)
    return ws, {'w': ws, 'texture': out['texture'], 'static': list(sta[:-1]) + list(out['static'][-1:])}


class GraphedOneShot:
    """hipGraph of ``one_shot_inversion`` for one input shape: the flow is ~2500 launches of batch-1 work that eager PyTorch issues more
    slowly than the GPU runs them.  Captured once per network (a clip-processing service keeps it); a call copies the four inputs into
    the graph's tensors, replays, and returns copies of the results.  The split-range watch brackets the replay as it brackets the
    eager call.  The graph holds the packed weights of the moment of capture: capture again after changing parameters."""

    def __init__(self, net, image, uv, cam, uvcoords, warmup=2):
        self.net = net
        self.inputs = [t.clone() for t in (image, uv, cam, uvcoords)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                one_shot_inversion(net, *self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.ws, self.res = one_shot_inversion(net, *self.inputs)

    def __call__(self, image, uv, cam, uvcoords):
        from .reenact_avatar_next3d import _check_split_range
        for dst, src in zip(self.inputs, (image, uv, cam, uvcoords)):
            dst.copy_(src)
        _check_split_range(self.net, start=True)
        self.graph.replay()
        _check_split_range(self.net)
        ws = self.ws.clone()
        return ws, {'w': ws, 'texture': [t.clone() for t in self.res['texture']], 'static': [t.clone() for t in self.res['static']]}
