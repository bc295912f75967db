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
