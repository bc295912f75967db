"""SURVEY §8(b), module-level boundary: the reference's scripts import its packages by their short names.  After
`invertavatar_amd.compat.install_aliases()` the import blocks of `reenact_avatar_next3d.py:5-19` and `eval_seq.py:7-16` (restated
below: the reference-package lines; third-party lines such as cv2 / click / imageio are the scripts' own business) resolve to
this package, and the generator is rebuilt the way `reenact_avatar_next3d.py:158-162` does it."""
import subprocess
import sys
import textwrap

SCRIPT = textwrap.dedent('''
    import sys
    import invertavatar_amd.compat as compat
    compat.install_aliases()
    compat.install_aliases()                          # idempotent

    # reenact_avatar_next3d.py:5-19 (reference-package imports)
    import legacy
    import dnnlib
    from torch_utils import misc
    from training_avatar_texture.camera_utils import LookAtPoseSampler, FOV_to_intrinsics
    from data_preprocess.FaceVerse.renderer import Faceverse_manager
    from training_avatar_texture.triplane_v20 import TriPlaneGenerator
    # eval_seq.py:7-16
    from encoder_inversion.models.uvnet import inversionNet
    from training_avatar_texture.dataset_new import ImageFolderDataset
    # eval_updated_os.py
    from encoder_inversion.models.uvnet_new import inversionNet as inversionNet_new
    # the operator level, as training/networks_stylegan2.py:16-20 imports it
    from torch_utils import persistence
    from torch_utils.ops import conv2d_resample, upfirdn2d, bias_act, fma
    from torch_utils.ops import filtered_lrelu, conv2d_gradfix, grid_sample_gradfix
    from torch_utils import custom_ops
    from training.networks_stylegan2 import Generator, SynthesisNetwork, MappingNetwork
    from training_avatar_texture.volumetric_rendering.renderer import ImportanceRenderer, ImportanceRenderer_bsMotion, fill_mouth
    from training_avatar_texture.volumetric_rendering.ray_sampler import RaySampler
    from training_avatar_texture.superresolution import SuperresolutionHybrid8XDC
    import camera_utils

    import invertavatar_amd
    import invertavatar_amd.training_avatar_texture.triplane_v20 as long_name
    import invertavatar_amd.encoder_inversion.models.uvnet as long_uvnet
    import invertavatar_amd.torch_utils.ops.bias_act as long_bias_act
    assert TriPlaneGenerator is long_name.TriPlaneGenerator
    assert inversionNet is long_uvnet.inversionNet
    assert bias_act is long_bias_act
    import torch_utils.ops.bias_act, training.networks_stylegan2
    assert sys.modules['torch_utils.ops.bias_act'] is long_bias_act
    assert sys.modules['training_avatar_texture'] is invertavatar_amd.training_avatar_texture
    assert misc is invertavatar_amd.torch_utils.misc and legacy is invertavatar_amd.legacy and dnnlib is invertavatar_amd.dnnlib
    assert long_name.__spec__.name == 'invertavatar_amd.training_avatar_texture.triplane_v20'      # not re-stamped
    assert long_name.__package__ == 'invertavatar_amd.training_avatar_texture'

    # reenact_avatar_next3d.py:158-162: rebuild from the pickle's recorded arguments and copy the tensors
    import copy, torch
    from invertavatar_amd import synthetic
    G = synthetic.fill_parameters(TriPlaneGenerator(**synthetic.generator_kwargs("small")).eval().requires_grad_(False))
    G_new = TriPlaneGenerator(*G.init_args, **G.init_kwargs).eval().requires_grad_(False).to('cpu')
    misc.copy_params_and_buffers(G, G_new, require_all=True)
    G_new.neural_rendering_resolution = G.neural_rendering_resolution
    G_new.rendering_kwargs = G.rendering_kwargs
    assert isinstance(G_new, long_name.TriPlaneGenerator) and isinstance(G, TriPlaneGenerator)
    for (n0, p0), (n1, p1) in zip(G.state_dict().items(), G_new.state_dict().items()):
        assert n0 == n1 and torch.equal(p0, p1), n0
    sr = dnnlib.util.construct_class_by_name(class_name='training_avatar_texture.superresolution.SuperresolutionHybrid8XDC',
                                             channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True)
    assert isinstance(sr, SuperresolutionHybrid8XDC)

    # eager form: every mirrored sub-module importable under its short name
    compat.install_aliases(eager=True)
    import pkgutil
    short = [n for n in sys.modules if n.split('.')[0] in compat.TOP_LEVEL]
    assert len(short) > 40, len(short)
    for n in short:
        assert sys.modules[n] is sys.modules['invertavatar_amd.' + n], n

    compat.remove_aliases()
    assert not [n for n in sys.modules if n.split('.')[0] in compat.TOP_LEVEL]
    print('ALIASES-OK')
''')


def _run(code, **kw):
    return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, **kw)


def test_reference_script_import_blocks_resolve_to_this_package():
    r = _run(SCRIPT)     # own interpreter: sys.modules / sys.meta_path of the test session stay untouched
    assert r.returncode == 0 and 'ALIASES-OK' in r.stdout, r.stderr[-3000:]


def test_aliases_win_over_a_checkout_on_sys_path_and_refuse_a_late_install(tmp_path):
    """A directory holding packages of the same names earlier on sys.path (= running a reference script from its checkout) does not
    shadow the aliases; installing AFTER such a module was imported is refused instead of mixing two class hierarchies."""
    (tmp_path / 'torch_utils').mkdir()
    (tmp_path / 'torch_utils' / '__init__.py').write_text('WHO = "checkout"\n')
    (tmp_path / 'torch_utils' / 'misc.py').write_text('WHO = "checkout"\n')
    code = textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {str(tmp_path)!r})
        import invertavatar_amd.compat as compat
        compat.install_aliases()
        from torch_utils import misc
        assert not hasattr(misc, 'WHO') and misc.__name__ == 'invertavatar_amd.torch_utils.misc'
        compat.remove_aliases()
        import torch_utils                      # now the checkout's
        assert torch_utils.WHO == 'checkout'
        try:
            compat.install_aliases()
        except ImportError as e:
            assert 'before' in str(e)
            print('LATE-REFUSED')
    ''')
    r = _run(code)
    assert r.returncode == 0 and 'LATE-REFUSED' in r.stdout, r.stderr[-3000:]
