"""bench.py's two encoder legs (few-shot inversion of 8 sources; one-shot inversion) under the route switches of the encoder
convolutions.  python tools/ab_encoder_legs.py [MODULE.ATTR=VALUE ...]   (e.g. invertavatar_amd.encoder_inversion.models.layers.HIP_CONVS=False)"""
import ast
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import encoder_bench, synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

GROUP_GRAPH, WHOLE = '--group-graph' in sys.argv, '--whole-graph' in sys.argv
for spec in [a for a in sys.argv[1:] if not a.startswith('--')]:
    target, value = spec.split('=', 1)
    mod, attr = target.rsplit('.', 1)
    setattr(importlib.import_module(mod), attr, ast.literal_eval(value))
gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).train().requires_grad_(False).cuda()
synthetic.fill_parameters(gen)
with torch.no_grad():
    few = encoder_bench.encoder_leg(gen, group_graph=GROUP_GRAPH, whole_graph=WHOLE)
    one = encoder_bench.oneshot_leg(gen)
print(sys.argv[1:], 'few-shot inversion_ms', few['inversion_ms'], few['inversion_ms_runs'], 'drive f/s', few['drive_frames_per_s'],
      '| one-shot inversion_ms', one['inversion_ms'], one['inversion_ms_runs'])
