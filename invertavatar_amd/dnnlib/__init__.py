"""Subset of the reference's ``dnnlib`` that the generator path and its callers use
(reference: dnnlib/util.py:42 EasyDict, :303 construct_class_by_name)."""
from . import util
from .util import EasyDict, construct_class_by_name, get_obj_by_name, call_func_by_name
