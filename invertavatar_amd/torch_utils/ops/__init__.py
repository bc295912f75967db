"""Operator API of the generator path (mirrors the reference's torch_utils.ops)."""
