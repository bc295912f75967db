"""Inversion encoders in front of the generator (mirror of the reference's ``encoder_inversion`` package, models only)."""
