"""Mirror of the reference's ``torch_utils`` package for the generator hot path."""
