"""Mirror of the reference's ``training`` package (generator-side modules only)."""
