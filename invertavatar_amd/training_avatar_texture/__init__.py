"""Mirror of the reference's ``training_avatar_texture`` package (Next3D++ v20 generator path)."""
