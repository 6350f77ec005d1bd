"""Code-tensor helpers of the reference's ``vall_e/emb`` package that sit on the generation path (no codec model)."""
