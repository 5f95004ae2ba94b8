"""Learner-side hot path: batch ABI, pad/pack collation, RL loss (MI355X kernels underneath)."""
