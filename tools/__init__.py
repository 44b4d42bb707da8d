"""Measurement and diagnostic tools of the MI355X vAttention hot path (kernel microbenchmarks, hardware probes, replay drivers)."""
