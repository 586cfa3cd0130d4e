"""MI355X-native MultiSWAG ensemble inference for the bnn_chaos_model Bayesian network.

Only the hot path is here: SWAG weight draws + the BNN forward, as hand-written
gfx950 kernels behind the reference's Python surface.  Nothing in this package falls back to a CPU path.
"""
__version__ = "0.1.0"
