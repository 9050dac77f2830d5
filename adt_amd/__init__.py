"""adt_amd: MI355X-native (gfx950) hot path of the Adaptive Disentangled Transformer sequential recommender.

Layout: csrc/ holds the hand-written HIP kernels and the C ABI (include/adt_hip.h); `ops` wraps the ABI for
torch tensors; `sasrec` mirrors the reference's sasrec/ module API (model, trainer loop, config surface).
"""
__all__ = ["_lib", "ops"]
