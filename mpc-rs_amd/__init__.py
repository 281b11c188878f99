"""mpc-rs_amd: MI355X-native batched MPC hot path (condensed-QP construction and state-space rollout).

``ffi``      ctypes binding of the C ABI in include/mpcx.h (libmpcx.so, hand-written HIP for gfx950)
``layout``   packing of per-instance matrices into the batch layouts the ABI takes
``control``  host-side mirror of the reference's ``control::{model, controller, simulator, trajectory}`` API
``synth``    seeded synthetic plant batches for tests and benchmarks
``shard``    batch sharding by instance index and the statistics record ranks exchange

The compute path is the GPU library only: importing ``ffi`` fails loudly when libmpcx.so is missing.
"""
__all__ = ["ffi", "layout", "control", "synth", "shard"]
