"""Host-side mirror of the reference's ``control`` module (src/control.rs:1-4): ``controller``, ``model``, ``simulator``,
``trajectory`` with the reference's names, argument meaning and error behaviour.  Matrices are 2-D numpy arrays where the
reference has ``nalgebra::DMatrix<f64>``.  The hot path (lifted matrices, Hessian, gradient, rollout) goes through the C
ABI to the GPU; one-shot set-up (discretisation, weights, trajectory) stays on the host as in the reference."""
from . import controller, model, simulator, trajectory  # noqa: F401
