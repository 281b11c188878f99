"""Mirror of src/control/controller.rs (module ``mpc``): the lifted matrices, Hessian and rollout step run on the GPU
through the C ABI; the weights of ``new`` are built on the host exactly as the reference builds them."""
from __future__ import annotations

import numpy as np

from .. import ffi
from .model import DiscreteStateSpaceModel
from .simulator import context


def _block_of(mat_w4: np.ndarray, f: int, r: int):
    """The r x r block if W4 is block-diagonal with one repeated block (what Controller::new builds), else None."""
    q = mat_w4[:r, :r]
    rebuilt = np.zeros_like(mat_w4)
    for i in range(f):
        rebuilt[i * r:(i + 1) * r, i * r:(i + 1) * r] = q
    return q if np.array_equal(rebuilt, mat_w4) else None


class mpc:  # noqa: N801  (module name in the reference)
    class Controller:
        def __init__(self, model, f, mat_o, gain_matrix, desired_ctrl_traj_total):
            self._model = model
            self._f = f
            self._mat_o = mat_o
            self._gain_matrix = gain_matrix
            self.desired_ctrl_traj_total = desired_ctrl_traj_total  # pub field, controller.rs:12

        @staticmethod
        def lifted_and_hessian(mat_a, mat_b, mat_c, f: int, v: int, mat_w3, mat_w4):
            """(O, M, H) on the GPU: controller.rs:28-93 up to the matrix handed to try_inverse."""
            mat_a, mat_b, mat_c = (np.asarray(x, dtype=np.float64) for x in (mat_a, mat_b, mat_c))
            mat_w3, mat_w4 = np.asarray(mat_w3, dtype=np.float64), np.asarray(mat_w4, dtype=np.float64)
            n, m, r = mat_a.shape[0], mat_b.shape[1], mat_c.shape[0]
            if mat_a.shape != (n, n) or mat_b.shape[0] != n or mat_c.shape[1] != n:
                raise ValueError("Matrix multiplication dimensions mismatch.")  # nalgebra's panic at :43 / :61
            if mat_w3.shape != (v * m, v * m) or mat_w4.shape != (f * r, f * r):
                raise ValueError("Matrix multiplication dimensions mismatch.")  # :92-93
            q = _block_of(mat_w4, f, r)
            mode, w4 = (ffi.WEIGHTS_BLOCK, q) if q is not None else (ffi.WEIGHTS_DENSE, mat_w4)
            out = context().qp_build(mat_a[None], mat_b[None], mat_c[None], f, v, W3=mat_w3, W4=w4, weights_mode=mode,
                                     want=("Phi", "Gamma", "H"))
            return out["Phi"][0], out["Gamma"][0], out["H"][0]

        @staticmethod
        def form_lifted_matrices(mat_a, mat_b, mat_c, f: int, v: int, mat_w3, mat_w4):
            """controller.rs:16-99: returns (mat_o, mat_m, gain_matrix); raises RuntimeError("Matrix inversion failed")
            where the reference returns that Err (:93-95).  The inversion and H^-1 M^T W4 run on the device
            (mpcx_qp_gain: LU with partial pivoting, status per instance)."""
            mat_o, mat_m, h = mpc.Controller.lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4)
            mat_w4 = np.asarray(mat_w4, dtype=np.float64)
            m, r = np.asarray(mat_b).shape[1], np.asarray(mat_c).shape[0]
            q = _block_of(mat_w4, f, r)
            mode, w4 = (ffi.WEIGHTS_BLOCK, q) if q is not None else (ffi.WEIGHTS_DENSE, mat_w4)
            gain, info = context().qp_gain(h[None], mat_m[None], w4, m, r, f, v, weights_mode=mode)
            if info[0] != 0:
                raise RuntimeError("Matrix inversion failed")
            return mat_o, mat_m, gain[0]

        @staticmethod
        def form_first_move_rows(mat_a, mat_b, mat_c, f: int, v: int, mat_w3, mat_w4):
            """(mat_o, rows) with rows = the first m rows of the gain matrix, which is all the control move uses
            (controller.rs:130-132).  An addition: m solves with H^T instead of f r solves (mpcx_qp_gain_rows); raises
            RuntimeError("Matrix inversion failed") where form_lifted_matrices does."""
            mat_o, mat_m, h = mpc.Controller.lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4)
            mat_w4 = np.asarray(mat_w4, dtype=np.float64)
            m, r = np.asarray(mat_b).shape[1], np.asarray(mat_c).shape[0]
            q = _block_of(mat_w4, f, r)
            mode, w4 = (ffi.WEIGHTS_BLOCK, q) if q is not None else (ffi.WEIGHTS_DENSE, mat_w4)
            rows, info = context().qp_gain_rows(h[None], mat_m[None], w4, m, r, f, v, weights_mode=mode)
            if info[0] != 0:
                raise RuntimeError("Matrix inversion failed")
            return mat_o, rows[0]

        @classmethod
        def new(cls, model: DiscreteStateSpaceModel, f: int, v: int, q0: float, q_other: float, pred_weight: float,
                desired_ctrl_traj: np.ndarray, all_elements: bool = True, first_move_only: bool = False) -> "mpc.Controller":
            """controller.rs:137-207.  `all_elements` (an addition, default = the reference) keeps the reference's weight
            blocks, every element of an m x m / r x r block set to the weight (:164-165, :184); False builds q I blocks,
            which is what a plant with several inputs or outputs needs for the Hessian to be invertible.
            `first_move_only` (an addition, default = the reference's full gain) keeps only the first m rows of the gain
            matrix: every method of the controller reads no more than those."""
            m = model.get_mat_b().shape[1]
            r = model.get_mat_c().shape[0]
            w1 = np.zeros((v * m, v * m))  # :149-161
            w2 = np.zeros((v * m, v * m))  # :164-175
            for i in range(v):
                w1[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.eye(m)
                if i >= 1:
                    w1[i * m:(i + 1) * m, (i - 1) * m:i * m] = -np.eye(m)
                qi = q0 if i == 0 else q_other
                w2[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.full((m, m), qi) if all_elements else qi * np.eye(m)
            w3 = w1.T @ (w2 @ w1)  # :177
            w4 = np.zeros((f * r, f * r))  # :179-185
            for i in range(f):
                w4[i * r:(i + 1) * r, i * r:(i + 1) * r] = np.full((r, r), pred_weight) if all_elements else pred_weight * np.eye(r)
            if first_move_only:
                mat_o, gain = cls.form_first_move_rows(model.get_mat_a(), model.get_mat_b(), model.get_mat_c(), f, v, w3, w4)
            else:
                mat_o, _, gain = cls.form_lifted_matrices(model.get_mat_a(), model.get_mat_b(), model.get_mat_c(), f, v, w3, w4)
            return cls(model, f, mat_o, gain, np.array(desired_ctrl_traj, dtype=np.float64, copy=True))

        def propagate_dynamics(self, control_input: np.ndarray, state: np.ndarray):
            """controller.rs:101-115: (x_next, y) with y the output at the current state; one rollout step on the GPU."""
            a, b, c = self._model.get_mat_a(), self._model.get_mat_b(), self._model.get_mat_c()
            u = np.asarray(control_input, dtype=np.float64).reshape(1, -1, 1)
            out = context().rollout(a[None], b[None], c[None], np.asarray(state, dtype=np.float64).reshape(1, -1), u)
            return out["X"][0, :, 1].copy(), out["Y"][0, :, 0].copy()

        def _first_move(self, timestep: int, state: np.ndarray) -> np.ndarray:
            """gain (window - O x)[0 .. m) on the device: one step of the receding-horizon loop started at `state`
            (mpcx_closed_loop with T = 1 keeps the reference's order: s first, then gain s)."""
            a, b, c = self._model.get_mat_a(), self._model.get_mat_b(), self._model.get_mat_c()
            r, v = c.shape[0], self._gain_matrix.shape[0] // b.shape[1]
            traj = np.asarray(self.desired_ctrl_traj_total, dtype=np.float64)
            if timestep + self._f > traj.shape[0]:
                raise IndexError("Matrix slicing out of bounds.")  # view_range panics at :121
            if traj.shape[1] != r:
                raise ValueError("Matrix addition/subtraction dimensions mismatch.")  # :125
            ref = np.zeros((r, self._f + 1))  # r x (T + f) with T = 1; the last column is never read
            ref[:, :self._f] = traj[timestep:timestep + self._f, :].T
            x = np.asarray(state, dtype=np.float64).reshape(1, -1)
            out = context().closed_loop(a[None], b[None], c[None], x, self._mat_o[None], self._gain_matrix[None], ref, 1,
                                        self._f, v)
            return out["U"][0, :, 0].copy()

        def compute_control_input(self, timestep: int, state: np.ndarray) -> np.ndarray:
            """controller.rs:117-135: first element of gain (traj[t..t+f] - O x), as a 1-vector (one input, one output)."""
            return self._first_move(timestep, state)[:1]  # :131-134

        # ---- additions to the reference API (it stops at one input and one output, :119-122 and :131-132) ----
        def compute_control_input_mimo(self, timestep: int, state: np.ndarray) -> np.ndarray:
            """The move for m inputs and r outputs: the first m entries of gain (window - O x), the window being rows
            timestep .. timestep + f of a (time_steps x r) trajectory stacked time-major."""
            return self._first_move(timestep, state)

        def closed_loop(self, steps: int, x0: np.ndarray):
            """The loop of src/main.rs:51-71 in one device call: returns (states n x (steps+1), inputs m x steps,
            outputs r x (steps+1)) with outputs[:, 0] = C x0 as main.rs:40-41 records it."""
            a, b, c = self._model.get_mat_a(), self._model.get_mat_b(), self._model.get_mat_c()
            r, v = c.shape[0], self._gain_matrix.shape[0] // b.shape[1]
            traj = np.asarray(self.desired_ctrl_traj_total, dtype=np.float64)
            if steps + self._f > traj.shape[0]:
                raise IndexError("Matrix slicing out of bounds.")
            ref = np.ascontiguousarray(traj[:steps + self._f, :].T).reshape(r, steps + self._f)
            x0 = np.asarray(x0, dtype=np.float64).reshape(1, -1)
            out = context().closed_loop(a[None], b[None], c[None], x0, self._mat_o[None], self._gain_matrix[None], ref, steps,
                                        self._f, v)
            outputs = np.concatenate([(c @ x0[0]).reshape(r, 1), out["Y"][0]], axis=1)
            return out["X"][0], out["U"][0], outputs
