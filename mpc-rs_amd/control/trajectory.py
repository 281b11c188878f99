"""Mirror of src/control/trajectory.rs."""
import numpy as np


def generate_pulse_trajectory(time_steps: int) -> np.ndarray:
    """trajectory.rs:3-11: (time_steps x 1), ones on [0, T/3) and [2T/3, T).  Raises where nalgebra's copy_from
    would panic on the shape mismatch (T - 2T/3 != T/3 in integer division)."""
    third = time_steps // 3
    if time_steps - 2 * time_steps // 3 != third:
        raise ValueError("Unable to copy from a matrix with incompatible shape.")
    traj = np.zeros((time_steps, 1))
    traj[:third, 0] = 1.0
    traj[2 * time_steps // 3:, 0] = 1.0
    return traj
