"""Mirror of src/control/model.rs: the DiscreteStateSpaceModel trait and the two plants, discretised with the
reference's backward-Euler rule A = (I - Ac dt)^-1, B = A (Bc dt) (model.rs:55-58, 138-141)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


class DiscreteStateSpaceModel:
    """model.rs:3-8."""

    def get_mat_a(self) -> np.ndarray:
        return self._a

    def get_mat_b(self) -> np.ndarray:
        return self._b

    def get_mat_c(self) -> np.ndarray:
        return self._c

    def get_sampling_dt(self) -> float:
        return self._dt

    def _discretise(self, ac: np.ndarray, bc: np.ndarray, cc: np.ndarray, dt: float) -> None:
        t = np.eye(ac.shape[0]) - ac * dt
        if np.linalg.det(t) == 0.0:  # try_inverse().unwrap() panics (model.rs:56, 139)
            raise ZeroDivisionError("called `Option::unwrap()` on a `None` value: I - Ac dt is singular")
        self._a = np.linalg.inv(t)
        self._b = self._a @ (bc * dt)
        self._c = cc
        self._dt = float(dt)


class dc_motor:  # noqa: N801  (module name in the reference)
    @dataclass
    class Parameters:
        """model.rs:16-34 (defaults)."""
        b: float = 0.1
        j: float = 0.01
        k: float = 0.01
        l: float = 0.5  # noqa: E741
        r: float = 1.0

    class Model(DiscreteStateSpaceModel):
        def __init__(self, params: "dc_motor.Parameters", sampling_dt: float):
            p = params
            ac = np.array([[-p.b / p.j, p.k / p.j], [-p.k / p.l, -p.r / p.l]])  # model.rs:47-50
            bc = np.array([[0.0], [1.0 / p.l]])
            cc = np.array([[1.0, 0.0]])
            self._discretise(ac, bc, cc, sampling_dt)


class mpc:  # noqa: N801
    @dataclass
    class Parameters:
        """model.rs:95-115 (defaults)."""
        m1: float = 2.0
        m2: float = 2.0
        k1: float = 100.0
        k2: float = 200.0
        d1: float = 1.0
        d2: float = 5.0

    class Model(DiscreteStateSpaceModel):
        def __init__(self, params: "mpc.Parameters", sampling_dt: float):
            p = params
            ac = np.array([  # model.rs:128-133
                [0.0, 1.0, 0.0, 0.0],
                [-(p.k1 + p.k2) / p.m1, -(p.d1 + p.d2) / p.m1, p.k2 / p.m1, p.d2 / p.m1],
                [0.0, 0.0, 0.0, 1.0],
                [p.k2 / p.m2, p.d2 / p.m2, -p.k2 / p.m2, -p.d2 / p.m2],
            ])
            bc = np.array([[0.0], [0.0], [0.0], [1.0 / p.m2]])
            cc = np.array([[1.0, 0.0, 0.0, 0.0]])
            self._discretise(ac, bc, cc, sampling_dt)
