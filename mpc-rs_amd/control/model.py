"""Mirror of src/control/model.rs: the DiscreteStateSpaceModel trait and the two plants.  Model::new runs on the device
(continuous matrices from the parameters, backward Euler A = (I - Ac dt)^-1, B = A (Bc dt), model.rs:55-58, 138-141)."""
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

    def _from_parameters(self, kind: int, params, dt: float) -> None:
        """Model::new on the device (mpcx_plants: continuous matrices from the parameters, then backward Euler)."""
        from .. import ffi
        from .simulator import context
        a, b, c, info = context().plants(kind, np.asarray([params], dtype=np.float64), float(dt))
        if info[0] != 0:  # try_inverse().unwrap() panics (model.rs:56, 139)
            raise ZeroDivisionError("called `Option::unwrap()` on a `None` value: I - Ac dt is singular")
        self._a, self._b, self._c, self._dt = a[0], b[0], c[0], float(dt)


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
            from .. import ffi
            p = params  # model.rs:47-58; parameter order of the device entry point: b, j, k, l, r
            self._from_parameters(ffi.PLANT_DC_MOTOR, [p.b, p.j, p.k, p.l, p.r], sampling_dt)


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
            from .. import ffi
            p = params  # model.rs:128-141; parameter order of the device entry point: m1, m2, k1, k2, d1, d2
            self._from_parameters(ffi.PLANT_TWO_MASS, [p.m1, p.m2, p.k1, p.k2, p.d1, p.d2], sampling_dt)
