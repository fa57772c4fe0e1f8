"""Sensor plug-in protocol - mirror of the reference's ``gym_quadruped/sensors/base_sensor.py`` (:4-41).

In the reference a sensor reads ``mjData`` after every ``mj_step``.  On the batched path the measurement is produced
inside the step kernel; a sensor object here carries the configuration and exposes the result tensors.  ``mj_data``
receives the owning :class:`~gym_quadruped_amd.quadruped_env.QuadrupedEnv` (the batch plays the role of MjData).
"""
from __future__ import annotations


class Sensor:
    """Base class for all sensors in the environment."""

    def __init__(self, mj_model, mj_data, **kwargs):
        self._mj_model = mj_model
        self._mj_data = mj_data  # the batched env

    def step(self, **kwargs) -> None:
        """Called by the environment every simulation step (a no-op for kernel-side sensors)."""
        raise NotImplementedError

    def get_observation(self, obs_name: str):
        raise NotImplementedError

    @staticmethod
    def available_observations() -> list[str]:
        raise NotImplementedError
