"""Batched IMU - mirror of the reference's ``gym_quadruped/sensors/imu.py`` (``IMU`` :20-139).

Same constructor arguments and observable names (``imu_acc``, ``imu_acc_noise``, ``imu_acc_bias``, ``imu_gyro``,
``imu_gyro_noise``, ``imu_gyro_bias``).  The reference reads MuJoCo's accelerometer / gyro ``sensordata`` and adds
``np.random.normal`` noise plus a random-walk bias in place; here the ground truth (site-frame acceleration minus
gravity, site-frame angular velocity) and the noise model are evaluated inside ``gq_step`` for all envs, with
counter-based Philox normals (tests/philox_ref.py documents the stream).  Only sites on the base body are supported,
which is where every registry robot mounts its IMU.
"""
from __future__ import annotations

import ctypes as C

import torch

from ..cabi import IMU_OBS, GqImuCfg
from .base_sensor import Sensor

LIN_ACC_OBS = ('imu_acc', 'imu_acc_noise', 'imu_acc_bias')
GYRO_OBS = ('imu_gyro', 'imu_gyro_noise', 'imu_gyro_bias')


class IMU(Sensor):
    ALL_OBS = LIN_ACC_OBS + GYRO_OBS

    def __init__(self, mj_model, mj_data, accel_name, gyro_name, imu_site_name, accel_noise: float = 0.01,
                 gyro_noise: float = 0.01, accel_bias_rate: float = 0.01, gyro_bias_rate: float = 0.01, seed: int = 0):
        super().__init__(mj_model, mj_data)
        names = [s[0] for s in mj_model.sensors]
        for nm, kind in ((accel_name, 'accelerometer'), (gyro_name, 'gyro')):
            if nm not in names or mj_model.sensors[names.index(nm)][1] != kind:
                raise ValueError(f'{kind} "{nm}" not found in the model sensors {names}')
        if imu_site_name not in mj_model.site_names:
            raise ValueError(f'site "{imu_site_name}" not found in {mj_model.site_names}')
        sid = mj_model.site_names.index(imu_site_name)
        if int(mj_model.site_bodyid[sid]) != 1:
            raise NotImplementedError('the batched IMU supports sites on the base body only')
        self._accel_name, self._gyro_name = accel_name, gyro_name
        self.cfg = GqImuCfg(site_pos=(C.c_double * 3)(*mj_model.site_pos[sid]), site_quat=(C.c_double * 4)(*mj_model.site_quat[sid]),
                            accel_noise=accel_noise, gyro_noise=gyro_noise, accel_bias_rate=accel_bias_rate,
                            gyro_bias_rate=gyro_bias_rate, seed=int(seed))
        env = mj_data
        #: [N, 6] accelerometer / gyro bias random walks (persist across resets, like the reference's IMU object)
        self.bias_state = torch.zeros(env.num_envs, 6, dtype=torch.float32, device=env.device)

    def step(self):
        """Nothing to do: the measurement of this step was produced by the step kernel."""

    def get_observation(self, obs_name):
        if obs_name not in self.ALL_OBS:
            raise ValueError(f'Invalid observation name {obs_name}')
        return self._mj_data._obs_views[obs_name]

    @staticmethod
    def available_observations():
        return IMU.ALL_OBS


assert tuple(IMU_OBS) == IMU.ALL_OBS
