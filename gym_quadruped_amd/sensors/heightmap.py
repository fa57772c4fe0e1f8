"""Batched HeightMap - mirror of the reference's ``gym_quadruped/sensors/heightmap.py`` (``HeightMap`` :17-221).

The reference casts ``num_rows x num_cols`` downward ``mujoco.mj_ray`` rays from a grid laid out in the robot's
heading frame (``create_sensor_matrix`` :106-169, ``raycast_sensor`` :66-104: ray origin z = center.z + 0.6 - 0.07,
direction -z, static geoms only).  Here one kernel launch (``gq_heightmap``) casts the rays of all envs; on the flat
scene the only static geom is the floor plane.  Like the reference class this is NOT a ``Sensor`` subclass: the user
calls ``update_height_map(center, yaw)`` after ``step``.
"""
from __future__ import annotations

import torch

from .. import _lib


class HeightMap:
    def __init__(self, num_rows, num_cols, dist_x, dist_y, mj_model, mj_data):
        self.mj_model, self.mj_data = mj_model, mj_data   # mj_data: the batched env
        env = mj_data
        self.num_rows, self.num_cols, self.dist_x, self.dist_y = int(num_rows), int(num_cols), float(dist_x), float(dist_y)
        self.sensor_data_matrix = torch.zeros(env.num_envs, self.num_rows, self.num_cols, 1, 3, dtype=torch.float32, device=env.device)
        self.data = None

    def create_sensor_matrix(self, center, yaw=0.0):
        """center: [N,3] (e.g. env.qpos[:, 0:3]); yaw: [N] or float.  Returns [N, rows, cols, 1, 3] hit points."""
        env = self.mj_data
        # views (env.qpos[:, 0:3], a column of the observation row, one broadcast yaw) are read in place through their row stride:
        # .contiguous() would put a staging copy per argument on the stream in front of the ray kernel
        c = torch.as_tensor(center, dtype=torch.float64, device=env.device).reshape(-1, 3)
        c = c.expand(env.num_envs, 3)
        if c.stride(1) != 1:
            c = c.contiguous()
        y = torch.as_tensor(yaw, dtype=torch.float32, device=env.device).reshape(-1).expand(env.num_envs)
        self._keep = (c, y)   # alive until the next call: the launch is asynchronous
        stream = torch.cuda.current_stream(env.device).cuda_stream
        _lib.check(_lib.lib().gq_heightmap_strided(env._hbatch, c.data_ptr(), int(c.stride(0)), y.data_ptr(), int(y.stride(0)), self.num_rows, self.num_cols,
                                                   self.dist_x, self.dist_y, self.sensor_data_matrix.data_ptr(), stream), 'gq_heightmap_strided')
        return self.sensor_data_matrix

    def update_height_map(self, center, yaw=0.0):
        self.data = self.create_sensor_matrix(center, yaw)
        return self.data
