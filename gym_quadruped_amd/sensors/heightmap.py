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
    def __init__(self, num_rows, num_cols, dist_x, dist_y, mj_model, mj_data, follow_base: bool = False):
        """``follow_base=True`` (extension; scenes with world boxes or a height field): the map is the one the reference's examples
        keep - ``update_height_map(qpos[0:3], yaw=base_ori_euler_xyz[2])`` after every step (examples/aliengo_with_heightmap.py) - and
        the STEP KERNEL casts its rays (``gq_batch_set_heightmap``): ``update_height_map()`` without arguments then returns what the last
        ``env.step`` wrote, with no kernel and no launch boundary of its own (9.4 us per step of BASELINE config 5).  After a reset, a rollout
        or a restored state it launches the ray kernel once, with the same centre and heading."""
        self.mj_model, self.mj_data = mj_model, mj_data   # mj_data: the batched env
        env = mj_data
        self.num_rows, self.num_cols, self.dist_x, self.dist_y = int(num_rows), int(num_cols), float(dist_x), float(dist_y)
        self.sensor_data_matrix = torch.zeros(env.num_envs, self.num_rows, self.num_cols, 1, 3, dtype=torch.float32, device=env.device)
        self.data = None
        self.follow_base = bool(follow_base)
        self._custom = None   # follow_base: hit points of update_height_map(center, yaw) calls go here - the kernel's tensor stays the base-following map
        if self.follow_base:
            import weakref
            other = env._hm_follow() if getattr(env, '_hm_follow', None) is not None else None
            if other is not None and other.follow_base:
                raise ValueError('this env already has a HeightMap(follow_base=True): the step kernel writes ONE base-following map per batch '
                                 '(close() the other one first, or build this one without follow_base)')
            _lib.check(_lib.lib().gq_batch_set_heightmap(env._hbatch, self.num_rows, self.num_cols, self.dist_x, self.dist_y,
                                                         self.sensor_data_matrix.data_ptr()), 'gq_batch_set_heightmap')
            env._hm_follow = weakref.ref(self)
            env._hm_fresh = False

    def create_sensor_matrix(self, center, yaw=0.0, out=None):
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
        out = self.sensor_data_matrix if out is None else out
        _lib.check(_lib.lib().gq_heightmap_strided(env._hbatch, c.data_ptr(), int(c.stride(0)), y.data_ptr(), int(y.stride(0)), self.num_rows, self.num_cols,
                                                   self.dist_x, self.dist_y, out.data_ptr(), stream), 'gq_heightmap_strided')
        return out

    def update_height_map(self, center=None, yaw=0.0):
        if center is None:
            if not self.follow_base:
                raise ValueError('update_height_map() without a centre needs HeightMap(..., follow_base=True)')
            env = self.mj_data
            mine = env._hm_follow is not None and env._hm_follow() is self
            if mine and env._hm_fresh and env._hm_version == env._qpos._version:   # the last env.step wrote it and nobody has touched the state since
                self.data = self.sensor_data_matrix
                return self.data
            q = env.qpos
            # the heading as the kernel takes it: (R10, R00) of the base rotation, normalised (csrc/gq_heightmap.h) - equal to the yaw of
            # base_ori_euler_xyz away from the gimbal pole
            w, x, y, z = q[:, 3].float(), q[:, 4].float(), q[:, 5].float(), q[:, 6].float()
            center, yaw = q[:, 0:3], torch.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
            self.data = self.create_sensor_matrix(center, yaw)
            if mine:
                env._hm_fresh, env._hm_version = True, env._qpos._version    # (until the state changes again)
            return self.data
        if self.follow_base:   # a custom centre on the base-following map: its own tensor, the kernel's stays what update_height_map() returns
            if self._custom is None:
                self._custom = torch.zeros_like(self.sensor_data_matrix)
            self.data = self.create_sensor_matrix(center, yaw, out=self._custom)
            return self.data
        self.data = self.create_sensor_matrix(center, yaw)
        return self.data

    def close(self):
        """Detach from the step kernel (``follow_base``): call before the tensor is dropped while the env lives on."""
        env = self.mj_data
        if self.follow_base and getattr(env, '_hbatch', None):
            if getattr(env, '_hm_follow', None) is not None and env._hm_follow() is self:   # only the registered map detaches the kernel's output slot
                _lib.check(_lib.lib().gq_batch_set_heightmap(env._hbatch, 0, 0, 0.0, 0.0, None), 'gq_batch_set_heightmap')
                env._hm_follow, env._hm_fresh = None, False
            self.follow_base = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
