"""ctypes mirror of ``include/gq.h`` (the C-ABI boundary) and the ModelDesc -> GqModelDesc marshaller.

The reference reaches its physics through the ``mujoco`` pybind API (SURVEY.md §8b lower boundary); this
module is the equivalent binding layer for ``libgq.so``: plain pointers and sizes, no torch types.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .mjcf import ModelDesc

GQ_NLEG = 4
GQ_ABI_VERSION = 500   # include/gq.h
# optional extra output rows of the step kernel (include/gq.h gq_batch_set_outputs)
GQ_DYN = dict(MC=0, MB=108, BIAS=144, XPOS=162, XMAT=201, FOOT=318, STRIDE=336)
GQ_CON_MAX, GQ_CON_REC = 12, 24
GQ_CON_STRIDE = 8 + GQ_CON_MAX * GQ_CON_REC

_I = C.POINTER(C.c_int32)
_D = C.POINTER(C.c_double)


class GqModelDesc(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32),
        ('nq', C.c_int32), ('nv', C.c_int32), ('nu', C.c_int32), ('nbody', C.c_int32), ('njnt', C.c_int32),
        ('ngeom', C.c_int32), ('ncloud', C.c_int32), ('nvert', C.c_int32),
        ('timestep', C.c_double), ('gravity', C.c_double * 3), ('cone', C.c_int32), ('impratio', C.c_double),
        ('integrator', C.c_int32),
        ('body_parentid', _I), ('body_pos', _D), ('body_quat', _D), ('body_ipos', _D), ('body_iquat', _D),
        ('body_mass', _D), ('body_inertia', _D), ('body_jntadr', _I), ('body_jntnum', _I), ('body_invweight0', _D),
        ('jnt_type', _I), ('jnt_bodyid', _I), ('jnt_qposadr', _I), ('jnt_dofadr', _I), ('jnt_pos', _D),
        ('jnt_axis', _D), ('jnt_limited', _I), ('jnt_range', _D), ('jnt_margin', _D), ('jnt_solref', _D),
        ('jnt_solimp', _D), ('jnt_actfrclimited', _I), ('jnt_actfrcrange', _D), ('qpos0', _D),
        ('dof_bodyid', _I), ('dof_jntid', _I), ('dof_parentid', _I), ('dof_damping', _D), ('dof_armature', _D),
        ('dof_frictionloss', _D), ('dof_solref', _D), ('dof_solimp', _D), ('dof_invweight0', _D),
        ('geom_bodyid', _I), ('geom_pos', _D), ('geom_quat', _D), ('geom_cloudid', _I), ('geom_friction', _D),
        ('geom_margin', _D), ('geom_gap', _D), ('geom_condim', _I), ('geom_priority', _I), ('geom_solref', _D),
        ('geom_solimp', _D), ('geom_solmix', _D), ('geom_rbound', _D),
        ('cloud_vertadr', _I), ('cloud_vertnum', _I), ('cloud_radius', _D), ('vert_pos', _D),
        ('actuator_trnid', _I), ('actuator_gear', _D), ('actuator_ctrllimited', _I), ('actuator_ctrlrange', _D),
        ('actuator_forcelimited', _I), ('actuator_forcerange', _D),
        ('floor_friction', C.c_double * 3), ('floor_margin', C.c_double), ('floor_gap', C.c_double),
        ('floor_solmix', C.c_double), ('floor_solref', C.c_double * 2), ('floor_solimp', C.c_double * 5),
        ('floor_condim', C.c_int32), ('floor_priority', C.c_int32),
        ('feet_geomid', C.c_int32 * GQ_NLEG), ('terrain_limits', C.c_double * 4), ('meaninertia', C.c_double),
        ('key_qpos', C.c_double * 19),
        ('solver', C.c_int32), ('iterations', C.c_int32), ('tolerance', C.c_double), ('noise_floor', C.c_double),
        ('nbox', C.c_int32), ('box_pos', _D), ('box_mat', _D), ('box_size', _D), ('box_friction', _D), ('box_margin', _D),
        ('box_gap', _D), ('box_solmix', _D), ('box_solref', _D), ('box_solimp', _D), ('box_condim', _I), ('box_priority', _I),
        ('hfield_nrow', C.c_int32), ('hfield_ncol', C.c_int32), ('hfield_data', C.POINTER(C.c_float)),
        ('hfield_size', C.c_double * 4), ('hfield_pos', C.c_double * 3),
        ('hfield_friction', C.c_double * 3), ('hfield_margin', C.c_double), ('hfield_gap', C.c_double),
        ('hfield_solmix', C.c_double), ('hfield_solref', C.c_double * 2), ('hfield_solimp', C.c_double * 5),
        ('hfield_condim', C.c_int32), ('hfield_priority', C.c_int32),
        ('nselfpair', C.c_int32), ('selfpair_geom1', _I), ('selfpair_geom2', _I), ('geom_capsule', _D), ('geom_type', _I),
    ]


class GqState(C.Structure):
    _fields_ = [('qpos', C.c_void_p), ('qvel', C.c_void_p), ('qacc', C.c_void_p), ('qacc_warmstart', C.c_void_p),
                ('qfrc_applied', C.c_void_p), ('time', C.c_void_p), ('friction', C.c_void_p), ('cmd', C.c_void_p)]


class GqResetCfg(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('random', C.c_int32), ('q_pos_amp', C.c_float), ('q_vel_amp', C.c_float),
                ('roll_sweep', C.c_float), ('pitch_sweep', C.c_float), ('hip_height', C.c_float),
                ('lin_vel_range', C.c_float * 2), ('ang_vel_range', C.c_float * 2), ('friction_range', C.c_float * 2),
                ('cmd_forward', C.c_int32), ('cmd_random', C.c_int32), ('cmd_rotate', C.c_int32), ('cmd_human', C.c_int32),
                ('env_id_offset', C.c_int32), ('autoreset_next_step', C.c_int32)]


class GqImuCfg(C.Structure):
    _fields_ = [('site_pos', C.c_double * 3), ('site_quat', C.c_double * 4), ('accel_noise', C.c_float),
                ('gyro_noise', C.c_float), ('accel_bias_rate', C.c_float), ('gyro_bias_rate', C.c_float), ('seed', C.c_uint64)]


class GqObsOut(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('terminated', C.c_void_p), ('truncated', C.c_void_p),
                ('invalid_contact', C.c_void_p), ('step_num', C.c_void_p), ('step_num_prev', C.c_void_p), ('contacts_dropped', C.c_void_p)]


class GqPolicyPd(C.Structure):
    _fields_ = [('kp', C.c_float * 12), ('kd', C.c_float * 12), ('q_des', C.c_float * 12), ('noise_sigma', C.c_float), ('noise_seed', C.c_uint64),
                ('noise_step0', C.c_int32)]


class GqMailboxView(C.Structure):
    _fields_ = [('action', C.c_void_p), ('steps_done', C.c_void_p), ('queue_items', C.c_void_p), ('queue_counters', C.c_void_p), ('status', C.c_void_p),
                ('n_queues', C.c_int32), ('queue_capacity', C.c_int32), ('counter_stride', C.c_int32), ('xcc_queue', C.c_int32 * 16)]


class GqResampleCfg(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('cmd_reset', C.c_int32), ('dist_reset', C.c_int32), ('dist_kind', C.c_int32 * 6),
                ('dist_range', (C.c_float * 2) * 6), ('env_id_offset', C.c_int32)]


# index into QuadrupedEnv.ALL_OBS (reference quadruped_env.py:35-66,81) == enum GqObsId
ALL_OBS = [
    'base_pos', 'base_lin_vel', 'base_lin_vel_err', 'base_lin_acc', 'base_ang_vel', 'base_ang_vel_err',
    'base_ori_euler_xyz', 'base_ori_quat_wxyz', 'base_ori_SO3', 'gravity_vector:base',
    'base_lin_vel:base', 'base_lin_vel_err:base', 'base_lin_acc:base', 'base_ang_vel:base', 'base_ang_vel_err:base',
    'qpos', 'qvel', 'tau_ctrl_setpoint', 'qpos_js', 'qvel_js', 'kinetic_energy', 'work',
    'feet_pos', 'feet_pos:base', 'feet_vel', 'feet_vel_rel', 'feet_vel:base', 'feet_vel_rel:base',
    'contact_state', 'contact_forces', 'contact_forces:base',
]
# sensor observables appended after ALL_OBS (enum GqObsId ids 31..36; reference sensors/imu.py:17-18)
IMU_OBS = ['imu_acc', 'imu_acc_noise', 'imu_acc_bias', 'imu_gyro', 'imu_gyro_noise', 'imu_gyro_bias']
OBS_NAMES = ALL_OBS + IMU_OBS
OBS_DIMS = [3, 3, 3, 3, 3, 3, 3, 4, 9, 3, 3, 3, 3, 3, 3, 19, 18, 12, 12, 12, 1, 1, 12, 12, 12, 12, 12, 12, 4, 12, 12] + [3] * 6
LEG_NAMES = ['FL', 'FR', 'RL', 'RR']

SOLVER_PGS, SOLVER_NEWTON = 0, 1


class MarshalledModel:
    """Owns the numpy buffers a GqModelDesc points into (keep alive for as long as the struct is in use)."""

    def __init__(self, md: ModelDesc, *, qpos0=None, feet_geom_names=None, terrain_limits=(1e4, -1e4, 1e4, -1e4),
                 timestep=None, solver=SOLVER_PGS, iterations=100, tolerance=1e-8, floor=None,
                 noise_floor=0.0, boxes=None, hfield=None, self_collision=None):
        self.md = md
        self._keep = []
        d = GqModelDesc()
        d.struct_size = C.sizeof(GqModelDesc)
        nvert = int(md.vert_pos.shape[0])
        for k in ('nq', 'nv', 'nu', 'nbody', 'njnt', 'ngeom'):
            setattr(d, k, int(getattr(md, k)))
        d.ncloud, d.nvert = int(len(md.cloud_vertnum)), nvert
        d.timestep = float(md.timestep if timestep is None else timestep)
        d.gravity = (C.c_double * 3)(*md.gravity)
        d.cone, d.impratio, d.integrator = int(md.cone), float(md.impratio), int(md.integrator)
        q0 = np.array(md.qpos0 if qpos0 is None else qpos0, dtype=np.float64)
        boxes = list(boxes or [])
        from scipy.spatial.transform import Rotation as _Rot
        box_arrays = dict(
            box_pos=[b['pos'] for b in boxes], box_size=[b['size'] for b in boxes],
            box_mat=[_Rot.from_quat(np.asarray(b['quat'], float), scalar_first=True).as_matrix().ravel() for b in boxes],
            box_friction=[b['friction'] for b in boxes], box_margin=[b['margin'] for b in boxes], box_gap=[b['gap'] for b in boxes],
            box_solmix=[b['solmix'] for b in boxes], box_solref=[b['solref'] for b in boxes], box_solimp=[b['solimp'] for b in boxes],
            box_condim=[b['condim'] for b in boxes], box_priority=[b['priority'] for b in boxes])
        d.nbox = len(boxes)
        self.boxes = boxes
        from .selfcol import geom_capsules, self_pairs
        if self_collision is None:   # MuJoCo's behaviour (robot geoms collide with each other) wherever the solver supports it
            self_collision = int(solver) == SOLVER_NEWTON
        pairs = self_pairs(md) if self_collision else np.zeros((0, 2), np.int32)
        box_arrays.update(selfpair_geom1=pairs[:, 0], selfpair_geom2=pairs[:, 1], geom_capsule=geom_capsules(md))
        d.nselfpair = int(len(pairs))
        self.self_pairs = pairs
        for name, ctype in GqModelDesc._fields_:
            if ctype in (_I, _D):
                src = q0 if name == 'qpos0' else (box_arrays[name] if name in box_arrays else getattr(md, name))
                arr = np.ascontiguousarray(src, dtype=np.int32 if ctype is _I else np.float64)
                if arr.size == 0:
                    arr = np.zeros(1, dtype=arr.dtype)
                self._keep.append(arr)
                setattr(d, name, arr.ctypes.data_as(ctype))
        fl = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                  solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
        fl.update(floor or {})
        d.floor_friction = (C.c_double * 3)(*fl['friction'])
        d.floor_margin, d.floor_gap, d.floor_solmix = fl['margin'], fl['gap'], fl['solmix']
        d.floor_solref = (C.c_double * 2)(*fl['solref'])
        d.floor_solimp = (C.c_double * 5)(*fl['solimp'])
        d.floor_condim, d.floor_priority = fl['condim'], fl['priority']
        names = feet_geom_names or {k: k for k in LEG_NAMES}
        d.feet_geomid = (C.c_int32 * 4)(*[md.geom_names.index(names[k]) for k in LEG_NAMES])
        d.terrain_limits = (C.c_double * 4)(*terrain_limits)
        d.meaninertia = float(md.meaninertia)
        kq = md.key_qpos[0] if len(md.key_qpos) else q0
        d.key_qpos = (C.c_double * 19)(*[float(v) for v in kq])
        d.solver, d.iterations, d.tolerance = int(solver), int(iterations), float(tolerance)
        d.noise_floor = float(noise_floor)
        # height field: dict(data=[nrow][ncol] in [0, 1], size=(rx, ry, elevation, base), pos=(x, y, z), + geom defaults)
        self.hfield = hfield
        if hfield is not None:
            data = np.ascontiguousarray(hfield['data'], dtype=np.float32)
            if data.ndim != 2:
                raise ValueError('hfield data must be a 2-D array [nrow][ncol]')
            self._keep.append(data)
            d.hfield_nrow, d.hfield_ncol = int(data.shape[0]), int(data.shape[1])
            d.hfield_data = data.ctypes.data_as(C.POINTER(C.c_float))
            d.hfield_size = (C.c_double * 4)(*[float(v) for v in hfield['size']])
            d.hfield_pos = (C.c_double * 3)(*[float(v) for v in hfield.get('pos', (0.0, 0.0, 0.0))])
            hg = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                      solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
            hg.update({k: v for k, v in hfield.items() if k in hg})
            d.hfield_friction = (C.c_double * 3)(*hg['friction'])
            d.hfield_margin, d.hfield_gap, d.hfield_solmix = hg['margin'], hg['gap'], hg['solmix']
            d.hfield_solref = (C.c_double * 2)(*hg['solref'])
            d.hfield_solimp = (C.c_double * 5)(*hg['solimp'])
            d.hfield_condim, d.hfield_priority = hg['condim'], hg['priority']
        self.desc = d


def obs_ids_from_names(names):
    ids = []
    for n in names:
        if n not in OBS_NAMES:
            raise ValueError(f'Invalid observation name: {n}, available obs: {ALL_OBS}')
        ids.append(OBS_NAMES.index(n))
    return ids
