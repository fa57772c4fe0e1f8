"""numpy restatement of the reference's observation / termination assembly (TEST INFRASTRUCTURE).

Follows ``gym_quadruped/quadruped_env.py``: ``_get_obs`` :1146-1226, ``base_configuration`` :961-970,
``heading_orientation_SO3`` :989-997, ``target_base_vel`` :488-499, ``base_*`` getters :501-541, ``feet_pos`` :597-629,
``feet_vel`` :631-679, ``feet_contact_state`` :799-872, ``gravity_vector`` :1007-1016, ``_check_for_invalid_contacts``
:1228-1248, ``_check_out_of_terrain_bounds`` :1250-1257 - including the quirks B3-B5 of SURVEY.md Appendix C.
PINNED by tests/golden/obs_algebra.npz, which holds outputs of the reference's own code for the same inputs
(tools/gen_golden.py).  ``kinetic_energy`` / ``work`` raise NameError in the reference (B1); the evident intent
(1/2 v'Mv and (M qacc).v) is implemented and is therefore NOT covered by the golden vectors.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

LEGS = ['FL', 'FR', 'RL', 'RR']


def get_obs(d: dict, obs_names, legs_order=(0, 1, 2, 3)):
    """d: qpos[19] qvel[18] qacc[18] ctrl[12] geom_xpos[ng,3] jacp[4,3,nv] (FL FR RL RR, world) contact_geom[k,2]
    contact_frame[k,9] contact_force[k,6] geom_bodyid[ng] feet_geom[4] cmd[4] cmd_none terrain_limits[4] M[nv,nv]."""
    qpos, qvel, qacc = d['qpos'], d['qvel'], d['qacc']
    R = Rotation.from_quat(np.roll(qpos[3:7], -1)).as_matrix()
    euler = Rotation.from_quat(np.roll(qpos[3:7], -1)).as_euler('xyz')
    Rh = Rotation.from_euler('xyz', Rotation.from_matrix(R).as_euler('xyz') * [0, 0, 1]).as_matrix()
    if d.get('cmd_none', False):
        tl_w, ta_w = np.zeros(3), np.zeros(3)
        tl_b, ta_b = np.zeros(3), np.zeros(3)  # target_base_vel returns zeros for BOTH frames when unset (:490-491)
    else:
        tl_w, ta_w = Rh @ d['cmd'][:3], np.array([0.0, 0.0, d['cmd'][3]])
        tl_b, ta_b = R.T @ tl_w, R.T @ ta_w
    gb, fg = np.asarray(d['geom_bodyid']), np.asarray(d['feet_geom'])
    fbody = gb[fg]
    fpos = d['geom_xpos'][fg]  # (4,3) FL FR RL RR
    fvel = np.einsum('lij,j->li', d['jacp'], qvel)
    fvel_rel = fvel - qvel[0:3] - np.cross(qvel[3:6], fpos - qpos[0:3])
    cstate, cf = np.zeros(4), np.zeros((4, 3))
    invalid = False
    for k in range(len(d['contact_geom'])):
        b1, b2 = gb[d['contact_geom'][k, 0]], gb[d['contact_geom'][k, 1]]
        if 0 in (b1, b2):
            other = b2 if b1 == 0 else b1
            if other in fbody:
                leg = int(np.where(fbody == other)[0][0])
                cstate[leg] = 1
                cf[leg] += d['contact_frame'][k].reshape(3, 3).T @ d['contact_force'][k][:3]
            else:
                invalid = True
    lo = list(legs_order)
    leg3 = lambda a: np.concatenate([a[i] for i in lo])
    out = {}
    for name in obs_names:
        base = name.endswith('base')
        if name == 'qpos': v = qpos.copy()
        elif name == 'qvel': v = qvel.copy()
        elif name == 'tau_ctrl_setpoint': v = np.array(d['ctrl'])
        elif name == 'qpos_js': v = qpos[7:].copy()
        elif name == 'qvel_js': v = qvel[6:].copy()
        elif name == 'base_pos': v = qpos[0:3].copy()
        elif 'base_lin_vel_err' in name: v = (tl_b - R.T @ qvel[0:3]) if base else (tl_w - qvel[0:3])
        elif 'base_lin_vel' in name: v = R.T @ qvel[0:3] if base else qvel[0:3].copy()
        elif 'base_lin_acc' in name: v = R.T @ qacc[0:3] if base else qacc[0:3].copy()
        elif 'base_ang_vel_err' in name: v = (ta_b - qvel[3:6]) if base else (ta_w - R @ qvel[3:6])
        elif 'base_ang_vel' in name: v = qvel[3:6].copy() if base else R @ qvel[3:6]
        elif name == 'base_ori_euler_xyz': v = euler
        elif name == 'base_ori_quat_wxyz': v = qpos[3:7].copy()
        elif name == 'base_ori_SO3': v = R.flatten()
        elif 'feet_pos' in name: v = leg3((fpos - qpos[0:3]) @ R) if base else leg3(fpos)
        elif 'feet_vel_rel' in name: v = leg3(fvel_rel @ R) if base else leg3(fvel_rel)
        elif 'feet_vel' in name: v = leg3(fvel @ R) if base else leg3(fvel)
        elif name == 'contact_state': v = cstate.copy()  # always FL FR RL RR (B5)
        elif 'contact_forces' in name: v = leg3(cf @ R) if base else leg3(cf)
        elif name == 'gravity_vector:base': v = R.T @ np.array([0.0, 0.0, -1.0])
        elif name == 'kinetic_energy': v = np.atleast_1d(0.5 * qvel @ d['M'] @ qvel)
        elif name == 'work': v = np.atleast_1d((d['M'] @ qacc) @ qvel)
        else: raise ValueError(f'Invalid observation name: {name}')
        out[name] = np.asarray(v, dtype=np.float64)
    tl = d['terrain_limits']
    oob = bool(qpos[0] > tl[0] or qpos[0] < tl[1] or qpos[1] > tl[2] or qpos[1] < tl[3])
    return out, invalid, oob
