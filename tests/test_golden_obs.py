"""Tier-1 pin: the numpy observation oracle against golden vectors produced by the reference's own code
(tools/gen_golden.py), and the C oracle's observation assembly against the numpy one."""
import json
from pathlib import Path

import numpy as np
import pytest

from helpers import ALL_OBS, OBS_DIMS, marshalled, random_states
from oracle import obs_oracle
from oracle.oracle import Oracle

G = Path(__file__).parent / 'golden'


def _case(z, i):
    pre = f'c{i}/'
    d = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and '/obs/' not in k}
    obs = {k.split('/obs/')[1]: z[k] for k in z.files if k.startswith(pre + 'obs/')}
    d['geom_bodyid'], d['feet_geom'] = z['meta/geom_bodyid'], z['meta/feet_geom']
    d['cmd_none'] = bool(d['cmd_none'])
    return d, obs


def test_obs_oracle_matches_reference_golden():
    z = np.load(G / 'obs_algebra.npz')
    n = int(z['meta/ncase'])
    assert n >= 20
    for i in range(n):
        d, ref = _case(z, i)
        got, invalid, oob = obs_oracle.get_obs(d, list(ref.keys()), legs_order=d['legs_order'])
        for k, v in ref.items():
            np.testing.assert_allclose(got[k], v, rtol=1e-12, atol=1e-12, err_msg=f'case {i} obs {k}')
        assert invalid == bool(d['invalid']) and oob == bool(d['oob'])


def test_known_answers():
    ka = json.loads((G / 'known_answers.json').read_text())
    from gym_quadruped_amd.cabi import ALL_OBS as mine, OBS_DIMS as dims
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    from gym_quadruped_amd.terrain import generate_terrain
    from gym_quadruped_amd.utils.math_utils import angle_between_vectors
    assert ka['all_obs'] == list(mine) == list(QuadrupedEnv.ALL_OBS)
    assert tuple(ka['default_obs']) == QuadrupedEnv._DEFAULT_OBS
    assert [ka['obs_dims_all'][k][0] for k in mine] == list(dims[:31]) and sum(dims[:31]) == 227
    for a, b, ang in ka['angle_between_vectors']:
        assert abs(angle_between_vectors(a, b) - ang) < 1e-15
    for name, ref in ka['robot_cfgs'].items():
        if ref == 'ValueError':
            with pytest.raises(ValueError):
                get_robot_config(name)
            continue
        c = get_robot_config(name)
        assert c.mjcf_filename == ref['mjcf_filename'] and c.hip_height == ref['hip_height']
        assert (c.qpos0_js is None) == (ref['qpos0_js'] is None)
        if ref['qpos0_js'] is not None:
            np.testing.assert_allclose(np.asarray(c.qpos0_js, float), ref['qpos0_js'])
        assert c.feet_geom_names == ref['feet_geom_names'] and c.leg_joints == ref['leg_joints']
    _, lim = generate_terrain('flat', 0.225)
    assert list(lim) == ka['flat_terrain_limits']


def test_c_obs_assembly_matches_numpy_oracle():
    """gqo_get_obs (C, used for the cpu_baseline and as the GPU checker) == obs_oracle fed with the C oracle's state."""
    mm = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-10)
    md = mm.md
    o = Oracle(mm)
    rng = np.random.default_rng(5)
    qpos, qvel = random_states(md, 40, rng)
    feet_geom = [md.geom_names.index(n) for n in ('FL', 'FR', 'RL', 'RR')]
    for e in range(40):
        o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18), 0.0, -1.0)
        o.step(rng.normal(0, 20, 12))
        cmd = rng.uniform(-1, 1, 4) * [1, 1, 0, 1]
        lo = [(0, 1, 2, 3), (1, 0, 3, 2)][e % 2]
        got, term, inv = o.get_obs(ALL_OBS, cmd, lo)
        gx = o.geom_xpos
        # geom ids in the oracle are robot geoms; the floor is "geom -1" -> build a table with the floor at the end
        gb = np.concatenate([md.geom_bodyid, [0]])
        nfloor = len(md.geom_bodyid)
        cg = np.array([[nfloor, int(g)] for g in o.contact_geom]).reshape(-1, 2)
        d = dict(qpos=o.qpos, qvel=o.qvel, qacc=o.qacc, ctrl=o.ctrl, geom_xpos=gx,
                 jacp=np.stack([o.jac(gx[g], md.geom_bodyid[g])[0] for g in feet_geom]),
                 contact_geom=cg, contact_frame=o.contact_frame.reshape(-1, 9), contact_force=o.contact_force,
                 geom_bodyid=gb, feet_geom=np.array(feet_geom), cmd=cmd, cmd_none=False,
                 terrain_limits=np.array([1e4, -1e4, 1e4, -1e4]), M=o.M)
        ref, invalid, oob = obs_oracle.get_obs(d, ALL_OBS, lo)
        for k in ALL_OBS:
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f'{e} {k}')
        assert inv == invalid and term == (invalid or oob)
