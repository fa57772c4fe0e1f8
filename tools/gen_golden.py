"""Generate the tier-1 golden vectors (tests/golden/*.npz|json) by RUNNING the importable numpy/scipy layer of the
reference under stubs of its absent third-party imports.

Run only in the build container:   python tools/gen_golden.py [/root/reference]
Nothing of the reference is copied: the fixtures hold inputs (fake mjData fields) and the outputs the reference's own
`QuadrupedEnv._get_obs`, `_check_*`, `math_utils`, `configure_observation_space`, `get_robot_config` and
`generate_terrain('flat')` computed for them.
"""
import json
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np

REF = Path(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
OUT = Path(__file__).resolve().parents[1] / 'tests' / 'golden'
sys.path.insert(0, str(REF))

# ---- stubs ---------------------------------------------------------------------------------------------------
class _Any:
    def __getattr__(self, k): return self
    def __call__(self, *a, **k): return self

mj = types.ModuleType('mujoco')
mj.MjData = object; mj.MjModel = object
for n in ('mjtObj', 'mjtJoint', 'mjtGeom', 'mjtCatBit', 'mjtRndFlag'):
    setattr(mj, n, _Any())
mjv = types.ModuleType('mujoco.viewer'); mjv.Handle = object
mj.viewer = mjv
_CUR = {}
def mj_jac(m, d, jacp, jacr, point, body):
    jacp[:] = _CUR['jacp'][body]
    if jacr is not None: jacr[:] = 0
def mj_contactForce(m, d, id, result):
    result[:] = _CUR['cforce'][id]
def mj_fullM(m, d, dst):
    dst[:] = _CUR['M']
mj.mj_jac, mj.mj_contactForce, mj.mj_fullM = mj_jac, mj_contactForce, mj_fullM
mj.mj_id2name = lambda m, t, i: f'body{i}'
gym = types.ModuleType('gymnasium')
class Env: pass
gym.Env = Env
sp = types.ModuleType('gymnasium.spaces')
class Box:
    def __init__(self, shape=None, low=None, high=None, dtype=None):
        self.shape, self.low, self.high, self.dtype = tuple(shape), np.asarray(low), np.asarray(high), dtype
class Dict_:
    def __init__(self, d): self.spaces = dict(d)
    def keys(self): return self.spaces.keys()
    def __getitem__(self, k): return self.spaces[k]
sp.Box, sp.Dict = Box, Dict_
gym.spaces = sp
for name, mod in [('mujoco', mj), ('mujoco.viewer', mjv), ('gymnasium', gym), ('gymnasium.spaces', sp),
                  ('cv2', types.ModuleType('cv2')), ('noise', types.ModuleType('noise'))]:
    sys.modules[name] = mod

from gym_quadruped.quadruped_env import QuadrupedEnv  # noqa: E402
from gym_quadruped.robot_cfgs import get_robot_config  # noqa: E402
from gym_quadruped.utils.math_utils import angle_between_vectors  # noqa: E402
from gym_quadruped.utils.quadruped_utils import LegsAttr, configure_observation_space  # noqa: E402

OBS = [o for o in QuadrupedEnv.ALL_OBS if o not in ('kinetic_energy', 'work')]  # both raise NameError in the reference (B1)
rng = np.random.default_rng(123)
NV, NCASE = 18, 24
# fake model: geoms 0 floor, 1..4 feet (FL FR RL RR) on bodies 4,7,10,13; geoms 5,6 on bodies 3 (thigh) and 1 (base)
geom_bodyid = np.array([0, 4, 7, 10, 13, 3, 1, 4])
feet_geom = dict(FL=1, FR=2, RL=3, RR=4)
feet_body = dict(FL=4, FR=7, RL=10, RR=13)
cases = []
for case in range(NCASE):
    legs_order = [('FL', 'FR', 'RL', 'RR'), ('FR', 'FL', 'RR', 'RL')][case % 2]
    env = object.__new__(QuadrupedEnv)
    quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
    qpos = np.concatenate([rng.uniform(-3, 3, 3), quat, rng.uniform(-1, 1, 12)])
    qvel, qacc, ctrl = rng.normal(size=18), rng.normal(size=18) * 10, rng.normal(size=12) * 20
    geom_xpos = rng.uniform(-1, 1, (8, 3))
    jacp = {b: rng.normal(size=(3, NV)) for b in feet_body.values()}
    # contacts: random subset; (geom1, geom2) with the floor on either side
    contacts, cforce = [], []
    for g in rng.permutation([1, 2, 3, 4, 5, 6, 7])[: rng.integers(0, 6)]:
        fr = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        g1, g2 = (0, int(g)) if rng.random() < 0.5 else (int(g), 0)
        contacts.append(SimpleNamespace(geom1=g1, geom2=g2, frame=fr.flatten(), dist=-rng.random() * 0.01))
        cforce.append(np.concatenate([rng.normal(size=3) * 30, np.zeros(3)]))
    if case % 5 == 4:  # a robot-robot contact that must be ignored
        contacts.append(SimpleNamespace(geom1=5, geom2=6, frame=np.eye(3).flatten(), dist=-0.001)); cforce.append(np.ones(6))
    _CUR.update(jacp=jacp, cforce=cforce, M=np.eye(NV))
    env.mjData = SimpleNamespace(qpos=qpos, qvel=qvel, qacc=qacc, ctrl=ctrl, geom_xpos=geom_xpos, contact=contacts)
    env.mjModel = SimpleNamespace(nv=NV, geom_bodyid=geom_bodyid)
    env._feet_geom_id = LegsAttr(**feet_geom); env._feet_body_id = LegsAttr(**feet_body)
    env.legs_order = legs_order
    cmd = None if case == 0 else (rng.uniform(-1, 1, 3) * [1, 1, 0], float(rng.uniform(-1, 1)))
    env._ref_base_lin_vel_H = None if cmd is None else cmd[0]
    env._ref_base_ang_yaw_dot = None if cmd is None else cmd[1]
    lim = 2.0 if case % 3 == 0 else 1e4
    env.terrain_limits = (lim, -lim, lim, -lim)
    env.sensors = []
    env.state_obs_names = tuple(OBS)
    env.observation_space = configure_observation_space(
        SimpleNamespace(nq=19, nv=18, nu=12, jnt_range=np.zeros((13, 2)), actuator_ctrlrange=np.zeros((12, 2))), OBS)
    obs = env._get_obs()
    invalid, info = env._check_for_invalid_contacts()
    oob = env._check_out_of_terrain_bounds()
    cases.append(dict(
        legs_order=list(legs_order), qpos=qpos, qvel=qvel, qacc=qacc, ctrl=ctrl, geom_xpos=geom_xpos,
        jacp=np.stack([jacp[feet_body[l]] for l in ('FL', 'FR', 'RL', 'RR')]),
        contact_geom=np.array([[c.geom1, c.geom2] for c in contacts]).reshape(-1, 2),
        contact_frame=np.array([c.frame for c in contacts]).reshape(-1, 9),
        contact_force=np.array(cforce).reshape(-1, 6),
        cmd=np.zeros(4) if cmd is None else np.concatenate([cmd[0], [cmd[1]]]), cmd_none=cmd is None,
        terrain_limits=np.array(env.terrain_limits), invalid=bool(invalid), n_invalid=len(info), oob=bool(oob),
        obs={k: np.asarray(v, dtype=np.float64) for k, v in obs.items()}))

OUT.mkdir(parents=True, exist_ok=True)
flat = {}
for i, c in enumerate(cases):
    for k, v in c.items():
        if k == 'obs':
            for n, a in v.items(): flat[f'c{i}/obs/{n}'] = a
        elif k == 'legs_order':
            flat[f'c{i}/legs_order'] = np.array([['FL', 'FR', 'RL', 'RR'].index(x) for x in v])
        else:
            flat[f'c{i}/{k}'] = np.asarray(v)
flat['meta/geom_bodyid'] = geom_bodyid
flat['meta/feet_geom'] = np.array([feet_geom[l] for l in ('FL', 'FR', 'RL', 'RR')])
flat['meta/ncase'] = np.array(NCASE)
np.savez_compressed(OUT / 'obs_algebra.npz', **flat)

# ---- small known answers -------------------------------------------------------------------------------------
small = {}
small['angle_between_vectors'] = [[list(a), list(b), angle_between_vectors(a, b)] for a, b in
                                  [([1, 1, 0], [0, 0, 0]), ([-3, 2, 0], [0, 0, 0]), ([0.5, -7, 0], [0, 0, 0]), ([1, 0, 0], [0, 2, 0])]]
fake = SimpleNamespace(nq=19, nv=18, nu=12, jnt_range=np.zeros((13, 2)), actuator_ctrlrange=np.zeros((12, 2)))
small['obs_dims_all'] = {k: list(configure_observation_space(fake, QuadrupedEnv.ALL_OBS)[k].shape) for k in QuadrupedEnv.ALL_OBS}
small['all_obs'] = list(QuadrupedEnv.ALL_OBS)
small['default_obs'] = list(QuadrupedEnv._DEFAULT_OBS)
cfgs = {}
for n in ['mini_cheetah', 'Mini_Cheetah_v2', 'go1', 'go2', 'aliengo', 'b2', 'hyqreal1', 'hyqreal2', 'my_spot', 'pegasus', 'hyqreal', 'GO2', 'unknown']:
    try:
        c = get_robot_config(n)
        cfgs[n] = dict(mjcf_filename=c.mjcf_filename, hip_height=c.hip_height,
                       qpos0_js=None if c.qpos0_js is None else [float(x) for x in c.qpos0_js],
                       feet_geom_names=c.feet_geom_names, leg_joints=c.leg_joints)
    except ValueError as e:
        cfgs[n] = 'ValueError'
small['robot_cfgs'] = cfgs
from gym_quadruped.utils.mujoco import terrain  # noqa: E402
base = REF / 'gym_quadruped' / 'robot_model' / 'scene_flat.xml'
_, lim = terrain.generate_terrain(base, REF / 'gym_quadruped' / 'utils' / 'mujoco' / 'assets', 0.225, 'flat', seed=10)
small['flat_terrain_limits'] = list(lim)
(OUT / 'known_answers.json').write_text(json.dumps(small, indent=1))
print('wrote', OUT / 'obs_algebra.npz', (OUT / 'obs_algebra.npz').stat().st_size, 'bytes;', len(cases), 'cases')
