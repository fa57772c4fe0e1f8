"""One cheap attempt to pin the oracle's mj_step against the real MuJoCo (SURVEY.md §8c tier 2, §8d "opportunistic").

Run on the GPU box (`gpurun -- python tools/mujoco_probe.py`): probes `import mujoco`, an offline `pip install mujoco`,
and the disk for a libmujoco; writes the evidence to gpurun_out/mujoco_probe.log.  Only if MuJoCo turns out to be
importable does it go on to generate fixtures from THIS repo's exported MJCF (tools/export_mjcf.py; the reference's
Python never travels): for every registry robot, 256 random contact-rich states -> mj_forward internals (qacc, efc_J /
aref / R / force, contact list) and the mj_step successor state, saved as gpurun_out/mujoco_fixtures/mujoco_<robot>.npz
(to be committed under tests/golden/ and checked by tests/test_oracle_vs_mujoco.py).
"""
from __future__ import annotations

import glob
import importlib
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
OUT = ROOT / 'gpurun_out'
OUT.mkdir(exist_ok=True)
log = open(OUT / 'mujoco_probe.log', 'w')


def say(*a):
    print(*a); print(*a, file=log); log.flush()


def sh(cmd, timeout=60):
    try:
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=timeout)
        return r.returncode, (r.stdout + r.stderr).strip()[-1500:]
    except subprocess.TimeoutExpired:
        return -9, f'timeout after {timeout}s'


def have_mujoco():
    try:
        m = importlib.import_module('mujoco')
        say('import mujoco: OK, version', getattr(m, '__version__', '?'), 'at', os.path.dirname(m.__file__))
        return m
    except Exception as e:   # noqa: BLE001
        say(f'import mujoco: FAILED: {type(e).__name__}: {e}')
        return None


say('== mujoco probe on', os.uname().nodename, '| python', sys.version.split()[0])
mj = have_mujoco()
if mj is None:
    for cmd in ('pip download mujoco --no-deps -d /tmp/mjwheel', 'pip install mujoco', 'pip index versions mujoco'):
        rc, out = sh(cmd, timeout=45)
        say(f'$ {cmd}\n  rc={rc}\n  ' + out.replace('\n', '\n  '))
    hits = [p for pat in ('/usr/**/libmujoco*', '/opt/**/libmujoco*', '/root/**/libmujoco*', '/usr/**/mujoco*.whl', '/opt/**/mujoco*.whl',
                          '/root/**/mujoco*.whl') for p in glob.glob(pat, recursive=True)][:20]
    say('libmujoco / wheels on disk:', hits or 'none')
    rc, out = sh('pip list 2>/dev/null | grep -i -E "mujoco|dm_control|gymnasium|mjx" || true')
    say('pip list (mujoco|dm_control|gymnasium|mjx):', out or 'none')
    importlib.invalidate_caches()
    mj = have_mujoco()
if mj is None:
    say('RESULT: MuJoCo is not available on the GPU box; the mj_step core of the oracle stays "parity unpinned".')
    sys.exit(0)

# ---------------------------------------------------------------- fixtures from the exported MJCF
from helpers import random_states  # noqa: E402
from tools.export_mjcf import ROBOTS, export  # noqa: E402
from gym_quadruped_amd.mjcf import load_compiled  # noqa: E402
from gym_quadruped_amd.robot_cfgs import get_robot_config  # noqa: E402

fx = OUT / 'mujoco_fixtures'
fx.mkdir(exist_ok=True)
for robot in ROBOTS:
    xml = export(robot)
    model = mj.MjModel.from_xml_string(xml)
    data = mj.MjData(model)
    md = load_compiled(Path(get_robot_config(robot).mjcf_filename).stem)
    rng = np.random.default_rng(21)
    hip = get_robot_config(robot).hip_height
    n = 256
    qpos, qvel = random_states(md, n, rng, z_range=(0.6 * hip, 1.6 * hip))
    ctrl = rng.normal(0, 1, (n, 12)) * 40
    rec = dict(qpos=qpos, qvel=qvel, ctrl=ctrl, qacc=np.zeros((n, 18)), qpos_next=np.zeros((n, 19)), qvel_next=np.zeros((n, 18)),
               nefc=np.zeros(n, int), ncon=np.zeros(n, int), qfrc_bias=np.zeros((n, 18)), M=np.zeros((n, 18, 18)))
    efc, con = [], []
    for e in range(n):
        mj.mj_resetData(model, data)
        data.qpos[:], data.qvel[:], data.ctrl[:] = qpos[e], qvel[e], ctrl[e]
        mj.mj_forward(model, data)
        rec['qacc'][e], rec['nefc'][e], rec['ncon'][e], rec['qfrc_bias'][e] = data.qacc, data.nefc, data.ncon, data.qfrc_bias
        mj.mj_fullM(model, rec['M'][e], data.qM)
        J = np.array(data.efc_J).reshape(data.nefc, -1) if data.nefc else np.zeros((0, 18))
        efc.append(dict(J=J, aref=np.array(data.efc_aref), R=np.array(data.efc_R), force=np.array(data.efc_force), type=np.array(data.efc_type)))
        con.append(np.array([[c.geom1, c.geom2, c.dist, *c.pos, *c.frame, c.dim, c.mu, *c.friction] for c in data.contact[:data.ncon]]).reshape(data.ncon, -1))
        mj.mj_resetData(model, data)
        data.qpos[:], data.qvel[:], data.ctrl[:] = qpos[e], qvel[e], ctrl[e]
        mj.mj_step(model, data)
        rec['qpos_next'][e], rec['qvel_next'][e] = data.qpos, data.qvel
    np.savez_compressed(fx / f'mujoco_{robot}.npz', mujoco_version=str(mj.__version__), xml=xml, **rec,
                        efc=np.array(efc, dtype=object), contacts=np.array(con, dtype=object), allow_pickle=True)
    say(f'{robot}: 256 states, mean ncon {rec["ncon"].mean():.2f}, mean nefc {rec["nefc"].mean():.1f} -> {fx / f"mujoco_{robot}.npz"}')
say('RESULT: fixtures written; copy gpurun_out/mujoco_fixtures/*.npz to tests/golden/')
