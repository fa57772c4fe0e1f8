"""Host logic + boundary: model compiler, registry, spaces, C-ABI export table, error behaviour without a GPU."""
import ctypes
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
REF_XML = Path('/root/reference/gym_quadruped/robot_model')


def test_compiled_tables_load_and_are_consistent():
    from gym_quadruped_amd.mjcf import load_compiled
    masses = dict(mini_cheetah=12.473, aliengo=24.638, go2=15.206, hyqreal1=107.573, hyqreal2=126.694, go1=12.743,
                  b2=83.498, spot=50.340)
    for r, mass in masses.items():
        md = load_compiled(r)
        assert (md.nq, md.nv, md.nu, md.nbody, md.njnt) == (19, 18, 12, 14, 13)
        assert abs(md.total_mass - mass) < 2e-3
        assert np.all(md.dof_invweight0 > 0) and np.all(md.body_invweight0[1:] > 0)
        assert md.body_parentid[1] == 0 and list(md.dof_parentid[:7]) == [-1, 0, 1, 2, 3, 4, 5]
        for leg, name in enumerate(['FL', 'FR', 'RL', 'RR']):
            g = md.geom_names.index(name)
            assert md.cloud_vertnum[md.geom_cloudid[g]] == 1 and md.cloud_radius[md.geom_cloudid[g]] > 0
    mc = load_compiled('mini_cheetah')
    # joints take the body's childclass, NOT the motor classes: no ranges -> no joint limits for mini_cheetah
    assert mc.jnt_limited.sum() == 0 and np.all(mc.dof_frictionloss[6:] == 0.2) and np.all(mc.dof_damping[6:] == 0.2)
    np.testing.assert_allclose(mc.actuator_ctrlrange[:3], [[-23.7, 23.7], [-23.7, 23.7], [-45.43, 45.43]])
    assert load_compiled('go2').cone == 1 and load_compiled('go2').impratio == 100 and load_compiled('spot').integrator == 3


@pytest.mark.skipif(not REF_XML.exists(), reason='reference checkout not present (GPU box)')
def test_compiler_reproduces_committed_tables():
    import dataclasses
    from gym_quadruped_amd.mjcf import compile_mjcf, load_compiled
    for r in ['mini_cheetah', 'aliengo']:
        a, b = compile_mjcf(REF_XML / r / f'{r}.xml'), load_compiled(r)
        for f in dataclasses.fields(a):
            x, y = getattr(a, f.name), getattr(b, f.name)
            if isinstance(x, np.ndarray):
                np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-14, err_msg=f'{r}.{f.name}')
            elif isinstance(x, float):
                assert abs(x - y) < 1e-12
            else:
                assert x == y, f.name


def test_mjcf_defaults_childclass_and_fromto(tmp_path):
    from gym_quadruped_amd.mjcf import compile_mjcf
    legs = ''.join(f'''<body name="L{i}_hip" pos="0.1 0 0"><inertial pos="0 0 0" mass="1" diaginertia="1e-3 1e-3 1e-3"/>
      <joint name="j{i}a" class="lim"/><body name="L{i}_thigh"><inertial pos="0 0 -0.1" mass="1" diaginertia="1e-3 1e-3 1e-3"/><joint name="j{i}b"/>
      <body name="L{i}_calf" pos="0 0 -0.2"><inertial pos="0 0 -0.1" mass="0.5" diaginertia="1e-3 1e-3 1e-3"/><joint name="j{i}c" axis="1 0 0"/>
      <geom name="{n}" size="0.02" pos="0 0 -0.2"/><geom type="capsule" fromto="0 0 0 0 0 -0.2" size="0.01"/></body></body></body>'''
                   for i, n in enumerate(['FL', 'FR', 'RL', 'RR']))
    xml = f'''<mujoco model="toy"><compiler angle="radian"/><default><default class="r"><joint axis="0 1 0" damping="0.3"/>
      <geom friction="0.7"/><default class="lim"><joint range="-1 1" armature="0.05"/></default></default></default>
      <worldbody><body name="base" pos="0 0 0.5" childclass="r"><inertial pos="0 0 0" mass="5" diaginertia="0.1 0.1 0.1"/><freejoint/>{legs}</body></worldbody>
      <actuator>{''.join(f'<motor name="m{i}{c}" joint="j{i}{c}" ctrlrange="-5 5"/>' for i in range(4) for c in 'abc')}</actuator></mujoco>'''
    p = tmp_path / 'toy.xml'; p.write_text(xml)
    md = compile_mjcf(p)
    assert md.nv == 18 and md.jnt_limited.tolist() == [0] + [1, 0, 0] * 4
    assert md.dof_damping[6] == 0.3 and md.dof_armature[6] == 0.05 and md.dof_armature[7] == 0
    np.testing.assert_allclose(md.jnt_axis[3], [1, 0, 0]); np.testing.assert_allclose(md.jnt_axis[2], [0, 1, 0])
    g = [i for i, t in enumerate(md.geom_type) if t == 3][0]   # capsule from fromto
    np.testing.assert_allclose(md.geom_pos[g], [0, 0, -0.1]); np.testing.assert_allclose(md.geom_size[g][:2], [0.01, 0.1])
    assert md.geom_friction[g][0] == 0.7 and md.actuator_ctrllimited.all()


def test_cabi_library_exports_every_declared_symbol():
    from gym_quadruped_amd import _lib
    assert _lib.LIB_PATH.exists(), 'build the HIP extension first (__graft_entry__.build())'
    header = (ROOT / 'include' / 'gq.h').read_text()
    declared = set(re.findall(r'\b(gq_[a-z_]+)\s*\(', header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    syms = subprocess.run(['nm', '-D', '--defined-only', str(_lib.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    for s in declared:
        assert f' T {s}\n' in syms, f'{s} not exported'


def test_ctypes_struct_layout_matches_header():
    """sizeof of the ctypes mirrors must equal what the C compiler sees (guards against field drift) - every struct that
    crosses the boundary, in the order of gq_struct_sizes."""
    from gym_quadruped_amd.cabi import GqImuCfg, GqMailboxView, GqModelDesc, GqObsOut, GqPolicyPd, GqResampleCfg, GqResetCfg, GqState
    names = ['GqModelDesc', 'GqState', 'GqObsOut', 'GqResetCfg', 'GqResampleCfg', 'GqImuCfg', 'GqPolicyPd', 'GqMailboxView']
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "gq.h"\nint main(){printf("' + '%zu ' * len(names) + '%zu\\n",' + ','.join(f'sizeof({n})' for n in names)
           + ',offsetof(GqPolicyPd, noise_seed));return 0;}')
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / 'a.c').write_text(src)
        subprocess.run(['gcc', '-I', str(ROOT / 'include'), str(Path(d) / 'a.c'), '-o', str(Path(d) / 'a')], check=True)
        out = subprocess.run([str(Path(d) / 'a')], check=True, capture_output=True, text=True).stdout.split()
    mirror = [GqModelDesc, GqState, GqObsOut, GqResetCfg, GqResampleCfg, GqImuCfg, GqPolicyPd, GqMailboxView]
    assert [int(x) for x in out[:-1]] == [ctypes.sizeof(t) for t in mirror]
    assert int(out[-1]) == GqPolicyPd.noise_seed.offset   # the 8-byte-aligned word after 37 floats
    # ... and the built library reports the same eight numbers (what _lib.lib() refuses a stale build with)
    from gym_quadruped_amd import _lib
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    sizes = (ctypes.c_int32 * 8)()
    assert L.gq_struct_sizes(sizes) == 0 and list(sizes) == [ctypes.sizeof(t) for t in mirror]


def integration_stub_text():
    """The first fenced python block of INTEGRATION.md section 2: the binding a reference maintainer would write."""
    text = (ROOT / 'INTEGRATION.md').read_text()
    a = text.index('```python\n', text.index('## 2. The stub')) + len('```python\n')
    return text[a:text.index('\n```', a)]


class _TypeCheckedLib:
    """Stands in for ctypes.CDLL on a machine without a GPU: calls that need no device go to the real library, every
    other call is checked against the argument table gym_quadruped_amd/_lib.py declares for it (count and convertibility
    of each argument - what ctypes itself would refuse) and answered with GQ_OK."""

    def __init__(self, real):
        self._real = real
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in ('gq_version', 'gq_struct_sizes', 'gq_last_error', 'gq_obs_dim'):
            return fn

        def call(*args):
            assert fn.argtypes is not None, f'{name}: no argument table in _lib.py'
            assert len(args) == len(fn.argtypes), f'{name}: {len(args)} arguments, the ABI takes {len(fn.argtypes)}'
            for k, (a, t) in enumerate(zip(args, fn.argtypes)):
                try:
                    t.from_param(a)
                except (TypeError, ctypes.ArgumentError) as e:
                    raise AssertionError(f'{name}: argument {k} ({type(a).__name__}) is not a {t.__name__}: {e}')
                if isinstance(a, ctypes.Structure):
                    assert type(a) is t, f'{name}: argument {k} is a {type(a).__name__}, the ABI takes {t.__name__} by value'
            self.calls.append(name)
            if name == 'gq_batch_obs_dim':
                return 19 + 18 + 12 + 12 + 12
            return 0
        return call


def test_integration_stub_text_matches_the_abi(monkeypatch):
    """INTEGRATION.md's stub, executed as written (device = cpu tensors; device calls type-checked, not run): the ABI
    version it asserts, the struct sizes it compares, every struct constructor and every call's argument list."""
    from gym_quadruped_amd import _lib
    real = _lib.lib()
    proxy = _TypeCheckedLib(real)
    monkeypatch.setattr(ctypes, 'CDLL', lambda path, *a, **k: proxy)
    monkeypatch.chdir(ROOT)
    text = integration_stub_text()
    assert "N, dev = 64, 'cuda'" in text
    exec(compile(text.replace("N, dev = 64, 'cuda'", "N, dev = 64, 'cpu'"), 'INTEGRATION.md', 'exec'), {'__name__': 'integration_stub'})
    assert proxy.calls == ['gq_model_create', 'gq_batch_create', 'gq_batch_obs_dim', 'gq_reset', 'gq_step', 'gq_step', 'gq_batch_destroy', 'gq_model_destroy']


def test_env_fails_loudly_without_gpu_or_library():
    import torch
    from gym_quadruped_amd import _lib
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    with pytest.raises(_lib.GqError):
        QuadrupedEnv('mini_cheetah', device='cpu')
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            QuadrupedEnv('mini_cheetah', num_envs=2)   # no silent CPU fallback
    with pytest.raises(ValueError):
        QuadrupedEnv('hyqreal', device='cpu')
    with pytest.raises(_lib.GqError):
        QuadrupedEnv('mini_cheetah', scene='perlin', device='cpu')


def test_pnoise2_restatement_reproduces_the_reference_image():
    """noise.pnoise2 (third party, absent here) restated in terrain.py: the image the reference ships - an output of its
    add_perlin_heightfield with default arguments (tools/gen_golden_perlin.py) - is reproduced bit for bit."""
    from gym_quadruped_amd.terrain import perlin_image, pnoise2
    g = np.load(Path(__file__).parent / 'golden' / 'perlin_default.npz')
    img = perlin_image(128, 128, smooth=float(g['smooth']), perlin_octaves=int(g['octaves']), perlin_persistence=float(g['persistence']),
                       perlin_lacunarity=float(g['lacunarity']))
    assert img.dtype == np.uint8 and np.array_equal(img, g['image'])
    # lattice points are zeros of every octave; one octave is bounded by sqrt(2) / 2 in 2-D
    assert np.all(pnoise2(np.arange(5.0), np.arange(5.0), octaves=3) == 0.0)
    xs = np.linspace(0.0, 7.3, 400)
    assert np.abs(pnoise2(xs, xs[::-1] * 1.37)).max() <= 0.7072


def test_legsattr_spaces_and_joint_maps():
    from gym_quadruped_amd.mjcf import load_compiled
    from gym_quadruped_amd.utils.quadruped_utils import LegsAttr, configure_observation_space, extract_mj_joint_info
    la = LegsAttr(FR=np.array([1.0]), FL=np.array([2.0]), RR=np.array([3.0]), RL=np.array([4.0]))
    assert [float(x[0]) for x in la.to_list()] == [2, 1, 4, 3] and [float(x[0]) for x in la.to_list(order=['RR', 'FL'])] == [3, 2]
    assert float((la + la).FL[0]) == 4 and float((la / 2).RR[0]) == 1.5 and la['RL'][0] == 4
    with pytest.raises(AssertionError):
        la['XX']
    md = load_compiled('aliengo')
    info = extract_mj_joint_info(md)
    assert list(info['FL_hip_joint'].qpos_idx) == [7] and list(info['FL_hip_joint'].qvel_idx) == [6] and info['FL_hip_joint'].tau_idx == (0,)
    assert list(info[md.jnt_names[0]].qpos_idx) == list(range(7))
    from gym_quadruped_amd.cabi import ALL_OBS
    sp = configure_observation_space(md, ALL_OBS)
    assert sum(sp[k].shape[0] for k in ALL_OBS) == 227 and abs(float(sp['qpos'].low[7]) - md.jnt_range[1, 0]) < 1e-6
    with pytest.raises(ValueError):
        configure_observation_space(md, ['nope'])


def test_dev_model_lowering_rejects_unsupported_models():
    """gq_build_dev_model (host part of libgq) through the emulator library: error text, not a crash."""
    from helpers import emu_step, marshalled
    from gym_quadruped_amd.mjcf import load_compiled
    mm = marshalled('mini_cheetah')
    mm.desc.cone = 1
    q = np.tile(mm.md.key_qpos[0], (1, 1))
    with pytest.raises(RuntimeError, match='elliptic'):
        emu_step(mm, np.zeros((1, 12)), q, np.zeros((1, 18)))


def test_reference_getter_surface_is_present():
    """Every public getter of the reference class (quadruped_env.py:488-1016) exists on the batched class (callable on a
    GPU box only); render / ghost / key-callback are viewer code and out of scope."""
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    for name in ('target_base_vel', 'base_lin_vel', 'base_lin_vel_err', 'base_ang_vel_err', 'base_ang_vel', 'base_lin_acc',
                 'get_base_inertia', 'hip_positions', 'feet_pos', 'feet_vel', 'feet_jacobians', 'feet_jacobians_dot', 'feet_contact_state', 'close',
                 'legs_mass_matrix', 'legs_qfrc_bias', 'legs_qfrc_passive', 'com', 'kinetic_energy', 'work',
                 'base_configuration', 'joint_space_state', 'base_pos', 'base_ori_euler_xyz', 'heading_orientation_SO3',
                 'torque_ctrl_setpoint', 'gravity_vector', 'simulation_dt', 'simulation_time', 'robot_model',
                 'get_hyperparameters', 'step', 'reset'):
        assert hasattr(QuadrupedEnv, name), name


def test_procedural_box_scenes_match_the_reference_generator():
    """random_boxes / random_pyramids: same numpy draws in the same order as the reference's add_world_of_boxes /
    add_world_of_pyramid -> the same boxes and spawn limits (golden: the reference's own generate_terrain, seed 10)."""
    import json
    from pathlib import Path
    from gym_quadruped_amd.terrain import generate_terrain
    gold = json.loads((Path(__file__).parent / 'golden' / 'terrain_boxes.json').read_text())
    state = np.random.get_state()[1][:5].copy()
    for key, ref in gold.items():
        name, hip = key.split('@')
        scene, lim = generate_terrain(name, float(hip), seed=10)
        assert len(scene['boxes']) == len(ref['boxes']), key
        np.testing.assert_allclose(lim, ref['terrain_limits'], rtol=0, atol=1e-12)
        for b, r in zip(scene['boxes'], ref['boxes']):
            np.testing.assert_allclose(b['pos'], r['pos'], atol=1e-12)
            np.testing.assert_allclose(b['size'], r['size'], atol=1e-12)
            np.testing.assert_allclose(b['quat'], r['quat'], atol=1e-12)
    assert np.array_equal(np.random.get_state()[1][:5], state), 'the global numpy generator must be restored (local_seed)'
    for name, nbox in (('ramp', 1), ('slippery', 2), ('stairs', 50)):
        scene, lim = generate_terrain(name, 0.3)
        assert len(scene['boxes']) == nbox and lim == (10000.0, -10000.0, 10000.0, -10000.0)
    assert generate_terrain('slippery', 0.3)[0]['boxes'][0]['priority'] == 2
    # perlin (terrain.py:345-356): 128 x 128 field, half extents 50 x hip, elevation 2 x hip, base 0.005; limits = 0.8 x half extent
    scene, lim = generate_terrain('perlin', 0.35)
    hf = scene['hfield']
    assert hf['data'].shape == (128, 128) and float(hf['data'].min()) == 0.0 and float(hf['data'].max()) == 1.0
    np.testing.assert_allclose(hf['size'], (17.5, 17.5, 0.7, 0.005))
    np.testing.assert_allclose(lim, (14.0, -14.0, 14.0, -14.0))
    assert not scene['boxes']
    with pytest.raises(ValueError):
        generate_terrain('moon', 0.3)


def test_dataset_hparams_encoding_follows_the_reference_conventions():
    """utils/data.py: env_hparams are stored the way the reference's save_dict_to_h5 does (h5py.py:22-48): lists / tuples
    as JSON, class references as 'TYPE:module.Class', None dropped by the HDF5 writer, nested dicts as groups."""
    import json
    from gym_quadruped_amd.sensors import IMU
    from gym_quadruped_amd.utils.data import _hparams_to_jsonable
    hp = dict(robot='aliengo', scene='flat', sim_dt=0.002, ref_base_lin_vel=(0.5, 1.0), legs_order=('FL', 'FR', 'RL', 'RR'),
              sensors=(IMU,), sensors_kwargs=({'accel_noise': 0.01},), external_disturbances_kwargs=None,
              state_obs_names=('qpos', 'qvel'), nested={'a': np.arange(3), 'b': np.float32(1.5)})
    enc = _hparams_to_jsonable(hp)
    assert enc['sensors'] == ['TYPE:gym_quadruped_amd.sensors.imu.IMU'] or enc['sensors'][0].startswith('TYPE:gym_quadruped_amd.sensors')
    assert enc['ref_base_lin_vel'] == [0.5, 1.0] and enc['external_disturbances_kwargs'] is None
    assert enc['nested'] == {'a': [0, 1, 2], 'b': 1.5} and enc['sensors_kwargs'] == [{'accel_noise': 0.01}]
    json.dumps(enc)   # everything is JSON-serialisable
    with pytest.raises(TypeError):
        _hparams_to_jsonable({'bad': object()})


def test_cloud_vertices_are_sorted_into_compact_chunks():
    """mjcf.sort_cloud_vertices: the hull clouds the kernels scan in 64-vertex chunks are ordered along their principal axis -
    same vertex SET (the deepest-vertex rule does not depend on the order), idempotent, and the chunks really are compact
    slices (what makes the per-chunk boxes of csrc/gq_host_model.cpp worth testing against a world box)."""
    import json
    from pathlib import Path
    from gym_quadruped_amd.mjcf import CLOUD_CHUNK, ModelDesc, load_compiled, sort_cloud_vertices
    for stem in ('mini_cheetah', 'hyqreal1'):
        raw = ModelDesc.from_json((Path(__file__).parents[1] / 'gym_quadruped_amd' / 'model_data' / f'{stem}.json').read_text())
        md = load_compiled(stem)
        before = np.array(md.vert_pos).copy()
        sort_cloud_vertices(md)
        assert np.array_equal(before, md.vert_pos)                      # idempotent
        big = 0
        for c in range(len(md.cloud_vertnum)):
            a, n = int(md.cloud_vertadr[c]), int(md.cloud_vertnum[c])
            v0, v1 = np.asarray(raw.vert_pos)[a:a + n], np.asarray(md.vert_pos)[a:a + n]
            assert sorted(map(tuple, np.round(v0, 12))) == sorted(map(tuple, np.round(v1, 12)))   # same set
            if n <= CLOUD_CHUNK:
                assert np.array_equal(v0, v1)
                continue
            big += 1
            ext = np.ptp(v1, axis=0).max()
            ax = np.linalg.svd(v1 - v1.mean(0))[2][0]
            chunk_ext = np.median([np.ptp(v1[k:k + CLOUD_CHUNK] @ ax) for k in range(0, n, CLOUD_CHUNK)])
            assert n < 4 * CLOUD_CHUNK or chunk_ext < 0.35 * ext, (stem, c, chunk_ext, ext)     # a typical chunk spans a fraction of the hull's long axis
                                                                         # (hull vertices crowd at the ends: the chunk across the sparse middle is long)
        assert big >= 3


@pytest.mark.parametrize('robot', ['mini_cheetah', 'hyqreal1', 'spot'])
def test_plane_support_tables_never_hide_the_support_vertex(robot):
    """cabi.plane_support_tables (the optional GqModelDesc.plane_* acceleration structure of the hull-versus-plane narrow phase): the
    direction-ordered copy of every cloud holds the same vertices, and for 20 000 random directions per hull the chunk of the support
    vertex (brute force over the whole cloud) is in the mask of the direction's cube-map cell - a missing bit would be a missed
    contact - and in the chunks its caps keep (the convex routine's selection).  Also: the masks do prune (fewer than half of the chunks on average) and directions on cell borders / cube edges
    (where the kernel's fp32 cell index may differ from this f64 one) are covered by both neighbours."""
    from gym_quadruped_amd.cabi import PLANE_GRID, plane_cell_of, plane_support_tables
    from gym_quadruped_amd.mjcf import load_compiled
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    md = load_compiled(Path(get_robot_config(robot).mjcf_filename).stem)
    pv, pm, _, caps = plane_support_tables(md)
    assert pv.shape == np.asarray(md.vert_pos).shape and pm.shape == (len(md.cloud_vertnum), 6 * PLANE_GRID ** 2)
    cap_kept = cap_chunks = 0
    rng = np.random.default_rng(4)
    seen = 0
    for cl in range(len(md.cloud_vertnum)):
        n, a = int(md.cloud_vertnum[cl]), int(md.cloud_vertadr[cl])
        V = pv[a:a + n]
        assert sorted(map(tuple, V)) == sorted(map(tuple, np.asarray(md.vert_pos, dtype=np.float64)[a:a + n]))
        if n <= 64:
            assert np.array_equal(V, np.asarray(md.vert_pos, dtype=np.float64)[a:a + n]) and (pm[cl] == 1).all()
            continue
        seen += 1
        nch = (n + 63) // 64
        dirs = rng.normal(size=(20000, 3))
        # directions on cell borders and cube edges: snap one or two raster coordinates to a grid line
        snap = dirs[:4000] / np.abs(dirs[:4000]).max(1)[:, None]
        k = rng.integers(0, 3, 4000)
        line = np.round((snap[np.arange(4000), k] + 1) * 0.5 * PLANE_GRID) / PLANE_GRID * 2 - 1
        snap[np.arange(4000), k] = np.where(np.abs(snap[np.arange(4000), k]) < 1, line, snap[np.arange(4000), k])
        dirs[:4000] = snap
        dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        kept = 0
        # the chunks' caps (the convex routine picks the chunks of a support query by them, csrc/gq_convex.h cvx_minkowski: kept when
        # cos(direction, axis) >= cosine - 1e-4): the support vertex's chunk is kept, or an exact tie in a kept one takes its place
        D = V @ dirs.T                                                  # [vertex][direction]
        sup_all = D.argmax(0)
        keep = (caps[cl, :nch, :3] @ dirs.T >= caps[cl, :nch, 3][:, None] - 1e-4)   # [chunk][direction]
        assert keep.any(0).all()
        miss = np.nonzero(~keep[sup_all // 64, np.arange(len(dirs))])[0]
        for i in miss:
            best = max(D[64 * q:64 * q + 64, i].max() for q in range(nch) if keep[q, i])
            assert best >= D[sup_all[i], i] - 1e-12, (robot, cl, dirs[i])
        if nch >= 5:   # (a cloud of two to four chunks has little to prune)
            cap_kept += keep.sum() / len(dirs); cap_chunks += nch
        for d in dirs:
            depth = V @ d
            sup = int(np.argmax(depth))
            cells = {plane_cell_of(d), plane_cell_of((d * (1 + 1e-6 * rng.normal(size=3))).astype(np.float32).astype(np.float64))}
            for c in cells:
                mk = int(pm[cl][c])
                kept += bin(mk).count('1') / len(cells)
                assert 0 < mk < (1 << nch)
                if not (mk >> (sup // 64)) & 1:   # only an exact tie with a vertex of a kept chunk may take its place
                    best = max(depth[64 * q:64 * q + 64].max() for q in range(nch) if (mk >> q) & 1)
                    assert best >= depth[sup] - 1e-12, (robot, cl, d)
        assert kept / len(dirs) < (0.5 * nch if nch >= 4 else nch)   # (a cloud of two or three chunks has little to prune)
    assert seen >= 1
    assert cap_chunks == 0 or cap_kept < 0.5 * cap_chunks, (cap_kept, cap_chunks)   # the caps do prune (patch-ordered chunks: rounds 4 - 5's cell-ordered strips kept 5 of 8)


def test_product_library_reads_no_environment_variable():
    """The profiling knobs (GQ_STOP_STAGE, GQ_FORCE_SELF, GQ_SELF_CUT, GQ_MB_FLAGS) belong to development builds (tools/dev_build.sh,
    -DGQ_DEV_KNOBS): the product sources call getenv only under that macro or under the emulator's trace macro, and the built library has no
    undefined reference to getenv."""
    import re
    import subprocess
    from gym_quadruped_amd import _lib
    csrc = Path(_lib.__file__).parent / 'csrc'
    for f in sorted(csrc.glob('*.h')) + sorted(csrc.glob('*.hip')) + sorted(csrc.glob('*.cpp')):
        guard = []
        for ln in f.read_text().splitlines():
            t = ln.strip()
            if t.startswith('#if'):
                guard.append(t)
            elif t.startswith('#else') and guard:
                guard[-1] = '#else of ' + guard[-1]
            elif t.startswith('#endif') and guard:
                guard.pop()
            if re.search(r'\bgetenv\s*\(', ln) and not t.startswith(('/*', '*', '//')):
                assert any(g.startswith(('#ifdef GQ_DEV_KNOBS', '#ifdef GQ_EMU_TRACE')) for g in guard), f'{f.name}: getenv outside GQ_DEV_KNOBS / GQ_EMU_TRACE: {t[:80]}'
    if _lib.LIB_PATH.exists():
        und = subprocess.run(['nm', '-D', '--undefined-only', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
        assert 'getenv' not in und, 'libgq.so imports getenv'


@pytest.mark.parametrize('robot', ['mini_cheetah', 'hyqreal1', 'spot'])
def test_support_grid_blend_is_an_upper_bound_and_tight(robot):
    """cabi.support_grids (GqModelDesc.support_grid: a hull's support function at the nodes of a cube map) as the kernel reads it
    (csrc/gq_convex.h cvx_hgrid, restated here in fp32): the bilinear blend of a cell's four nodes is never below the true support of the
    direction - the lane-parallel mid phase may only cull pairs that are apart - and within millimetres of it (second order in the cell)."""
    from gym_quadruped_amd.cabi import support_grids
    from gym_quadruped_amd.mjcf import load_compiled
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    md = load_compiled(Path(get_robot_config(robot).mjcf_filename).stem)
    T = support_grids(md).astype(np.float32)
    V = np.asarray(md.vert_pos, dtype=np.float64)
    rng = np.random.default_rng(3)
    f32 = np.float32
    worst = 0.0
    for cl in range(len(md.cloud_vertnum)):
        n, a = int(md.cloud_vertnum[cl]), int(md.cloud_vertadr[cl])
        if n < 3:
            continue
        D = rng.normal(size=(20000, 3)) * rng.uniform(0.2, 3.0, (20000, 1))     # any length
        D[:3000] /= np.abs(D[:3000]).max(1)[:, None]                            # on cube edges / cell borders
        D[:1000, 1] = np.round(D[:1000, 1] * 8) / 8
        D = D.astype(f32)
        ad = np.abs(D)
        major = np.where((ad[:, 0] >= ad[:, 1]) & (ad[:, 0] >= ad[:, 2]), 0, np.where(ad[:, 1] >= ad[:, 2], 1, 2))
        idx = np.arange(len(D))
        mj = ad[idx, major]
        face = 2 * major + (D[idx, major] < 0)
        oa = np.where(major == 0, 1, 0); ob = np.where(major == 2, 1, 2)
        from gym_quadruped_amd.cabi import SUPPORT_GRID as NG
        fa = np.clip((D[idx, oa] / mj + f32(1)) * f32(0.5 * NG), 0, NG).astype(f32); fb = np.clip((D[idx, ob] / mj + f32(1)) * f32(0.5 * NG), 0, NG).astype(f32)
        ia = np.minimum(fa.astype(np.int32), NG - 1); ib = np.minimum(fb.astype(np.int32), NG - 1)
        ta = (fa - ia).astype(f32); tb = (fb - ib).astype(f32)
        t00, t01, t10, t11 = T[cl, face, ia, ib], T[cl, face, ia, ib + 1], T[cl, face, ia + 1, ib], T[cl, face, ia + 1, ib + 1]
        h0 = t00 + tb * (t01 - t00); h1 = t10 + tb * (t11 - t10)
        bound = (mj * (h0 + ta * (h1 - h0)) + f32(2e-6) * mj).astype(np.float64)
        true = (V[a:a + n] @ D.astype(np.float64).T).max(0)
        assert (bound >= true).all(), (robot, cl, float((true - bound).max()))
        worst = max(worst, float(((bound - true) / np.linalg.norm(D, axis=1)).max()))
    assert 0.0 < worst < 0.03, worst   # per unit direction: the blend's slack stays below three centimetres on the largest hull (spot's 0.8 m trunk: 2.1 cm; typically millimetres)
