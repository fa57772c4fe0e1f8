"""Scene description - host-side mirror of the reference's ``utils/mujoco/terrain.py::generate_terrain`` (:309-365).

The reference builds an MJCF scene tree and hands it to MuJoCo; here a scene is a plain dict consumed by the model
marshaller (``cabi.MarshalledModel``): the floor plane plus a list of static world boxes.

* ``flat``: infinite plane named ``floor`` with MuJoCo default contact parameters
  (``assets/scene_flat.xml:31``: ``<geom name="floor" size="0 0 0.05" type="plane"/>``), terrain limits
  ``(10000, -10000, 10000, -10000)`` as ``(max_x, min_x, max_y, min_y)`` (terrain.py:357-359).
* ``random_boxes`` (terrain.py:145-238, parameters :324-335) and ``random_pyramids`` (:241-295, :336-344): the same
  draws from numpy's global generator under ``local_seed(seed)`` in the same order, so the box lists are the reference's
  (pinned by ``tests/golden/terrain_boxes.json``, produced by running the reference's own function).
* ``ramp`` / ``slippery`` / ``stairs``: the box geoms of the reference's static scene files, shipped as data
  (``model_data/static_scenes.json``, extracted by ``tools/gen_golden_terrain.py``).
* ``perlin`` needs a height-field narrow phase and ``noise.pnoise2``: not built - raises.

Scene generation is host-side set-up code; the boxes travel in ``GqModelDesc`` (``box_*`` tables) and are simulated by the
BOXES variants of the step kernel (``csrc/gq_boxes.h``; Newton solver only - ``gq_model_create`` rejects them with PGS).
"""
from __future__ import annotations

import contextlib
import json
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation

_FLOOR_DEFAULT = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                      solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
_BOX_DEFAULT = dict(friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0, solmix=1.0, solref=(0.02, 1.0),
                    solimp=(0.9, 0.95, 0.001, 0.5, 2.0), condim=3, priority=0)
_STATIC = ('ramp', 'slippery', 'stairs')
_FLAT_LIMITS = (10000.0, -10000.0, 10000.0, -10000.0)


@contextlib.contextmanager
def local_seed(seed):
    """numpy's GLOBAL generator seeded for the duration of the block, restored afterwards (terrain.py:299-306)."""
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


def _box(pos, euler, size, **over):
    """World box from a centre, xyz Euler angles and FULL extents (add_box, terrain.py:121-142: MuJoCo takes half sizes)."""
    quat = Rotation.from_euler('xyz', euler).as_quat(canonical=True, scalar_first=True)
    b = dict(_BOX_DEFAULT)
    b.update(pos=[float(x) for x in pos], size=[0.5 * float(x) for x in size], quat=[float(x) for x in quat])
    b.update(over)
    return b


def _world_of_boxes(init_pos, euler, nums, box_size, box_euler, separation, box_size_rand, box_euler_rand, separation_rand,
                    random_roll_pitch):
    """Grid of randomly sized / oriented boxes (add_world_of_boxes, terrain.py:145-238).  Draw order per box: size xy (2),
    size z (1), Euler angles (3, or yaw only), x separation (1), y separation (1); two draws for the first column step."""
    U = np.random.uniform
    init_pos, separation, separation_rand = np.asarray(init_pos, float), np.asarray(separation, float), np.asarray(separation_rand, float)
    box_size, box_size_rand, box_euler_rand = np.asarray(box_size, float), np.asarray(box_size_rand, float), np.asarray(box_euler_rand, float)
    Rw = Rotation.from_euler('xyz', euler).as_matrix()
    boxes, local = [], np.zeros(3)
    ext = [0.0, 0, 0.0, 0]   # largest |x|, its sign, largest |y|, its sign
    step = separation + separation_rand * U(-1.0, 1.0, 2)
    for _ in range(nums[0]):
        local[0] += step[0]
        local[1] = 0.0
        for _ in range(nums[1]):
            size_xy = box_size[0:2] + box_size_rand[0:2] * U(-0.2, 0.2, 2)
            size_z = box_size[2] + box_size_rand[2] * U(-0.1, 0.15, 1)
            if random_roll_pitch:
                ang = np.asarray(box_euler, float) + box_euler_rand * U(-1.0, 1.0, 3)
            else:
                ang = np.array(box_euler, float)
                ang[2] = ang[2] + box_euler_rand[2] * U(-1, 1, 1)[0]
            step = np.array([separation[0] + separation_rand[0] * U(0, 0.5, 1)[0], separation[1] + separation_rand[1] * U(-0.5, 0.5, 1)[0]])
            local[1] += step[1]
            boxes.append(_box(Rw @ local + init_pos, ang, [size_xy[0], size_xy[1], size_z[0]]))
            ax, ay = abs(local[0] + init_pos[0]), abs(local[1] + init_pos[1])
            if ax >= ext[0]:
                ext[0], ext[1] = ax, (1 if ax > 0 else -1)
            if ay >= ext[2]:
                ext[2], ext[3] = ay, (1 if ay > 0 else -1)
    max_x, max_y = ext[0] * ext[1], ext[2] * ext[3]
    cx, cy = (max_x + init_pos[0]) / 2, (max_y + init_pos[1]) / 2
    radius = 1.2 * np.sqrt(2 * (max_x - cx) ** 2) if ext[0] >= ext[2] else 1.2 * np.sqrt(2 * (max_y - cy) ** 2)
    return boxes, (float(cx + radius), float(cx - radius), float(cy + radius), float(cy - radius))


def _world_of_pyramid(init_pos, yaw, width, max_height, length, stair_nums):
    """Stack of shrinking slabs (add_world_of_pyramid, terrain.py:241-295); two draws: slab height, stride."""
    U = np.random.uniform
    boxes, local = [], np.array([0.0, 0.0, -0.05])
    height = U(0.08, max_height, 1)[0]
    stride = U(0.5, 1.0, 1)[0]
    Rz = Rotation.from_euler('xyz', [0, 0, yaw]).as_matrix()
    mx = my = 0.0
    center = (0.0, 0.0)
    for i in range(int(stair_nums)):
        local[2] += height
        x, y, _ = Rz @ local
        w, l = width - stride * i, length - stride * i
        if w < 0.3 or l < 0.3:
            break
        boxes.append(_box([x + init_pos[0], y + init_pos[1], local[2]], [0.0, 0.0, yaw], [w, l, height]))
        if i == 0:
            mx, my = abs(x + init_pos[0] + w / 2.0), abs(y + init_pos[1] + l / 2.0)
            center = (x + init_pos[0], y + init_pos[1])
    radius = 1.5 * np.sqrt(2 * (mx - center[0]) ** 2) if mx >= my else 1.5 * np.sqrt(2 * (my - center[1]) ** 2)
    return boxes, (float(center[0] + radius), float(center[0] - radius), float(center[1] + radius), float(center[1] - radius))


def generate_terrain(terrain_name: str = 'flat', hip_height: float = 0.3, seed: int = 10):
    """Returns ``(scene_desc, terrain_limits)``: ``scene_desc = {'name', 'floor': {...}, 'boxes': [{pos, size (half
    extents), quat (wxyz), friction, priority, condim, solref, solimp, solmix, margin, gap}, ...]}``."""
    scene = {'name': terrain_name, 'floor': dict(_FLOOR_DEFAULT), 'boxes': []}
    if terrain_name == 'flat':
        return scene, _FLAT_LIMITS
    if terrain_name in _STATIC:   # robot_model/scene_<name>.xml exists in the reference: static file, flat limits (:319-321)
        data = json.loads((Path(__file__).parent / 'model_data' / 'static_scenes.json').read_text())[terrain_name]
        for b in data['boxes']:
            box = dict(_BOX_DEFAULT)
            box.update({k: v for k, v in b.items() if k != 'name'})
            box['friction'] = tuple(b.get('friction', _BOX_DEFAULT['friction']))
            scene['boxes'].append(box)
        return scene, _FLAT_LIMITS
    with local_seed(seed):
        h = float(hip_height)
        if terrain_name == 'random_boxes':      # parameters of terrain.py:324-335
            scene['boxes'], limits = _world_of_boxes(
                init_pos=[0.5, -3.0, 0.02], euler=[0, 0, 0.0], nums=[10, 10], box_size=[2 * h, 2 * h, h / 2.0], box_euler=[0.0, 0.0, 0.0],
                separation=[2 * h, 2 * h], box_size_rand=[0.5 * h, 0.5 * h, h / 2], box_euler_rand=[0.1, 0.1, 2 * np.pi],
                separation_rand=[0, 1], random_roll_pitch=True)
            return scene, limits
        if terrain_name == 'random_pyramids':   # :336-344; the stair count is drawn before the slab height and stride
            stair_nums = np.random.uniform(2, 8, 1)[0]
            scene['boxes'], limits = _world_of_pyramid(init_pos=[3, 0, 0.02], yaw=0.0, width=10 * h, max_height=5 * h, length=10 * h,
                                                       stair_nums=stair_nums)
            return scene, limits
    if terrain_name == 'perlin':
        raise NotImplementedError("scene 'perlin' needs a height-field narrow phase and noise.pnoise2 (SURVEY.md §8f rank 2): not built")
    raise ValueError(f'Invalid scene name: {terrain_name}, available are: flat, random_boxes, random_pyramids, '
                     f'perlin, stairs, ramp, slippery')
